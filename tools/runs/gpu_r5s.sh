#!/bin/bash
# round 5, GPU session s: the tree as it stands -- full GPU suite, smoke, the bench line (plain and through the collective path), rocprofv3 --stats of the roofline loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r5s_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5s_bench.json 2> gpurun_out/r5s_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5s_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5s_bench.json
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 > gpurun_out/r5s_bench_gather.json 2> gpurun_out/r5s_bench_gather.err; echo "gather bench rc=$?"
python tools/bench_summary.py gpurun_out/r5s_bench_gather.json | head -1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5s_roofline_stats -- python $R/bench.py --roofline-only > $R/gpurun_out/r5s_roofline_only.json 2> $R/gpurun_out/r5s_roofline_only.err); echo "roofline-only under rocprof rc=$?"
cp $(find gpurun_out/r5s_roofline_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r5s_roofline_only_kernel_stats.csv; rm -rf gpurun_out/r5s_roofline_stats
head -3 gpurun_out/r5s_roofline_only_kernel_stats.csv | cut -c1-200
