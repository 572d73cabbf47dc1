#!/bin/bash
# round 5, GPU session h: the full GPU suite, smoke, the bench line for profiles/, the collective path on one rank (JSON + a kernel trace of its loop)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r5h_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5h_bench.json 2> gpurun_out/r5h_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r5h_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5h_bench.json
S=$(date +%s); timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5h_bench_20steps.json 2> gpurun_out/r5h_bench_20steps.err; echo "bench.py --steps 20 rc=$? wall $(( $(date +%s) - S )) s"
python tools/bench_summary.py gpurun_out/r5h_bench_20steps.json | head -3
LAMD_BENCH_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --skip-extra > gpurun_out/r5h_bench_gather.json 2> gpurun_out/r5h_bench_gather.err; echo "gather bench rc=$?"
python tools/bench_summary.py gpurun_out/r5h_bench_gather.json | head -2
(cd /tmp && export TMPDIR=/tmp && LAMD_BENCH_GATHER=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5h_trace_gather -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29712 $R/bench.py --gpus 1 --ab --steps 12 --warmup 3 > $R/gpurun_out/r5h_trace_gather.json 2> $R/gpurun_out/r5h_trace_gather.err)
find gpurun_out/r5h_trace_gather -name "*.csv" | xargs ls -la | cut -c1-150
find gpurun_out/r5h_trace_gather -name "*.csv" | xargs gzip -9
