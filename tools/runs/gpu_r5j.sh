#!/bin/bash
# round 5, GPU session j: the scheduling-variants test with the round's new knobs, then the full suite, smoke, and the bench lines for profiles/
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r5j_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5j_bench.json 2> gpurun_out/r5j_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5j_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5j_bench.json
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 > gpurun_out/r5j_bench_gather.json 2> gpurun_out/r5j_bench_gather.err; echo "gather bench rc=$?"
python tools/bench_summary.py gpurun_out/r5j_bench_gather.json | head -1
S=$(date +%s); timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r5j_bench_20steps.json 2> gpurun_out/r5j_bench_20steps.err; echo "bench.py --steps 20 rc=$? wall $(( $(date +%s) - S )) s"
python tools/bench_summary.py gpurun_out/r5j_bench_20steps.json | head -1
