#!/bin/bash
# round 5, last GPU session: the final tree -- full GPU suite, smoke, bench line plain and through the collective path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r5final_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5final_bench.json 2> gpurun_out/r5final_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5final_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5final_bench.json | grep -E "^value|^config|^sweep|^latency|^cfg"
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 > gpurun_out/r5final_bench_gather.json 2> gpurun_out/r5final_bench_gather.err; echo "gather bench rc=$?"
python tools/bench_summary.py gpurun_out/r5final_bench_gather.json | head -1
