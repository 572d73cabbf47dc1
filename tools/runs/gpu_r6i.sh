#!/bin/bash
# round 6, session i: the final tree as the driver will run it -- GPU suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6i
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert|served streaming" | tail -6 | tee gpurun_out/r6i/pytest.txt; echo "pytest wall $(( $(date +%s) - S )) s"
cp gpurun_out/served_stream.json gpurun_out/r6i/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r6i/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6i/bench20.json 2> gpurun_out/r6i/bench20.err; cp bench_details.json gpurun_out/r6i/details20.json
grep ^real gpurun_out/r6i/bench20.err; wc -c gpurun_out/r6i/bench20.json; cat gpurun_out/r6i/bench20.json
