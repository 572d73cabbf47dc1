#!/bin/bash
# round 5, GPU session w: the cold-row ladder in its hot form (signed odd digits, bare additions, complete ladder on suspicion): the full GPU suite, then the bench line
# (key-reuse sweep's all-distinct point, configs[3], first-sight latencies)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee gpurun_out/r5w_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5w_bench.json 2> gpurun_out/r5w_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5w_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5w_bench.json
