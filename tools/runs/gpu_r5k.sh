#!/bin/bash
# round 5, GPU session k: the full GPU suite on the final tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -8 | tee gpurun_out/r5k_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
