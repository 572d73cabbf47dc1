#!/bin/bash
# round 5, GPU session t: configs[4] shards with ramped first flushes (tools/cfg5_ramp_probe.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
PROBE_FLUSH_SIZES=1 timeout 600 python tools/cfg5_ramp_probe.py 2>&1 | grep -E "ramp|Error|error" | tee gpurun_out/r5t_flush_sizes.txt
