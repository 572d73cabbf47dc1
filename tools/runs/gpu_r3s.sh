#!/bin/bash
# round 3, GPU session s: full GPU suite on the build with the multi-block latency path; where a commitment's latency goes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3s_pytest.log; tail -3 gpurun_out/r3s_pytest.log
python tools/commit_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/r3s_commit_probe.txt
