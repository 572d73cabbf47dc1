#!/bin/bash
# round 3, GPU session v: full GPU suite with the latency path behind check_tx_sig / bolt12 / gossip host calls
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3v_pytest.log; tail -4 gpurun_out/r3v_pytest.log
