#!/bin/bash
# round 5, GPU session p: k_txsig_tx_hash rewritten (per-lane LDS message buffers, one compression site, 4.4 k instructions instead of 72 k):
# the template tests incl. the new stream-boundary test, the commitment tests, then the commitment probe with a kernel trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_commitment.py tests/test_served.py -m gpu -q -x -k "tx_sig or templates or transactions or commitment or served or native" 2>&1 | tail -5 | tee gpurun_out/r5p_tests.txt
timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee gpurun_out/r5p_commit_probe.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5p_trace_commit -- python $R/tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee -a $R/gpurun_out/r5p_commit_probe.txt)
find gpurun_out/r5p_trace_commit -name "*.csv" | xargs gzip -9
