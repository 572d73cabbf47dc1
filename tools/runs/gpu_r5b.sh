#!/bin/bash
# round 5, GPU session b: the new parity tests with their full output kept, the multi H2D A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for T in tests/test_gpu_commitment.py tests/test_cln_shim.py tests/test_gpu_multi.py; do
  timeout 600 python -X faulthandler -m pytest $T -m gpu -v -x > gpurun_out/r5b_$(basename $T .py).log 2>&1; echo "$T rc=$?"
  grep -E "passed|failed|error|PASSED|FAILED|Fatal|fault|Abort" gpurun_out/r5b_$(basename $T .py).log | head -20
done
timeout 300 python tools/multi_h2d_probe.py 2>&1 | tail -2 | tee gpurun_out/r5b_multi_h2d.txt
LAMD_MULTI_PINNED=0 timeout 300 python tools/multi_h2d_probe.py 2>&1 | tail -2 | tee -a gpurun_out/r5b_multi_h2d.txt
