#!/bin/bash
# round 3, GPU session aa: small flushes of the streaming queue as one latency-kernel launch over the staging rows -- streaming tests, one
# commitment per flush with recurring channels (both paths)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q -k "stream or queue or reserve or commit or stress or scheduling or learn or small" 2>&1 | tail -3
for v in 1 0; do LAMD_SMALL_KERNEL=$v timeout 300 python tools/stream_small_batches.py 2>&1 | grep -v "^W\|amdgpu.ids"; done | tee gpurun_out/r3aa_small_flushes.txt
