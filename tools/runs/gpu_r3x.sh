#!/bin/bash
# round 3, GPU session x: the new boundary test of the one-launch path; the ingest flood after the apply-pass work
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "block_boundaries or scheduling" 2>&1 | tail -5
timeout 300 python tools/ingest_gpu_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/r3x_ingest.txt
