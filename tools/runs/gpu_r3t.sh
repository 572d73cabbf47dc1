#!/bin/bash
# round 3, GPU session t: k_small_verify with one comb body for both shapes -- parity tests of the small paths, the commitment probe, latency probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py -m gpu -x -q -k "small or learn or veneers or degenerate or golden or ragged or shim or cfg1 or commit or keyed" 2>&1 | tail -3
python tools/commit_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/r3t_commit_probe.txt
PROBE_SIZES=1,64,484 timeout 300 python tools/latency_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/r3t_latency.txt
