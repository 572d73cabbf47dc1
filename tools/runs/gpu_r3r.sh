#!/bin/bash
# round 3, GPU session r: the one-launch path for calls of up to 4096 rows (k_small_verify as a grid of 64-row blocks) -- parity tests of the
# small paths, latency against the general path (LAMD_SMALL_KERNEL=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py -m gpu -x -q -k "small or learn or veneers or degenerate or golden or ragged or shim or cfg1 or commit or keyed" 2>&1 | tail -4
for v in 1 0; do
  echo "== LAMD_SMALL_KERNEL=$v"; LAMD_SMALL_KERNEL=$v timeout 300 python tools/latency_probe.py 2>&1 | grep -v "^W\|amdgpu.ids"
done > gpurun_out/r3r_latency.txt 2>&1
cat gpurun_out/r3r_latency.txt
