#!/bin/bash
# round 3, GPU session u: gossip messages a few at a time through the latency path -- gossip / shim parity tests, latency probe (both paths)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py tests/test_gossip_ingest.py -m gpu -x -q -k "gossip or shim or small or learn or golden or ingest or store or cfg4" 2>&1 | tail -3
for v in 1 0; do
  echo "== LAMD_SMALL_KERNEL=$v"; LAMD_SMALL_KERNEL=$v PROBE_SIZES=1 timeout 300 python tools/latency_probe.py 2>&1 | grep -v "^W\|amdgpu.ids"
done | tee gpurun_out/r3u_latency.txt
