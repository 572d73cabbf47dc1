#!/bin/bash
set -u
mkdir -p gpurun_out
for ct in 4 8 16; do
  echo "== LAMD_COPY_THREADS=$ct cache on"; LAMD_COPY_THREADS=$ct timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
done
echo "== LAMD_COPY_THREADS=8 LAMD_CACHE=0"; LAMD_CACHE=0 LAMD_COPY_THREADS=8 timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
numactl -H 2>/dev/null | head -20
lscpu | head -25
