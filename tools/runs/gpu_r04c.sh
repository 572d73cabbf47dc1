#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== cache on"; timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
echo "== LAMD_CACHE=0"; LAMD_CACHE=0 timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests -m gpu -x -q -k "queue or stream or stress or flush" 2>&1 | tail -3
