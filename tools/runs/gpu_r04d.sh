#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r04d.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r04d.log
echo "== cache on"; timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
echo "== LAMD_CACHE=0"; LAMD_CACHE=0 timeout 300 python tools/host_path_probe.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout 600 python bench.py --cpu-sample 0 > gpurun_out/bench_r04d_$i.json 2> gpurun_out/bench_r04d_$i.err
echo "bench rc=$?"
done
