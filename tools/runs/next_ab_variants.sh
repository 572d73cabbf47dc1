#!/bin/bash
# Queued A/B of build-time knobs that change the dominant kernel (none of them is in the shipped build; the shipped .hip_fatbin is
# byte-identical with and without the knobs' source).  Step 1 (container, ~3 min per variant, in parallel): build the variants.
# Step 2 (GPU box, ~10 s per run): alternate them against the shipped library.
#   LAMD_TOUCH_NEXT        touch the next table entry before each addition (verify_core.h)
#   LAMD_KEYED_THREADS=64  / 128: block size of the table-driven ecmult launches (tail of a 3 907-block grid on 768 block slots)
set -u
cd "$(dirname "$0")/../.."
if [ "${1:-build}" = build ]; then
  mkdir -p tools/variants
  for v in "touch:-DLAMD_TOUCH_NEXT" "t64:-DLAMD_KEYED_THREADS=64" "t128:-DLAMD_KEYED_THREADS=128"; do
    name=${v%%:*}; flag=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unknown-pragmas -mllvm -amdgpu-codegenprepare-mul24=false \
      $flag -o tools/variants/liblightning_amd_$name.so lightning_amd/csrc/lamd_engine.hip &
  done
  wait
  ls -la tools/variants/*.so
else
  mkdir -p gpurun_out
  for v in base touch t64 t128 base touch t64 t128; do
    if [ $v = base ]; then unset LAMD_LIB_PATH; else export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_$v.so; fi
    timeout 300 python bench.py --roofline-only > gpurun_out/ab_$v.json 2> /dev/null
    python -c "
import json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$v: cold %.1f M/s, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
  done
fi
