#!/bin/bash
# A/B of build-time knobs that change the dominant kernel.  Step 1 (container, ~2.5 min per variant, in parallel): build the variants.
# Step 2 (GPU box, ~15 s per run): alternate them against the shipped library.
#   LAMD_TOUCH_NEXT        touch the next table entry before each addition (verify_core.h)
#   LAMD_KEYED_THREADS=64  / 128: block size of the table-driven ecmult launches (tail of a 3 907-block grid on 768 block slots)
#   LAMD_TABLE_LIMBS=0     table entries as 8 x 32-bit words (the round-2 layout) instead of nine 29-bit limbs
set -u
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unknown-pragmas -mllvm -amdgpu-codegenprepare-mul24=false"
if [ "${1:-build}" = build ]; then
  mkdir -p tools/variants
  for v in "touch:-DLAMD_TOUCH_NEXT" "t64:-DLAMD_KEYED_THREADS=64" "t128:-DLAMD_KEYED_THREADS=128" "words:-DLAMD_TABLE_LIMBS=0"; do
    name=${v%%:*}; flag=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS $flag -o tools/variants/liblightning_amd_$name.so lightning_amd/csrc/lamd_engine.hip &
  done
  wait
  # the signer library reads the engine's G table: the 8-word layout needs its own build
  /opt/rocm/bin/hipcc $FLAGS -DLAMD_TABLE_LIMBS=0 -o tools/variants/liblightning_amd_testgen_words.so lightning_amd/csrc/lamd_testgen.hip -Ltools/variants -l:liblightning_amd_words.so '-Wl,-rpath,$ORIGIN'
  ls -la tools/variants/*.so
else
  mkdir -p gpurun_out
  for v in base words touch t64 t128 base words touch t64 t128; do
    unset LAMD_LIB_PATH LAMD_TESTGEN_LIB_PATH
    if [ $v != base ]; then export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_$v.so; fi
    if [ $v = words ]; then export LAMD_TESTGEN_LIB_PATH=$PWD/tools/variants/liblightning_amd_testgen_words.so; fi
    timeout 300 python bench.py --roofline-only > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$v: cold %.1f M/s, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
  done
fi
