#!/bin/bash
# round 3, GPU session g: the north star's "LDS-staged precomputed G table" as a measured A/B (-DLAMD_G_LDS -DLAMD_KEYED_THREADS=768: 104 KB of
# 5-bit windows staged into LDS per block, 52 additions for u1*G) against the shipped 3 GiB / 22-bit-window table in HBM (12 additions)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for v in base glds base glds; do
  unset LAMD_LIB_PATH
  if [ $v != base ]; then export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_$v.so; fi
  timeout 300 python bench.py --roofline-only > gpurun_out/r3g_$v.json 2> gpurun_out/r3g_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/r3g_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$v: cold %.1f M/s, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))" || tail -3 gpurun_out/r3g_$v.err
done | tee gpurun_out/r3g_glds.txt
