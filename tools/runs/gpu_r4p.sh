#!/bin/bash
# round 4, GPU session p: static G table with 26-bit windows (10 additions for u1*G, 43 GB) against the 24 bits / 11 GiB that ship, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for v in base g26 base g26; do
  k=$((k+1))
  unset LAMD_LIB_PATH LAMD_TESTGEN_LIB_PATH
  if [ $v != base ]; then export LAMD_LIB_PATH=$PWD/tools/variants/$v/liblightning_amd.so LAMD_TESTGEN_LIB_PATH=$PWD/tools/variants/$v/liblightning_amd_testgen.so; fi
  timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4p_ab_$k.json 2> gpurun_out/r4p_ab_$k.err || tail -3 gpurun_out/r4p_ab_$k.err
  line gpurun_out/r4p_ab_$k.json "$v"
done | tee gpurun_out/r4p_g26.txt
