#!/bin/bash
# round 3, GPU session o: the large ecmult launches chained one after the other (LAMD_ECMULT_CHAIN=1) against overlapping (=0), 4 and 6 lanes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=d['rates']['kernel_ms_ecdsa_isolated']
print('$2: cold %.1f M/s, step %.2f ms, launch in the loop %.3f ms (frac %.3f), isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
for cfg in "0 4" "1 4" "1 6" "0 4" "1 4" "1 5" "1 8"; do
  set -- $cfg
  LAMD_ECMULT_CHAIN=$1 LAMD_LANES=$2 timeout 300 python bench.py --roofline-only > gpurun_out/r3o_c$1_l$2.json 2> gpurun_out/r3o_c$1_l$2.err
  line gpurun_out/r3o_c$1_l$2.json "LAMD_ECMULT_CHAIN=$1 LAMD_LANES=$2"
done | tee gpurun_out/r3o_chain.txt
