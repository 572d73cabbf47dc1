#!/bin/bash
# round 3, GPU session af: a large ecmult launch cut in two (LAMD_ECMULT_CHAIN=2): part 1 chained behind the previous call's part 1, part 2 (the last
# LAMD_ECMULT_TAIL work items) on a lowest-priority stream -- meant to fill part 1's tail and the head of the next call's part 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s, step %.2f ms, (first-part) launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for cfg in "0 0" "2 131072" "2 65536" "2 196608" "1 0" "2 131072" "0 0"; do
  set -- $cfg; k=$((k+1))
  LAMD_ECMULT_CHAIN=$1 LAMD_ECMULT_TAIL=$2 timeout 300 python bench.py --roofline-only > gpurun_out/r3af_$k.json 2> gpurun_out/r3af_$k.err
  line gpurun_out/r3af_$k.json "LAMD_ECMULT_CHAIN=$1 LAMD_ECMULT_TAIL=$2"
done | tee gpurun_out/r3af_tail.txt
