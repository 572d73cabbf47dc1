#!/bin/bash
# round 3, GPU session l: the fused front end (six launches instead of 19 + 5 fills) -- full GPU suite, then A/B in the cold loop against
# LAMD_FUSED_FRONT=0; lanes 4 / 5 / 6 / 7 with two streams per lane (LAMD_MERGE_SIDE=1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3l_pytest.log
tail -3 gpurun_out/r3l_pytest.log
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s, step %.2f ms, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
for v in 1 0 1 0; do
  LAMD_FUSED_FRONT=$v timeout 300 python bench.py --roofline-only > gpurun_out/r3l_f$v.json 2> gpurun_out/r3l_f$v.err
  line gpurun_out/r3l_f$v.json "LAMD_FUSED_FRONT=$v"
done | tee gpurun_out/r3l_fused.txt
for l in 4 5 6 7; do
  LAMD_MERGE_SIDE=1 LAMD_LANES=$l timeout 300 python bench.py --roofline-only > gpurun_out/r3l_m$l.json 2> gpurun_out/r3l_m$l.err
  line gpurun_out/r3l_m$l.json "LAMD_MERGE_SIDE=1 LAMD_LANES=$l"
done | tee -a gpurun_out/r3l_fused.txt
