#!/bin/bash
# round 4, GPU session i: rows grouped by key (k_group_*) and the pairs-first kernel with one-wave workgroups: parity subset, 2 x 2 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random or diagnostics or keyed or degenerate or cfg4_gossip_replay_small or cfg5_commit_storm_streaming or cache_warm or repeated" 2>&1 | tail -5 | tee gpurun_out/r4i_pytest_subset.log
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 0" "0 1" "1 1"; do
  set -- $v
  k=$((k+1))
  LAMD_GROUP=$1 LAMD_PAIRS=$2 timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4i_ab_$k.json 2> gpurun_out/r4i_ab_$k.err || tail -3 gpurun_out/r4i_ab_$k.err
  line gpurun_out/r4i_ab_$k.json "LAMD_GROUP=$1 LAMD_PAIRS=$2"
done | tee gpurun_out/r4i_ab.txt
