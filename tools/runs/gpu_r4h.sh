#!/bin/bash
# round 4, GPU session h: pairs-first table-driven ecmult (k_ecmult_keyed_pairs): parity subset, then A/B against one mixed addition per entry (LAMD_PAIRS=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random or diagnostics or keyed or degenerate or cfg4_gossip_replay_small or cfg5_commit_storm_streaming or cache_warm" 2>&1 | tail -5 | tee gpurun_out/r4h_pytest_subset.log
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for v in 0 1 0 1 0 1; do
  k=$((k+1))
  LAMD_PAIRS=$v timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4h_ab_$k.json 2> gpurun_out/r4h_ab_$k.err || tail -3 gpurun_out/r4h_ab_$k.err
  line gpurun_out/r4h_ab_$k.json "LAMD_PAIRS=$v"
done | tee gpurun_out/r4h_ab.txt
