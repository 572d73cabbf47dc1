#!/bin/bash
# round 6, session ac: wave priority (LAMD_PRIO bit mask: 1 front end, 2 cold-row ladder, 4 scalar preparation, 8 key tables, 16 parity stage; default 0) against the
# strong-scaling sweeps -- the timeline of a 1/8 gossip shard (session ab) shows the 30 waves of the node keys' doubling chains stretched 4x by the ladder waves
# they share their SIMDs with
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ac
export GPU_MAX_HW_QUEUES=16
for rep in $(seq 1 ${REPS:-2}); do
  for pr in ${@:-0 8 9 13 29}; do
    LAMD_PRIO=$pr timeout 300 python bench.py --cpu-sample 0 --no-h2h --details gpurun_out/r6ac/d_${pr}_$rep.json > gpurun_out/r6ac/l_${pr}_$rep.json 2> gpurun_out/r6ac/e_${pr}_$rep.err
    python - $pr $rep <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6ac/d_%s_%s.json" % (sys.argv[1], sys.argv[2])))
s = d["strong_scaling_1gpu"]; c, e = s["cfg4_gossip_replay"], s["cfg5_commit_storm_streaming"]
print("LAMD_PRIO=%-3s value %.1f M/s | cfg4 T1 %.2f W8 %.2f ms x%.2f | cfg5 T1 %.2f W8 %.2f ms x%.2f | mismatches %d %d %d" % (
    sys.argv[1], d["value"] / 1e6, c["1"]["slowest_ms"], c["8"]["slowest_ms"], c["predicted_speedup_8"],
    e["1"]["slowest_ms"], e["8"]["slowest_ms"], e["predicted_speedup_8"], d["parity"]["mismatches"], c["mismatches"], e["mismatches"]))
PY
  done
done 2>&1 | tee gpurun_out/r6ac/ab.txt
