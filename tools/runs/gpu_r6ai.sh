#!/bin/bash
# round 6, session ai: the bench process bound to its GPU's NUMA node (the default from here on) against the floating process (LAMD_BENCH_NUMA=0), alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ai
export GPU_MAX_HW_QUEUES=16
for rep in $(seq 1 ${REPS:-4}); do
  for nb in 1 0; do
    LAMD_BENCH_NUMA=$nb timeout 300 python bench.py --cpu-sample 0 --details gpurun_out/r6ai/d_${nb}_$rep.json > gpurun_out/r6ai/l_${nb}_$rep.json 2> gpurun_out/r6ai/e_${nb}_$rep.err
    python - $nb $rep <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6ai/d_%s_%s.json" % (sys.argv[1], sys.argv[2])))
s = d["strong_scaling_1gpu"]; c, e = s["cfg4_gossip_replay"], s["cfg5_commit_storm_streaming"]
print("numa_bound=%s %s value %.1f M/s  host->host %.3f x | cfg4 T1 %.2f W8 %.2f x%.2f | cfg5 T1 %.2f W8 %.2f ms x%.2f | mismatches %d" % (
    sys.argv[1], d.get("host_numa"), d["value"] / 1e6, d["config"]["host_to_host_over_value"], c["1"]["slowest_ms"], c["8"]["slowest_ms"], c["predicted_speedup_8"],
    e["1"]["slowest_ms"], e["8"]["slowest_ms"], e["predicted_speedup_8"], d["parity"]["mismatches"] + c["mismatches"] + e["mismatches"]))
PY
  done
done 2>&1 | tee gpurun_out/r6ai/ab.txt
