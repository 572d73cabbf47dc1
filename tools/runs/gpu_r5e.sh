#!/bin/bash
# round 5, GPU session e: events per flush (1 / 2 / 3) x copy streams on the host->host cold loop; the full GPU suite; the bench line (strong-scaling sweep with two-chunk shards)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
: > gpurun_out/r5e_copy_events.txt
for CFG in "3 1" "1 1" "2 1" "1 2" "3 1" "1 1"; do
  set -- $CFG
  echo "LAMD_COPY_EVENTS=$1 LAMD_COPY_STREAMS=$2" | tee -a gpurun_out/r5e_copy_events.txt
  LAMD_COPY_EVENTS=$1 LAMD_COPY_STREAMS=$2 PROBE_STEPS=30 timeout 300 python tools/call_trace_probe.py stream 2>&1 | grep -E "loop" | tee -a gpurun_out/r5e_copy_events.txt
done
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r5e_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5e_bench.json 2> gpurun_out/r5e_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r5e_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5e_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5e_bench.json").read().strip().splitlines()[-1])
ss = d.get("strong_scaling_1gpu", {})
if "cfg4_gossip_replay" in ss:
    for W in ("2", "4", "8"):
        r = ss["cfg4_gossip_replay"][W]
        print("cfg4 W=%s one chunk %s | two chunks %s | chunks=%s" % (W, [round(x, 2) for x in r["shard_ms"]], [round(x, 2) for x in r["shard_ms_two_chunks"]], r["chunks"]))
o = d["other_configs_1gpu"]["cfg5_commit_storm_one_commitment_per_flush"]
for k, v in o.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items()})
PY
