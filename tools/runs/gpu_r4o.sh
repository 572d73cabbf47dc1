#!/bin/bash
# round 4, GPU session o: final build of the round -- full GPU suite, smoke, the full bench line (250 steps)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4o_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r4o_bench.json 2> gpurun_out/r4o_bench.err; echo "bench.py wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4o_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4o_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms (%d steps) | roofline %s frac %.3f launch %.3f ms sum %.3f <= %.3f | iso %.3f ms | pipeline %.3f | traffic x%s" % (
    d["value"] / 1e6, d["ms_per_step"], d["steps"], r["mode"], r["frac"], r["avg_launch_ms"], r["sum_of_launch_ms_per_step"], r["ms_per_step"], r["isolated"]["launch_ms"], r["pipeline"]["frac"], r["traffic_over_algorithmic"]))
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k != "note" and k != "shape"})
h = d["value_host_to_host"]
print("warm %.1f h2h %.1f ratio %.3f cfg4 %.1f cfg5 %.1f" % (d["warm_cache"]["value"] / 1e6, h["value"] / 1e6, h["ratio_to_value"], o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6),
      {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("cfg5 streaming", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in o["cfg5_commit_storm_streaming"].items() if isinstance(v, dict)}, "one per flush", {k: round(v["batches_per_s"]) for k, v in o["cfg5_commit_storm_one_commitment_per_flush"].items() if isinstance(v, dict)})
print("sweep", {k: round(v["verifies_per_s"]/1e6,1) for k,v in o["key_reuse_sweep"].items() if isinstance(v, dict)}, "latency", {k: round(v["p50_ms"], 3) for k, v in d["latency"].items() if isinstance(v, dict) and "p50_ms" in v}, "cfg1 ns/call", round(d["latency"]["cfg1_one_by_one_check_signed_hash"]["ns_per_call"]))
print("cpu", d["cpu_baseline"]["kind"], round(d["cpu_baseline"]["value"]), "parity", d["parity"]["mismatches"])
PY
