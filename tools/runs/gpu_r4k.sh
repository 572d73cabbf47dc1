#!/bin/bash
# round 4, GPU session k: the ingest with sharded open-addressing maps and parallel announcement / reply passes; the new GPU tests (pairs-first kernel on the goldens,
# grouping / pairs among the scheduling variants); full suite, host-logic bench on the box, full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4k_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 2>&1 | grep -E "host logic|txout replies n|sub-batch 1/1|sub-batch 2/4|process n=400000" | tail -6; done 2>&1 | tee gpurun_out/r4k_ingest_host.txt
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err; echo "bench.py wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4k_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4k_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms (%d steps) | roofline %s frac %.3f launch %.3f ms | iso %.3f ms | pipeline %.3f | traffic x%s" % (
    d["value"] / 1e6, d["ms_per_step"], d["steps"], r["mode"], r["frac"], r["avg_launch_ms"], r["isolated"]["launch_ms"], r["pipeline"]["frac"], r["traffic_over_algorithmic"]))
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k != "note" and k != "shape"})
print("warm %.1f h2h %.3f cfg4 %.1f cfg5 %.1f" % (d["warm_cache"]["value"] / 1e6, d["value_host_to_host"]["ratio_to_value"], o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6))
PY
