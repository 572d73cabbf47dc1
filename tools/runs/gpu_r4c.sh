#!/bin/bash
# round 4, GPU session c: the libsecp-names test with its traceback, the ingest host bench after the false-sharing fix (16 / 32 threads), the full
# GPU suite, the full bench line at the default 250 steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_cln_shim.py -m gpu -x -q -k libsecp_names 2>&1 | tail -30
for t in 16 32; do
  echo "== ingest host bench, LAMD_INGEST_THREADS=$t"
  LAMD_INGEST_THREADS=$t LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 2>&1 | grep -E "sub-batch|host logic" | tail -6
done 2>&1 | tee gpurun_out/r4c_ingest_host.txt
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r4c_pytest.log
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err; echo "bench.py wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms (%d steps) | roofline %s frac %.3f launch %.3f ms sum/step %.3f <= %.3f | iso %.3f ms | pipeline %.3f | traffic x%.1f" % (
    d["value"] / 1e6, d["ms_per_step"], d["steps"], r["mode"], r["frac"], r["avg_launch_ms"], r["sum_of_launch_ms_per_step"], r["ms_per_step"], r["isolated"]["launch_ms"], r["pipeline"]["frac"], r["traffic_over_algorithmic"] or 0))
h = d["value_host_to_host"]
print("chained %.1f warm %.1f h2h %.1f ratio %.3f" % (r["verifies_per_s"] / 1e6, d["warm_cache"]["value"] / 1e6, h["value"] / 1e6, h["ratio_to_value"]), {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k != "note" and k != "shape"})
print("cfg4", round(o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, 1), "cfg5", round(o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6, 1))
print("cpu", d["cpu_baseline"]["kind"], round(d["cpu_baseline"]["value"]), "parity", d["parity"]["mismatches"])
PY
