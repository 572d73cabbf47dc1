#!/bin/bash
# round 4, GPU session n: the flushes' verdict copies on a stream of their own (LAMD_D2H_STREAM): A/B on the streaming probe, streaming tests, then the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for v in 0 1 0 1; do
  echo "== LAMD_D2H_STREAM=$v"; LAMD_D2H_STREAM=$v PROBE_INFLIGHT=8,8,8 PROBE_RESIDENT=100 timeout 300 python tools/stream_probe.py 2>&1 | grep -E "in flight|resident"
done | tee gpurun_out/r4n_d2h_stream.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x -k "stream or queue or flush or poll or stress or small" 2>&1 | grep -E "passed|failed|error" | tail -3
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r4n_bench.json 2> gpurun_out/r4n_bench.err; echo "bench.py wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4n_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4n_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms (%d steps) | roofline %s frac %.3f launch %.3f ms | iso %.3f ms | pipeline %.3f | traffic x%s" % (
    d["value"] / 1e6, d["ms_per_step"], d["steps"], r["mode"], r["frac"], r["avg_launch_ms"], r["isolated"]["launch_ms"], r["pipeline"]["frac"], r["traffic_over_algorithmic"]))
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k != "note" and k != "shape"})
h = d["value_host_to_host"]
print("warm %.1f h2h %.1f ratio %.3f cfg4 %.1f cfg5 %.1f" % (d["warm_cache"]["value"] / 1e6, h["value"] / 1e6, h["ratio_to_value"], o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6),
      {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("cfg5 streaming", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in o["cfg5_commit_storm_streaming"].items() if isinstance(v, dict)}, "one per flush", {k: round(v["batches_per_s"]) for k, v in o["cfg5_commit_storm_one_commitment_per_flush"].items() if isinstance(v, dict)})
PY
