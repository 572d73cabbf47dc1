#!/bin/bash
# round 3, GPU session h: tests (incl. the learning of recurring keys in the latency path), lanes 4 / 5 / 6 in the cold loop, the full bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3h_pytest.log
tail -3 gpurun_out/r3h_pytest.log
for l in 4 5 6 4; do
  LAMD_LANES=$l timeout 300 python bench.py --roofline-only > gpurun_out/r3h_l$l.json 2> /dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/r3h_l$l.json').read().strip().splitlines()[-1]); r=d['roofline']
print('LAMD_LANES=$l: cold %.1f M/s, launch in the loop %.3f ms, isolated %.3f ms, mismatches %d' % (d['value']/1e6, r['avg_launch_ms'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
done | tee gpurun_out/r3h_lanes.txt
timeout 900 python bench.py > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err
tail -2 gpurun_out/r3h_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3h_bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f  h2h %.1f (%.2f)" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6, d["value_host_to_host"]["value"] / 1e6, d["value_host_to_host"]["ratio_to_value"]))
r = d["roofline"]
print("roofline frac %.3f isolated %.3f pipeline %.3f rows_in_launch %d launch %.2f iso %.2f" % (r["frac"], r["frac_isolated"], r["pipeline"]["frac"], r["rows_in_launch"], r["avg_launch_ms"], r["isolated"]["launch_ms"]))
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("lat", {k: (round(v.get("p50_ms", v.get("ns_per_call", 0) / 1e6), 3)) for k, v in d["latency"].items() if isinstance(v, dict)})
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in d["other_configs_1gpu"]["gossip_ingest_flood"].items() if k.endswith("_per_s") or k.endswith("overall")})
print("cfg4", round(d["other_configs_1gpu"]["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, 1), "cfg5", round(d["other_configs_1gpu"]["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6, 1))
print("parity", d["parity"]["mismatches"], d["parity"].get("oracle_mismatches"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
