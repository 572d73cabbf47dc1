#!/bin/bash
# round 4, GPU session a: the new tests (link-compatible mirror, libsecp names, lamd_multi_* engine back end on one device, edge-class rows,
# learning filter), the restructured bench line, wave-priority / bulk-stream A/B (bench.py --ab)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_multi.py tests/test_cln_shim.py tests/test_gpu_soak.py -m gpu -x -q -k "not soak_10m" 2>&1 | tail -5
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -x -q -k "learn or small or veneers or stream or scheduling or golden" 2>&1 | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; tail -3 gpurun_out/r4a_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4a_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f M/s step %.3f ms | steady %s | roofline mode %s frac %.3f avg launch %.3f ms (e %.3f s %.3f) sum/step %.3f <= %.3f %s | iso %.3f ms frac %.3f | overlapped %s | pipeline %.3f" % (
    d["value"] / 1e6, d["ms_per_step"], d["steady_state"] and round(d["steady_state"]["value"] / 1e6, 1), r["mode"], r["frac"], r["avg_launch_ms"], r["avg_launch_ms_ecdsa"], r["avg_launch_ms_schnorr"],
    r["sum_of_launch_ms_per_step"], r["ms_per_step"], r["sum_of_launches_le_step"], r["isolated"]["launch_ms"], r["frac_isolated"], r["overlapped"] and round(r["overlapped"]["avg_launch_ms"], 3), r["pipeline"]["frac"]))
print("chained value %.1f" % (r["verifies_per_s"] / 1e6), "warm", round(d["warm_cache"]["value"] / 1e6, 1), "h2h", round(d["value_host_to_host"]["value"] / 1e6, 1), round(d["value_host_to_host"]["ratio_to_value"], 3))
c = d["cpu_baseline"]
print("cpu", c["kind"], c["cores"], round(c["value"]), "legs", {k: (round(v["value"]) if isinstance(v, dict) else v[:20]) for k, v in c["legs"].items()}, "mism", c["gpu_vs_cpu_verdict_mismatches"])
print("lat", {k: (round(v.get("p50_ms", v.get("ns_per_call", 0) / 1e6), 3)) for k, v in d["latency"].items() if isinstance(v, dict)})
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in o["gossip_ingest_flood"].items() if k.endswith("_per_s") or k.endswith("overall")})
print("parity", d["parity"]["mismatches"])
PY
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for cfg in "0 0" "3 0" "15 0" "31 0" "0 1" "15 1" "0 0" "2 0" "1 0" "31 0"; do
  set -- $cfg; k=$((k+1))
  LAMD_PRIO=$1 LAMD_BULK_STREAM=$2 timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4a_ab_$k.json 2> gpurun_out/r4a_ab_$k.err || tail -3 gpurun_out/r4a_ab_$k.err
  line gpurun_out/r4a_ab_$k.json "LAMD_PRIO=$1 LAMD_BULK_STREAM=$2"
done | tee gpurun_out/r4a_prio.txt
