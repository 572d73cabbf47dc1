#!/bin/bash
# round 3, GPU session y: the profile pass on the build with the fused front end / copy stream -- full GPU suite, rocprofv3 passes behind the
# roofline block (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ sets), kernel trace of bench --roofline-only, lane timeline, soak, full bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3y_pytest.log
tail -2 gpurun_out/r3y_pytest.log
bash tools/pmc_run.sh r03 > gpurun_out/r3y_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r03 gpurun_out/r03 > gpurun_out/r3y_pmc_summary.log 2>&1; tail -3 gpurun_out/r3y_pmc_summary.log; cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3y_rf -- python $R/bench.py --roofline-only > $R/gpurun_out/r03_roofline_only_bench_under_rocprof.json 2> $R/gpurun_out/r3y_rf.err )
cp $(find gpurun_out/r3y_rf -name "*kernel_stats.csv" | head -1) gpurun_out/r03_roofline_only_kernel_stats.csv
gzip -c $(find gpurun_out/r3y_rf -name "*kernel_trace.csv" | head -1) > gpurun_out/r03_roofline_only_kernel_trace.csv.gz
head -4 gpurun_out/r03_roofline_only_kernel_stats.csv | cut -c1-60,250-400
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3y_steady -- python $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --skip-extra > $R/gpurun_out/r3y_steady.json 2> $R/gpurun_out/r3y_steady.err )
python tools/lane_timeline.py $(find gpurun_out/r3y_steady -name "*kernel_trace.csv" | head -1) 6 2 > gpurun_out/r03_lane_timeline.txt 2>&1; head -50 gpurun_out/r03_lane_timeline.txt
timeout 600 python tests/soak_10m.py > gpurun_out/r03_soak_10M.json 2> gpurun_out/r3y_soak.err; tail -c 300 gpurun_out/r03_soak_10M.json
rm -rf gpurun_out/r3y_rf gpurun_out/r3y_steady gpurun_out/pmc_r03/*/runc 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r3y_bench.err
tail -2 gpurun_out/r3y_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_n1.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f  h2h %.1f (%.2f)" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6, d["value_host_to_host"]["value"] / 1e6, d["value_host_to_host"]["ratio_to_value"]))
r = d["roofline"]
print("roofline frac %.3f isolated %.3f pipeline %.3f rows_in_launch %d launch %.2f iso %.2f" % (r["frac"], r["frac_isolated"], r["pipeline"]["frac"], r["rows_in_launch"], r["avg_launch_ms"], r["isolated"]["launch_ms"]))
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("chained", r["chained"]["frac"], r["chained"]["avg_launch_ms"], r["chained"]["verifies_per_s"] / 1e6)
print("lat", {k: (round(v.get("p50_ms", v.get("ns_per_call", 0) / 1e6), 3)) for k, v in d["latency"].items() if isinstance(v, dict)})
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in d["other_configs_1gpu"]["gossip_ingest_flood"].items() if k.endswith("_per_s") or k.endswith("overall")})
print("cfg4", round(d["other_configs_1gpu"]["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, 1), "cfg5", round(d["other_configs_1gpu"]["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6, 1))
print("parity", d["parity"]["mismatches"], d["parity"].get("oracle_mismatches"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
du -sh gpurun_out
