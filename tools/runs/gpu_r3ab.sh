#!/bin/bash
# round 3, GPU session ab: final checks -- full GPU suite, smoke(), the bench line, the collective path on one rank (RCCL all-gathers inside the
# timed region, sharded configs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3ab_pytest.log; tail -3 gpurun_out/r3ab_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r3ab_bench.err; tail -2 gpurun_out/r3ab_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_n1.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f  h2h %.1f (%.2f)" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6, d["value_host_to_host"]["value"] / 1e6, d["value_host_to_host"]["ratio_to_value"]))
r = d["roofline"]
print("roofline frac %.3f isolated %.3f pipeline %.3f chained %.3f (%.2f ms, %.1f M/s) launch %.2f iso %.2f" % (r["frac"], r["frac_isolated"], r["pipeline"]["frac"], r["chained"]["frac"], r["chained"]["avg_launch_ms"], r["chained"]["verifies_per_s"] / 1e6, r["avg_launch_ms"], r["isolated"]["launch_ms"]))
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("lat", {k: (round(v.get("p50_ms", v.get("ns_per_call", 0) / 1e6), 3)) for k, v in d["latency"].items() if isinstance(v, dict)})
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in o["gossip_ingest_flood"].items() if k.endswith("_per_s") or k.endswith("overall")})
print("cfg4", round(o["cfg4_gossip_replay"]["verifies_per_s"] / 1e6, 1), "cfg5", round(o["cfg5_commit_storm_superbatch"]["verifies_per_s"] / 1e6, 1),
      {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in o["cfg5_commit_storm_streaming"].items() if isinstance(v, dict)})
print("per-commitment", o["cfg5_commit_storm_one_commitment_per_flush"])
print("parity", d["parity"]["mismatches"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
LAMD_BENCH_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/r03_bench_n1_collective_path.json 2> gpurun_out/r3ab_gather.err; tail -2 gpurun_out/r3ab_gather.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_n1_collective_path.json").read().strip().splitlines()[-1])
print("collective path on one rank: value %.1f M/s, mismatches %d, sharded %s" % (d["value"] / 1e6, d["parity"]["mismatches"], {k: (round(v["verifies_per_s"] / 1e6, 1), v["mismatches"]) for k, v in (d.get("sharded_configs") or {}).items()}))
PY
