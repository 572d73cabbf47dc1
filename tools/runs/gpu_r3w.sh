#!/bin/bash
# round 3, GPU session w: the reordered bench (cold engine alone while `value` is measured) in its short form; the ingest flood with its phase clock
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err; tail -3 gpurun_out/r3w_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3w_bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f  h2h %.1f (%.2f)" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6, d["value_host_to_host"]["value"] / 1e6, d["value_host_to_host"]["ratio_to_value"]))
r = d["roofline"]
print("roofline frac %.3f isolated %.3f pipeline %.3f chained %.3f / %.1f M/s" % (r["frac"], r["frac_isolated"], r["pipeline"]["frac"], r["chained"]["frac"], r["chained"]["verifies_per_s"] / 1e6))
print("lat", {k: (round(v.get("p50_ms", v.get("ns_per_call", 0) / 1e6), 3)) for k, v in d["latency"].items() if isinstance(v, dict)})
print("parity", d["parity"]["mismatches"])
PY
timeout 300 python tools/ingest_gpu_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | tee gpurun_out/r3w_ingest.txt
