#!/bin/bash
# round 6, session d: the GPU suite (incl. the streaming service test and the sighash-gate case), smoke, the bench lines for profiles/, one --extras run
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6d
export GPU_MAX_HW_QUEUES=16
S=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert|served streaming" | tail -8 | tee gpurun_out/r6d/pytest.txt; echo "pytest wall $(( $(date +%s) - S )) s"
cp gpurun_out/served_stream.json gpurun_out/r6d/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6d/bench20.json 2> gpurun_out/r6d/bench20.err; cp bench_details.json gpurun_out/r6d/details20.json
( time python bench.py ) > gpurun_out/r6d/bench250.json 2> gpurun_out/r6d/bench250.err; cp bench_details.json gpurun_out/r6d/details250.json
( time python bench.py --extras ) > gpurun_out/r6d/bench_extras.json 2> gpurun_out/r6d/bench_extras.err; cp bench_details.json gpurun_out/r6d/details_extras.json
grep real gpurun_out/r6d/*.err
cat gpurun_out/r6d/bench20.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6d/details_extras.json"))
print({k: (round(v["p50_ms"], 3) if "p50_ms" in v else v) for k, v in d.get("latency", {}).items() if isinstance(v, dict)})
print("warm", d.get("warm_cache", {}).get("value"), "h2h", d["config"].get("value_host_to_host"), d["config"].get("predicted_speedup_8"))
oc = d.get("other_configs_1gpu", {})
print({k: (v.get("verifies_per_s") or v.get("messages_per_s_overall") or v.get("recoveries_per_s")) for k, v in oc.items() if isinstance(v, dict)})
print(d["phase_seconds"])
PY
