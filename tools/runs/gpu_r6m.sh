#!/bin/bash
# round 6, session m: the gossip replay split by SIGNER KEY (one spans call per shard) beside the per-kind and one-cut splits, in the one-GPU sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6m
export GPU_MAX_HW_QUEUES=16
LAMD_BENCH_BY_KEY=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --no-h2h --details gpurun_out/r6m/details.json > gpurun_out/r6m/bench.json 2> gpurun_out/r6m/bench.err; echo rc=$?
python - <<'PY' | tee gpurun_out/r6m/sweep.txt
import json
d = json.load(open("gpurun_out/r6m/details.json")); s = d["strong_scaling_1gpu"]["cfg4_gossip_replay"]
for W in "1248":
    print("W=%s per kind (two calls): %s | by signer key (one spans call): %s | one cut: %s" % (
        W, [round(x, 2) for x in s[W]["shard_ms"]], [round(x, 2) for x in s[W].get("by_signer_key_shard_ms", [])], [round(x, 2) for x in s["one_cut"][W]["shard_ms"]]))
print("t1 %.2f mismatches %d" % (s["t1_ms"], s["mismatches"]))
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spans" 2>&1 | tail -2
