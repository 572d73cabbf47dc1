#!/bin/bash
# round 6, session l: the spans form of the gossip entry point -- parity test, the one-GPU strong-scaling sweep with ONE call per shard, the sharded
# configs through the collective path on one rank (nccl)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6l
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gossip" 2>&1 | tail -3 | tee gpurun_out/r6l/pytest_gossip.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r6l/bench20.json 2> gpurun_out/r6l/bench20.err; cp bench_details.json gpurun_out/r6l/details20.json
grep real gpurun_out/r6l/bench20.err
python - <<'PY' | tee gpurun_out/r6l/sweep.txt
import json
d = json.load(open("gpurun_out/r6l/details20.json")); s = d["strong_scaling_1gpu"]["cfg4_gossip_replay"]
print("value %.1f M/s; predicted_speedup_8 %s" % (d["value"] / 1e6, d["config"]["predicted_speedup_8"]))
for W in "1248":
    print("W=%s kinds, one spans call: slowest %.2f ms %s | two range calls: %s | one cut: slowest %.2f ms %s" % (
        W, s[W]["slowest_ms"], [round(x, 2) for x in s[W]["shard_ms"]], [round(x, 2) for x in s[W]["two_calls_shard_ms"]], s["one_cut"][W]["slowest_ms"], [round(x, 2) for x in s["one_cut"][W]["shard_ms"]]))
print("t1 %.2f  kinds x%.2f  one cut x%.2f  mismatches %d" % (s["t1_ms"], s["predicted_speedup_8"], s["one_cut"]["predicted_speedup_8"], s["mismatches"]))
PY
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6l/details_collective.json > gpurun_out/r6l/bench_collective.json 2> gpurun_out/r6l/bench_collective.err; echo "collective bench rc=$?"
python - <<'PY' | tee -a gpurun_out/r6l/sweep.txt
import json
d = json.load(open("gpurun_out/r6l/details_collective.json"))
print({k: {kk: vv for kk, vv in v.items() if kk in ("ms", "one_cut_ms", "verifies_per_s", "mismatches", "ranks")} for k, v in d["sharded_configs"].items()}, d["parity"]["mismatches"])
PY
