#!/bin/bash
# round 6, session ad: the driver's bench command with configs[3] sharded as ONE spans call per rank (the new primary form), and the same through torchrun on one
# rank (the collective path: sharded_configs with nccl)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6t gpurun_out/r6e2
export GPU_MAX_HW_QUEUES=16
( time python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6t/details20.json ) > gpurun_out/r6t/bench20.json 2> gpurun_out/r6t/bench20.err
grep ^real gpurun_out/r6t/bench20.err; wc -c gpurun_out/r6t/bench20.json
LAMD_BENCH_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6e2/details_collective.json > gpurun_out/r6e2/bench_collective.json 2> gpurun_out/r6e2/bench_collective.err; echo "collective rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6t/details20.json"))
c = d["strong_scaling_1gpu"]["cfg4_gossip_replay"]
print(d["value"], d["config"]["predicted_speedup_8"], d["config"]["host_to_host_over_value"], d["phase_seconds"]["total"])
print({k: (round(max(c[k]["shard_ms"]), 2), round(max(c[k]["two_calls_shard_ms"]), 2)) for k in "1248"}, c["t1_ms"], c["mismatches"])
e = json.load(open("gpurun_out/r6e2/details_collective.json"))
def find(o, k):
    if isinstance(o, dict):
        if k in o:
            return o[k]
        for v in o.values():
            r = find(v, k)
            if r is not None:
                return r
print({k: v for k, v in find(e, "cfg4_gossip_replay_sharded").items() if k in ("ms", "two_calls_ms", "one_cut_ms", "mismatches", "split")}, e["value"])
PY
