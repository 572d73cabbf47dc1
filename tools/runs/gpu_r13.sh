#!/bin/bash
# final pass of the round-2 build: GPU suite, the bench line, the collective path on one rank
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r13.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r13.log
timeout 600 python bench.py > gpurun_out/bench_r13.json 2> gpurun_out/bench_r13.err
echo "bench rc=$?"; tail -c 300 gpurun_out/bench_r13.err
LAMD_BENCH_GATHER=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --cpu-sample 0 > gpurun_out/bench_r13_gather.json 2> gpurun_out/bench_r13_gather.err
echo "gather rc=$?"
python - <<'PY'
import json
for f in ("bench_r13", "bench_r13_gather"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "%.1f M/s" % (d["value"] / 1e6), "warm %.1f" % (d["warm_cache"]["value"] / 1e6), "mism", d["parity"]["mismatches"], "frac", round(d["roofline"]["frac"], 3), "iso frac", round(d["roofline"]["isolated"]["frac"], 3))
    ms = d.get("pcie_inclusive", {}).get("mix_streaming", {})
    for k, v in ms.items():
        if isinstance(v, dict):
            print("   ", k, "%.1f M/s" % (v["verifies_per_s"] / 1e6), "mism", v["mismatches"])
PY
