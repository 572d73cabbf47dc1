#!/bin/bash
# hardware-queue count against the cost of the collective path: plain and single-rank RCCL path of `bench.py --roofline-only` at 8/16/24/32/48 queues
set -u
mkdir -p gpurun_out
for q in 16 32 24 48 8 16 32; do
  export GPU_MAX_HW_QUEUES=$q
  timeout 200 python bench.py --roofline-only > gpurun_out/q${q}_plain.json 2> /dev/null
  LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --roofline-only > gpurun_out/q${q}_gather.json 2> gpurun_out/q${q}_gather.err
  python - <<PY
import json
o = []
for f in ("plain", "gather"):
    try:
        d = json.loads(open("gpurun_out/q${q}_%s.json" % f).read().strip().splitlines()[-1])
        o.append("%s %.1f M/s (launch %.2f ms, mism %d)" % (f, d["value"] / 1e6, d["roofline"]["avg_launch_ms_both_kinds"], d["parity"]["mismatches"]))
    except Exception as e:
        o.append("%s failed %r" % (f, e))
print("queues=$q:", "; ".join(o))
PY
done
