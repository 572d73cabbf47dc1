#!/bin/bash
# collective path with per-call dependencies (one gather per batch kind, issued one step late; buffer reuse by event)
set -u
mkdir -p gpurun_out
for rep in 1 2 3; do
LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --roofline-only > gpurun_out/r12_gather_$rep.json 2> gpurun_out/r12_gather_$rep.err
timeout 200 python bench.py --roofline-only > gpurun_out/r12_plain_$rep.json 2> /dev/null
python - <<PY
import json
for f in ("gather", "plain"):
    try:
        d = json.loads(open("gpurun_out/r12_%s_$rep.json" % f).read().strip().splitlines()[-1])
        print("%s: %.1f M/s (launch %.2f ms, mism %d)" % (f, d["value"] / 1e6, d["roofline"]["avg_launch_ms_both_kinds"], d["parity"]["mismatches"]))
    except Exception as e:
        print(f, "failed", repr(e))
PY
done
tail -3 gpurun_out/r12_gather_1.err
