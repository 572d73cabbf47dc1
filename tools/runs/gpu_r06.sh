#!/bin/bash
# G-table window A/B: 22 bits (3 GiB, 12 additions, shipped) against 24 (11.8 GB, 11) and 26 (42.9 GB, 10), alternating runs
set -u
mkdir -p gpurun_out
for v in g22 g24 g26 g22 g26 g24; do
  if [ $v = g22 ]; then unset LAMD_LIB_PATH; else export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_$v.so; fi
  T0=$(date +%s)
  timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  rc=$?
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench_$v.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$v rc=$rc %ds cold %.1f M/s warm %.1f M/s isolated ecmult %.3f ms in-loop %.3f ms mismatches %d lat1 %.3f ms" % (
        $(date +%s) - $T0, d["value"] / 1e6, d["warm_cache"]["value"] / 1e6, r["isolated"]["launch_ms"], r["avg_launch_ms"], d["parity"]["mismatches"],
        d["latency"]["ecdsa65_batch_1"]["p50_ms"]))
except Exception as e:
    print("$v rc=$rc failed:", e)
PY
done
unset LAMD_LIB_PATH
