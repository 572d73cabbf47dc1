#!/bin/bash
# roofline cross-check: rocprofv3 --kernel-trace --stats of `bench.py --roofline-only` (every k_ecmult_keyed launch covers a full batch) beside the
# bench's own HIP-event average; the collective path on one rank at the default 20 steps
set -u
mkdir -p gpurun_out
R=$(pwd)
timeout 300 python bench.py --roofline-only > gpurun_out/bench_r07_roofline_only.json 2> gpurun_out/bench_r07_roofline_only.err
echo "plain rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rf -- python $R/bench.py --roofline-only > $R/gpurun_out/bench_r07_roofline_only_rocprof.json 2> $R/gpurun_out/rf.err )
echo "rocprof rc=$?"
for f in $(find gpurun_out/rf -name "*kernel_stats.csv"); do cp $f gpurun_out/r07_roofline_only_kernel_stats.csv; done
for f in $(find gpurun_out/rf -name "*kernel_trace.csv"); do gzip -c $f > gpurun_out/r07_roofline_only_kernel_trace.csv.gz; done
rm -rf gpurun_out/rf
LAMD_BENCH_GATHER=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --cpu-sample 0 --skip-extra > gpurun_out/bench_r07_gather20.json 2> gpurun_out/bench_r07_gather20.err
echo "gather rc=$?"
timeout 300 python bench.py --cpu-sample 0 --skip-extra > gpurun_out/bench_r07_plain20.json 2> /dev/null
python - <<'PY'
import json
for f in ("bench_r07_roofline_only", "bench_r07_roofline_only_rocprof", "bench_r07_gather20", "bench_r07_plain20"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print("%-36s value %.1f M/s  avg_launch %.3f ms (schnorr %.3f, both %.3f) frac %.3f  isolated %.3f ms  mism %d" % (
            f, d["value"] / 1e6, r["avg_launch_ms"], r["avg_launch_ms_schnorr"], r["avg_launch_ms_both_kinds"], r["frac"], r["isolated"]["launch_ms"], d["parity"]["mismatches"]))
    except Exception as e:
        print(f, "failed", e)
PY
grep "k_ecmult_keyed<false" gpurun_out/r07_roofline_only_kernel_stats.csv | sed 's/(.*)"//' | cut -c1-200
