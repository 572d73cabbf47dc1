#!/bin/bash
# collective path (one rank through RCCL): all-gather issued right away / one step late, two gathers / one fused gather; plain run for reference
set -u
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --roofline-only > gpurun_out/r09_$name.json 2> gpurun_out/r09_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r09_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M/s (launch %.2f ms, mism %d)" % (d["value"] / 1e6, d["roofline"]["avg_launch_ms_both_kinds"], d["parity"]["mismatches"]))
except Exception as e:
    print("$name failed", repr(e))
PY
}
timeout 200 python bench.py --roofline-only > gpurun_out/r09_plain.json 2> /dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r09_plain.json').read().strip().splitlines()[-1]); print('plain: %.1f M/s mism %d' % (d['value']/1e6, d['parity']['mismatches']))"
for rep in 1 2; do
run now_two LAMD_BENCH_LATE_GATHER=0 LAMD_BENCH_FUSED_GATHER=0
run late_two LAMD_BENCH_LATE_GATHER=1 LAMD_BENCH_FUSED_GATHER=0
run now_fused LAMD_BENCH_LATE_GATHER=0 LAMD_BENCH_FUSED_GATHER=1
run late_fused LAMD_BENCH_LATE_GATHER=1 LAMD_BENCH_FUSED_GATHER=1
done
run late_fused_lanes3 LAMD_LANES=3
run late_fused_q12 GPU_MAX_HW_QUEUES=12
run late_fused_q20 GPU_MAX_HW_QUEUES=20
