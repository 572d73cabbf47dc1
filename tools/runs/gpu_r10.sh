#!/bin/bash
# collective path without the side stream (late gather: the engine waits on torch's stream itself)
set -u
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 1 --roofline-only > gpurun_out/r10_$name.json 2> gpurun_out/r10_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r10_$name.json").read().strip().splitlines()[-1])
    print("$name: %.1f M/s (launch %.2f ms, mism %d)" % (d["value"] / 1e6, d["roofline"]["avg_launch_ms_both_kinds"], d["parity"]["mismatches"]))
except Exception as e:
    print("$name failed", repr(e))
PY
}
for rep in 1 2; do
run late_fused_noside LAMD_BENCH_LATE_GATHER=1
run now_fused_side LAMD_BENCH_LATE_GATHER=0
run late_noside_q18 LAMD_BENCH_LATE_GATHER=1 GPU_MAX_HW_QUEUES=18
run late_noside_nowatchdog LAMD_BENCH_LATE_GATHER=1 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
done
timeout 200 python bench.py --roofline-only > gpurun_out/r10_plain.json 2> /dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r10_plain.json').read().strip().splitlines()[-1]); print('plain: %.1f M/s mism %d' % (d['value']/1e6, d['parity']['mismatches']))"
