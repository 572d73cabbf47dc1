#!/bin/bash
# round 5, GPU session f: the service under 8 client processes (real engine); the collective path on ONE rank (RCCL all-gather inside the timed region)
# against the plain loop for 6 / 5 / 4 lanes; the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 600 python -X faulthandler -m pytest tests/test_served.py -m gpu -v -x > gpurun_out/r5f_test_served.log 2>&1; echo "test_served rc=$?"
grep -E "passed|failed|error|PASSED|FAILED|Fatal|fault|Abort|assert" gpurun_out/r5f_test_served.log | head -12
: > gpurun_out/r5f_copy_one.txt
for CO in 1 0 1 0; do
  echo "LAMD_COPY_ONE=$CO" | tee -a gpurun_out/r5f_copy_one.txt
  LAMD_COPY_ONE=$CO PROBE_STEPS=30 timeout 300 python tools/call_trace_probe.py stream 2>&1 | grep -E "loop" | tee -a gpurun_out/r5f_copy_one.txt
done
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5f_trace_stream -- python $R/tools/call_trace_probe.py stream 2>&1 | grep -E "loop" | tee $R/gpurun_out/r5f_probe_stream.txt)
find gpurun_out/r5f_trace_stream -name "*.csv" | xargs gzip -9
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms) mismatches %d' % (d['value']/1e6, d['ms_per_step'], d['parity']['mismatches']))"
}
: > gpurun_out/r5f_collective.txt
k=0
for CFG in "plain 6" "gather 6" "gather 5" "gather 4" "plain 6" "gather 6"; do
  set -- $CFG
  k=$((k+1))
  if [ $1 = plain ]; then
    LAMD_LANES=$2 timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r5f_c$k.json 2> gpurun_out/r5f_c$k.err || tail -3 gpurun_out/r5f_c$k.err
  else
    LAMD_LANES=$2 LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + k)) bench.py --gpus 1 --ab --steps 100 --warmup 5 > gpurun_out/r5f_c$k.json 2> gpurun_out/r5f_c$k.err || tail -3 gpurun_out/r5f_c$k.err
  fi
  line gpurun_out/r5f_c$k.json "$1 lanes=$2" | tee -a gpurun_out/r5f_collective.txt
done
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r5f_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5f_bench.json
