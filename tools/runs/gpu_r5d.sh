#!/bin/bash
# round 5, GPU session d: copy-stream count A/B on the resident / host->host cold loops, lamd_multi staging A/B, the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
: > gpurun_out/r5d_copy_streams.txt
for CS in 1 2 3 2 1; do
  echo "LAMD_COPY_STREAMS=$CS" | tee -a gpurun_out/r5d_copy_streams.txt
  LAMD_COPY_STREAMS=$CS PROBE_STEPS=30 timeout 300 python tools/call_trace_probe.py stream 2>&1 | grep -E "loop" | tee -a gpurun_out/r5d_copy_streams.txt
done
: > gpurun_out/r5d_multi_h2d.txt
timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | sed 's/^/helper on  /' | tee -a gpurun_out/r5d_multi_h2d.txt
LAMD_MULTI_HELPER=0 timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | sed 's/^/helper off /' | tee -a gpurun_out/r5d_multi_h2d.txt
LAMD_MULTI_PINNED=0 timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | sed 's/^/pageable   /' | tee -a gpurun_out/r5d_multi_h2d.txt
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r5d_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5d_bench.json
