#!/bin/bash
# round 6, session ae: launch-by-launch timeline of ONE 1/8 shard of configs[4] (1 250 commitments streamed from host memory), cache off
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ae
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
LAMD_CACHE=0 PROBE_REPS=3 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r6ae/trace -- python $R/tools/call_trace_probe.py storm > $R/gpurun_out/r6ae/probe.txt 2> $R/gpurun_out/r6ae/probe.err
cd $R
cat gpurun_out/r6ae/probe.txt
F=$(find gpurun_out/r6ae/trace -name "*_kernel_trace.csv" | head -1)
python tools/trace_calls.py $F -v 8 > gpurun_out/r6ae/timeline.txt 2>&1
head -150 gpurun_out/r6ae/timeline.txt | cut -c1-130
gzip -9 $F; find gpurun_out/r6ae -name "*.csv" -size +1M -delete
