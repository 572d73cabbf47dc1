#!/bin/bash
# round 6, session ab: launch-by-launch timeline of ONE 1/8 shard of configs[3] cut per message kind (two asynchronous calls), cache off, idle chip
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ab
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
LAMD_CACHE=0 PROBE_REPS=3 PROBE_SPANS=${PROBE_SPANS:-0} timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6ab/trace -- python $R/tools/call_trace_probe.py gossip8 > $R/gpurun_out/r6ab/probe.txt 2> $R/gpurun_out/r6ab/probe.err
cd $R
cat gpurun_out/r6ab/probe.txt
F=$(find gpurun_out/r6ab/trace -name "*_kernel_trace.csv" | head -1)
python tools/trace_calls.py $F -v 10 > gpurun_out/r6ab/timeline.txt 2>&1
head -75 gpurun_out/r6ab/timeline.txt | cut -c1-150
gzip -9 $F; find gpurun_out/r6ab -name "*.csv" -delete
