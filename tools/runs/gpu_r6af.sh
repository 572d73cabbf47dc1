#!/bin/bash
# round 6, session af: launch-by-launch timeline of the resident cold loop (the headline's loop) under the new defaults
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6af
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
PROBE_STEPS=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6af/trace -- python $R/tools/call_trace_probe.py stream > $R/gpurun_out/r6af/probe.txt 2> $R/gpurun_out/r6af/probe.err
cd $R
cat gpurun_out/r6af/probe.txt | cut -c1-200
F=$(find gpurun_out/r6af/trace -name "*_kernel_trace.csv" | head -1)
python tools/trace_calls.py $F -v 5 > gpurun_out/r6af/timeline.txt 2>&1
head -30 gpurun_out/r6af/timeline.txt | cut -c1-130
gzip -9 $F; find gpurun_out/r6af -name "*.csv" -size +1M -delete
