#!/bin/bash
# round 6, session h: ONE call on an idle chip with the key-table cache OFF (the engine a multi-rank bench run has): 1/8 shards of configs[3], launch by launch
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6h
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
LAMD_CACHE=0 PROBE_REPS=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6h/trace -- python $R/tools/call_trace_probe.py gossip > $R/gpurun_out/r6h/probe_cold.txt 2> $R/gpurun_out/r6h/probe_cold.err
cd $R
cat gpurun_out/r6h/probe_cold.txt
F=$(find gpurun_out/r6h/trace -name "*_kernel_trace.csv" | head -1)
python tools/trace_calls.py $F -v 1 2 > gpurun_out/r6h/timeline_cold.txt 2>&1
head -120 gpurun_out/r6h/timeline_cold.txt
gzip -9 $F; find gpurun_out/r6h -name "*.csv" -delete
