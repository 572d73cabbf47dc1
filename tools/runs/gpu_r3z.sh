#!/bin/bash
# round 3, GPU session z: lane timeline of the three timed loops (cold, cold chained, warm) with the tool that follows the bench's loop order
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3z_steady -- python $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --skip-extra > $R/gpurun_out/r3z_steady.json 2> $R/gpurun_out/r3z_steady.err )
python tools/lane_timeline.py $(find gpurun_out/r3z_steady -name "*kernel_trace.csv" | head -1) 6 2 > gpurun_out/r03_lane_timeline.txt 2>&1; cat gpurun_out/r03_lane_timeline.txt
rm -rf gpurun_out/r3z_steady
