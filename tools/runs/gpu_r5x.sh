#!/bin/bash
# round 5, GPU session x: the per-signature ladder, complete form (tools/variants/r5q_base.so) against the hot form (the tree's library): all-distinct rows per second and
# k_ecmult<3>'s duration in a kernel trace, one box, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
: > gpurun_out/r5x_cold.txt
for lib in tools/variants/r5q_base.so lightning_amd/liblightning_amd.so tools/variants/r5q_base.so lightning_amd/liblightning_amd.so; do
  LAMD_LIB_PATH=$R/$lib timeout 300 python tools/cold_rows_probe.py 2>&1 | grep -E "all-distinct|Error|error" | sed "s/^/$(basename $lib): /" | tee -a gpurun_out/r5x_cold.txt
done
for lib in tools/variants/r5q_base.so lightning_amd/liblightning_amd.so; do
  (cd /tmp && export TMPDIR=/tmp && LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5x_trace -- python $R/tools/cold_rows_probe.py > /dev/null 2>&1)
  grep -h "k_ecmult<" $(find gpurun_out/r5x_trace -name "*kernel_stats.csv") | sed 's/(unsigned long[^"]*"/"/' | cut -c1-120 | sed "s/^/$(basename $lib): /" | tee -a gpurun_out/r5x_cold.txt
  rm -rf gpurun_out/r5x_trace
done
