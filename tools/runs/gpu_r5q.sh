#!/bin/bash
# round 5, GPU session q: the BOLT #12 hashing kernel before (tools/variants/r5q_base.so: every shs_update / shs_final site inlined, k_bolt12_hash 61 k instructions)
# and after (the streaming SHA-256 is a call in device code: 5 k instructions); the BOLT #12 GPU tests on the new build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
run() {  # tag, library
  LAMD_LIB_PATH=$2 timeout 300 python tools/bolt12_probe.py 2>&1 | grep bolt12 | sed "s/^/$1: /" | tee -a gpurun_out/r5q_bolt12.txt
  (cd /tmp && export TMPDIR=/tmp && LAMD_LIB_PATH=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5q_trace_$1 -- python $R/tools/bolt12_probe.py > /dev/null 2>&1)
  grep -h "k_bolt12_hash" $(find gpurun_out/r5q_trace_$1 -name "*kernel_stats.csv") | cut -c1-200 | sed "s/^/$1: /" | tee -a gpurun_out/r5q_bolt12.txt
  rm -rf gpurun_out/r5q_trace_$1
}
rm -f gpurun_out/r5q_bolt12.txt
run inlined $R/tools/variants/r5q_base.so
run calls $R/lightning_amd/liblightning_amd.so
timeout 600 python -m pytest tests -m gpu -q -x -k "bolt12" 2>&1 | tail -3 | tee -a gpurun_out/r5q_bolt12.txt
