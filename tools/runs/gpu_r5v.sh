#!/bin/bash
# round 5, GPU session v: lamd_check_commitment_signed with the host's SHA-256 portable / byte-fed (tools/variants/r5q_base.so) against SHA extensions + whole blocks
# (the tree's library), alternating on one box; the kernels' durations from a kernel trace of each
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
: > gpurun_out/r5v_commit_probe.txt
for lib in tools/variants/r5q_base.so lightning_amd/liblightning_amd.so tools/variants/r5q_base.so lightning_amd/liblightning_amd.so; do
  echo "== $lib" | tee -a gpurun_out/r5v_commit_probe.txt
  LAMD_LIB_PATH=$R/$lib timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "cached" | tee -a gpurun_out/r5v_commit_probe.txt
done
for lib in tools/variants/r5q_base.so lightning_amd/liblightning_amd.so; do
  (cd /tmp && export TMPDIR=/tmp && LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5v_trace -- python $R/tools/commit_trace_probe.py 2>&1 | grep cached | sed "s/^/under rocprofv3, $(basename $lib): /" | tee -a $R/gpurun_out/r5v_commit_probe.txt)
  grep -h "k_small_verify\|k_txsig_tx_hash" $(find gpurun_out/r5v_trace -name "*kernel_stats.csv") | cut -c1-110 | tee -a gpurun_out/r5v_commit_probe.txt
  rm -rf gpurun_out/r5v_trace
done
