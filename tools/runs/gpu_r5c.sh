#!/bin/bash
# round 5, GPU session c: the mirror tests again (ctypes argtypes fixed), the multi H2D A/B, kernel traces of single shard calls and of the
# resident / host->host cold loops
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 600 python -X faulthandler -m pytest tests/test_cln_shim.py -m gpu -v -x > gpurun_out/r5c_test_cln_shim.log 2>&1; echo "test_cln_shim rc=$?"
grep -E "passed|failed|error|PASSED|FAILED|Fatal|fault|Abort|assert" gpurun_out/r5c_test_cln_shim.log | head -20
timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | tee gpurun_out/r5c_multi_h2d.txt
LAMD_MULTI_PINNED=0 timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | tee -a gpurun_out/r5c_multi_h2d.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5c_trace_gossip -- python $R/tools/call_trace_probe.py gossip 2>&1 | grep -E "shard|whole" | tee $R/gpurun_out/r5c_probe_gossip.txt
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r5c_trace_stream -- python $R/tools/call_trace_probe.py stream 2>&1 | grep -E "loop" | tee $R/gpurun_out/r5c_probe_stream.txt
cd $R
find gpurun_out/r5c_trace_gossip gpurun_out/r5c_trace_stream -name "*.csv" | xargs ls -la
find gpurun_out/r5c_trace_gossip gpurun_out/r5c_trace_stream -name "*.csv" | xargs gzip -9
