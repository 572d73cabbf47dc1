#!/bin/bash
# round 6, session s: the head of the GPU suite (mirror, ingest, commitments, multi) over and over on one box -- a run of the whole suite died there once
# ("Fatal Python error: Aborted" after 14 tests and 500 s); per-test time limit with a stack dump, so that a repeat names the call that hangs
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6s
export GPU_MAX_HW_QUEUES=16
for i in $(seq 1 ${1:-12}); do
  timeout 400 python -X faulthandler -m pytest tests/test_cln_shim.py tests/test_gossip_ingest.py tests/test_gpu_commitment.py tests/test_gpu_multi.py -m gpu -x -q --timeout 120 > gpurun_out/r6s/loop_$i.txt 2>&1
  rc=$?; echo "loop $i rc=$rc $(tail -1 gpurun_out/r6s/loop_$i.txt)"
  [ $rc -ne 0 ] && { tail -80 gpurun_out/r6s/loop_$i.txt; dmesg 2>/dev/null | tail -20; break; }
done
