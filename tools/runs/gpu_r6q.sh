#!/bin/bash
# round 6, session q: flush rows queued IN PLACE (lamd_queue_*_batch_inplace, lamd_host_register): parity of the new engine path, then the streaming
# service with its clients' blocks pinned (default) against --copy-flushes, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6q
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "in_place or reserve or streaming" 2>&1 | tail -3 | tee gpurun_out/r6q/parity.txt
for rep in 1 2; do
  for mode in inplace copy; do
    args=""; [ $mode = copy ] && args="--copy-flushes"
    LAMD_SERVED_TEST_ARGS="$args" timeout 600 python -m pytest tests/test_served.py -m gpu -q -x -k stream -s 2>&1 | grep -E "served streaming|passed|failed|Error|assert" | tee -a gpurun_out/r6q/served_ab.txt
    cp gpurun_out/served_stream.json gpurun_out/r6q/served_stream_${mode}_$rep.json
  done
done
