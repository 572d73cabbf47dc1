#!/bin/bash
# round 3, GPU session k: the flushes' H2D copies on a copy stream of their own (LAMD_COPY_STREAM=1, new) against the lane's prep stream (=0),
# flushes in flight 4 / 6 / 8 (nine staging sets); the copying form with 4 / 8 copy threads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream or queue or reserve" 2>&1 | tail -3
for cs in 1 0 1 0; do
  echo "== LAMD_COPY_STREAM=$cs"
  LAMD_COPY_STREAM=$cs PROBE_INFLIGHT=4,6,8,4,8 PROBE_RESIDENT=100 timeout 300 python tools/stream_probe.py 2>&1 | grep -v "^W\|amdgpu.ids"
done > gpurun_out/r3k_copy_stream.txt 2>&1
for th in 4 8; do
  echo "== LAMD_COPY_THREADS=$th (copy stream on)"
  LAMD_COPY_THREADS=$th PROBE_INFLIGHT=8 PROBE_COPYING=4,8 PROBE_RESIDENT= timeout 300 python tools/stream_probe.py 2>&1 | grep -v "^W\|amdgpu.ids"
done >> gpurun_out/r3k_copy_stream.txt 2>&1
cat gpurun_out/r3k_copy_stream.txt
