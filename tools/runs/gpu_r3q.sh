#!/bin/bash
# round 3, GPU session q: the scheduling-variant parity test; latency of small calls on this build against the build of commit 4cd113f
# (tools/variants/r3_prev, a checkout of that commit built in place) on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "scheduling_variants or streaming" 2>&1 | tail -4
for rep in 1 2; do
  echo "== this build"; python tools/latency_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | grep "1 rows\|64 rows"
  echo "== commit 4cd113f"; ( cd tools/variants/r3_prev && python tools/latency_probe.py 2>&1 | grep -v "^W\|amdgpu.ids" | grep "1 rows\|64 rows" )
done > gpurun_out/r3q_latency_ab.txt 2>&1
cat gpurun_out/r3q_latency_ab.txt
