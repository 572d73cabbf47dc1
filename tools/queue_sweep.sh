#!/bin/bash
# Runs GPU tests under different hardware-queue counts (the engine's verdicts must not depend on them).
#   tools/queue_sweep.sh <reps_subset> <reps_full>   -> gpurun_out/queue_sweep.txt
set -u
OUT=gpurun_out/queue_sweep.txt
mkdir -p gpurun_out
: > $OUT
SUBSET="tests/test_gpu_stress.py tests/test_gpu_parity.py::test_cfg4_gossip_replay_small_vs_oracle_and_construction tests/test_gpu_parity.py::test_lanes_back_to_back_calls_without_host_sync tests/test_gpu_parity.py::test_chunk_splitting_small_chunks tests/test_gpu_parity.py::test_streaming_pipelined_flushes"
for q in 4 16 32; do
  pass=0; fail=0
  for i in $(seq 1 ${1:-5}); do
    if GPU_MAX_HW_QUEUES=$q LAMD_STRESS_SEED=$i timeout 600 python -m pytest $SUBSET -x -q -m gpu > gpurun_out/sweep_q${q}_$i.log 2>&1; then pass=$((pass+1)); rm -f gpurun_out/sweep_q${q}_$i.log; else fail=$((fail+1)); fi
  done
  echo "queues=$q subset(stress+gossip+lanes+chunks+flushes) pass=$pass fail=$fail" >> $OUT
  pass=0; fail=0
  for i in $(seq 1 ${2:-2}); do
    if GPU_MAX_HW_QUEUES=$q timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/sweep_full_q${q}_$i.log 2>&1; then pass=$((pass+1)); tail -1 gpurun_out/sweep_full_q${q}_$i.log >> $OUT; rm -f gpurun_out/sweep_full_q${q}_$i.log; else fail=$((fail+1)); fi
  done
  echo "queues=$q full-suite pass=$pass fail=$fail" >> $OUT
done
cat $OUT
