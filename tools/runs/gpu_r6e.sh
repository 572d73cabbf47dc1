#!/bin/bash
# round 6, session e: the streaming service against one in-process producer at three caps of the merged engine flush; the bench through the collective
# path on one rank (the code a --gpus N run executes, with nccl); the sighash-gate case
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6e
export GPU_MAX_HW_QUEUES=16
for cap in 65536 131072 262144; do
  LAMD_SERVED_TEST_ARGS="--max-flush-rows $cap" timeout 600 python -m pytest tests/test_served.py -m gpu -q -x -k stream -s 2>&1 | grep -E "served streaming|passed|failed" | tee -a gpurun_out/r6e/served_caps.txt
  cp gpurun_out/served_stream.json gpurun_out/r6e/served_stream_$cap.json
done
timeout 600 python -m pytest tests/test_gpu_commitment.py tests/test_gpu_multi.py -m gpu -q -x 2>&1 | tail -2
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29921 bench.py --gpus 1 --details gpurun_out/r6e/details_collective.json > gpurun_out/r6e/bench_collective.json 2> gpurun_out/r6e/bench_collective.err; echo "collective bench rc=$?"
cat gpurun_out/r6e/bench_collective.json | cut -c1-1500
