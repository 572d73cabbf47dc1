#!/bin/bash
# round-2 final measurement pass of the HEAD build: bench line, rocprofv3 trace + PMC passes, the collective path on one rank, GPU suite
set -u
mkdir -p gpurun_out
T0=$(date +%s)
timeout 600 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))"; tail -c 300 gpurun_out/bench_r05.err
bash tools/pmc_run.sh r05 > gpurun_out/pmc_r05.log 2>&1
echo "pmc rc=$? t=$(( $(date +%s) - T0 ))"
LAMD_BENCH_GATHER=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 5 --cpu-sample 0 > gpurun_out/bench_r05_gather.json 2> gpurun_out/bench_r05_gather.err
echo "gather rc=$? t=$(( $(date +%s) - T0 ))"; tail -c 300 gpurun_out/bench_r05_gather.err
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r05.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s) - T0 ))"; tail -3 gpurun_out/pytest_gpu_r05.log
