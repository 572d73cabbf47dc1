#!/bin/bash
# round-2 late measurement pass: full bench (cfg1 leg, host->host mix), copy-thread A/B, GPU suite
set -u
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err
echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r04.err
LAMD_COPY_THREADS=8 timeout 600 python bench.py --cpu-sample 0 --steps 5 > gpurun_out/bench_r04_ct8.json 2> gpurun_out/bench_r04_ct8.err
echo "bench ct8 rc=$?"
LAMD_COPY_THREADS=2 timeout 600 python bench.py --cpu-sample 0 --steps 5 --skip-extra > gpurun_out/bench_r04_ct2.json 2> gpurun_out/bench_r04_ct2.err
echo "bench ct2 rc=$?"
nproc > gpurun_out/nproc_r04.txt
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r04.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r04.log
