#!/bin/bash
# kernel trace of the collective path on one rank (what RCCL launches, where the lanes stall)
set -u
mkdir -p gpurun_out
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && LAMD_BENCH_GATHER=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/gt -- python $R/bench.py --gpus 1 --roofline-only --steps 6 --warmup 2 > $R/gpurun_out/r11_gather_trace.json 2> $R/gpurun_out/r11.err )
echo "rc=$?"
for f in $(find gpurun_out/gt -name "*kernel_trace.csv"); do gzip -c $f > gpurun_out/r11_gather_kernel_trace.csv.gz; done
for f in $(find gpurun_out/gt -name "*memory_copy_trace.csv"); do gzip -c $f > gpurun_out/r11_gather_memcpy_trace.csv.gz; done
rm -rf gpurun_out/gt
tail -2 gpurun_out/r11.err
