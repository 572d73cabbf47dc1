#!/bin/bash
set -u
mkdir -p gpurun_out
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && PROBE_INFLIGHT=4,4 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/sp -- python $R/tools/stream_probe.py > $R/gpurun_out/r14_probe.txt 2> $R/gpurun_out/r14.err )
echo rc=$?
for f in $(find gpurun_out/sp -name "*kernel_trace.csv"); do gzip -c $f > gpurun_out/r14_stream_kernel_trace.csv.gz; done
for f in $(find gpurun_out/sp -name "*memory_copy_trace.csv"); do gzip -c $f > gpurun_out/r14_stream_memcpy_trace.csv.gz; done
rm -rf gpurun_out/sp
grep -v amdgpu.ids gpurun_out/r14_probe.txt
