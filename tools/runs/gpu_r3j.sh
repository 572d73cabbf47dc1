#!/bin/bash
# round 3, GPU session j: rocprofv3 kernel + memory-copy timeline of the host->host streaming loop (LAMD_CACHE=0, four flushes in flight)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small or learn or veneers or degenerate" 2>&1 | tail -3
( cd /tmp && export TMPDIR=/tmp && LAMD_CACHE=0 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r3j_tl -- python $R/tools/host_path_trace.py 4 > $R/gpurun_out/r3j_tl.txt 2>&1 )
tail -2 gpurun_out/r3j_tl.txt
for f in $(find gpurun_out/r3j_tl -name "*kernel_trace.csv" -o -name "*memory_copy_trace.csv"); do gzip -c $f > gpurun_out/r3j_$(basename $f).gz; done
rm -rf gpurun_out/r3j_tl
ls -la gpurun_out/r3j_*
