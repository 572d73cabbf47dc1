set -x
mkdir -p gpurun_out
timeout 600 python tools/diag_mismatch.py > gpurun_out/diag_mismatch.txt 2> gpurun_out/diag_mismatch.err; tail -40 gpurun_out/diag_mismatch.txt; tail -3 gpurun_out/diag_mismatch.err
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in 3 4; do
LAMD_CACHE=0 LAMD_KEYED_WAVES=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02c_w$w -- python $R/tools/prof_calls.py > $R/gpurun_out/prof_r02c_w$w.log 2>&1
done
cd $R
tail -12 gpurun_out/prof_r02c_w3.log | cut -c1-200
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r02c.log 2>&1; tail -15 gpurun_out/pytest_gpu_r02c.log
