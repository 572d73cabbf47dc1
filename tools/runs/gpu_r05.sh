#!/bin/bash
# round-2 final measurement pass of the HEAD build: bench line, rocprofv3 trace + PMC passes (summarised on the box: the raw
# counter CSVs exceed what gpurun copies back), the collective path on one rank, a longer kernel trace for the lane timeline
set -u
mkdir -p gpurun_out
T0=$(date +%s)
R=$(pwd)
timeout 600 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))"; tail -c 300 gpurun_out/bench_r05.err
bash tools/pmc_run.sh r05 > gpurun_out/pmc_r05.log 2>&1
echo "pmc rc=$? t=$(( $(date +%s) - T0 ))"
python tools/pmc_summary.py gpurun_out/pmc_r05 gpurun_out/r05 > /dev/null 2> gpurun_out/pmc_summary_r05.err
echo "summary rc=$?"; tail -3 gpurun_out/pmc_summary_r05.err
cp profiles/pmc_latest.json gpurun_out/pmc_latest_r05.json
for f in $(find gpurun_out/pmc_r05/trace -name "*kernel_stats.csv" -o -name "*kernel_trace.csv"); do gzip -c $f > gpurun_out/r05_$(basename $f).gz; done
rm -rf gpurun_out/pmc_r05
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace6 -- python $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --skip-extra > $R/gpurun_out/trace6.json 2> $R/gpurun_out/trace6.err )
echo "trace6 rc=$? t=$(( $(date +%s) - T0 ))"
for f in $(find gpurun_out/trace6 -name "*kernel_trace.csv"); do gzip -c $f > gpurun_out/r05_steady_$(basename $f).gz; done
rm -rf gpurun_out/trace6
LAMD_BENCH_GATHER=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 5 --cpu-sample 0 > gpurun_out/bench_r05_gather.json 2> gpurun_out/bench_r05_gather.err
echo "gather rc=$? t=$(( $(date +%s) - T0 ))"
du -sh gpurun_out
