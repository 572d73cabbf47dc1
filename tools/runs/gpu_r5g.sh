#!/bin/bash
# round 5, GPU session g: mul32 peak with its shader clock; collective path with one event per call against seven; lamd_multi from pageable / page-locked caller
# memory; rocprofv3 stats + PMC passes of the roofline loop (profiles/r05_*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
python tools/mul32_peak_probe.py 2>&1 | grep waves | tee gpurun_out/r5g_mul32_peak.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r5g_peak_pmc -- python $R/tools/mul32_peak_probe.py > /dev/null 2>&1)
python tools/mul32_peak_clock.py gpurun_out/r5g_peak_pmc 2>&1 | tee -a gpurun_out/r5g_mul32_peak.txt | tail -32
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2: cold %.1f M/s (step %.3f ms) mismatches %d' % (d['value']/1e6, d['ms_per_step'], d['parity']['mismatches']))"
}
: > gpurun_out/r5g_collective.txt
k=0
for CFG in "plain 0" "gather 0" "gather 1" "gather 0" "gather 1" "plain 0"; do
  set -- $CFG
  k=$((k+1))
  if [ $1 = plain ]; then
    timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r5g_c$k.json 2> gpurun_out/r5g_c$k.err || tail -3 gpurun_out/r5g_c$k.err
  else
    LAMD_BENCH_MARK_ALL=$2 LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + k)) bench.py --gpus 1 --ab --steps 100 --warmup 5 > gpurun_out/r5g_c$k.json 2> gpurun_out/r5g_c$k.err || tail -3 gpurun_out/r5g_c$k.err
  fi
  line gpurun_out/r5g_c$k.json "$1 mark_all=$2" | tee -a gpurun_out/r5g_collective.txt
done
: > gpurun_out/r5g_multi_h2d.txt
timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | sed 's/^/pageable caller (default)   /' | tee -a gpurun_out/r5g_multi_h2d.txt
PROBE_PINNED_CALLER=1 timeout 300 python tools/multi_h2d_probe.py 2>&1 | grep "pinned=" | sed 's/^/page-locked caller          /' | tee -a gpurun_out/r5g_multi_h2d.txt
PMC_STEPS=12 bash tools/pmc_run.sh r05 > gpurun_out/r5g_pmc.log 2>&1; tail -4 gpurun_out/r5g_pmc.log
python tools/pmc_summary.py gpurun_out/pmc_r05 gpurun_out/r05 > gpurun_out/r5g_pmc_summary.log 2>&1; tail -3 gpurun_out/r5g_pmc_summary.log; tail -22 gpurun_out/r05_pmc_summary.txt | cut -c1-200
# keep what the summaries were made from, small: the stats CSVs and the roofline-only bench line; the raw counter CSVs stay on the box
cp gpurun_out/pmc_r05/trace.json gpurun_out/r05_roofline_only_bench_under_rocprof.json 2>/dev/null
cp $(find gpurun_out/pmc_r05/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r05_roofline_only_kernel_stats.csv 2>/dev/null
find gpurun_out/pmc_r05 -name "*.csv" -size +200k -delete
