#!/bin/bash
# rocprofv3 passes for the roofline block (run on the GPU box from the repo root):
#   1. kernel trace + stats (CSV)     2. FETCH_SIZE     3. WRITE_SIZE     4. SQ VALU counters
# Counter passes use --kernel-trace only (gpurun refuses --pmc with sys/hip tracing).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${1:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the roofline loop only (cold engine, chained ecmult launches): every k_ecmult_keyed<false, 3> launch of the process is one of that loop's
CMD="python $R/bench.py --roofline-only --steps ${PMC_STEPS:-6} --warmup 2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.json 2> $OUT/write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.json 2> $OUT/sq.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.json 2> $OUT/sq2.err
find $OUT -name "*.csv" | head -30
du -sh $OUT
# FETCH_SIZE calibration on gathers of known size (the kernel's access pattern: random 64-byte G entries, random 96-byte comb entries)
for E in 64 96; do
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib$E -- $R/tools/gather_calib $E 3 4194304 16 > $OUT/calib$E.txt 2> $OUT/calib$E.err
done
