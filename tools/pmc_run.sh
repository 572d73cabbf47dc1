#!/bin/bash
# rocprofv3 passes for the roofline block (run on the GPU box from the repo root):
#   1. kernel trace + stats (CSV)     2. FETCH_SIZE     3. WRITE_SIZE     4. SQ VALU counters
# Counter passes use --kernel-trace only (gpurun refuses --pmc with sys/hip tracing).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${1:-r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --skip-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.json 2> $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.json 2> $OUT/write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.json 2> $OUT/sq.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.json 2> $OUT/sq2.err
find $OUT -name "*.csv" | head -30
du -sh $OUT
