#!/bin/bash
# round 3, GPU session n: hardware-queue count and lanes with the fused front end + two streams per lane; kernel trace of the cold loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; k=d['rates']['kernel_ms_ecdsa_isolated']
print('$2: cold %.1f M/s, step %.2f ms, launch in the loop %.3f ms, isolated %.3f ms, isolated front %.2f tables %.2f, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['isolated']['launch_ms'], k['prep'], k['keys_and_tables'], d['parity']['mismatches']))"
}
for q in 16 8 12 16 10; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --roofline-only > gpurun_out/r3n_q$q.json 2> gpurun_out/r3n_q$q.err
  line gpurun_out/r3n_q$q.json "GPU_MAX_HW_QUEUES=$q"
done | tee gpurun_out/r3n_queues.txt
for l in 3 4 5; do
  LAMD_LANES=$l timeout 300 python bench.py --roofline-only > gpurun_out/r3n_l$l.json 2> gpurun_out/r3n_l$l.err
  line gpurun_out/r3n_l$l.json "LAMD_LANES=$l"
done | tee -a gpurun_out/r3n_queues.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3n_rf -- python $R/bench.py --roofline-only > $R/gpurun_out/r3n_roofline_only.json 2> $R/gpurun_out/r3n_rf.err )
cp $(find gpurun_out/r3n_rf -name "*kernel_stats.csv" | head -1) gpurun_out/r3n_kernel_stats.csv
gzip -c $(find gpurun_out/r3n_rf -name "*kernel_trace.csv" | head -1) > gpurun_out/r3n_kernel_trace.csv.gz
rm -rf gpurun_out/r3n_rf
head -12 gpurun_out/r3n_kernel_stats.csv | cut -c1-60,200-330
