#!/bin/bash
# round 4, GPU session j: multiplier schedule v2 (column 16 not cut, h8 kept as a 64-bit addend), rows grouped by key on by default, pairs-first off:
# micro-benchmark, the FULL GPU suite, A/B of the grouping, the full bench line, rocprofv3 passes of the roofline loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for v in new new2 new new2; do echo "== fe_bench $v"; timeout 120 tools/variants/fe_bench_$v; done 2>&1 | tee gpurun_out/r4j_fe_bench.txt
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4j_pytest.log; echo "pytest wall $(( $(date +%s) - S )) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for v in 0 1 0 1; do
  k=$((k+1))
  LAMD_GROUP=$v timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4j_ab_$k.json 2> gpurun_out/r4j_ab_$k.err || tail -3 gpurun_out/r4j_ab_$k.err
  line gpurun_out/r4j_ab_$k.json "LAMD_GROUP=$v"
done | tee gpurun_out/r4j_ab.txt
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r4j_bench.json 2> gpurun_out/r4j_bench.err; echo "bench.py wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r4j_bench.err
PMC_STEPS=6 bash tools/pmc_run.sh r4j > gpurun_out/pmc_r4j.log 2>&1; python tools/pmc_summary.py gpurun_out/pmc_r4j gpurun_out/r4j 2>&1 | tail -3
