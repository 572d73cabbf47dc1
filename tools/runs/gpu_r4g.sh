#!/bin/bash
# round 4, GPU session g: the field multiplier with the wrap-around folded BEFORE the low chain (one asm statement per product, tail included):
# (1) micro-benchmark old / new on the real headers, (2) device self-test + fuzz + goldens on the new build, (3) A/B of the two engines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for v in old new old new; do echo "== fe_bench $v"; timeout 120 tools/variants/fe_bench_$v; done 2>&1 | tee gpurun_out/r4g_fe_bench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random or diagnostics or keyed or degenerate" 2>&1 | tail -3 | tee gpurun_out/r4g_pytest_subset.log
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2: cold %.1f M/s (step %.3f ms), chained %.1f M/s launch %.3f ms frac %.3f, isolated %.3f ms, mismatches %d' % (d['value']/1e6, d['ms_per_step'], r['verifies_per_s']/1e6, r['avg_launch_ms'], r['frac'], r['isolated']['launch_ms'], d['parity']['mismatches']))"
}
k=0
for v in oldtail base oldtail base oldtail base; do
  k=$((k+1))
  unset LAMD_LIB_PATH LAMD_TESTGEN_LIB_PATH
  if [ $v != base ]; then export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_$v.so LAMD_TESTGEN_LIB_PATH=$PWD/tools/variants/liblightning_amd_testgen_$v.so; fi
  timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r4g_ab_$k.json 2> gpurun_out/r4g_ab_$k.err || tail -3 gpurun_out/r4g_ab_$k.err
  line gpurun_out/r4g_ab_$k.json "$v"
done | tee gpurun_out/r4g_ab.txt
