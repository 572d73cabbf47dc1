#!/bin/bash
# round 5, GPU session r: the latency schedule of the field multiplier (tools/gen_fe_asm.py --ilp, fe_asm_ilp.inc) against the throughput schedule:
# fe_bench at 1..8 waves per SIMD; then the engine built entirely with the latency schedule (tools/variants/r5r_ilp.so): field fuzz / self-test / goldens,
# and the latency probes (k_small_verify's duration in the kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
for v in r5base r5ilp r5base r5ilp; do echo "== fe_bench $v"; timeout 120 tools/variants/fe_bench_$v; done 2>&1 | tee gpurun_out/r5r_fe_bench.txt
if [ -f tools/variants/r5r_ilp.so ]; then
  LAMD_LIB_PATH=$R/tools/variants/r5r_ilp.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "selftest or fuzz or golden or native or edge" 2>&1 | tail -3 | tee gpurun_out/r5r_tests_ilp.txt
  for lib in lightning_amd/liblightning_amd.so tools/variants/r5r_ilp.so; do
    echo "== $lib" | tee -a gpurun_out/r5r_commit_probe.txt
    LAMD_LIB_PATH=$R/$lib timeout 300 python tools/commit_trace_probe.py 2>&1 | grep -E "sight|cached" | tee -a gpurun_out/r5r_commit_probe.txt
    (cd /tmp && export TMPDIR=/tmp && LAMD_LIB_PATH=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5r_trace -- python $R/tools/commit_trace_probe.py > /dev/null 2>&1)
    grep -h "k_small_verify\|k_txsig_tx_hash" $(find gpurun_out/r5r_trace -name "*kernel_stats.csv") | cut -c1-120 | tee -a gpurun_out/r5r_commit_probe.txt
    rm -rf gpurun_out/r5r_trace
  done
fi
