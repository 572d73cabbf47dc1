#!/bin/bash
# round 6, session b: (1) XYZZ G run against the Jacobian run of rounds 1-5, alternating; (2) the clock experiment: the dominant kernel
# with the G windows read from a 64 MiB slice of the table (ghot: same instructions, cache-resident) and with 16-bit windows (g16: 64 MiB table, 16 additions)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6b
export GPU_MAX_HW_QUEUES=16
one() {  # label lib
  unset LAMD_LIB_PATH; [ "$2" != base ] && export LAMD_LIB_PATH=$R/tools/variants/liblightning_amd_$2.so
  timeout 300 python bench.py --ab --no-parity --cpu-sample 0 --details gpurun_out/r6b/$1.json > gpurun_out/r6b/$1.line 2> gpurun_out/r6b/$1.err
  python - "$1" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6b/%s.json" % sys.argv[1])); r = d["roofline"]
print("%-8s cold %.1f M/s step %.3f ms | chained launch %.3f ms (ecdsa %.3f schnorr %.3f) frac %.3f | isolated %.3f ms | peak %.2f T | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["avg_launch_ms_ecdsa"], r["avg_launch_ms_schnorr"], r["frac"], r["isolated"]["launch_ms"], r["peak"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2 3; do one xyzz$rep base; one jac$rep jac; done 2>&1 | tee gpurun_out/r6b/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in base ghot g16; do
  unset LAMD_LIB_PATH LAMD_TESTGEN_LIB_PATH; [ $v != base ] && export LAMD_LIB_PATH=$R/tools/variants/liblightning_amd_$v.so; [ $v = g16 ] && export LAMD_TESTGEN_LIB_PATH=$R/tools/variants/liblightning_amd_testgen_g16.so
  CMD="python $R/bench.py --roofline-only --steps 6 --warmup 2 --no-parity --cpu-sample 0 --details $R/gpurun_out/r6b/clk_$v.json"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6b/${v}_trace -- $CMD > /dev/null 2> $R/gpurun_out/r6b/${v}_trace.err
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r6b/${v}_pmc -- $CMD > /dev/null 2> $R/gpurun_out/r6b/${v}_pmc.err
done
cd $R
python tools/clock_probe_summary.py gpurun_out/r6b base ghot g16 | tee gpurun_out/r6b/clock.txt
find gpurun_out/r6b -name "*.csv" -size +4M -delete; du -sh gpurun_out/r6b
