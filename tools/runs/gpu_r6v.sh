#!/bin/bash
# round 6, session v: the multiplier's unused carry-out parked in an SGPR pair (tools/gen_fe_asm.py --sgpr-carry: v_mad_u64_u32 ..., s[20:21], ... instead of
# vcc; +1.1 % on the micro-benchmark's mixed addition, profiles/r06_ab_variants.txt 6) as a build of the engine, against the shipped one, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6v
export GPU_MAX_HW_QUEUES=16
V=$R/tools/variants/liblightning_amd_sc.so
LAMD_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -m gpu -q -x -k "not scheduling_variants" 2>&1 | tail -2 | tee gpurun_out/r6v/parity_sc.txt
one() {  # label env...
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6v/$lab.json > gpurun_out/r6v/$lab.line 2> gpurun_out/r6v/$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6v/%s.json" % sys.argv[1])); r = d["roofline"]; k = d["rates"]
print("%-10s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated: tables %.3f ecmult %.3f ms | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], k["kernel_ms_ecdsa_isolated"]["keys_and_tables"], r["isolated"]["launch_ms"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2 3 4; do
  one base$rep LAMD_X=0
  one sc$rep LAMD_LIB_PATH=$V
done 2>&1 | tee gpurun_out/r6v/ab.txt
