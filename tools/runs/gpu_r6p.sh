#!/bin/bash
# round 6, session p: the dominant kernel at TWO blocks per CU (LAMD_KEYED_LDS_PAD=65536: a dynamic-LDS request as occupancy limiter) so that a front-end wave
# fits beside two ecmult waves on a SIMD -- alone and with the tree builder -- against the defaults, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6p
export GPU_MAX_HW_QUEUES=16
one() {  # label env...
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6p/$lab.json > gpurun_out/r6p/$lab.line 2> gpurun_out/r6p/$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6p/%s.json" % sys.argv[1])); r = d["roofline"]; k = d["rates"]
print("%-10s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated: tables %.3f ecmult %.3f ms | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], k["kernel_ms_ecdsa_isolated"]["keys_and_tables"], r["isolated"]["launch_ms"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2; do
  one base$rep LAMD_X=0
  one pad64k_$rep LAMD_KEYED_LDS_PAD=65536
  one pad64ktree_$rep LAMD_KEYED_LDS_PAD=65536 LAMD_KC_TREE=1
  one pad48k_$rep LAMD_KEYED_LDS_PAD=49152
done 2>&1 | tee gpurun_out/r6p/ab.txt
