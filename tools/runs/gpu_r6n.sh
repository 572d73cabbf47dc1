#!/bin/bash
# round 6, session n: the affine tree builder of the key tables (LAMD_KC_TREE=1: k_kc_tree_both) against the Gray-code chains (k_kc_finish_both):
# parity first (the GPU parity / commitment / stress tests with the knob on), then timing, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6n
export GPU_MAX_HW_QUEUES=16
LAMD_KC_TREE=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_commitment.py tests/test_gpu_stress.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 | tee gpurun_out/r6n/pytest_tree.txt
one() {  # label env...
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6n/$lab.json > gpurun_out/r6n/$lab.line 2> gpurun_out/r6n/$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6n/%s.json" % sys.argv[1])); r = d["roofline"]; k = d["rates"]
print("%-8s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated: tables %.3f ecmult %.3f ms | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], k["kernel_ms_ecdsa_isolated"]["keys_and_tables"], r["isolated"]["launch_ms"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2 3; do one chains$rep LAMD_KC_TREE=0; one tree$rep LAMD_KC_TREE=1; done 2>&1 | tee gpurun_out/r6n/ab.txt
