#!/bin/bash
# round 6, session ah: the lane count again under the new defaults (LAMD_PRIO=13, early cold list): `python bench.py --ab`, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ah
export GPU_MAX_HW_QUEUES=16
one() {  # label env...
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6ah/$lab.json > gpurun_out/r6ah/$lab.line 2> gpurun_out/r6ah/$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6ah/%s.json" % sys.argv[1])); r = d["roofline"]
print("%-10s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated ecmult %.3f ms | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["isolated"]["launch_ms"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2; do
  one base$rep LAMD_X=0
  one lanes4_$rep LAMD_LANES=4
  one lanes5_$rep LAMD_LANES=5
  one lanes8_$rep LAMD_LANES=8
  one prio29_$rep LAMD_PRIO=29
done 2>&1 | tee gpurun_out/r6ah/ab.txt
