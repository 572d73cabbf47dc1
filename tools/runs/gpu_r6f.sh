#!/bin/bash
# round 6, session f: run-time knobs against the XYZZ kernel, alternating with the defaults on one box (`bench.py --ab`: cold loop, chained loop, isolated)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6f
export GPU_MAX_HW_QUEUES=16
one() {  # label env...
  local lab=$1; shift
  env "$@" timeout 300 python bench.py --ab --cpu-sample 0 --details gpurun_out/r6f/$lab.json > gpurun_out/r6f/$lab.line 2> gpurun_out/r6f/$lab.err
  python - "$lab" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6f/%s.json" % sys.argv[1])); r = d["roofline"]
print("%-12s cold %.1f M/s step %.3f ms | chained launch %.3f ms frac %.3f | isolated %.3f ms | peak %.2f T | mismatches %d" % (
    sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["isolated"]["launch_ms"], r["peak"], d["parity"]["mismatches"]))
PY
}
for rep in 1 2; do
  one base$rep LAMD_X=0
  one waves4_$rep LAMD_KEYED_WAVES=4
  one lanes5_$rep LAMD_LANES=5
  one lanes7_$rep LAMD_LANES=7
  one lanes8_$rep LAMD_LANES=8
done 2>&1 | tee gpurun_out/r6f/knobs.txt
