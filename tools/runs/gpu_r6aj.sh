#!/bin/bash
# round 6, session aj: the streaming service with every device's engine thread on its GPU's NUMA node (default) against --no-numa, N alternating pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6aj
export GPU_MAX_HW_QUEUES=16
for rep in $(seq 1 ${1:-6}); do
  for mode in numa nonuma; do
    args=""; [ $mode = nonuma ] && args="--no-numa"
    LAMD_SERVED_TEST_ARGS="$args" timeout 600 python -m pytest tests/test_served.py -m gpu -q -x -k stream -s 2>&1 | grep -E "served streaming" > gpurun_out/r6aj/one.txt
    python - $mode <<'PY'
import ast, sys
d = ast.literal_eval(open("gpurun_out/r6aj/one.txt").read().split("served streaming:", 1)[1].strip())
print("%-8s in-process %.1f ms (in place %.1f)  served %.1f ms  ratio %.2f  engine flushes %d" % (sys.argv[1], d["in_process_s"] * 1e3, d["in_process_in_place_s"] * 1e3, d["served_s"] * 1e3, d["served_over_in_process"], d["engine_flushes"]))
PY
  done
done 2>&1 | tee gpurun_out/r6aj/served_ab.txt
