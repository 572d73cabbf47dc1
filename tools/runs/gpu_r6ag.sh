#!/bin/bash
# round 6, session ag: one 1/8 shard of configs[4] (tools/call_trace_probe.py storm) against the flush schedule: rows of the first flush, commitments per flush,
# copying or in-place producer.  median and best of 10 repetitions, host wall ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6ag
export GPU_MAX_HW_QUEUES=16
for ip in 0 1; do for ff in 32 64 128; do for g in 192 256 384 512; do
  LAMD_CACHE=0 PROBE_REPS=10 PROBE_INPLACE=$ip PROBE_GROUP=$g LAMD_BENCH_FIRST_FLUSH=$ff timeout 120 python tools/call_trace_probe.py storm 2>&1 | grep "storm shard" > gpurun_out/r6ag/one.txt
  python - $ip $ff $g <<'PY'
import re, sys
t = sorted(float(re.search(r": ([0-9.]+) ms host wall", l).group(1)) for l in open("gpurun_out/r6ag/one.txt"))
print("in_place=%s first=%3s group=%3s  median %.2f  best %.2f ms" % (sys.argv[1], sys.argv[2], sys.argv[3], t[len(t) // 2], t[0]))
PY
done; done; done 2>&1 | tee gpurun_out/r6ag/grid.txt
