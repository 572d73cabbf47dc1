#!/bin/bash
# round 6, session x: the in-place flush with its copies in the staged form's order (keys first, the same events): parity, the service A/B again, and the bench's
# configs[4] sweep with the in-place producer beside the copying one
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r6x
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "in_place or reserve or streaming" 2>&1 | tail -2 | tee gpurun_out/r6x/parity.txt
for rep in 1 2; do
  for mode in inplace copy; do
    args=""; [ $mode = copy ] && args="--copy-flushes"
    LAMD_SERVED_TEST_ARGS="$args" timeout 600 python -m pytest tests/test_served.py -m gpu -q -x -k stream -s 2>&1 | grep -E "served streaming|passed|failed|Error|assert" | tee -a gpurun_out/r6x/served_ab.txt
  done
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r6x/details20.json ) > gpurun_out/r6x/bench20.json 2> gpurun_out/r6x/bench20.err
grep ^real gpurun_out/r6x/bench20.err; wc -c gpurun_out/r6x/bench20.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6x/details20.json"))
c = d["strong_scaling_1gpu"]["cfg5_commit_storm_streaming"]
print({k: (round(c[k]["slowest_ms"], 2), c[k].get("predicted_speedup")) for k in "1248"}, "in place:", c.get("in_place_producer"))
print(d["value"], d["config"]["predicted_speedup_8"], d["phase_seconds"]["total"])
PY
