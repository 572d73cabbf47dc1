#!/bin/bash
# round 3, GPU session b: tests on the new build (limb tables, cold-list compaction, stage-in kernels), A/B of the build variants,
# streaming queue with / without the stage-in kernels, full bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3b_pytest.log
tail -3 gpurun_out/r3b_pytest.log
bash tools/runs/next_ab_variants.sh run 2>&1 | tee gpurun_out/r3b_ab.txt
for sk in 0 1 0 1; do
  echo "== LAMD_STAGE_KERNEL=$sk"
  LAMD_STAGE_KERNEL=$sk PROBE_INFLIGHT=3,4,4 timeout 300 python tools/stream_probe.py 2>&1 | grep -v "^$"
done | tee gpurun_out/r3b_stream.txt
timeout 600 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3b_bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6))
print("iso", d["rates"]["kernel_ms_ecdsa_isolated"], d["roofline"]["isolated"])
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("lat", {k: v for k, v in d["latency"].items() if isinstance(v, dict)})
print("parity", d["parity"]["mismatches"], d["parity"].get("oracle_mismatches"))
PY
