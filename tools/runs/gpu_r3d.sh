#!/bin/bash
# round 3, GPU session d: the fused latency path (k_small_verify): tests with it and without it, latency probe both ways, bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3d_pytest.log
tail -3 gpurun_out/r3d_pytest.log
LAMD_SMALL_KERNEL=0 python -m pytest tests/test_gpu_parity.py tests/test_cln_shim.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3d_latency.txt
LAMD_SMALL_KERNEL=0 timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3d_latency.txt
timeout 900 python bench.py > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err
tail -3 gpurun_out/r3d_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3d_bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6))
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("lat", {k: v for k, v in d["latency"].items() if isinstance(v, dict)})
print("parity", d["parity"]["mismatches"], d["parity"].get("oracle_mismatches"))
PY
