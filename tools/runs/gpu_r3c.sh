#!/bin/bash
# round 3, GPU session c: full GPU tests on the build with early reject + cold-list compaction (8-word tables), then the full bench
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3c_pytest.log
tail -3 gpurun_out/r3c_pytest.log
timeout 900 python bench.py > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
tail -3 gpurun_out/r3c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c_bench.json").read().strip().splitlines()[-1])
print("value %.1f M/s  step %.2f ms  warm %.1f" % (d["value"] / 1e6, d["ms_per_step"], d["warm_cache"]["value"] / 1e6))
print("iso", d["rates"]["kernel_ms_ecdsa_isolated"], d["roofline"]["isolated"])
print("mix", {k: round(v["verifies_per_s"] / 1e6, 1) for k, v in d["pcie_inclusive"]["mix_streaming"].items() if isinstance(v, dict)})
print("sweep", {k: (round(v["verifies_per_s"] / 1e6, 1), v["rows_on_ladder"]) for k, v in d["other_configs_1gpu"]["key_reuse_sweep"].items() if isinstance(v, dict)})
print("lat", {k: v for k, v in d["latency"].items() if isinstance(v, dict)})
print("ingest", d["other_configs_1gpu"]["gossip_ingest_flood"])
print("parity", d["parity"]["mismatches"], d["parity"].get("oracle_mismatches"))
PY
