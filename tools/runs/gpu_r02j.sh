mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu_r02j.log 2>&1; tail -4 gpurun_out/pytest_gpu_r02j.log
for cfg in "16 0" "32 32768" "64 16384"; do set -- $cfg
LAMD_PREP_BATCH=$1 LAMD_PREP_MIN_THREADS=$2 timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_r02j_pb$1.json 2>gpurun_out/bench_r02j_pb$1.err || tail -2 gpurun_out/bench_r02j_pb$1.err; done
python - <<'PY'
import json
for b in (16,32,64):
    try:
        d=json.load(open("gpurun_out/bench_r02j_pb%d.json"%b))
        print(b, "value %.1fM"%(d["value"]/1e6), "warm %.1fM"%(d["warm_cache"]["value"]/1e6), "pcie %.1fM"%(d["pcie_inclusive"]["ecdsa65_verifies_per_s"]/1e6), "iso", {k:round(v,3) for k,v in d["rates"]["kernel_ms_ecdsa_isolated"].items()}, d["parity"]["mismatches_by_leg"])
    except Exception as e: print(b, "ERR", e)
PY
