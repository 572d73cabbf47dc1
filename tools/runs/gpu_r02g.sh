mkdir -p gpurun_out
for cfg in "16 0" "32 32768" "64 16384" "128 8192"; do set -- $cfg
LAMD_PREP_BATCH=$1 LAMD_PREP_MIN_THREADS=$2 timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_r02g_pb$1.json 2>gpurun_out/bench_r02g_pb$1.err || tail -2 gpurun_out/bench_r02g_pb$1.err; done
python - <<'PY'
import json
for b in (16,32,64,128):
    try:
        d=json.load(open("gpurun_out/bench_r02g_pb%d.json"%b))
        print(b, "value %.1fM"%(d["value"]/1e6), "warm %.1fM"%(d["warm_cache"]["value"]/1e6), "pcie %.1fM"%(d["pcie_inclusive"]["ecdsa65_verifies_per_s"]/1e6), "iso", {k:round(v,3) for k,v in d["rates"]["kernel_ms_ecdsa_isolated"].items()}, d["parity"]["mismatches"])
    except Exception as e: print(b, "ERR", e)
PY
