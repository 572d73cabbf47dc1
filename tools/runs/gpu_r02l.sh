mkdir -p gpurun_out
timeout 100 tools/variants/fe_bench_fused > gpurun_out/fe_bench_fused.txt 2>&1; cat gpurun_out/fe_bench_fused.txt
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu_r02l.log 2>&1; tail -4 gpurun_out/pytest_gpu_r02l.log
timeout 700 python bench.py > gpurun_out/bench_r02l.json 2> gpurun_out/bench_r02l.err; tail -2 gpurun_out/bench_r02l.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r02l.json"))
print("value %.1fM"%(d["value"]/1e6), "warm %.1fM"%(d["warm_cache"]["value"]/1e6), "pcie %.1fM"%(d["pcie_inclusive"]["ecdsa65_verifies_per_s"]/1e6))
print("iso", {k:round(v,3) for k,v in d["rates"]["kernel_ms_ecdsa_isolated"].items()}, d["parity"]["mismatches_by_leg"])
print("lat", {k:(round(v["p50_ms"],3) if isinstance(v,dict) else v) for k,v in d["latency"].items() if k!="note"})
for k,v in d["other_configs_1gpu"].items(): print("  ",k, {kk:(round(vv/1e6,2) if isinstance(vv,float) and vv>1e5 else vv) for kk,vv in v.items() if kk not in ("note","batch","check")})
PY
bash tools/pmc_run.sh r02l > gpurun_out/pmc_r02l.log 2>&1; tail -3 gpurun_out/pmc_r02l.log
