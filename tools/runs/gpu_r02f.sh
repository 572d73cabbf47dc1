mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu_r02f.log 2>&1; tail -4 gpurun_out/pytest_gpu_r02f.log
timeout 600 python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err; tail -2 gpurun_out/bench_r02f.err
for b in 32 64; do LAMD_PREP_BATCH=$b timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_r02f_pb$b.json 2>/dev/null; done
LAMD_BENCH_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/bench_r02f_gather.json 2> gpurun_out/bench_r02f_gather.err; tail -3 gpurun_out/bench_r02f_gather.err
python - <<'PY'
import json
for f in ("bench_r02f","bench_r02f_pb32","bench_r02f_pb64","bench_r02f_gather"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f))
        print(f, "value %.1fM"%(d["value"]/1e6), "warm %.1fM"%(d["warm_cache"]["value"]/1e6), "pcie", d.get("pcie_inclusive",{}).get("ecdsa65_verifies_per_s"), "iso", d["rates"]["kernel_ms_ecdsa_isolated"])
        oc=d.get("other_configs_1gpu",{})
        for k,v in oc.items(): print("   ",k, v.get("verifies_per_s"), v.get("mismatches"))
        for k,v in d.get("sharded_configs",{}).items(): print("   SH",k, v.get("verifies_per_s"), v.get("mismatches"), v.get("ms"))
        print("    latency", d.get("latency"))
    except Exception as e: print(f, "ERR", e)
PY
