mkdir -p gpurun_out
for rep in 1 2; do
for v in g22 g16; do
  if [ $v = g16 ]; then export LAMD_LIB_PATH=$PWD/tools/variants/liblightning_amd_g16.so; else unset LAMD_LIB_PATH; fi
  timeout 300 python bench.py --skip-extra --cpu-sample 0 > gpurun_out/bench_gtab_${v}_$rep.json 2>/dev/null
done; done
unset LAMD_LIB_PATH
python - <<'PY' | tee gpurun_out/gtable_window_comparison.txt
import json
print("# G table: 12 windows x 2^22 entries (3 GiB, shipped) against 16 windows x 2^16 (64 MiB): `python bench.py --skip-extra --cpu-sample 0`, two runs each, alternating")
for v in ("g22","g16"):
    for rep in (1,2):
        d=json.load(open("gpurun_out/bench_gtab_%s_%d.json"%(v,rep)))
        print(v, "run", rep, "cold %.1f M/s"%(d["value"]/1e6), "warm %.1f M/s"%(d["warm_cache"]["value"]/1e6), "isolated ecmult %.3f ms"%d["rates"]["kernel_ms_ecdsa_isolated"]["ecmult"], "mismatches", d["parity"]["mismatches"])
PY
bash tools/queue_sweep.sh 4 1 > /dev/null 2>&1; cat gpurun_out/queue_sweep.txt
timeout 900 python tests/soak_10m.py > gpurun_out/soak_r02.json 2> gpurun_out/soak_r02.err; tail -c 600 gpurun_out/soak_r02.json
