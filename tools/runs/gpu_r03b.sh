run() { timeout 300 python bench.py --skip-extra --cpu-sample 0 --no-parity > gpurun_out/bench_sweep_tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_sweep_tmp.json')); print('$1: cold %.1f warm %.1f mism %d' % (d['value']/1e6, d['warm_cache']['value']/1e6, d['parity']['mismatches']))"; }
for rep in 1 2; do
run "default"
LAMD_LANES=3 run "lanes3"
LAMD_LANES=6 run "lanes6"
LAMD_LANES=8 run "lanes8"
GPU_MAX_HW_QUEUES=24 run "queues24"
LAMD_KEYED_MIN_USES=4 run "min_uses4"
LAMD_KEYED_MIN_USES=8 run "min_uses8"
done
