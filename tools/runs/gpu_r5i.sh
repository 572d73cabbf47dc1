#!/bin/bash
# round 5, GPU session i: the collective path on one rank with 2 / 4 verdict buffers per kind against the plain loop; the whole collective-path bench line
# (sharded configs included); the bench line with the 100-step host->host region
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
line() {
  python -c "
import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$2: cold %.1f M/s (step %.3f ms) mismatches %d' % (d['value']/1e6, d['ms_per_step'], d['parity']['mismatches']))"
}
: > gpurun_out/r5i_collective.txt
k=0
for CFG in "plain 0" "gather 2" "gather 4" "gather 2" "gather 4" "gather 6" "plain 0"; do
  set -- $CFG
  k=$((k+1))
  if [ $1 = plain ]; then
    timeout 300 python bench.py --ab --steps 100 --warmup 5 > gpurun_out/r5i_c$k.json 2> gpurun_out/r5i_c$k.err || tail -3 gpurun_out/r5i_c$k.err
  else
    LAMD_BENCH_GATHER_BUFS=$2 LAMD_BENCH_GATHER=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29800 + k)) bench.py --gpus 1 --ab --steps 100 --warmup 5 > gpurun_out/r5i_c$k.json 2> gpurun_out/r5i_c$k.err || tail -3 gpurun_out/r5i_c$k.err
  fi
  line gpurun_out/r5i_c$k.json "$1 buffers=$2" | tee -a gpurun_out/r5i_collective.txt
done
LAMD_BENCH_GATHER=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29821 bench.py --gpus 1 > gpurun_out/r5i_bench_gather.json 2> gpurun_out/r5i_bench_gather.err; echo "gather bench rc=$?"; tail -2 gpurun_out/r5i_bench_gather.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5i_bench_gather.json | head -2
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5i_bench_gather.json").read().strip().splitlines()[-1])
for k, v in (d.get("sharded_configs") or {}).items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a not in ("note",)})
PY
S=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5i_bench.json 2> gpurun_out/r5i_bench.err; echo "bench.py rc=$? wall $(( $(date +%s) - S )) s"; tail -2 gpurun_out/r5i_bench.err | cut -c1-300
python tools/bench_summary.py gpurun_out/r5i_bench.json
