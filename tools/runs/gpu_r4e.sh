#!/bin/bash
# round 4, GPU session e: where the host -> host form loses time (kernel + memory-copy trace of the streaming queue, cache off and on), and the
# ingest flood of the bench with the ingest's phase clock
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  LAMD_CACHE=$c PROBE_INFLIGHT=8,8 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r4e_h2h_c$c -- python $R/tools/stream_probe.py > $R/gpurun_out/r4e_h2h_c$c.txt 2>&1
  tail -3 $R/gpurun_out/r4e_h2h_c$c.txt
done
cd $R
for f in gpurun_out/r4e_h2h_c*/runc/*_kernel_trace.csv gpurun_out/r4e_h2h_c*/runc/*_memory_copy_trace.csv; do gzip -f $f; done
ls -la gpurun_out/r4e_h2h_c*/runc/ | head -20
LAMD_INGEST_PROFILE=1 timeout 900 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
grep "ingest\]" gpurun_out/r4e_bench.err | tail -45 | tee gpurun_out/r4e_ingest_phases.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench.json").read().strip().splitlines()[-1])
o = d["other_configs_1gpu"]
print("ingest", {k: (round(v / 1e6, 2) if isinstance(v, float) and v > 1000 else v) for k, v in o["gossip_ingest_flood"].items() if k != "note" and k != "shape"})
r = d["roofline"]
print("sum/step %.3f <= %.3f" % (r["sum_of_launch_ms_per_step"], r["ms_per_step"]), "h2h", d["value_host_to_host"]["ratio_to_value"])
PY
