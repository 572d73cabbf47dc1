#!/bin/bash
# round 4, GPU session b: host-side probes of the GPU box (thread wake-up, page faults), the ingest host bench by thread count (no GPU work: the box's
# CPU), the full GPU suite incl. the 10.84 M-row soak, rocprofv3 --kernel-trace --stats of the roofline loop, the PMC passes + FETCH_SIZE calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/kernel/mm/transparent_hugepage/enabled
g++ -O2 -pthread -o /tmp/par tools/thread_scaling_probe.cpp && /tmp/par
gcc -O2 -o /tmp/pf tools/page_fault_probe.c && /tmp/pf
for t in 1 4 16; do
  echo "== ingest host bench, LAMD_INGEST_THREADS=$t"
  LAMD_INGEST_THREADS=$t LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 2>&1 | grep -E "sub-batch|host logic" | tail -7
done 2>&1 | tee gpurun_out/r4b_ingest_host.txt
echo "== one-by-one path"; LAMD_INGEST_RUN_MIN=0 LAMD_INGEST_SUB=100000000 LAMD_INGEST_PROFILE=1 timeout 300 python tools/ingest_host_bench.py 100000 4 2>&1 | grep -E "sub-batch|host logic" | tail -3 | tee -a gpurun_out/r4b_ingest_host.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4b_pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4b_roofline -- python $R/bench.py --roofline-only --steps 100 --warmup 5 > $R/gpurun_out/r4b_roofline_only.json 2> $R/gpurun_out/r4b_roofline.err
cd $R
find gpurun_out/r4b_roofline -name "*kernel_stats.csv" | head -2
python - <<'PY'
import csv, glob, json
d = json.loads(open("gpurun_out/r4b_roofline_only.json").read().strip().splitlines()[-1])
r = d["roofline"]
f = glob.glob("gpurun_out/r4b_roofline/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "k_ecmult_keyed<false" in row["Name"]:
        print("rocprofv3 stats:", row["Name"][:40], "calls", row["Calls"], "avg ms", float(row["AverageNs"]) / 1e6)
print("bench.py roofline: mode %s launches %d avg_launch_ms %.4f frac %.4f sum/step %.3f <= %.3f" % (r["mode"], r["launches_timed"], r["avg_launch_ms"], r["frac"], r["sum_of_launch_ms_per_step"], r["ms_per_step"]))
PY
PMC_STEPS=6 bash tools/pmc_run.sh r04 > gpurun_out/r4b_pmc.log 2>&1; tail -3 gpurun_out/r4b_pmc.log
python tools/pmc_summary.py gpurun_out/pmc_r04 gpurun_out/r04 > gpurun_out/r4b_pmc_summary.log 2>&1; tail -25 gpurun_out/r4b_pmc_summary.log
