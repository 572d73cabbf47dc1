#!/bin/bash
# round 3, GPU session e: tests; kernel durations of the fused latency path; the rocprofv3 passes behind the roofline block (kernel trace +
# stats, FETCH_SIZE, WRITE_SIZE, SQ sets); kernel trace of bench --roofline-only; lane timeline of the steady loops; the 10 M-row soak
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3e_pytest.log
tail -2 gpurun_out/r3e_pytest.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e_lat -- python $R/tools/latency_probe.py > $R/gpurun_out/r3e_lat.txt 2>&1 )
grep -h "k_small_verify\|k_ecmult_keyed<false\|k_ecmult<3>" $(find gpurun_out/r3e_lat -name "*kernel_stats.csv") | cut -c1-60,200-400 | head -5
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3e_lat/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith("k_small_verify")]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"]))]
import statistics
print("k_small_verify launches", len(d))
for lo in range(0, len(d), 300):
    seg = d[lo:lo + 300]
    if len(seg) >= 50:
        print("  launches %5d..%5d: median %.1f us  min %.1f  p90 %.1f" % (lo, lo + len(seg), statistics.median(seg), min(seg), sorted(seg)[int(len(seg) * 0.9)]))
PY
bash tools/pmc_run.sh r03 > gpurun_out/r3e_pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_r03 gpurun_out/r03 > gpurun_out/r3e_pmc_summary.log 2>&1; tail -3 gpurun_out/r3e_pmc_summary.log
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3e_rf -- python $R/bench.py --roofline-only > $R/gpurun_out/r3e_roofline_only.json 2> $R/gpurun_out/r3e_rf.err )
cp $(find gpurun_out/r3e_rf -name "*kernel_stats.csv" | head -1) gpurun_out/r03_roofline_only_kernel_stats.csv
gzip -c $(find gpurun_out/r3e_rf -name "*kernel_trace.csv" | head -1) > gpurun_out/r03_roofline_only_kernel_trace.csv.gz
head -4 gpurun_out/r03_roofline_only_kernel_stats.csv | cut -c1-80,250-400
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3e_steady -- python $R/bench.py --steps 6 --warmup 2 --cpu-sample 0 --skip-extra > $R/gpurun_out/r3e_steady.json 2> $R/gpurun_out/r3e_steady.err )
python tools/lane_timeline.py $(find gpurun_out/r3e_steady -name "*kernel_trace.csv" | head -1) 6 2 > gpurun_out/r03_lane_timeline.txt 2>&1; head -30 gpurun_out/r03_lane_timeline.txt
timeout 600 python tests/soak_10m.py > gpurun_out/r03_soak_10M.json 2> gpurun_out/r3e_soak.err; tail -c 400 gpurun_out/r03_soak_10M.json
rm -rf gpurun_out/r3e_lat gpurun_out/r3e_rf gpurun_out/r3e_steady gpurun_out/pmc_r03/*/runc 2>/dev/null
du -sh gpurun_out
