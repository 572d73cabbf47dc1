#!/bin/bash
# round 3, GPU session i: the four-wave k_small_verify (u1*G spread over the task waves, three-level merge): tests, latency probe, kernel durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3i_pytest.log
tail -3 gpurun_out/r3i_pytest.log
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3i_latency.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3i_lat -- python $R/tools/latency_probe.py > /dev/null 2>&1 )
python - <<'PY' | tee -a gpurun_out/r3i_latency.txt
import csv, glob, statistics
f = glob.glob("gpurun_out/r3i_lat/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if r["Kernel_Name"].startswith("k_small_verify")]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"]))]
print("k_small_verify launches", len(d))
for lo in range(0, len(d), 300):
    seg = d[lo:lo + 300]
    if len(seg) >= 50:
        print("  launches %5d..%5d: median %.1f us  min %.1f  p90 %.1f" % (lo, lo + len(seg), statistics.median(seg), min(seg), sorted(seg)[int(len(seg) * 0.9)]))
PY
rm -rf gpurun_out/r3i_lat
