#!/usr/bin/env python3
"""usage: clock_probe_summary.py DIR LABEL...: DIR/LABEL_pmc (rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE ...) and DIR/LABEL_trace (--kernel-trace --stats)
of `bench.py --roofline-only` under one engine build each -> one line per build: the dominant kernel's mean duration without counters, and its shader
clock = GRBM_GUI_ACTIVE / XCDs / duration of the same dispatch in the counter pass (profiles/r06_clock.txt)."""
import collections
import csv
import glob
import sys

base, labels = sys.argv[1], sys.argv[2:]
KERNEL = "k_ecmult_keyed<false"


def rows(d, suffix):
    f = glob.glob("%s/%s/**/*_%s.csv" % (base, d, suffix), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


print("%-8s %10s %12s %12s %14s %10s" % ("build", "launches", "trace ms", "pmc-pass ms", "GUI_ACTIVE/XCD", "clock GHz"))
for lab in labels:
    tr = [r for r in rows(lab + "_trace", "kernel_trace") if KERNEL in r["Kernel_Name"]]
    big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in tr if int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) >= 100000]
    pm = rows(lab + "_pmc", "counter_collection")
    ptr = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows(lab + "_pmc", "kernel_trace")}
    gui = collections.defaultdict(float)
    for r in pm:
        if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and int(r["Grid_Size"]) >= 100000:
            gui[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    clk, pms = [], []
    for d, g in gui.items():
        if d in ptr and ptr[d] > 0:
            clk.append(g / 8 / (ptr[d] * 1e-3) / 1e9)
            pms.append(ptr[d])
    mean = lambda l: sum(l) / len(l) if l else float("nan")
    print("%-8s %10d %12.3f %12.3f %14.4g %10.3f" % (lab, len(big), mean(big), mean(pms), mean(list(gui.values())) / 8 if gui else float("nan"), mean(clk)))
