#!/usr/bin/env python3
"""shader clock of every k_mul32_peak launch from a `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv` run of tools/mul32_peak_probe.py:
usage: mul32_peak_clock.py <dir with *_counter_collection.csv and *_kernel_trace.csv>"""
import csv
import glob
import gzip
import sys


def rd(pat):
    f = (glob.glob(sys.argv[1] + "/**/" + pat, recursive=True) + glob.glob(sys.argv[1] + "/**/" + pat + ".gz", recursive=True))[0]
    return list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))


kt = {r["Dispatch_Id"]: r for r in rd("*kernel_trace.csv") if "k_mul32_peak" in r["Kernel_Name"]}
cc = [r for r in rd("*counter_collection.csv") if "k_mul32_peak" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
for r in cc:
    k = kt.get(r["Dispatch_Id"])
    if not k:
        continue
    dur = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e9
    grid = int(k["Grid_Size_X"]) if "Grid_Size_X" in k else int(k["Grid_Size"])
    print("grid %8d (waves/SIMD %d)  %.3f ms  GRBM_GUI_ACTIVE %.0f  ->  %.2f GHz" % (grid, grid // 256 // 256, dur * 1e3, float(r["Counter_Value"]), float(r["Counter_Value"]) / 8 / dur / 1e9))
