#!/usr/bin/env python3
"""Per-phase timeline of a tools/call_trace_probe.py run from its rocprofv3 kernel trace (CSV[.gz]): phases are delimited by the marker fills
(Grid_Size_X = 1000 * 256 * k / 4 ... any fill whose element count is a multiple of 256000).  For each phase: wall span, busy time (union of kernel
intervals), per-kernel totals and -- with `-v` -- every launch in start order with the gap to the previous end.
usage: trace_calls.py kernel_trace.csv[.gz] [-v] [phase numbers...]"""
import collections
import csv
import gzip
import sys

path = sys.argv[1]
verbose = "-v" in sys.argv
want = [int(a) for a in sys.argv[2:] if a.isdigit()]
op = gzip.open if path.endswith(".gz") else open
rows = sorted(csv.DictReader(op(path, "rt")), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.split("(")[0].replace("void ", "")[:60]


marks = []
for i, r in enumerate(rows):
    nm = r["Kernel_Name"]
    if "FillFunctor" in nm or "fill" in nm.lower():
        g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
        marks.append((i, g, int(r["Start_Timestamp"])))
# marker k = a fill of 256000 * k floats: grid 64000 * k (four elements per work item)
big = [m for m in marks if m[1] % 64000 == 0 and 1 <= m[1] // 64000 <= 16]
sizes = sorted({m[1] for m in big})
print("markers:", [s // 64000 for s in sizes])
bounds = []
for s in sizes:
    idx = [m[0] for m in big if m[1] == s]
    bounds.append(idx[-1])
for p in range(len(bounds) - 1):
    if want and (sizes[p] // 64000) not in want:
        continue
    seg = rows[bounds[p] + 1:bounds[p + 1]]
    seg = [r for r in seg if "FillFunctor" not in r["Kernel_Name"]]
    if not seg:
        continue
    t0 = int(seg[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in seg)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = collections.Counter()
    cnt = collections.Counter()
    for r in seg:
        tot[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        cnt[short(r["Kernel_Name"])] += 1
    print("\n== after marker %d: %d launches, span %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms" % (sizes[p] // 64000, len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    for k, v in tot.most_common(24):
        print("   %-62s n=%4d  total %8.3f ms  avg %7.3f ms" % (k, cnt[k], v / 1e6, v / cnt[k] / 1e6))
    if verbose:
        last_end = t0
        for r in seg:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print("      +%9.3f ms  dur %8.3f  gap %8.3f  q=%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - last_end) / 1e6, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
            last_end = max(last_end, e)
