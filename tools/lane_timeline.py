#!/usr/bin/env python3
"""Lane timeline of the timed loops from a rocprofv3 kernel trace (CSV, optionally .gz) of
`bench.py --steps K --warmup W --cpu-sample 0 --skip-extra`: for each loop (cold, cold with the large ecmult launches
chained, then the warm engine's) the window between its first and last 1 M-row table-driven ecmult launch, how busy each kernel class was inside it and
how the launches overlap.
usage: lane_timeline.py trace.csv[.gz] [steps] [warmup]"""
import collections
import csv
import gzip
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
op = gzip.open if path.endswith(".gz") else open
rows = sorted(csv.DictReader(op(path, "rt")), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.split("(")[0].replace("void ", "")


CLASS = (("k_ecmult_keyed<false", "table-driven ecmult"), ("k_ecmult_keyed<true", "complete-formula ecmult"), ("k_ecmult<", "cold-row ladder"),
         ("k_kc_", "table building"), ("k_keys_bases", "table building"), ("k_keys", "key parse"), ("k_call_init", "de-duplication"), ("k_cache_", "cache lookup/publish"), ("k_dedupe", "de-duplication"),
         ("k_partition", "partition"), ("k_ecdsa_prep", "scalar prep"), ("k_schnorr_prep", "scalar prep"), ("k_schnorr_final", "BIP-340 parity stage"))


def cls(name):
    for p, c in CLASS:
        if name.startswith(p):
            return c
    return "other"


big = [r for r in rows if short(r["Kernel_Name"]).startswith("k_ecmult_keyed<false") and int(r["Grid_Size_X"]) >= 500000]
# launch order of bench.py: cold loop (W warm-up + K timed steps, two launches per step), cold loop with chained launches (same), 2 x 2
# isolated calls, then -- on the default engine, created only now -- the warm loop
K, W = steps, warmup
per = 2 * W + 2 * K
legs = [big[2 * W:per], big[per + 2 * W:2 * per], big[2 * per + 4 + 2 * W:3 * per + 4]]
iso = big[2 * per:2 * per + 4]
print("isolated calls (one at a time): %s ms" % ", ".join("%.3f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in iso))
for name, leg in zip(("cold loop (tables rebuilt every call: `value`)", "cold loop, large ecmult launches chained (lamd_set_ecmult_chain(1): roofline.chained)",
                      "warm loop (key-table cache on)"), legs):
    if not leg:
        continue
    t0, t1 = int(leg[0]["Start_Timestamp"]), int(leg[-1]["End_Timestamp"])
    span = (t1 - t0) / 1e6
    print("== %s: %d launches, window %.2f ms (%.2f ms per 2-launch step)" % (name, len(leg), span, span / (len(leg) / 2)))
    busy = collections.defaultdict(float)
    edges = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e <= t0 or s >= t1:
            continue
        s, e = max(s, t0), min(e, t1)
        c = cls(short(r["Kernel_Name"]))
        busy[c] += (e - s) / 1e6
        if c == "table-driven ecmult" and int(r["Grid_Size_X"]) >= 500000:
            edges += [(s, 1), (e, -1)]
    edges.sort()
    depth, last, hist = 0, t0, collections.defaultdict(float)
    for t, d in edges:
        hist[depth] += (t - last) / 1e6
        depth, last = depth + d, t
    hist[depth] += (t1 - last) / 1e6
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in leg]
    print("   table-driven ecmult launches: mean %.3f ms, min %.3f, max %.3f" % (sum(durs) / len(durs), min(durs), max(durs)))
    print("   time with 0 / 1 / 2 / 3+ such launches in flight: %s" % " / ".join("%.1f %%" % (100 * sum(v for k, v in hist.items() if (k == i if i < 3 else k >= 3)) / span) for i in range(4)))
    for c, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        print("   %-26s kernel-time %8.2f ms = %5.2f of the window" % (c, v, v / span))
