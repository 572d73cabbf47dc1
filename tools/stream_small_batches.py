#!/usr/bin/env python3
"""Small-batch streaming (configs[4] as BASELINE.json words it: batches of 484 in arrival order): one commitment_signed per flush, the channels
recur; flushes in flight 1 / 2 / 4 / 8.  Pass 0 meets every key for the first time (ladder), pass 1 builds their tables, later passes are cache hits:
batches/s, signatures/s and the per-batch latency (flush -> verdicts collected) of the steady passes.
usage: [LAMD_SMALL_KERNEL=0] python tools/stream_small_batches.py [channels]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightning_amd import Engine, workload

eng = Engine(0)
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
st = workload.make_commit_storm(eng, nch, bip340_every=0)
w = st["ecdsa"]
per = st["per"]
nb = w.n // per
out = {"channels": nb, "signatures_per_batch": per, "small_kernel": os.environ.get("LAMD_SMALL_KERNEL", "1 (default)")}


def one_pass(depth):
    pend, bad, lat = [], 0, []
    t0 = time.perf_counter()
    for b in range(nb):
        a = b * per
        eng.queue_ecdsa_batch(w.cols[0][a:a + per], w.cols[1][a:a + per], w.cols[2][a:a + per])
        eng.flush()
        pend.append((a, time.perf_counter()))
        if len(pend) == depth:
            a0, t1 = pend.pop(0)
            bad += int((eng.wait() != w.expect[a0:a0 + per]).sum())
            lat.append(time.perf_counter() - t1)
    while pend:
        a0, t1 = pend.pop(0)
        bad += int((eng.wait() != w.expect[a0:a0 + per]).sum())
        lat.append(time.perf_counter() - t1)
    assert bad == 0, bad
    return nb / (time.perf_counter() - t0), np.sort(np.array(lat)) * 1e3


first = one_pass(1)
second = one_pass(1)
out["first_sight_pass"] = {"batches_per_s": first[0], "p50_ms": float(first[1][len(first[1]) // 2])}
out["table_building_pass"] = {"batches_per_s": second[0], "p50_ms": float(second[1][len(second[1]) // 2])}
for depth in (1, 2, 4, 8):
    best, lat = 0.0, None
    for rep in range(3):
        r, l = one_pass(depth)
        if r > best:
            best, lat = r, l
    out["in_flight_%d" % depth] = {"batches_per_s": best, "signatures_per_s": best * per, "p50_ms": float(lat[len(lat) // 2]), "p99_ms": float(lat[int(len(lat) * 0.99)])}
print(json.dumps(out))
eng.close()
