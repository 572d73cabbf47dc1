#!/usr/bin/env python3
"""Small-batch streaming: commitments of 484 signatures, one flush each, 1 / 2 / 4 flushes in flight.
usage: [LAMD_LANES=1] python tools/stream_small_batches.py   (prints batches/s and signatures/s)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lightning_amd import Engine, workload

eng = Engine(0)
st = workload.make_commit_storm(eng, 400, bip340_every=0)
w = st["ecdsa"]
per = st["per"]
nb = w.n // per
out = {}
for depth in (1, 2, 4):
    best = 0.0
    for rep in range(3):
        pend, bad = [], 0
        t0 = time.perf_counter()
        for b in range(nb):
            a = b * per
            eng.queue_ecdsa_batch(w.cols[0][a:a + per], w.cols[1][a:a + per], w.cols[2][a:a + per])
            eng.flush()
            pend.append(a)
            if len(pend) > depth - 1 and depth > 1 or depth == 1:
                a0 = pend.pop(0)
                bad += int((eng.wait() != w.expect[a0:a0 + per]).sum())
        while pend:
            a0 = pend.pop(0)
            bad += int((eng.wait() != w.expect[a0:a0 + per]).sum())
        dt = time.perf_counter() - t0
        assert bad == 0
        best = max(best, nb / dt)
    out["in_flight_%d" % depth] = {"batches_per_s": best, "signatures_per_s": best * per}
out["lanes"] = os.environ.get("LAMD_LANES", "4 (default)")
print(json.dumps(out))
