#!/usr/bin/env python3
"""configs[4] streamed from host memory in flushes of 256 / 1024 commitments: verifies/s against the number of flushes kept in flight."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from lightning_amd import Engine, workload

eng = Engine(0)
st = workload.make_commit_storm(eng, 10_000, device="cuda:0")
nv = st["ecdsa"].n + st["schnorr"].n
for cpf in (256, 1024):
    per, grp = st["per"], cpf * st["per"]
    jobs = []
    for kind in ("ecdsa", "schnorr"):
        wl = st[kind]
        for o in range(0, wl.n, grp):
            jobs.append((kind, wl, o, min(wl.n, o + grp)))
    jobs.sort(key=lambda j: j[2])
    for depth in (3, 8, 3, 8, 5):
        ts = []
        for it in range(3):
            pend, bad = [], 0
            t1 = time.perf_counter()
            for kind, wl, a, b in jobs:
                if kind == "ecdsa":
                    eng.queue_ecdsa_batch(wl.cols[0][a:b], wl.cols[1][a:b], wl.cols[2][a:b])
                else:
                    eng.queue_schnorr_batch(wl.cols[0][a:b], wl.cols[1][a:b], wl.cols[2][a:b])
                eng.flush()
                pend.append((wl, a, b))
                if len(pend) == depth:
                    wl0, a0, b0 = pend.pop(0)
                    bad += int((eng.wait() != wl0.expect[a0:b0]).sum())
            while pend:
                wl0, a0, b0 = pend.pop(0)
                bad += int((eng.wait() != wl0.expect[a0:b0]).sum())
            ts.append(time.perf_counter() - t1)
        print("%4d commitments per flush, %d in flight: %.1f M verifies/s (best of the last two of three), mismatches %d" % (cpf, depth, nv / min(ts[1:]) / 1e6, bad), flush=True)
eng.close()
