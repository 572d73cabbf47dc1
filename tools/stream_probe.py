#!/usr/bin/env python3
"""Streaming queue from pinned staging memory (lamd_queue_reserve): verifies/s against the number of flushes kept in flight."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LAMD_CACHE", "0")
from lightning_amd import Engine, workload

n = 1_000_000
eng = Engine(0)
we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65, device="cuda:0")
ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536, device="cuda:0")


def loop(reps, inflight, filled):
    pend, bad = [], 0
    t1 = time.perf_counter()
    for r in range(reps):
        for wl in (we, ws):
            _, a, b, c = eng.queue_reserve(n, 65 if wl is we else 32)
            if a.ctypes.data not in filled:
                filled.add(a.ctypes.data)
                a[:] = wl.cols[0]
                if wl is we:
                    b[:], c[:] = wl.cols[1], wl.cols[2]
                else:
                    c[:], b[:] = wl.cols[1], wl.cols[2]
            eng.flush()
            pend.append(wl)
            if len(pend) == inflight:
                bad += int((eng.wait(cap=n) != pend.pop(0).expect).sum())
    while pend:
        bad += int((eng.wait(cap=n) != pend.pop(0).expect).sum())
    return time.perf_counter() - t1, bad


seen = set()
loop(12, max(int(x) for x in os.environ.get("PROBE_INFLIGHT", "4").split(",")), seen)
for inflight in [int(x) for x in os.environ.get("PROBE_INFLIGHT", "2,3,4,3,4").split(",")]:
    dt, bad = loop(10, inflight, seen)
    print("in flight %d: %.1f M verifies/s (%.2f ms per 2 M-row step), mismatches %d" % (inflight, 20 * n / dt / 1e6, dt / 10 * 1e3, bad))


def loop_copying(reps, inflight):
    """the copying form: caller-owned (pageable) rows -> lamd_queue_*_batch -> staging set (LAMD_COPY_THREADS threads)"""
    pend, bad = [], 0
    t1 = time.perf_counter()
    for r in range(reps):
        for wl in (we, ws):
            if wl is we:
                eng.queue_ecdsa_batch(wl.cols[0], wl.cols[1], wl.cols[2])
            else:
                eng.queue_schnorr_batch(wl.cols[0], wl.cols[1], wl.cols[2])
            eng.flush()
            pend.append(wl)
            if len(pend) == inflight:
                bad += int((eng.wait(cap=n) != pend.pop(0).expect).sum())
    while pend:
        bad += int((eng.wait(cap=n) != pend.pop(0).expect).sum())
    return time.perf_counter() - t1, bad


for inflight in [int(x) for x in os.environ.get("PROBE_COPYING", "").split(",") if x]:
    loop_copying(3, inflight)
    dt, bad = loop_copying(10, inflight)
    print("copying form, in flight %d: %.1f M verifies/s (%.2f ms per 2 M-row step), mismatches %d" % (inflight, 20 * n / dt / 1e6, dt / 10 * 1e3, bad))
# the same calls with the inputs resident in HBM and at most `inflight` calls outstanding (host waits for the oldest)
import torch
for inflight in [int(x) for x in os.environ.get("PROBE_RESIDENT", "3,4,100").split(",") if x]:
    eng.synchronize()
    t1 = time.perf_counter()
    k = 0
    for r in range(10):
        for wl in (we, ws):
            if wl is we:
                eng.verify_ecdsa_device(wl.dev[0], wl.dev[1], wl.dev[2], wl.d_ok)
            else:
                eng.verify_schnorr_device(wl.dev[0], wl.dev[1], wl.dev[2], wl.d_ok)
            k += 1
            if inflight < 100 and k % inflight == 0:
                eng.synchronize()
    eng.synchronize()
    dt = time.perf_counter() - t1
    print("resident, synchronise every %d calls: %.1f M verifies/s" % (inflight, 20 * n / dt / 1e6))
eng.close()
