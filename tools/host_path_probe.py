#!/usr/bin/env python3
"""Where the host->host rate of the streaming queue goes: staging memcpy alone, flush+wait alone, and the pipelined loop,
for 1 M ECDSA-65 + 1 M BIP-340 per step.  usage: [LAMD_COPY_THREADS=k] [LAMD_CACHE=0] python tools/host_path_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
with Engine(0) as eng:
    we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65)
    ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536)
    torch.cuda.synchronize()

    def push(wl):
        if wl is we:
            eng.queue_ecdsa_batch(wl.cols[0], wl.cols[1], wl.cols[2])
        else:
            eng.queue_schnorr_batch(wl.cols[0], wl.cols[1], wl.cols[2])
    for rep in range(6):    # every staging set of both kinds gets allocated
        for wl in (we, ws):
            push(wl); eng.flush(); assert (eng.wait(cap=n) == wl.expect).all()
    t_copy, t_run = [], []
    for rep in range(4):
        for wl in (we, ws):
            t0 = time.perf_counter(); push(wl); t1 = time.perf_counter()
            eng.flush(); t2 = time.perf_counter(); eng.wait(cap=n); t3 = time.perf_counter()
            t_copy.append(t1 - t0); t_run.append((t2 - t1, t3 - t2))
    print("staging memcpy per 1M-row push (ms):", [round(x * 1e3, 2) for x in t_copy])
    print("flush call / wait (H2D + verify + D2H) per push (ms):", [(round(a * 1e3, 2), round(b * 1e3, 2)) for a, b in t_run])
    # plain numpy copy of the same bytes by one thread, for scale
    dst = [np.empty_like(c) for c in we.cols]
    t0 = time.perf_counter()
    for d, c in zip(dst, we.cols):
        np.copyto(d, c)
    print("numpy single-thread copy of the ECDSA columns (ms): %.2f" % ((time.perf_counter() - t0) * 1e3))
    for depth in (1, 2, 3, 4):
        pend = []
        t0 = time.perf_counter()
        for rep in range(10):
            for wl in (we, ws):
                push(wl); eng.flush(); pend.append(wl)
                if len(pend) == depth:
                    eng.wait(cap=n); pend.pop(0)
        while pend:
            eng.wait(cap=n); pend.pop(0)
        dt = time.perf_counter() - t0
        print("pipelined, %d flushes in flight: %.2f ms per 2M step, %.1f M verifies/s" % (depth, dt / 10 * 1e3, 20 * n / dt / 1e6))
    # device-resident reference on the same engine
    for rep in range(3):
        torch.cuda.synchronize(); eng.synchronize()
        t0 = time.perf_counter()
        for k in range(4):
            eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
            eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok)
        eng.synchronize()
        dt = time.perf_counter() - t0
    print("resident: %.2f ms per 2M step, %.1f M verifies/s" % (dt / 4 * 1e3, 8 * n / dt / 1e6))
