#!/usr/bin/env python3
"""The pipelined host->host loop only (for a rocprofv3 --kernel-trace --memory-copy-trace timeline)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

n, depth = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 3
with Engine(0) as eng:
    we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65)
    ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536)
    torch.cuda.synchronize()

    def push(wl):
        if wl is we:
            eng.queue_ecdsa_batch(wl.cols[0], wl.cols[1], wl.cols[2])
        else:
            eng.queue_schnorr_batch(wl.cols[0], wl.cols[1], wl.cols[2])
    for rep in range(6):
        for wl in (we, ws):
            push(wl); eng.flush(); eng.wait(cap=n)
    time.sleep(0.2)          # a gap in the timeline marks where the pipelined loop starts
    pend = []
    t0 = time.perf_counter()
    for rep in range(8):
        for wl in (we, ws):
            push(wl); eng.flush(); pend.append(wl)
            if len(pend) == depth:
                eng.wait(cap=n); pend.pop(0)
    while pend:
        eng.wait(cap=n); pend.pop(0)
    dt = time.perf_counter() - t0
    print("pipelined, %d in flight: %.2f ms per 2M step" % (depth, dt / 8 * 1e3))
