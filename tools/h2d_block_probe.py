#!/usr/bin/env python3
"""Does hipMemcpyAsync(H2D, pinned, 64 MB) return before the copy has run?  Host time of the CALL for copies queued back to back on one stream, on
several streams, and with an event record + cross-stream wait in between (the pattern of lamd_flush).  Then the host time of lamd_flush / lamd_wait
in the steady cold streaming loop."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p


def chk(rc):
    assert rc == 0, rc


N = 64 << 20
h = [vp() for _ in range(6)]
d = [vp() for _ in range(6)]
for i in range(6):
    chk(hip.hipHostMalloc(ctypes.byref(h[i]), ctypes.c_size_t(N), 0))
    ctypes.memset(h[i], i + 1, N)
    chk(hip.hipMalloc(ctypes.byref(d[i]), ctypes.c_size_t(N)))
streams = [vp() for _ in range(3)]
for s in streams:
    chk(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1))
ev = [vp() for _ in range(6)]
for e in ev:
    chk(hip.hipEventCreateWithFlags(ctypes.byref(e), 2))
hip.hipMemcpyAsync.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, vp]


def run(label, plan):
    chk(hip.hipDeviceSynchronize())
    t0 = time.perf_counter()
    calls = []
    for i, (s, record, waiter) in enumerate(plan):
        t = time.perf_counter()
        chk(hip.hipMemcpyAsync(d[i % 6], h[i % 6], N, 1, streams[s]))
        calls.append((time.perf_counter() - t) * 1e3)
        if record:
            chk(hip.hipEventRecord(ev[i % 6], streams[s]))
            if waiter is not None:
                chk(hip.hipStreamWaitEvent(streams[waiter], ev[i % 6], 0))
    t_issue = (time.perf_counter() - t0) * 1e3
    chk(hip.hipDeviceSynchronize())
    t_all = (time.perf_counter() - t0) * 1e3
    print("%-58s host ms per hipMemcpyAsync call: %s | all issued after %.2f ms, all done after %.2f ms" % (label, " ".join("%.2f" % c for c in calls), t_issue, t_all))


for rep in range(2):
    run("6 x 64 MB, one stream", [(0, False, None)] * 6)
    run("6 x 64 MB, one stream, event after each", [(0, True, None)] * 6)
    run("6 x 64 MB, one stream, event + other stream waits", [(0, True, 1)] * 6)
    run("6 x 64 MB, alternating two streams", [(i % 2, False, None) for i in range(6)])

# ---- the engine's own loop
import numpy as np
os.environ["LAMD_CACHE"] = "0"
from lightning_amd import Engine, workload
eng = Engine(0)
n = 1_000_000
we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65, device="cuda:0")
ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536, device="cuda:0")
filled = set()


def stream(steps, depth=8):
    pend = []
    tf, tw, tr = [], [], []
    t0 = time.perf_counter()
    for r in range(steps):
        for wl in (we, ws):
            t = time.perf_counter()
            _, a, b_, c = eng.queue_reserve(n, 65 if wl is we else 32)
            if a.ctypes.data not in filled:
                filled.add(a.ctypes.data)
                a[:] = wl.cols[0]
                if wl is we:
                    b_[:], c[:] = wl.cols[1], wl.cols[2]
                else:
                    c[:], b_[:] = wl.cols[1], wl.cols[2]
            tr.append(time.perf_counter() - t)
            t = time.perf_counter()
            eng.flush()
            tf.append(time.perf_counter() - t)
            pend.append(wl)
            if len(pend) == depth:
                t = time.perf_counter()
                eng.wait(cap=n)
                tw.append(time.perf_counter() - t)
                pend.pop(0)
    while pend:
        eng.wait(cap=n)
        pend.pop(0)
    dt = time.perf_counter() - t0
    return dt, np.array(tf) * 1e3, np.array(tw) * 1e3, np.array(tr) * 1e3


stream(9)
dt, tf, tw, tr = stream(12)
print("cold in-place streaming loop: %.1f M verifies/s | host ms per call: flush mean %.3f (min %.3f max %.3f), wait mean %.3f, reserve mean %.3f" % (
    2 * n * 12 / dt / 1e6, tf.mean(), tf.min(), tf.max(), tw.mean(), tr.mean()))
print("flush ms:", " ".join("%.2f" % x for x in tf))
eng.close()
