#!/usr/bin/env python3
"""Latency of small host-buffer calls (submit -> verdicts in host memory): 1 / 8 / 64 rows, key known to the cache or not, both
signature kinds, with the fused launch (k_small_verify) and with the general path (LAMD_SMALL_KERNEL=0 in a second process)."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_amd import Engine, workload

eng = Engine(0)
we = workload.make_ecdsa(eng, 200_000, seed=7, nkeys=64, publen=33, device="cuda:0", invalid_frac=0.1)
ws = workload.make_schnorr(eng, 200_000, seed=8, nkeys=64, device="cuda:0", invalid_frac=0.1)
# make the 64 keys known to the cache (a large call builds and publishes their tables)
assert (eng.verify_ecdsa(*we.cols) == we.expect).all() and (eng.verify_schnorr(*ws.cols) == ws.expect).all()
eng.synchronize()
cold = workload.make_ecdsa(eng, 4096, seed=9, nkeys=1 << 40, publen=33, device="cuda:0", group=1, invalid_frac=0.1)   # every row its own key


def p50(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts = np.sort(np.array(ts[reps // 10:])) * 1e3
    return ts[len(ts) // 2], ts[int(len(ts) * 0.99)]


print("LAMD_SMALL_KERNEL =", os.environ.get("LAMD_SMALL_KERNEL", "1 (default)"))
for bs in (1, 8, 64):
    for name, wl, fn in (("ecdsa33 cached key", we, eng.verify_ecdsa), ("schnorr cached key", ws, eng.verify_schnorr), ("ecdsa33 unknown key", cold, eng.verify_ecdsa)):
        k = [0]

        def call():
            o = (k[0] * bs) % (wl.n - bs)
            k[0] += 1
            got = fn(*[c[o:o + bs] for c in wl.cols])
            assert (got == wl.expect[o:o + bs]).all()
        a, b = p50(call, 300)
        print("  %2d rows, %-20s p50 %.3f ms  p99 %.3f ms" % (bs, name, a, b))
eng.close()
