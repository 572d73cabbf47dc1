#!/usr/bin/env python3
"""Diagnostic: where a small batch's latency goes (engine HIP events + host clock)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

with Engine(0) as eng:
    eng.set_timing(True)
    st = workload.make_commit_storm(eng, 4, seed=4242)["ecdsa"]
    rnd = workload.make_ecdsa(eng, 484, seed=5, nkeys=1 << 40, publen=33)
    for name, cols in (("commitment (one htlc key)", [np.ascontiguousarray(x[:484]) for x in st.cols]), ("484 distinct keys", rnd.cols)):
        for it in range(6):
            t = time.perf_counter()
            eng.verify_ecdsa(*cols)
            dt = (time.perf_counter() - t) * 1e3
            inf = eng.info()
            print(name, "call", it, "host %.3f ms" % dt, "kernel ms [front, keys, ecmult, final]", [round(x, 3) for x in inf["last_kernel_ms"]], "hits", inf["last_cache_hits"],
                  "new tables", inf["last_new_tables"], flush=True)
