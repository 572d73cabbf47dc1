#!/usr/bin/env python3
"""Experiment: does the ORDER of the rows (keys scattered as in cfg2, or rows of one key adjacent) change the table-driven ecmult?
Same number of rows, keys and rows per key; isolated calls, kernel durations from the engine's HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["LAMD_CACHE"] = "0"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

n = 1_000_000
with Engine(0) as eng:
    eng.set_timing(True)
    for name, kw in (("scattered (cfg2: 65536 keys, uniform reuse)", dict(nkeys=65536)), ("grouped (runs of 15 rows per key)", dict(nkeys=1 << 40, group=15))):
        w = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, publen=65, **kw)
        ts = []
        for it in range(5):
            eng.verify_ecdsa_device(w.dev[0], w.dev[1], w.dev[2], w.d_ok)
            eng.synchronize()
            ts.append(eng.info()["last_kernel_ms"])
        bad = int((w.d_ok.cpu().numpy().astype(bool) != w.expect).sum())
        inf = eng.info()
        print(name, "| ms [front, keys+tables, ecmult, parity]", [round(x, 3) for x in np.min(np.array(ts[1:]), axis=0)], "| unique keys", inf["last_unique_keys"],
              "hot rows", inf["last_hot_rows"], "cold", inf["last_cold_rows"], "mismatches", bad, flush=True)
