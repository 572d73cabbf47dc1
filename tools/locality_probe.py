#!/usr/bin/env python3
"""How much does the table-driven ecmult gain when the rows of one key sit next to each other?  1 M ECDSA-65 rows under 65 536 keys, key-table
cache off, one call at a time (kernel-stage times from the engine's own HIP events) and 8 calls pipelined: keys drawn per row (configs[1]) against
keys drawn per group of 16 consecutive rows.  usage: [LAMD_PAIRS=0|1] python tools/locality_probe.py"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["LAMD_CACHE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lightning_amd import engine as E, workload

eng = E.Engine(0)
n = 1000000
for label, grp in (("keys per row", 0), ("keys per 16 rows", 16)):
    wk = workload.make_ecdsa(eng, n, nkeys=65536, publen=65, group=grp)
    for _ in range(eng.info()["lanes"] + 1):
        eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], wk.d_ok)
    eng.synchronize()
    iso = []
    for _ in range(6):
        eng.set_timing(True)
        eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], wk.d_ok)
        eng.synchronize()
        lanes = range(eng.info()["lanes"])
        iso.append(sum(eng.info(l)["keyed_ecmult_ms_sum"][0] for l in lanes) / max(1, sum(eng.info(l)["keyed_ecmult_launches"][0] for l in lanes)))
    eng.set_timing(False)
    t1 = time.perf_counter()
    for _ in range(16):
        eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], wk.d_ok)
    eng.synchronize()
    dt = time.perf_counter() - t1
    bad = int((wk.d_ok.cpu().numpy().astype(bool) != wk.expect).sum())
    inf = eng.info()
    print("%-18s LAMD_PAIRS=%s: ecmult launch alone %s ms, 16 calls pipelined %.1f M/s, distinct keys %d, mismatches %d" % (
        label, os.environ.get("LAMD_PAIRS", "1"), " ".join("%.3f" % x for x in iso[1:]), 16 * n / dt / 1e6, inf["last_unique_keys"], bad), flush=True)
