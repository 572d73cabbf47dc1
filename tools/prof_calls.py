#!/usr/bin/env python3
"""A few isolated calls (synchronised in between) for per-kernel durations under `rocprofv3 --kernel-trace --stats`.
usage: [LAMD_CACHE=0] [LAMD_KEYED_WAVES=3] python tools/prof_calls.py [rows]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
with Engine(0) as eng:
    eng.set_timing(True)
    we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65)
    ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536)
    for it in range(5):
        eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
        eng.synchronize()
        print("ecdsa", [round(x, 3) for x in eng.info()["last_kernel_ms"]], eng.info()["last_cache_hits"], flush=True)
        eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok)
        eng.synchronize()
        print("schnorr", [round(x, 3) for x in eng.info()["last_kernel_ms"]], flush=True)
    assert (we.d_ok.cpu().numpy().astype(bool) == we.expect).all() and (ws.d_ok.cpu().numpy().astype(bool) == ws.expect).all()
