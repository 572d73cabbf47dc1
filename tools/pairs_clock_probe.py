#!/usr/bin/env python3
"""Experiment build -DLAMD_PAIRS_CLOCK: where the lanes of k_ecmult_keyed_pairs spend their time.  usage: LAMD_LIB_PATH=tools/variants/liblightning_amd_pclk.so python tools/pairs_clock_probe.py"""
import ctypes
import os
import sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["LAMD_CACHE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_amd import engine as E, workload

eng = E.Engine(0)
lib = eng._lib
n = 1000000
wk = workload.make_ecdsa(eng, n, nkeys=65536, publen=65)
for _ in range(eng.info()["lanes"] + 1):
    eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], wk.d_ok)
eng.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.lamd_debug_pairs_clock(out, 1)
for _ in range(4):
    eng.set_timing(True)
    eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], wk.d_ok)
    eng.synchronize()
    lanes = range(eng.info()["lanes"])
    ms = sum(eng.info(l)["keyed_ecmult_ms_sum"][0] for l in lanes) / max(1, sum(eng.info(l)["keyed_ecmult_launches"][0] for l in lanes))
    lib.lamd_debug_pairs_clock(out, 1)
    nl = 768 * 256
    print("launch %.3f ms | per lane mean (ms): pass 1 %.3f, inversion %.3f, pass 2 %.3f | slowest lane: total %.3f, pass 1 %.3f, pass 2 %.3f | first start -> last end %.3f ms" % (
        ms, out[0] / nl / 1e5, out[1] / nl / 1e5, out[2] / nl / 1e5, out[3] / 1e5, out[6] / 1e5, out[7] / 1e5, (out[5] - out[4]) / 1e5), flush=True)
