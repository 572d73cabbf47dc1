#!/usr/bin/env python3
"""One commitment_signed as ONE call under a kernel + memory-copy trace: lamd_check_commitment_signed on a synthetic 1 + 483-signature commitment --
first sight of its keys, the learning call, then R calls under cached keys (marker 10 | ... | marker 11) -- to show what the call is on the device:
one small copy (the templates, pinned -> HBM) and two launches (k_txsig_tx_hash, k_small_verify)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import orc
import test_gpu_commitment as C
from lightning_amd import Engine

orc.lib()
eng = Engine(0)
rnd = random.Random(7)
ctx, fund, csig, htxs, hkey, hsigs, tys = C._synthetic_commitment(orc, rnd, 483, False)


def marker(k):
    torch.cuda.synchronize()
    torch.zeros(1000 * 256 * k, dtype=torch.float32, device="cuda:0").fill_(1.0)
    torch.cuda.synchronize()


cc = eng.commitment_call(ctx, fund, csig, 1, htxs, hkey, hsigs, tys)   # arguments marshalled once: the clock holds the C call only
ts = []
for rep in range(3):
    t = time.perf_counter()
    fb, ok = cc()
    ts.append((time.perf_counter() - t) * 1e3)
    assert fb == -1 and ok.all()
print("first sight %.3f ms, learning call %.3f ms, third call %.3f ms" % tuple(ts))
marker(10)
ts = []
for rep in range(40):
    t = time.perf_counter()
    fb, ok = cc()
    ts.append((time.perf_counter() - t) * 1e3)
    assert fb == -1
marker(11)
ts.sort()
print("cached keys: p50 %.3f ms, min %.3f ms, p99 %.3f ms over 40 calls of lamd_check_commitment_signed (1 + 483 signatures)" % (ts[20], ts[0], ts[39]))
eng.close()
