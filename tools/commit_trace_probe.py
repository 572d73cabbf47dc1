#!/usr/bin/env python3
"""One commitment_signed as ONE call under a kernel + memory-copy trace: lamd_check_commitment_signed on a synthetic 1 + 483-signature commitment --
first sight of its keys, the learning call, then R calls under cached keys (marker 10 | ... | marker 11) -- to show what the call is on the device:
two launches (k_txsig_tx_hash, k_small_verify), no copy command."""
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


ts = []
for rep in range(3):
    t = time.perf_counter()
    fb, ok = eng.check_commitment_signed(ctx, fund, csig, 1, htxs, hkey, hsigs, tys)
    ts.append((time.perf_counter() - t) * 1e3)
    assert fb == -1 and ok.all()
print("first sight %.3f ms, learning call %.3f ms, third call %.3f ms (Python wrapper included)" % tuple(ts))
marker(10)
ts = []
for rep in range(20):
    t = time.perf_counter()
    fb, ok = eng.check_commitment_signed(ctx, fund, csig, 1, htxs, hkey, hsigs, tys)
    ts.append((time.perf_counter() - t) * 1e3)
    assert fb == -1
marker(11)
ts.sort()
print("cached keys: p50 %.3f ms, min %.3f ms over 20 calls (the Python wrapper flattens 484 templates per call: ~0.5 ms of that is Python)" % (ts[10], ts[0]))
eng.close()
