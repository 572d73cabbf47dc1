#!/usr/bin/env python3
"""All-distinct keys (every row on the per-signature ladder, k_ecmult<3>): n rows per call, resident inputs, `calls` calls back to back; prints rows/s.  Run under
rocprofv3 --kernel-trace --stats for the ladder kernel's own duration.  usage: python tools/cold_rows_probe.py [n] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lightning_amd import Engine, workload


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 196_608 * 4
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    os.environ["LAMD_CACHE"] = "0"   # as the bench's cold legs: no key-table cache
    eng = Engine(0)
    wk = workload.make_ecdsa(eng, n, seed=77, nkeys=n, publen=65, device="cuda:0")
    ok = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], ok)
    eng.synchronize()
    bad = int((ok.cpu().numpy().astype(bool) != wk.expect).sum())
    inf = eng.info()
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.verify_ecdsa_device(wk.dev[0], wk.dev[1], wk.dev[2], ok)
    eng.synchronize()
    dt = time.perf_counter() - t0
    print("all-distinct: %d rows x %d calls: %.1f M rows/s (%.3f ms per call), mismatches %d, rows on the ladder %d" % (n, calls, n * calls / dt / 1e6, 1e3 * dt / calls, bad, int(inf["last_cold_rows"])), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
