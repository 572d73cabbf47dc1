#!/usr/bin/env python3
"""Diagnostic: the bench's un-synchronised loop (calls rotate over the lanes), then a per-row look at every mismatching verdict."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lightning_amd import Engine, workload  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
idle = Engine(0) if os.environ.get('DIAG_IDLE_ENGINE') == '1' else None
with Engine(0) as eng:
    we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65)
    ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536)
    # every step gets its own verdict buffers so that a wrong step cannot be overwritten by a later correct one
    oke = [torch.full((n,), 9, dtype=torch.uint8, device="cuda:0") for _ in range(steps)]
    oks = [torch.full((n,), 9, dtype=torch.uint8, device="cuda:0") for _ in range(steps)]
    torch.cuda.synchronize()
    shared, poison, timing = (os.environ.get(k, "0") == "1" for k in ("DIAG_SHARED", "DIAG_POISON", "DIAG_TIMING"))
    if shared:
        oke, oks = [oke[0]] * steps, [oks[0]] * steps
    if timing:
        eng.set_timing(True)
    tstream = torch.cuda.current_stream().cuda_stream
    eng.auto_order = False
    import time
    for rep in range(3):
        torch.cuda.synchronize(); eng.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if poison and k == steps - 1:
                oke[k].fill_(7); oks[k].fill_(7)
                eng.wait_stream(tstream)
            eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], oke[k])
            eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], oks[k])
        eng.synchronize()
        torch.cuda.synchronize()
        print("rep %d: %.3f ms per step, %.1f M verifies/s" % (rep, (time.perf_counter() - t0) / steps * 1e3, 2 * n * steps / (time.perf_counter() - t0) / 1e6), flush=True)
        for k in (range(steps) if not shared else [steps - 1]):
            for kind, w, ok in (("ecdsa", we, oke[k]), ("schnorr", ws, oks[k])):
                got = ok.cpu().numpy()
                bad = np.nonzero(got != w.expect.astype(np.uint8))[0]
                if len(bad):
                    print(json.dumps({"rep": rep, "step": k, "kind": kind, "mismatches": int(len(bad)), "values": dict(collections.Counter(int(v) for v in got[bad])),
                                      "expected": dict(collections.Counter(int(v) for v in w.expect[bad])),
                                      "classes": dict(collections.Counter(int(c) for c in w.classes[bad])), "first_rows": [int(b) for b in bad[:8]],
                                      "row_mod_16384": dict(collections.Counter(int(b) % 16384 for b in bad).most_common(4)),
                                      "span": [int(bad.min()), int(bad.max())]}), flush=True)
                ok.fill_(9)
        torch.cuda.synchronize()
    print("finished")
