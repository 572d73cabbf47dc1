#!/usr/bin/env python3
"""Diagnostic: run the cfg2/cfg3 batches call by call on engines with / without the key-table cache and with both
occupancy variants of the bare-formula kernel; for every call report the rows whose verdict differs from construction
(value written, corruption class, whether the same rows fail again)."""
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
res = []
for waves in ("3", "4"):
    for cache in ("0", "1"):
        os.environ["LAMD_KEYED_WAVES"], os.environ["LAMD_CACHE"] = waves, cache
        with Engine(0) as eng:
            we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65)
            ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536)
            prev = {}
            for call in range(4):
                for kind, w, f in (("ecdsa", we, eng.verify_ecdsa_device), ("schnorr", ws, eng.verify_schnorr_device)):
                    w.d_ok.fill_(9)
                    torch.cuda.synchronize()
                    f(w.dev[0], w.dev[1], w.dev[2], w.d_ok)
                    if call == 3:     # last round: no sync between the two kinds (lanes overlap)
                        continue
                    eng.synchronize()
                    got = w.d_ok.cpu().numpy()
                    bad = np.nonzero(got != w.expect.astype(np.uint8))[0]
                    inf = eng.info()
                    r = {"waves": waves, "cache": cache, "call": call, "kind": kind, "mismatches": int(len(bad)),
                         "values": dict(collections.Counter(int(v) for v in got[bad])), "classes": dict(collections.Counter(int(c) for c in w.classes[bad])),
                         "first_rows": [int(b) for b in bad[:6]], "same_as_prev_call": bool(kind in prev and np.array_equal(prev[kind], bad)),
                         "hits": inf["last_cache_hits"], "cold": inf["last_cold_rows"], "new": inf["last_new_tables"], "suspect": inf["last_suspect_rows"],
                         "ms": [round(x, 3) for x in inf["last_kernel_ms"]]}
                    prev[kind] = bad
                    res.append(r)
                    print(json.dumps(r), flush=True)
            eng.synchronize()
            for kind, w in (("ecdsa", we), ("schnorr", ws)):
                got = w.d_ok.cpu().numpy()
                bad = np.nonzero(got != w.expect.astype(np.uint8))[0]
                print(json.dumps({"waves": waves, "cache": cache, "call": "overlapped", "kind": kind, "mismatches": int(len(bad)),
                                  "values": dict(collections.Counter(int(v) for v in got[bad]))}), flush=True)
