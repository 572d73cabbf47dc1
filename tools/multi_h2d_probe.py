#!/usr/bin/env python3
"""lamd_multi_verify_ecdsa_batch from PAGEABLE caller memory on the devices of this box: rows/s through the runtime's staged hipMemcpy (the
per-device pinned ring this probe compared it with in round 5 lost and was deleted in round 6: lamd_multi.cpp).  Also checks the NULL-node_ids refusal."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    import torch
    import orc
    from lightning_amd import _ffi
    lib = _ffi.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    ndev = min(torch.cuda.device_count(), 8)
    base = 65536
    h, s, p, c, e = orc.gen_ecdsa_edge_batch(0xABCD, base, 65, 8)
    reps = (n + base - 1) // base
    H, S, P = (np.ascontiguousarray(np.tile(x, (reps, 1))[:n]) for x in (h, s, p))
    E = np.tile(e, reps)[:n]
    if os.environ.get("PROBE_PINNED_CALLER", "0") == "1":   # the caller's rows in page-locked memory (torch's pinned allocator): no staging at all
        keep = []
        def pin(a):
            t = torch.empty(a.shape, dtype=torch.uint8, pin_memory=True)
            t.numpy()[:] = a
            keep.append(t)
            return t.numpy()
        H, S, P = pin(H), pin(S), pin(P)
    m = ctypes.c_void_p()
    assert lib.lamd_multi_init(ctypes.byref(m), None, ndev) == 0, lib.lamd_multi_last_error(m)
    ok = np.zeros(n, np.uint8)
    ts = []
    for it in range(6):
        ok[:] = 7
        t = time.perf_counter()
        rc = lib.lamd_multi_verify_ecdsa_batch(m, n, H.ctypes.data, S.ctypes.data, P.ctypes.data, 65, 65, 1, ok.ctypes.data)
        ts.append(time.perf_counter() - t)
        assert rc == 0, lib.lamd_multi_last_error(m)
        assert np.array_equal(ok, E), int((ok != E).sum())
    # a channel_update without node ids is refused, not read through a placeholder
    msg = bytes([1, 2]) + bytes(136)
    blob = np.frombuffer(msg + b"\x00", dtype=np.uint8).copy()
    off = np.array([0, len(msg)], dtype=np.uint64)
    v = np.zeros(1, np.int8)
    rc = lib.lamd_multi_sigcheck_gossip_batch(m, 1, blob.ctypes.data, off.ctypes.data, None, v.ctypes.data)
    print("pinned=%s devices=%d rows=%d: best %.1f M ECDSA-65/s from pageable memory (%.2f ms; calls: %s) | NULL node_ids rc=%d (%s)" % (
        "runtime-staged", ndev, n, n / min(ts[1:]) / 1e6, min(ts[1:]) * 1e3, " ".join("%.1f" % (x * 1e3) for x in ts), rc,
        lib.lamd_multi_last_error(m).decode()))
    lib.lamd_multi_shutdown(m)


if __name__ == "__main__":
    main()
