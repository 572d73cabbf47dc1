#!/usr/bin/env python3
"""lamd_served under load: C client processes (1, 4, 8, 16), each validating commitments of its own channels one request at a time -- 1 + 483 signatures per
request through lamd_verify_ecdsa_batch of the client library, the channels' keys recurring (20 channels per client, 60 passes over them) -- against the
same requests issued by ONE process straight into the engine.  Reports requests/s, signatures/s, per-request latency and how many requests the server merged."""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
PER, CH, PASSES = 484, 20, 60

CLIENT = r"""
import ctypes, os, sys, time, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import test_served as T
L = T._client()
rc, ctx = T._connect(L, sys.argv[2])
assert rc == 0, L.lamd_last_error(ctx)
d = np.load(sys.argv[3])
h, s, k, e = d["h"], d["s"], d["k"], d["e"]
n = len(h) // T.PER
ok = np.zeros(T.PER, np.uint8)
lat, bad = [], 0
for p in range(int(sys.argv[4])):
    for c in range(n):
        a = c * T.PER
        t = time.perf_counter()
        assert L.lamd_verify_ecdsa_batch(ctx, T.PER, h[a:a + T.PER].ctypes.data, s[a:a + T.PER].ctypes.data, k[a:a + T.PER].ctypes.data, 33, 33, ok.ctypes.data) == 0
        lat.append(time.perf_counter() - t)
        bad += int((ok != e[a:a + T.PER]).sum())
lat = np.sort(np.array(lat[2 * n:])) * 1e3     # (the first two passes are the keys' first and second sight)
print("RESULT %d %.4f %.4f %d" % (len(lat), lat[len(lat) // 2], lat[int(len(lat) * 0.99)], bad))
"""


def main():
    import torch
    import test_served as T
    from lightning_amd import Engine, workload
    eng = Engine(0)
    files = []
    tmp = "/tmp/served_probe"
    os.makedirs(tmp, exist_ok=True)
    NC = 16
    st = workload.make_commit_storm(eng, NC * CH, bip340_every=0, device="cuda:0")["ecdsa"]
    for c in range(NC):
        a, z = c * CH * PER, (c + 1) * CH * PER
        hs, ss, ks = (np.ascontiguousarray(x[a:z]) for x in st.cols)
        k33 = ks if ks.shape[1] == 33 else None
        assert k33 is not None
        f = "%s/rows%d.npz" % (tmp, c)
        np.savez(f, h=hs, s=ss, k=k33, e=st.expect[a:z].astype(np.uint8))
        files.append(f)
    # ---- in-process: one caller straight into the engine
    d = {k: v for k, v in np.load(files[0]).items()}   # (an NpzFile re-reads the archive on every access)
    lat = []
    for p in range(PASSES):
        for c in range(CH):
            a = c * PER
            t = time.perf_counter()
            got = eng.verify_ecdsa(d["h"][a:a + PER], d["s"][a:a + PER], d["k"][a:a + PER])
            lat.append(time.perf_counter() - t)
            assert np.array_equal(got.astype(np.uint8), d["e"][a:a + PER])
    lat = np.sort(np.array(lat[2 * CH:])) * 1e3
    print("in-process, one caller: %.0f requests/s (%.2f M signatures/s), p50 %.3f ms p99 %.3f ms" % (1e3 / lat.mean(), PER * 1e3 / lat.mean() / 1e6, lat[len(lat) // 2], lat[int(len(lat) * 0.99)]))
    eng.close()
    del eng
    sock = tmp + "/probe.sock"
    for linger in (0, 100):
        srv = T._start(sock, None, ["--linger-us", str(linger)] if linger else [])
        try:
            for nc in (1, 4, 8, 16):
                t0 = time.perf_counter()
                procs = [subprocess.Popen([sys.executable, "-c", CLIENT, ROOT, sock, files[i], str(PASSES)], stdout=subprocess.PIPE, text=True) for i in range(nc)]
                outs = [p.communicate(timeout=600)[0] for p in procs]
                wall = time.perf_counter() - t0
                res = [o.split("RESULT")[1].split() for o in outs]
                nreq = sum(int(r[0]) for r in res)
                p50 = np.median([float(r[1]) for r in res])
                p99 = max(float(r[2]) for r in res)
                bad = sum(int(r[3]) for r in res)
                # rate from the clients' own latencies (each client has one request in flight): requests/s = sum over clients of 1 / mean latency ~ nc / p50
                print("service, linger %3d us, %2d client processes: p50 %.3f ms p99 %.3f ms -> ~%.0f requests/s (%.2f M signatures/s), mismatches %d (wall %.1f s incl. start-up)" % (
                    linger, nc, p50, p99, nc * 1e3 / p50, nc * PER * 1e3 / p50 / 1e6, bad, wall))
            L = T._client()
            rc, ctx = T._connect(L, sock)
            stt = T.Stats()
            L.lamd_client_server_stats(ctx, ctypes.byref(stt))
            print("  server: %d requests, %d engine calls, %d requests in merged calls, largest merge %d" % (stt.requests, stt.engine_calls, stt.merged_requests, stt.largest_merge_requests))
            L.lamd_shutdown(ctx)
        finally:
            T._stop(srv)


if __name__ == "__main__":
    main()
