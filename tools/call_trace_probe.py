#!/usr/bin/env python3
"""Workloads for a rocprofv3 --kernel-trace (--memory-copy-trace) timeline, phases separated by marker kernels (a torch fill whose element count
is the marker: 1000 * 256 * k elements -> Grid/Workgroup sizes that nothing else launches):
  gossip : marker 1 | cann shard (1/8 of configs[3] by cost) x R, a synchronise after each | marker 2 | cupd shard x R | marker 3 | whole job x 2 | marker 4
  stream : marker 5 | resident cold loop, S steps | marker 6 | host->host in-place cold streaming loop, S steps, 8 flushes in flight | marker 7
Prints the host-side wall time of every repetition."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from lightning_amd import Engine, sharding, workload

mode = sys.argv[1] if len(sys.argv) > 1 else "gossip"
R = int(os.environ.get("PROBE_REPS", "5"))
S = int(os.environ.get("PROBE_STEPS", "12"))
dev = "cuda:0"


def marker(k):
    torch.cuda.synchronize()
    torch.zeros(1000 * 256 * k, dtype=torch.float32, device=dev).fill_(1.0)
    torch.cuda.synchronize()


if mode == "gossip":
    eng = Engine(0)
    g = workload.make_gossip(eng, 500_000, 2_000_000, n_nodes=15000, device=dev)
    gw = sharding.gossip_weights(g.msgs, g.off)
    b = sharding.shard_bounds(g.n, 8, None, gw)

    def shard(lo, hi):
        rb = (g.d_rowbase[lo:hi + 1] - g.d_rowbase[lo]).contiguous()
        rows = int(g.rowbase[hi] - g.rowbase[lo])
        d_v = torch.zeros(hi - lo, dtype=torch.int8, device=dev)
        torch.cuda.synchronize()

        def one():
            t = time.perf_counter()
            eng.sigcheck_gossip_device(hi - lo, g.d_msgs, g.d_off[lo:hi + 1], g.d_ids[lo:hi], rb, rows, d_v)
            eng.synchronize()
            return (time.perf_counter() - t) * 1e3
        return one, d_v
    whole, _ = shard(0, g.n)
    for _ in range(eng.info()["lanes"] + 1):
        whole()
    for name, k, mk in (("cann shard 0", 0, 1), ("cupd shard 7", 7, 2)):
        one, d_v = shard(int(b[k]), int(b[k + 1]))
        for _ in range(eng.info()["lanes"]):
            one()
        marker(mk)
        ts = [one() for _ in range(R)]
        bad = int((d_v.cpu().numpy() != g.expect[int(b[k]):int(b[k + 1])]).sum())
        print("%s: %d messages, host wall ms %s, mismatches %d" % (name, int(b[k + 1] - b[k]), " ".join("%.3f" % t for t in ts), bad))
    marker(3)
    print("whole job: host wall ms", " ".join("%.3f" % whole() for _ in range(2)))
    marker(4)
elif mode == "gossip8":
    # one 1/8 shard of configs[3] cut PER MESSAGE KIND (sharding.segment_bounds, as bench.py and lamd_multi cut it): range 0 of the announcements and
    # range 0 of the updates as two asynchronous calls, one synchronise.  marker 10 | the shard x R | marker 11
    eng = Engine(0)
    g = workload.make_gossip(eng, 500_000, 2_000_000, n_nodes=15000, device=dev)
    sb = sharding.segment_bounds([0, g.n_cann, g.n], 8, sharding.gossip_weights(g.msgs, g.off))
    calls = []
    for s_ in range(2):
        lo, hi = int(sb[s_, 0]), int(sb[s_, 1])
        rb = (g.d_rowbase[lo:hi + 1] - g.d_rowbase[lo]).contiguous()
        calls.append((lo, hi, rb, int(g.rowbase[hi] - g.rowbase[lo]), torch.zeros(hi - lo, dtype=torch.int8, device=dev)))
    torch.cuda.synchronize()

    SPANS = os.environ.get("PROBE_SPANS", "0") == "1"      # ... or as ONE spans call over both ranges (the form bench.py runs)
    if SPANS:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sp = bench.gossip_spans(g, [(lo, hi) for lo, hi, *_ in calls], dev)
        torch.cuda.synchronize()

    def one():
        t = time.perf_counter()
        if SPANS:
            eng.sigcheck_gossip_spans_device(sp[0], g.d_msgs, sp[1], sp[2], sp[3], sp[4], sp[5], sp[6])
        else:
            for lo, hi, rb, rows, d_v in calls:
                eng.sigcheck_gossip_device(hi - lo, g.d_msgs, g.d_off[lo:hi + 1], g.d_ids[lo:hi], rb, rows, d_v)
        eng.synchronize()
        return (time.perf_counter() - t) * 1e3
    for _ in range(eng.info()["lanes"] + 2):
        one()
    marker(10)
    ts = [one() for _ in range(R)]
    marker(11)
    bad = int((sp[6].cpu().numpy() != np.concatenate([g.expect[lo:hi] for lo, hi, *_ in calls])).sum()) if SPANS else \
        sum(int((d_v.cpu().numpy() != g.expect[lo:hi]).sum()) for lo, hi, rb, rows, d_v in calls)
    print("per-kind 1/8 shard: %s messages, host wall ms %s, mismatches %d" % ([hi - lo for lo, hi, *_ in calls], " ".join("%.3f" % t for t in ts), bad))
elif mode == "storm":
    # one 1/8 shard of BASELINE configs[4] (1 250 commitments: 937 ECDSA + 313 BIP-340) through the streaming queue as bench.py's strong-scaling sweep
    # runs it; marker 8 | the shard x R (host wall printed per repetition, with the host time spent inside queue_*_batch / flush / wait) | marker 9
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    eng = Engine(0)
    st = workload.make_commit_storm(eng, 10_000, device=dev)
    per = st["per"]
    bb = {kind: sharding.shard_bounds(st[kind].n, 8, [per] * (st[kind].n // per)) for kind in ("ecdsa", "schnorr")}
    tq = {"queue": 0.0, "flush": 0.0, "wait": 0.0}
    real = (eng.queue_ecdsa_batch, eng.queue_schnorr_batch, eng.flush, eng.wait)

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            r = fn(*a, **k)
            tq[name] += time.perf_counter() - t
            return r
        return w
    eng.queue_ecdsa_batch, eng.queue_schnorr_batch = timed("queue", real[0]), timed("queue", real[1])
    eng.queue_ecdsa_batch_inplace, eng.queue_schnorr_batch_inplace = timed("queue", eng.queue_ecdsa_batch_inplace), timed("queue", eng.queue_schnorr_batch_inplace)
    eng.flush, eng.wait = timed("flush", real[2]), timed("wait", real[3])
    depth = min(8, eng.info()["queue_sets"] - 1)
    INPLACE = os.environ.get("PROBE_INPLACE", "0") == "1" and bench.pin_storm(eng, st)     # rows queued where they are (pinned columns)
    GRP = int(os.environ.get("PROBE_GROUP", "256"))
    _ss = bench.stream_shard
    bench.stream_shard = lambda e, s_, b_, k, g_, d, f=0: _ss(e, s_, b_, k, GRP * per, d, f, inplace=INPLACE)
    for _ in range(3):
        bench.stream_shard(eng, st, bb, 0, 256 * per, depth, bench.storm_first_flush(per))
    marker(8)
    for r in range(R):
        for k in tq:
            tq[k] = 0.0
        t = time.perf_counter()
        got = bench.stream_shard(eng, st, bb, 0, 256 * per, depth, bench.storm_first_flush(per))
        dt = time.perf_counter() - t
        bad = sum(int((got[kind].astype(bool) != st[kind].expect[int(bb[kind][0]):int(bb[kind][1])]).sum()) for kind in got)
        print("storm shard 0 of 8: %.3f ms host wall (queue_*_batch %.3f, flush %.3f, wait %.3f ms), mismatches %d" % (dt * 1e3, tq["queue"] * 1e3, tq["flush"] * 1e3, tq["wait"] * 1e3, bad))
    marker(9)
else:
    os.environ["LAMD_CACHE"] = "0"
    eng = Engine(0)
    n = 1_000_000
    we = workload.make_ecdsa(eng, n, seed=workload.SEED_CFG2, nkeys=65536, publen=65, device=dev)
    ws = workload.make_schnorr(eng, n, seed=workload.SEED_CFG3, nkeys=65536, device=dev)
    eng.auto_order = False

    def resident(steps):
        torch.cuda.synchronize(); eng.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
            eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok)
        eng.synchronize()
        return time.perf_counter() - t

    def stream(steps, filled, depth=8):
        pend, bad = [], 0
        got = []
        t = time.perf_counter()
        for r in range(steps):
            for wl in (we, ws):
                _, a, b_, c = eng.queue_reserve(n, 65 if wl is we else 32)
                if a.ctypes.data not in filled:
                    filled.add(a.ctypes.data)
                    a[:] = wl.cols[0]
                    if wl is we:
                        b_[:], c[:] = wl.cols[1], wl.cols[2]
                    else:
                        c[:], b_[:] = wl.cols[1], wl.cols[2]
                eng.flush()
                pend.append(wl)
                if len(pend) == depth:
                    got.append((eng.wait(cap=n), pend.pop(0)))
        while pend:
            got.append((eng.wait(cap=n), pend.pop(0)))
        dt = time.perf_counter() - t
        return dt, sum(int((v != wl.expect).sum()) for v, wl in got)
    resident(6)
    seen = set()
    stream(9, seen)
    marker(5)
    dt = resident(S)
    print("resident cold loop: %.1f M verifies/s (%.3f ms per step)" % (2 * n * S / dt / 1e6, dt / S * 1e3))
    marker(6)
    dt, bad = stream(S, seen)
    print("host->host in-place cold loop: %.1f M verifies/s (%.3f ms per step), mismatches %d" % (2 * n * S / dt / 1e6, dt / S * 1e3, bad))
    marker(7)
eng.close()
