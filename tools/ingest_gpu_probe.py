#!/usr/bin/env python3
"""The gossip ingest flood of bench.py by itself, with the ingest's own phase clock (LAMD_INGEST_PROFILE=1): channel_announcements,
their txout replies, channel_updates through csrc/gossip_ingest.cpp around the device calls.
usage: ingest_gpu_probe.py [n_announcements=100000] [n_updates=400000]"""
import hashlib
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LAMD_INGEST_PROFILE", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from lightning_amd import Engine, workload
from lightning_amd.gossipd import GossipIngest

eng = Engine(0)
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
NU = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
g = workload.make_gossip(eng, NA, NU, n_nodes=15000, corrupt_frac=0.01, device="cuda:0")
chain = bytes(g.msgs[260:292])
peer = bytes(g.ids[g.n_cann])
cann_blob, cann_off = g.msgs[:int(g.off[g.n_cann]) + 1], g.off[:g.n_cann + 1].copy()
cupd_blob = g.msgs[int(g.off[g.n_cann]):]
cupd_off = (g.off[g.n_cann:] - g.off[g.n_cann]).copy()
spk = []
for i in range(g.n_cann):
    m = g.msgs[int(g.off[i]):int(g.off[i + 1])]
    k1, k2 = sorted([bytes(m[366:399]), bytes(m[399:432])])
    spk.append(b"\x00\x20" + hashlib.sha256(b"\x52\x21" + k1 + b"\x21" + k2 + b"\x52\xae").digest())
spk_blob = np.frombuffer(b"".join(spk) + b"\x00", dtype=np.uint8)
spk_off = (np.arange(g.n_cann + 1, dtype=np.uint64) * 34)
scids = np.arange(g.n_cann, dtype=np.uint64)
sats = np.full(g.n_cann, 1_000_000, dtype=np.uint64)
for rep in range(3):
    with GossipIngest(eng, chain, peer, 700_000, 1 << 32, prune_interval=0xFFFFFFFF, collect_events=False) as ing:
        t1 = time.perf_counter()
        ing.push_batch(peer, cann_blob, cann_off)
        tp = time.perf_counter()
        ing.process()
        t2 = time.perf_counter()
        ing.txout_reply_batch(scids, sats, spk_blob, spk_off)
        t3 = time.perf_counter()
        tq = t3
        for o in range(0, g.n_cupd, 500_000):      # connectd's queue bound: the updates arrive as queues of 500 k
            e_ = min(g.n_cupd, o + 500_000)
            tq0 = time.perf_counter()
            ing.push_batch(peer, cupd_blob[int(cupd_off[o]):int(cupd_off[e_]) + 1], (cupd_off[o:e_ + 1] - cupd_off[o]).copy())
            tq += time.perf_counter() - tq0
            ing.process()
        t4 = time.perf_counter()
        st = ing.stats()
    print("rep %d: announcements %.2f M/s (push %.1f ms, process %.1f ms), replies %.2f M/s, updates %.2f M/s (push %.1f ms, process %.1f ms); channels %d late %d"
          % (rep, g.n_cann / (t2 - t1) / 1e6, (tp - t1) * 1e3, (t2 - tp) * 1e3, g.n_cann / (t3 - t2) / 1e6, g.n_cupd / (t4 - t3) / 1e6, (tq - t3) * 1e3, (t4 - tq) * 1e3,
             st["channels"], st["late_verifies"]), flush=True)
eng.close()
