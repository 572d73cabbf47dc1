#!/usr/bin/env python3
"""Host side of the batched gossip ingest by itself (csrc/gossip_ingest.cpp): a flood of structurally valid channel_announcements,
their txout replies and channel_updates with a verification back end that answers "all good" at once -- what the C++ host logic
(framing, filters, maps, store records, events) costs per message when the GPU is infinitely fast.  Runs without a GPU.
usage: ingest_host_bench.py [n_channels] [updates_per_channel]"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_amd import gossipd

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_cann = int(args[0]) if len(args) > 0 else 100_000
per = int(args[1]) if len(args) > 1 else 4
if "--with-engine" in sys.argv:   # the same host logic inside a process that holds a GPU context (pinned memory, the runtime's threads)
    from lightning_amd import Engine
    _eng = Engine(0)
rng = np.random.default_rng(0xC1A00004)
chain = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
n_nodes = 15000
nodes = rng.integers(0, 256, (n_nodes, 33), dtype=np.uint8)
nodes[:, 0] = 2 + (nodes[:, 0] & 1)
order = np.lexsort(nodes.T[::-1])          # lexicographic order of the node ids
nodes = nodes[order]
now = 1_700_000_000

# channel_announcement: type 256 | 4 x 64-byte signatures | flen = 0 | chain_hash | scid | node_id_1 < node_id_2 | bitcoin_key_1 | bitcoin_key_2
cann = np.zeros((n_cann, 432), dtype=np.uint8)
cann[:, 0:2] = (1, 0)
cann[:, 2:258] = rng.integers(0, 256, (n_cann, 256), dtype=np.uint8)
cann[:, 2:258:32] &= 0x7F                 # r, s below n
cann[:, 260:292] = np.frombuffer(chain, dtype=np.uint8)
scid = (np.arange(n_cann, dtype=np.uint64) + 1) | (np.uint64(600_000) << np.uint64(40))
cann[:, 292:300] = scid.astype(">u8").view(np.uint8).reshape(n_cann, 8)
a = rng.integers(0, n_nodes, n_cann)
b = (a + 1 + rng.integers(0, n_nodes - 1, n_cann)) % n_nodes
lo, hi = np.minimum(a, b), np.maximum(a, b)
cann[:, 300:333] = nodes[lo]
cann[:, 333:366] = nodes[hi]
bk = rng.integers(0, 256, (n_cann, 66), dtype=np.uint8)
bk[:, 0] = 2 + (bk[:, 0] & 1)
bk[:, 33] = 2 + (bk[:, 33] & 1)
cann[:, 366:432] = bk
# channel_update: type 258 | signature | chain_hash | scid | timestamp | message_flags | channel_flags | cltv | htlc_min | base | prop | htlc_max
n_cupd = n_cann * per
cupd = np.zeros((n_cupd, 138), dtype=np.uint8)
cupd[:, 0:2] = (1, 2)
cupd[:, 2:66] = rng.integers(0, 256, (n_cupd, 64), dtype=np.uint8)
cupd[:, 2:66:32] &= 0x7F
cupd[:, 66:98] = np.frombuffer(chain, dtype=np.uint8)
ch = np.repeat(np.arange(n_cann), per)
cupd[:, 98:106] = cann[ch, 292:300]
ts = (now - 1000 + (np.arange(n_cupd) % per) // 2).astype(">u4")
cupd[:, 106:110] = ts.view(np.uint8).reshape(n_cupd, 4)
cupd[:, 110] = 1
cupd[:, 111] = np.arange(n_cupd) & 1      # direction
cupd[:, 113] = 40
cupd[:, 130:138] = (10**9 // 2**np.arange(56, -8, -8) % 256).astype(np.uint8)
peer = bytes(nodes[7])
spk = []
for i in range(n_cann):
    k1, k2 = sorted([bytes(cann[i, 366:399]), bytes(cann[i, 399:432])])
    spk.append(b"\x00\x20" + hashlib.sha256(b"\x52\x21" + k1 + b"\x21" + k2 + b"\x52\xae").digest())
spk_blob = np.frombuffer(b"".join(spk) + b"\x00", dtype=np.uint8)
spk_off = np.arange(n_cann + 1, dtype=np.uint64) * 34
sats = np.full(n_cann, 1_000_000, dtype=np.uint64)
cann_off = np.arange(n_cann + 1, dtype=np.uint64) * 432
cupd_off = np.arange(n_cupd + 1, dtype=np.uint64) * 138
cann_blob = np.concatenate([cann.reshape(-1), np.zeros(1, dtype=np.uint8)])
cupd_blob = np.concatenate([cupd.reshape(-1), np.zeros(1, dtype=np.uint8)])


# node_announcement: type 257 | signature | flen = 0 | timestamp | node_id | rgb | alias | addrlen = 0 -- three per node that has a channel, rising timestamps,
# interleaved over the nodes (a run of plain node_announcements of known nodes: apply_nann_run)
used = np.unique(np.concatenate([lo, hi]))
n_nann = 3 * len(used)
nann = np.zeros((n_nann, 2 + 64 + 2 + 4 + 33 + 3 + 32 + 2), dtype=np.uint8)
nann[:, 0:2] = (1, 1)
nann[:, 2:66] = rng.integers(0, 256, (n_nann, 64), dtype=np.uint8)
nann[:, 2:66:32] &= 0x7F
who = np.tile(used, 3)
nann[:, 68:72] = (now - 500 + np.repeat(np.arange(3), len(used))).astype(">u4").view(np.uint8).reshape(n_nann, 4)
nann[:, 72:105] = nodes[who]
nann[:, 108:140] = 0x41
nann_off = np.arange(n_nann + 1, dtype=np.uint64) * nann.shape[1]
nann_blob = np.concatenate([nann.reshape(-1), np.zeros(1, dtype=np.uint8)])


def all_good_sig(_u, n, msgs, off, ids, verdict):
    ctypes.memset(verdict, 0, n)
    return 0


def all_good_key(_u, n, pub, ok):
    ctypes.memset(ok, 1, n)
    return 0


best = [0.0, 0.0, 0.0, 0.0]
for rep in range(5):
    ing = gossipd.GossipIngest(None, chain, bytes(nodes[3]), 700_000, now, prune_interval=0xFFFFFFFF, backend=(lambda *a: [], lambda *a: []), collect_events=False)
    be = (gossipd.SIGCHECK_FN(all_good_sig), gossipd.KEYPARSE_FN(all_good_key))
    ing._L.lamd_gossipd_set_backend(ing._g, be[0], be[1], None)
    t1 = time.perf_counter()
    ing.push_batch(peer, cann_blob, cann_off)
    ing.process()
    t2 = time.perf_counter()
    ing.txout_reply_batch(scid, sats, spk_blob, spk_off)
    t3 = time.perf_counter()
    ing.push_batch(peer, cupd_blob, cupd_off)
    t3b = time.perf_counter()
    ing.process()
    t4 = time.perf_counter()
    if os.environ.get("LAMD_INGEST_PROFILE"):
        print("[bench] updates: push_batch %.1f ms, process %.1f ms" % ((t3b - t3) * 1e3, (t4 - t3b) * 1e3), file=sys.stderr)
    ing.push_batch(peer, nann_blob, nann_off)
    ing.process()
    t5 = time.perf_counter()
    st = ing.stats()
    ing.close()
    assert st["channels"] == n_cann and st["store_records"] == 2 * n_cann + n_cupd + n_nann + 1 and st["verified_sigs"] == 4 * n_cann + n_cupd + n_nann, st
    assert st["nodes"] == len(used)
    best = [max(b, v) for b, v in zip(best, (n_cann / (t2 - t1), n_cann / (t3 - t2), n_cupd / (t4 - t3), n_nann / (t5 - t4)))]
print("host logic only (verification answers at once), best of 5 per phase: %.2f M channel_announcements/s, %.2f M txout replies/s, "
      "%.2f M channel_updates/s, %.2f M node_announcements/s (%d channels, %d updates, %d node_announcements of %d nodes, %d of them applied by all cores)" % (
          best[0] / 1e6, best[1] / 1e6, best[2] / 1e6, best[3] / 1e6, n_cann, n_cupd, n_nann, len(used), st["run_nodes"]))
