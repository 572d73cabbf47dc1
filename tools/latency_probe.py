#!/usr/bin/env python3
"""Latency of small host-buffer calls (submit -> verdicts in host memory): 1 ... 4096 rows (PROBE_SIZES), key known to the cache or not, both
signature kinds, with the one-launch path (k_small_verify) or, under LAMD_SMALL_KERNEL=0, the general path; a commitment_signed (1 + 483 rows
under one cached htlc key)."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_amd import Engine, workload

eng = Engine(0)
we = workload.make_ecdsa(eng, 200_000, seed=7, nkeys=64, publen=33, device="cuda:0", invalid_frac=0.1)
ws = workload.make_schnorr(eng, 200_000, seed=8, nkeys=64, device="cuda:0", invalid_frac=0.1)
# make the 64 keys known to the cache (a large call builds and publishes their tables)
assert (eng.verify_ecdsa(*we.cols) == we.expect).all() and (eng.verify_schnorr(*ws.cols) == ws.expect).all()
eng.synchronize()
cold = workload.make_ecdsa(eng, 4096, seed=9, nkeys=1 << 40, publen=33, device="cuda:0", group=1, invalid_frac=0.1)   # every row its own key
# (the 10 % invalid rows of `we` include damaged and foreign KEYS, which no cache knows: a call that carries one waits for that row's ladder.
# `wc`: the same keys, every row valid -- a call whose keys are all cached)
wc = workload.make_ecdsa(eng, 200_000, seed=7, nkeys=64, publen=33, device="cuda:0", invalid_frac=0.0)


def p50(fn, reps):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts = np.sort(np.array(ts[reps // 10:])) * 1e3
    return ts[len(ts) // 2], ts[int(len(ts) * 0.99)]


print("LAMD_SMALL_KERNEL =", os.environ.get("LAMD_SMALL_KERNEL", "1 (default)"))
for bs in [int(x) for x in os.environ.get("PROBE_SIZES", "1,8,64,484,1024,4096").split(",")]:
    for name, wl, fn in (("ecdsa33 cached key", we, eng.verify_ecdsa), ("ecdsa33 all keys cached", wc, eng.verify_ecdsa), ("schnorr cached key", ws, eng.verify_schnorr),
                         ("ecdsa33 unknown key", cold, eng.verify_ecdsa)):
        k = [0]

        def call():
            o = (k[0] * bs) % (wl.n - bs)
            k[0] += 1
            got = fn(*[c[o:o + bs] for c in wl.cols])
            assert (got == wl.expect[o:o + bs]).all()
        if (bs > 256 and wl is cold) or (bs <= 64 and wl is wc):
            continue          # (4096 fresh keys per call: the workload holds 4096 rows)
        a, b = p50(call, 300 if bs <= 64 else 100)
        print("  %4d rows, %-24s p50 %.3f ms  p99 %.3f ms" % (bs, name, a, b))
# a key the latency path itself learnt (ladder at first sight, table built at the second: such keys take the 10-tooth comb)
one = workload.make_ecdsa(eng, 256, seed=11, nkeys=1, publen=33, device="cuda:0", invalid_frac=0.0)
k = [0]


def call1():
    o = k[0] % 256
    k[0] += 1
    assert eng.verify_ecdsa(*[c[o:o + 1] for c in one.cols])[0]
for _ in range(4):
    call1()
a, b = p50(call1, 300)
print("     1 rows, ecdsa33 key learnt one call at a time  p50 %.3f ms  p99 %.3f ms  (comb teeth %s)" % (a, b, eng.info()["last_keyed"]))
# gossip, one message per call as an unmodified gossipd does it (sigcheck_channel_announcement / _update, gossmap_manage.c:687,924): the nodes recur
g = workload.make_gossip(eng, 300, 300, n_nodes=20, device="cuda:0", corrupt_frac=0.05)
gm = [bytes(g.msgs[int(g.off[i]):int(g.off[i + 1])]) for i in range(g.n)]
gi = [bytes(g.ids[i]) for i in range(g.n)]
for name, lo in (("channel_announcement (4 signatures)", 0), ("channel_update", g.n_cann)):
    kk = [0]

    def gcall():
        i = lo + kk[0] % 300
        kk[0] += 1
        assert int(eng.sigcheck_gossip([gm[i]], [gi[i]])[0]) == int(g.expect[i]), i
    for _ in range(120):
        gcall()           # node ids learnt (first sight: ladder, second: table); a channel's bitcoin keys never recur
    a, b = p50(gcall, 300)
    print("     1 message, %-38s p50 %.3f ms  p99 %.3f ms" % (name, a, b))
# one commitment_signed: 1 signature under the funding key + 483 under the channel's htlc key (both known to the cache after two sights)
cs = workload.make_commit_storm(eng, 2, device="cuda:0")["ecdsa"]
hh, ss, pp = [np.ascontiguousarray(x[:484]) for x in cs.cols]
for _ in range(3):
    assert (eng.verify_ecdsa(hh, ss, pp) == cs.expect[:484]).all()
a, b = p50(lambda: eng.verify_ecdsa(hh, ss, pp), 100)
print("  commitment_signed (484 rows, cached keys) p50 %.3f ms  p99 %.3f ms" % (a, b))
eng.close()
