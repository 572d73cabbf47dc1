"""N2: the batched gossip ingest (csrc/gossip_ingest.cpp, one device call per drained queue) against the sequential model of
gossipd's receive path (oracle/gossipd_model.py, one sigcheck per message as gossipd/gossmap_manage.c does it): same events --
warnings with the reference's exact texts, txout requests, store appends / deletions / timestamp rewrites, peer updates,
traces -- in the same order, on traffic with duplicates, reordering, orphans, damaged and malformed messages.
CPU: the host logic with the C oracle as verification back end.  GPU: the engine as back end."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gossip_stream as gs  # noqa: E402
import gossipd_model  # noqa: E402


class ModelReceiver:
    """the sequential model behind the same driving interface: queued messages are applied one by one at process()"""

    def __init__(self, orc, net):
        def sigcheck(m, signer):
            t = int.from_bytes(m[:2], "big")
            if t == 256:
                return orc.sigcheck_channel_announcement(m)
            if t == 258:
                return orc.sigcheck_channel_update(m, signer)
            return orc.sigcheck_node_announcement(m)
        self.m = gossipd_model.Model(gs.CHAIN, net.our_id, net.height, gs.NOW, sigcheck, lambda k: orc.pubkey_parse(k) is not None)
        self.q = []

    events = property(lambda self: self.m.events)

    def push(self, peer, msg):
        self.q.append((peer, msg))

    def process(self):
        q, self.q = self.q, []
        for peer, msg in q:
            self.m.recv(peer, msg)

    def txout_reply(self, scid, sat, script):
        self.m.txout_reply(scid, sat, script)

    def new_block(self, h):
        self.m.new_block(h)


def oracle_backend(orc):
    def sig(blob, off, ids):
        n = len(off) - 1
        return list(orc.sigcheck_gossip_batch(np.frombuffer(blob + b"\x00", dtype=np.uint8), np.array(off, dtype=np.uint64),
                                              np.frombuffer(ids + b"\x00", dtype=np.uint8)[:33 * n].reshape(n, 33), 1))

    def key(keys):
        return [1 if orc.pubkey_parse(keys[33 * i:33 * i + 33]) is not None else 0 for i in range(len(keys) // 33)]
    return sig, key


def _kinds(events):
    out = {}
    for e in events:
        out[e[0]] = out.get(e[0], 0) + 1
    return out


def _compare(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        assert x == y, "event %d differs:\n ingest: %r\n model:  %r" % (i, x, y)
    assert len(a) == len(b), (len(a), len(b), a[len(b):][:3], b[len(a):][:3])


def test_batched_ingest_equals_sequential_model_cpu_backend(orc):
    from lightning_amd.gossipd import GossipIngest
    k, texts, tot = {}, [], {}
    for seed in (11, 12, 13, 14):
        net, ops = gs.make_script(orc, seed)
        model = ModelReceiver(orc, net)
        gs.drive(net, ops, model, seed)
        with GossipIngest(None, gs.CHAIN, net.our_id, net.height, gs.NOW, backend=oracle_backend(orc)) as ing:
            gs.drive(net, ops, ing, seed)
            _compare(ing.events, model.events)
            st = ing.stats()
        for kind, c in _kinds(model.events).items():
            k[kind] = k.get(kind, 0) + c
        texts += [e[2] for e in model.events if e[0] == "WARNING"]
        assert st["late_verifies"] == 0
        assert st["batches"] < st["messages"] / 5          # batched: far fewer verification calls than messages
        assert st["channels"] > 5 and st["store_records"] > 50
        for key, v in st.items():
            tot[key] = tot.get(key, 0) + v
    # the traffic really exercises the paths: every kind of event occurs, warnings of every family are present
    for kind in ("WARNING", "GET_TXOUT", "STORE_ADD", "STORE_DEL", "STORE_SET_TS", "TRACE", "QUERY_CHANNEL", "QUERY_NODE", "GOOD_GOSSIP", "TXOUT_FAILED", "PEER_UPDATE"):
        assert k.get(kind, 0) > 0, (kind, k)
    for needle in ("Malformed channel_announcement", "node_id_1 must be the lesser", "Bad node_signature_", "Bad bitcoin_signature_", "Bad signature for",
                   "channel_update: malformed", "Do not set DONT_FORWARD", "node_announcement: malformed", "malformed wireaddrs", "channel_announcement: txout",
                   "Bad gossip order: ignoring channel_announcement", "channel_update: Bad signature for"):
        assert any(needle in t for t in texts), needle
    assert tot["duplicates"] > 0 and tot["keyparse_messages"] > 0


def test_ingest_ordering_dependencies_explicit(orc):
    """the dependency VERDICT names: a channel_update is only accepted once its channel_announcement has been accepted AND
    confirmed (gossmap_manage.c:900-924, 1060-1097) -- in one batch, across batches, and when the announcement is bad"""
    from lightning_amd.gossipd import GossipIngest
    net = gs.Net(orc, 99, n_nodes=6, n_chans=4)
    for ch in net.chans:                       # all deep enough
        ch["scid"] = ((net.height - 100) << 40) | (ch["scid"] & 0xFFFFFFFFFF)
    peer = net.compress(orc.pubkey_create(b"\x07" * 32))      # a relaying peer that is no party to these channels
    good, bad = net.cann(0), gs.damage(net.rnd, net.cann(1), "sig")
    u0, u1, u_orphan = net.cupd(0, 0, gs.NOW - 50), net.cupd(1, 0, gs.NOW - 50), net.cupd(2, 1, gs.NOW - 50)
    with GossipIngest(None, gs.CHAIN, net.our_id, net.height, gs.NOW, backend=oracle_backend(orc)) as ing:
        for m in (u0, good, u0, bad, u1, u_orphan):          # an update BEFORE its announcement, one after, one for a rejected announcement
            ing.push(peer, m)
        ing.process()
        ev = ing.events
        assert [e for e in ev if e[0] == "GET_TXOUT"] == [("GET_TXOUT", net.chans[0]["scid"])]
        assert sum(1 for e in ev if e[0] == "WARNING" and "Bad " in e[2]) == 1
        # u0 (first copy): unknown channel then; second copy waits for the txout; u1 / orphan: unknown channel
        assert sum(1 for e in ev if e[0] == "TRACE" and "Unknown channel" in e[2]) == 3
        assert not any(e[0] == "STORE_ADD" for e in ev)
        assert ing.stats()["queued_updates"] == 1
        ing.txout_reply(net.chans[0]["scid"], 5000, net.spk(0))
        adds = [e for e in ing.events if e[0] == "STORE_ADD"]
        assert [a[2] for a in adds] == [256, 4101, 258] and adds[2][4] == u0.hex()
        assert ing.stats()["queued_updates"] == 0 and ing.stats()["late_verifies"] == 0


@pytest.mark.gpu
def test_batched_ingest_on_the_engine_equals_sequential_model(orc):
    """the same comparison with the GPU engine deciding the signatures (lamd_sigcheck_gossip_batch / lamd_pubkey_parse_batch)"""
    from lightning_amd import Engine
    from lightning_amd.gossipd import GossipIngest
    for seed in (21, 22):
        net, ops = gs.make_script(orc, seed, n_nodes=20, n_chans=60, n_ops=1500)
        model = ModelReceiver(orc, net)
        gs.drive(net, ops, model, seed)
        with Engine(0) as eng, GossipIngest(eng, gs.CHAIN, net.our_id, net.height, gs.NOW) as ing:
            gs.drive(net, ops, ing, seed)
            _compare(ing.events, model.events)
            st = ing.stats()
            assert st["late_verifies"] == 0 and st["batches"] > 3 and st["verified_sigs"] > 300
