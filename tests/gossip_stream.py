"""Synthetic gossip traffic for the ingest tests: a small network (nodes, channels) whose announcements, updates and node
announcements arrive from several peers duplicated, reordered, orphaned, damaged -- plus the script that drives a receiver
(batches, txout replies, new blocks).  Messages are built like devtools/mkgossip.c:131-147,235-322 builds them; signatures
come from the oracle's signer (tests only)."""
import hashlib
import random

CHAIN = bytes.fromhex("6fe28c0ab6f1b372c1a6a246ae63f74f931e8365e15a089c68d6190000000000")  # bitcoin mainnet genesis, as on the wire
OTHER_CHAIN = bytes.fromhex("43497fd7f826957108f4a30fd9cec3aeba79972084e90ead01ea330900000000")
NOW = 1_700_000_000


def sha256d(b):
    return hashlib.sha256(hashlib.sha256(b).digest()).digest()


class Net:
    def __init__(self, orc, seed, n_nodes=12, n_chans=30, height=800_000):
        self.orc, self.rnd, self.height = orc, random.Random(seed), height
        r = self.rnd
        self.node_sk = [bytes(r.randrange(1, 256) for _ in range(32)) for _ in range(n_nodes)]
        self.node_id = [self.compress(orc.pubkey_create(sk)) for sk in self.node_sk]
        self.peers = [self.node_id[i] for i in range(min(4, n_nodes))]   # who relays to us
        self.our_id = self.node_id[0]
        self.chans = []
        for c in range(n_chans):
            a, b = r.sample(range(n_nodes), 2)
            if not self.node_id[a] < self.node_id[b]:
                a, b = b, a
            bsk = [bytes(r.randrange(1, 256) for _ in range(32)) for _ in range(2)]
            blk = height - 6 - r.randrange(0, 1000) if c % 7 else height + r.choice([1, 3, 10, 20])   # every 7th: not deep enough yet / far ahead
            scid = (blk << 40) | (r.randrange(1, 3000) << 16) | r.randrange(0, 3)
            self.chans.append(dict(n=(a, b), bsk=bsk, bkey=[self.compress(orc.pubkey_create(k)) for k in bsk], scid=scid, sat=r.randrange(10**5, 10**8)))

    @staticmethod
    def compress(pub65):
        return bytes([2 + (pub65[64] & 1)]) + pub65[1:33]

    def sign(self, sk, tail):
        return self.orc.ecdsa_sign(sha256d(tail), sk, bytes(self.rnd.randrange(1, 256) for _ in range(32)))

    def cann(self, c, features=b"", chain=CHAIN, swap_ids=False):
        ch = self.chans[c]
        a, b = ch["n"]
        ids = [self.node_id[a], self.node_id[b]]
        if swap_ids:
            ids.reverse()
        tail = len(features).to_bytes(2, "big") + features + chain + ch["scid"].to_bytes(8, "big") + ids[0] + ids[1] + ch["bkey"][0] + ch["bkey"][1]
        sks = [self.node_sk[a], self.node_sk[b]]
        if swap_ids:
            sks.reverse()
        sigs = [self.sign(k, tail) for k in sks + ch["bsk"]]
        return b"\x01\x00" + b"".join(sigs) + tail

    def spk(self, c):
        k1, k2 = sorted(self.chans[c]["bkey"])
        return b"\x00\x20" + hashlib.sha256(b"\x52\x21" + k1 + b"\x21" + k2 + b"\x52\xae").digest()

    def cupd(self, c, d, ts, chain=CHAIN, mflags=1, disabled=False, signer=None, extra=b""):
        ch = self.chans[c]
        r = self.rnd
        body = chain + ch["scid"].to_bytes(8, "big") + ts.to_bytes(4, "big") + bytes([mflags, d | (2 if disabled else 0)]) + r.randrange(1, 144).to_bytes(2, "big")
        body += r.randrange(1, 1000).to_bytes(8, "big") + r.randrange(0, 5000).to_bytes(4, "big") + r.randrange(0, 1000).to_bytes(4, "big") + r.randrange(10**6, 10**9).to_bytes(8, "big") + extra
        sk = self.node_sk[ch["n"][d]] if signer is None else signer
        return b"\x01\x02" + self.sign(sk, body) + body

    def nann(self, n, ts, addrs=b"", tlvs=b"", features=b"", sk=None):
        r = self.rnd
        body = len(features).to_bytes(2, "big") + features + ts.to_bytes(4, "big") + self.node_id[n] + bytes(r.randrange(256) for _ in range(3))
        body += bytes(r.randrange(32, 127) for _ in range(32)) + len(addrs).to_bytes(2, "big") + addrs + tlvs
        return b"\x01\x01" + self.sign(self.node_sk[n] if sk is None else sk, body) + body


def damage(rnd, m, kind):
    b = bytearray(m)
    if kind == "sig":            # a signature bit: "Bad ..." (or another verdict index)
        first = 2 + 64 * rnd.randrange(4 if m[:2] == b"\x01\x00" else 1)
        b[first + rnd.randrange(64)] ^= 1 << rnd.randrange(8)
    elif kind == "tail":         # signed bytes
        b[len(b) - 1 - rnd.randrange(20)] ^= 0x10
    elif kind == "trunc":
        b = b[:rnd.randrange(3, len(b) - 1)]
    elif kind == "r>=n":
        b[2:34] = b"\xff" * 32
    elif kind == "badkey" and m[:2] == b"\x01\x00":      # a bitcoin key that is not on the curve (x = 5 has no square root... check by parse)
        flen = int.from_bytes(m[258:260], "big")
        off = 260 + flen + 40 + 66 + 33 * rnd.randrange(2)
        b[off:off + 33] = b"\x02" + (5).to_bytes(32, "big")
    return bytes(b)


def make_script(orc, seed, n_nodes=12, n_chans=30, n_ops=900, lifecycle=False):
    """-> (net, ops): ops is a list of ("push", peer, msg) | ("process",) | ("txout", scid, sat, script) | ("block", height)"""
    net = Net(orc, seed, n_nodes, n_chans)
    rnd = random.Random(seed ^ 0x5EED)
    ops = []
    announced = set()
    ts0 = NOW - 3600
    height = net.height
    for step in range(n_ops):
        x = rnd.random()
        peer = rnd.choice(net.peers)
        c = rnd.randrange(n_chans)
        if x < 0.22:
            kind = rnd.choice(["ok"] * 10 + ["dup", "dup", "sig", "tail", "trunc", "r>=n", "badkey", "chain", "swap", "features"])
            if kind == "chain":
                m = net.cann(c, chain=OTHER_CHAIN)
            elif kind == "swap":
                m = net.cann(c, swap_ids=True)
            elif kind == "features":
                m = net.cann(c, features=bytes(rnd.randrange(256) for _ in range(3)))
            else:
                m = net.cann(c)
                if kind not in ("ok", "dup"):
                    m = damage(rnd, m, kind)
            ops.append(("push", peer, m))
            if kind == "dup":
                for p2 in rnd.sample(net.peers, 2):
                    ops.append(("push", p2, m))
            announced.add(c)
        elif x < 0.62:
            kind = rnd.choice(["ok"] * 12 + ["old", "same", "sig", "tail", "trunc", "r>=n", "chain", "future", "ancient", "dontfwd", "private", "extra", "nosrc"])
            d = rnd.randrange(2)
            ts = ts0 + step * 3 + rnd.randrange(3)
            if kind == "old":
                ts -= 3000
            elif kind == "future":
                ts = NOW + 2 * 86400
            elif kind == "ancient":
                ts = NOW - 15 * 86400
            if kind == "chain":
                m = net.cupd(c, d, ts, chain=OTHER_CHAIN)
            elif kind == "dontfwd":
                m = net.cupd(c, d, ts, mflags=3)
            elif kind == "private":       # an update for a channel nobody announced, signed by the relaying peer itself
                pi = net.node_id.index(peer)
                m = net.cupd(c, d, ts, signer=net.node_sk[pi])
            elif kind == "extra":
                m = net.cupd(c, d, ts, extra=bytes(rnd.randrange(256) for _ in range(7)))
            else:
                m = net.cupd(c, d, ts, disabled=rnd.random() < 0.1)
                if kind in ("sig", "tail", "trunc", "r>=n"):
                    m = damage(rnd, m, kind)
            ops.append(("push", None if kind == "nosrc" else peer, m))
            if kind == "same":
                ops.append(("push", rnd.choice(net.peers), m))
        elif x < 0.80:
            kind = rnd.choice(["ok"] * 8 + ["old", "sig", "tail", "trunc", "addrs", "addrs-bad", "tlv", "tlv-bad", "wrongkey", "unknown-type-addr"])
            n = rnd.randrange(n_nodes)
            ts = ts0 + step * 3
            if kind == "old":
                ts -= 5000
            if kind == "addrs":
                m = net.nann(n, ts, addrs=b"\x01" + bytes(4) + b"\x26\x07" + b"\x05\x03abc\x00\x50")
            elif kind == "addrs-bad":
                m = net.nann(n, ts, addrs=b"\x02" + bytes(9))                      # an ipv6 descriptor cut short
            elif kind == "unknown-type-addr":
                m = net.nann(n, ts, addrs=b"\x01" + bytes(4) + b"\x26\x07" + b"\x09\x01\x02")
            elif kind == "tlv":
                m = net.nann(n, ts, tlvs=b"\x01\x0a" + bytes(range(1, 11)) + b"\x03\x02ab")
            elif kind == "tlv-bad":
                m = net.nann(n, ts, tlvs=b"\x02\x02ab")
            elif kind == "wrongkey":
                m = net.nann(n, ts, sk=net.node_sk[(n + 1) % n_nodes])
            else:
                m = net.nann(n, ts)
                if kind in ("sig", "tail", "trunc"):
                    m = damage(rnd, m, kind)
            ops.append(("push", peer, m))
        elif x < 0.90:
            ops.append(("process",))
        elif x < 0.97:
            ops.append(("txouts", rnd.choice(["all", "some", "some", "one"]), rnd.random()))
        else:
            height += rnd.choice([1, 1, 2, 6])
            ops.append(("block", height))
    ops.append(("process",))
    ops.append(("txouts", "all", 0.5))
    ops.append(("block", height + 30))
    ops.append(("process",))
    ops.append(("txouts", "all", 0.9))
    if lifecycle:
        # the channel life cycle after the flood: funding outputs get spent (dying, removed 72 blocks later), time passes and
        # prune_network runs, more gossip arrives in between (updates for dying / removed channels, node_announcements that move)
        lr = random.Random(seed ^ 0x11FE)
        for rnd_round in range(6):
            for _ in range(lr.randrange(1, 5)):
                ops.append(("spent", height, lr.randrange(n_chans)))
            for _ in range(lr.randrange(3, 12)):
                c, d = lr.randrange(n_chans), lr.randrange(2)
                ops.append(("push", lr.choice(net.peers), net.cupd(c, d, NOW + 100 + 50 * rnd_round + lr.randrange(40))))
            for _ in range(lr.randrange(1, 4)):
                ops.append(("push", lr.choice(net.peers), net.nann(lr.randrange(n_nodes), NOW + 100 + 50 * rnd_round)))
            ops.append(("process",))
            height += lr.choice([10, 40, 80])
            ops.append(("block", height))
            if rnd_round in (2, 4):
                ops.append(("time", NOW + (8 + 4 * rnd_round) * 86400))
                ops.append(("prune",))
        ops.append(("block", height + 100))
        ops.append(("process",))
    return net, ops


def drive(net, ops, receiver, seed):
    """receiver: object with push(peer, msg), process(), txout_reply(scid, sat, script), new_block(h), and an `events` list.
    GET_TXOUT events are answered by later ("txouts", ...) ops: with the right script mostly, sometimes a wrong one or none."""
    rnd = random.Random(seed ^ 0x7E57)
    by_scid = {ch["scid"]: i for i, ch in enumerate(net.chans)}
    asked, seen = [], 0
    for op in ops:
        if op[0] == "push":
            receiver.push(op[1], op[2])
        elif op[0] == "process":
            receiver.process()
        elif op[0] == "block":
            receiver.new_block(op[1])
        elif op[0] == "spent":
            receiver.channel_spent(op[1], net.chans[op[2]]["scid"])
        elif op[0] == "time":
            receiver.set_time(op[1])
        elif op[0] == "prune":
            receiver.prune()
        elif op[0] == "txout":       # an explicit reply (a receiver without a listener emits no GET_TXOUT to answer)
            receiver.txout_reply(op[1], op[2], op[3])
        elif op[0] == "txouts":
            ev = receiver.events
            for e in ev[seen:]:
                if e[0] == "GET_TXOUT":
                    asked.append(e[1])
            seen = len(ev)
            take = asked if op[1] == "all" else (asked[:1] if op[1] == "one" else [s for s in asked if rnd.random() < 0.5])
            for scid in take:
                asked.remove(scid)
                c = by_scid[scid]
                roll = rnd.random()
                if roll < 0.08:
                    receiver.txout_reply(scid, 0, b"")                                    # spent / unknown output
                elif roll < 0.14:
                    receiver.txout_reply(scid, net.chans[c]["sat"], b"\x00\x20" + bytes(32))   # not the 2-of-2 we expect
                else:
                    receiver.txout_reply(scid, net.chans[c]["sat"], net.spk(c))
                # replies can produce more requests only through new_block; keep scanning
            for e in receiver.events[seen:]:
                if e[0] == "GET_TXOUT":
                    asked.append(e[1])
            seen = len(receiver.events)
