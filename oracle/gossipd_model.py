"""TEST INFRASTRUCTURE (oracle): gossipd's receive path restated ONE MESSAGE AT A TIME, the way the reference runs it.

Follows /root/reference/gossipd/gossipd.c:172-286 (handle_recv_gossip) and gossipd/gossmap_manage.c function by function:
  gossmap_manage_channel_announcement :620-753      gossmap_manage_handle_get_txout_reply :753-872
  process_channel_update              :878-998      gossmap_manage_channel_update         :1014-1120
  process_node_announcement           :1122-1160    gossmap_manage_node_announcement      :1162-1243
  process_pending_cupdate / reprocess_* :1245-1342  gossmap_manage_new_block              :1358-1390
Every signature is checked when the reference checks it, by `sigcheck(msg, signer_or_None)` (default: the C oracle's
restatement of gossipd/sigcheck.c, one call per message); fromwire_* is restated from wire/peer_wire.csv:344-381 with
pyref's framing helpers.  Only tests import this: the product's batched ingest (lightning_amd/csrc/gossip_ingest.cpp) is
compared against it event by event.  Also modelled: remove_channel :296-375, prune_network :398-470, kill_spent_channel /
the dying loop of new_block :1369-1437, gossmap_manage_channel_spent :1439-1497.  Not modelled (neither is it in the product): the
seeker's internals, local announcements with a known amount, store compaction.  (The store's FILE FORMAT is not restated here: the
product's image is compared byte for byte with files the reference's gossipd wrote, tests/test_gossip_ingest.py.)

Events are tuples (kind, peer_hex_or_None, ...) in the vocabulary of include/lightning_amd_gossipd.h."""
import hashlib

import pyref

CANN, NANN, CUPD = 256, 257, 258
ORDER_N = pyref.N


def sha256d(b):
    return hashlib.sha256(hashlib.sha256(b).digest()).digest()


def fmt_scid(scid):  # bitcoin/short_channel_id.c:56-62
    return "%dx%dx%d" % (scid >> 40, (scid >> 16) & 0xFFFFFF, scid & 0xFFFF)


def der_hex(sig64):  # fmt_secp256k1_ecdsa_signature: the DER form (bitcoin/signature.c:325-335)
    def enc(v):
        b = v.lstrip(b"\x00") or b"\x00"
        if b[0] & 0x80:
            b = b"\x00" + b
        return b"\x02" + bytes([len(b)]) + b
    body = enc(sig64[:32]) + enc(sig64[32:])
    return (b"\x30" + bytes([len(body)]) + body).hex()


def sig_in_range(sig64):  # secp256k1_ecdsa_signature_parse_compact via wire/fromwire.c:188-199
    return int.from_bytes(sig64[:32], "big") < ORDER_N and int.from_bytes(sig64[32:], "big") < ORDER_N


def wireaddrs_ok(b):  # common/wireaddr.c:30-68, 858-891
    pos = 0
    while pos < len(b):
        t = b[pos]
        pos += 1
        if t == 1:
            alen = 4
        elif t == 2:
            alen = 16
        elif t == 3:
            alen = 10
        elif t == 4:
            alen = 35
        elif t == 5:
            if pos >= len(b):
                return False
            alen = b[pos]
            pos += 1
        else:
            return True
        if len(b) - pos < alen + 2:
            return False
        pos += alen + 2
    return True


class Model:
    def __init__(self, chain_hash, our_id, blockheight, now, sigcheck, key_valid, prune_interval=1209600, store_version=16):
        self.chain_hash, self.our_id, self.blockheight, self.now, self.prune = chain_hash, our_id, blockheight, now, prune_interval
        self.sigcheck = sigcheck      # (msg, signer33 or None) -> 0 ok / k first bad signature / -1 malformed
        self.key_valid = key_valid    # 33 bytes -> bool (fromwire_pubkey)
        self.chans, self.nodes = {}, {}
        self.pending_ann, self.early_ann = {}, {}
        self.pending_cupdates, self.early_cupdates, self.pending_nannounces = [], [], []
        self.txout_failures = set()
        # [type, timestamp, deleted, dying]; record numbers grow with the file offset.  A v16 store opens with its uuid record (gossip_store.c:186-197)
        self.store = [[4107, 0, False, False]] if (store_version & 0x1F) >= 16 else []
        self.dying = []               # [scid, deadline, record] (struct chan_dying)
        self.events = []

    # ---- helpers
    def ev(self, *e):
        self.events.append(tuple(e))

    @staticmethod
    def ph(peer):
        return peer.hex() if peer is not None else None

    def warning(self, peer, text):
        self.ev("WARNING", self.ph(peer), text)

    def bad_gossip(self, peer, text):  # :576-579
        self.ev("TRACE", self.ph(peer), "Bad gossip order: " + text)

    def peer_warning(self, peer, text):  # :582-597
        self.bad_gossip(peer, text)
        if peer is not None:
            self.warning(peer, text)

    def store_add(self, typ, ts, data):
        self.store.append([typ, ts, False, False])
        self.ev("STORE_ADD", len(self.store) - 1, typ, ts, data.hex())
        return len(self.store) - 1

    def store_del(self, idx):  # gossip_store_del :622-638 (a channel_announcement takes its amount record with it)
        self.store[idx][2] = True
        if self.store[idx][0] == CANN and idx + 1 < len(self.store) and self.store[idx + 1][0] == 4101:
            self.store[idx + 1][2] = True
        self.ev("STORE_DEL", idx, self.store[idx][0])

    def store_set_dying(self, idx):  # gossip_store_set_flag(.., GOSSIP_STORE_DYING_BIT, ..)
        self.store[idx][3] = True
        self.ev("STORE_FLAG", idx, self.store[idx][0], 0x0800)

    def sigcheck_text(self, typ, which, m):  # gossipd/sigcheck.c
        off = 258 if typ == CANN else 66
        what = ["Bad node_signature_1", "Bad node_signature_2", "Bad bitcoin_signature_1", "Bad bitcoin_signature_2"][which - 1] if typ == CANN else "Bad signature for"
        sig = m[2 + (64 * (which - 1) if typ == CANN else 0):][:64]
        kind = {CANN: "channel_announcement", CUPD: "channel_update", NANN: "node_announcement"}[typ]
        return "%s %s hash %s on %s %s" % (what, der_hex(sig), sha256d(m[off:]).hex(), kind, m.hex())

    # ---- gossipd.c:172-286
    def recv(self, peer, msg):
        typ = int.from_bytes(msg[:2], "big") if len(msg) >= 2 else 0
        if typ == CANN:
            err = self.channel_announcement(msg, peer)
        elif typ == CUPD:
            err = self.channel_update(msg, peer)
        elif typ == NANN:
            err = self.node_announcement(msg, peer)
        else:
            return
        if err:
            self.warning(peer, err)  # :277-283

    # ---- :620-753
    def channel_announcement(self, m, peer):
        ok = len(m) >= 260
        if ok:
            flen = int.from_bytes(m[258:260], "big")
            keyoff = 260 + flen + 40
            ok = len(m) >= keyoff + 132
        if ok:
            ok = all(sig_in_range(m[2 + 64 * i:66 + 64 * i]) for i in range(4))
        if ok:
            ok = self.key_valid(m[keyoff + 66:keyoff + 99]) and self.key_valid(m[keyoff + 99:keyoff + 132])
        if not ok:
            return "Malformed channel_announcement " + m.hex()
        chain, scid = m[260 + flen:292 + flen], int.from_bytes(m[292 + flen:300 + flen], "big")
        id1, id2 = m[keyoff:keyoff + 33], m[keyoff + 33:keyoff + 66]
        if not id1 < id2:
            return "node_id_1 must be the lesser node id! 1=%s, 2=%s" % (id1.hex(), id2.hex())
        if chain != self.chain_hash:
            return None
        if scid in self.txout_failures:
            return None
        if scid in self.chans or scid in self.pending_ann or scid in self.early_ann:
            return None
        v = self.sigcheck(m, None)
        assert v != -1, "fromwire accepted what the signature check calls malformed"
        if v:
            return self.sigcheck_text(CANN, v, m)
        k1, k2 = sorted([m[keyoff + 66:keyoff + 99], m[keyoff + 99:keyoff + 132]])
        script = b"\x52\x21" + k1 + b"\x21" + k2 + b"\x52\xae"
        pca = dict(msg=m, peer=peer, node=(id1, id2), spk=b"\x00\x20" + hashlib.sha256(script).digest())
        if not (scid >> 40) + 6 - 1 <= self.blockheight:
            if self.blockheight != 0 and (scid >> 40) > self.blockheight + 12:
                return "Bad gossip order: ignoring channel_announcement %s at blockheight %u" % (fmt_scid(scid), self.blockheight)
            self.early_ann[scid] = pca
            return None
        self.pending_ann[scid] = pca
        self.ev("GET_TXOUT", scid)
        return None

    # ---- :753-872
    def txout_reply(self, scid, sat, script):
        pca = self.pending_ann.pop(scid, None)
        if pca is None:
            return
        bad = False
        if len(script) == 0:
            bad = True
        elif script != pca["spk"]:
            self.peer_warning(pca["peer"], "channel_announcement: txout %s expected %s, got %s" % (fmt_scid(scid), pca["spk"].hex(), script.hex()))
            bad = True
        if bad:
            self.txout_failures.add(scid)
            self.ev("TXOUT_FAILED", scid)
            return
        if scid in self.chans:
            return
        rec = self.store_add(CANN, 0, pca["msg"])
        self.store_add(4101, 0, b"\x10\x05" + sat.to_bytes(8, "big"))
        self.chans[scid] = dict(node=pca["node"], cann=rec, cupd=[None, None], dying=False)
        for n in pca["node"]:
            self.nodes.setdefault(n, dict(nann=None))
        self.reprocess_queued_msgs()

    # ---- :878-998
    def process_channel_update(self, u):
        scid, d = u["scid"], u["cflags"] & 1
        chan = self.chans.get(scid)
        if chan is None:
            if scid in self.txout_failures:
                return None
            self.ev("QUERY_CHANNEL", self.ph(u["peer"]), scid)
            self.bad_gossip(u["peer"], "Unknown channel " + fmt_scid(scid))
            return None
        v = self.sigcheck(u["msg"], chan["node"][d])
        if v:
            return self.sigcheck_text(CUPD, 1, u["msg"])
        if u["mflags"] & 2:
            return "Do not set DONT_FORWARD on public channel_updates (%s)" % fmt_scid(scid)
        if chan["cupd"][d] is not None:
            if self.store[chan["cupd"][d]][1] >= u["ts"]:
                return None
        elif chan["cupd"][1 - d] is None:
            self.store[chan["cann"]][1] = u["ts"]
            self.ev("STORE_SET_TS", chan["cann"], u["ts"])
        rec = self.store_add(CUPD, u["ts"], u["msg"])
        if chan["cupd"][d] is not None:
            self.store_del(chan["cupd"][d])
        chan["cupd"][d] = rec
        if chan["node"][1 - d] == self.our_id:
            self.ev("PEER_UPDATE", self.ph(u["peer"]), scid, u["fee_base"], u["fee_ppm"], u["cltv"], u["hmin"], u["hmax"])
        if u["peer"] is not None:
            self.ev("GOOD_GOSSIP", self.ph(u["peer"]))
        self.ev("TRACE", self.ph(u["peer"]), "Received channel_update for channel %s/%d now %s" % (fmt_scid(scid), d, "DISABLED" if u["cflags"] & 2 else "ACTIVE"))
        return None

    # ---- :1014-1120
    def channel_update(self, m, peer):
        if len(m) < 138 or not sig_in_range(m[2:66]):
            return "channel_update: malformed " + m.hex()
        u = dict(msg=m, peer=peer, scid=int.from_bytes(m[98:106], "big"), ts=int.from_bytes(m[106:110], "big"), mflags=m[110], cflags=m[111],
                 cltv=int.from_bytes(m[112:114], "big"), hmin=int.from_bytes(m[114:122], "big"), fee_base=int.from_bytes(m[122:126], "big"),
                 fee_ppm=int.from_bytes(m[126:130], "big"), hmax=int.from_bytes(m[130:138], "big"))
        if m[66:98] != self.chain_hash:
            return None
        if u["ts"] > self.now + 24 * 60 * 60 or u["ts"] < self.now - self.prune:  # :1001-1012
            return None
        if u["scid"] in self.pending_ann:
            self.pending_cupdates.append(u)
            return None
        if u["scid"] in self.early_ann:
            self.early_cupdates.append(u)
            return None
        if u["scid"] not in self.chans and peer is not None and self.sigcheck(m, peer) == 0:
            self.ev("PEER_UPDATE", self.ph(peer), u["scid"], u["fee_base"], u["fee_ppm"], u["cltv"], u["hmin"], u["hmax"])
            return None
        return self.process_channel_update(u)

    # ---- :1122-1160
    def process_node_announcement(self, node, ts, nid, m, peer):
        if node["nann"] is not None and self.store[node["nann"]][1] >= ts:
            return
        rec = self.store_add(NANN, ts, m)
        if node["nann"] is not None:
            self.store_del(node["nann"])
        node["nann"] = rec
        node["msg"] = m
        if peer is not None:
            self.ev("GOOD_GOSSIP", self.ph(peer))
        self.ev("TRACE", self.ph(peer), "Received node_announcement for node " + nid.hex())

    def unknown_node(self, peer, nid):
        self.ev("QUERY_NODE", self.ph(peer), nid.hex())
        self.bad_gossip(peer, "node_announcement: unknown node " + nid.hex())

    # ---- :1162-1243
    def node_announcement(self, m, peer):
        ok = len(m) >= 68
        if ok:
            flen = int.from_bytes(m[66:68], "big")
            keyoff = 68 + flen + 4
            ok = len(m) >= keyoff + 70
        if ok:
            alen = int.from_bytes(m[keyoff + 68:keyoff + 70], "big")
            end = keyoff + 70 + alen
            ok = len(m) >= end and pyref.node_ann_tlvs_ok(m[end:]) and sig_in_range(m[2:66])
        if not ok:
            return "node_announcement: malformed " + m.hex()
        if not wireaddrs_ok(m[keyoff + 70:end]):
            return "node_announcement: malformed wireaddrs  in " + m.hex()  # tal_hex(tmpctx, NULL) is the empty string
        v = self.sigcheck(m, None)
        if v:
            return self.sigcheck_text(NANN, 1, m)
        nid, ts = m[keyoff:keyoff + 33], int.from_bytes(m[keyoff - 4:keyoff], "big")
        node = self.nodes.get(nid)
        if node is None:
            if self.pending_ann or self.early_ann:
                self.pending_nannounces.append(dict(id=nid, ts=ts, msg=m, peer=peer))
                return None
            self.unknown_node(peer, nid)
            return None
        self.process_node_announcement(node, ts, nid, m, peer)
        return None

    # ---- :1245-1342
    def process_pending_cupdate(self, u):
        err = self.process_channel_update(u)
        if err:
            self.peer_warning(u["peer"], "channel_update: " + err)

    def reprocess_queued_msgs(self):
        pending_empty, early_empty = not self.pending_ann, not self.early_ann
        if pending_empty:
            l, self.pending_cupdates = self.pending_cupdates, []
            for u in l:
                self.process_pending_cupdate(u)
        if early_empty:
            l, self.early_cupdates = self.early_cupdates, []
            for u in l:
                if u["scid"] in self.pending_ann:
                    self.pending_cupdates.append(u)
                    continue
                self.process_pending_cupdate(u)
        if early_empty and pending_empty:
            l, self.pending_nannounces = self.pending_nannounces, []
            for pn in l:
                node = self.nodes.get(pn["id"])
                if node is None:
                    self.unknown_node(pn["peer"], pn["id"])
                    continue
                self.process_node_announcement(node, pn["ts"], pn["id"], pn["msg"], pn["peer"])

    # ---- :1358-1390
    def new_block(self, height):
        self.blockheight = height
        for scid in sorted(self.early_ann):
            if not (scid >> 40) + 6 - 1 <= height:
                break
            pca = self.early_ann.pop(scid)
            if scid in self.pending_ann:
                continue
            self.pending_ann[scid] = pca
            self.ev("GET_TXOUT", scid)

        # :1419-1436 dying channels whose deadline has come
        i = 0
        while i < len(self.dying):
            scid, deadline, rec = self.dying[i]
            if deadline > height:
                i += 1
                continue
            if scid in self.chans:  # kill_spent_channel :1369-1387
                self.ev("TRACE", None, "Deleting channel %s due to the funding outpoint being spent" % fmt_scid(scid))
                self.remove_channel(scid)
            self.store_del(rec)
            del self.dying[i]

    # ---- gossmap views the removal code needs
    def node_chans(self, nid):
        return [s for s, c in self.chans.items() if nid in c["node"]]

    def any_cannounce_precedes(self, nid, exclude, off):  # :267-286
        for s in self.node_chans(nid):
            if s == exclude:
                continue
            c = self.chans[s]
            if c["cann"] > off or c["dying"]:
                continue
            return True
        return False

    def all_node_channels_dying(self, nid, ignore):  # :289-298
        return all(self.chans[s]["dying"] for s in self.node_chans(nid) if s != ignore)

    # ---- :296-375
    def remove_channel(self, scid):
        chan = self.chans[scid]
        self.txout_failures.add(scid)
        self.pending_ann.pop(scid, None)
        self.early_ann.pop(scid, None)
        self.store_add(4103, 0, b"\x10\x07" + scid.to_bytes(8, "big"))
        self.store_del(chan["cann"])
        for d in (0, 1):
            if chan["cupd"][d] is not None:
                self.store_del(chan["cupd"][d])
        for d in (0, 1):
            nid = chan["node"][d]
            if d == 1 and nid == chan["node"][0]:
                continue
            node = self.nodes.get(nid)
            if node is None or node["nann"] is None:
                continue
            if len(self.node_chans(nid)) == 1:
                self.store_del(node["nann"])
                node["nann"] = None
                continue
            if chan["cann"] < node["nann"] and not self.any_cannounce_precedes(nid, scid, node["nann"]):
                ts, msg = self.store[node["nann"]][1], node["msg"]
                self.store_del(node["nann"])
                off = self.store_add(NANN, ts, msg)
                node["nann"] = off
            else:
                if chan["dying"]:
                    continue
                off = node["nann"]
            if self.all_node_channels_dying(nid, scid):
                self.store_set_dying(off)
        del self.chans[scid]
        for nid in set(chan["node"]):   # the next gossmap refresh drops nodes without channels
            if not self.node_chans(nid):
                self.nodes.pop(nid, None)

    # ---- :1439-1497
    def channel_spent(self, blockheight, scid):
        chan = self.chans.get(scid)
        if chan is None:
            return
        if any(d[0] == scid for d in self.dying):
            return
        deadline = blockheight + 72
        self.ev("TRACE", None, "channel %s closing soon due to the funding outpoint being spent" % fmt_scid(scid))
        rec = self.store_add(4106, 0, b"\x10\x0a" + scid.to_bytes(8, "big") + deadline.to_bytes(4, "big"))
        self.dying.append([scid, deadline, rec])
        self.store_set_dying(chan["cann"])
        chan["dying"] = True
        for d in (0, 1):
            if chan["cupd"][d] is not None:
                self.store_set_dying(chan["cupd"][d])
        for d in (0, 1):
            nid = chan["node"][d]
            if d == 1 and nid == chan["node"][0]:
                continue
            node = self.nodes.get(nid)
            if node is None or node["nann"] is None:
                continue
            if self.all_node_channels_dying(nid, scid):
                self.store_set_dying(node["nann"])

    # ---- :398-470
    def prune_network(self):
        highwater = self.now - self.prune
        pruned = 0
        for scid in sorted(self.chans, key=lambda s: self.chans[s]["cann"]):   # gossmap's channel index order = store order
            chan = self.chans.get(scid)
            if chan is None:
                continue
            ts = [self.store[chan["cupd"][d]][1] if chan["cupd"][d] is not None else 0xFFFFFFFF for d in (0, 1)]
            if ts[0] >= highwater and ts[1] >= highwater:
                continue
            if any(d[0] == scid for d in self.dying):
                continue
            if self.our_id in chan["node"]:
                local = 1 if chan["node"][1] == self.our_id else 0
                self.ev("TRACE", None, "Pruning local channel %s from gossip_store: local channel_update time %u, remote %u" % (fmt_scid(scid), ts[local], ts[1 - local]))
            self.ev("TRACE", None, "Pruning channel %s from network view (ages %u and %u)" % (fmt_scid(scid), ts[0], ts[1]))
            self.remove_channel(scid)
            pruned += 1
        return pruned
