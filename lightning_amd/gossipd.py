"""Python front end of the batched gossip ingest (include/lightning_amd_gossipd.h, csrc/gossip_ingest.cpp): what a
GPU-backed lightning_gossipd would drive.  Events come back as tuples in the vocabulary of that header, e.g.
("WARNING", peer_hex, text), ("GET_TXOUT", scid), ("STORE_ADD", index, type, timestamp, hex)."""
import ctypes

from . import _build


class Event(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("has_peer", ctypes.c_int), ("peer", ctypes.c_ubyte * 33), ("scid", ctypes.c_uint64),
                ("index", ctypes.c_uint64), ("type", ctypes.c_uint32), ("timestamp", ctypes.c_uint32), ("values", ctypes.c_uint64 * 5),
                ("text", ctypes.c_char_p), ("data", ctypes.POINTER(ctypes.c_ubyte)), ("len", ctypes.c_size_t)]


class Config(ctypes.Structure):
    _fields_ = [("chain_hash", ctypes.c_ubyte * 32), ("our_id", ctypes.c_ubyte * 33), ("blockheight", ctypes.c_uint32), ("now", ctypes.c_uint64),
                ("prune_interval", ctypes.c_uint32), ("store_version", ctypes.c_uint8), ("emit_store_writes", ctypes.c_uint8),
                ("store_uuid", ctypes.c_ubyte * 32)]


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("messages", "batches", "verified_messages", "verified_sigs", "keyparse_messages", "duplicates",
                                               "late_verifies", "channels", "nodes", "pending", "early", "queued_updates", "queued_nodes",
                                               "store_records", "run_updates", "sub_batches", "overlapped_stages", "run_announcements", "run_nodes")]


EVENT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(Event))
SIGCHECK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
KEYPARSE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)
KINDS = {1: "WARNING", 2: "GET_TXOUT", 3: "STORE_ADD", 4: "STORE_DEL", 5: "STORE_SET_TS", 6: "PEER_UPDATE", 7: "TRACE", 8: "QUERY_CHANNEL",
         9: "QUERY_NODE", 10: "GOOD_GOSSIP", 11: "TXOUT_FAILED", 12: "STORE_FLAG", 13: "STORE_WRITE"}

_lib = None


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_build.build_shim())
        L.lamd_gossipd_new.restype = ctypes.c_void_p
        L.lamd_gossipd_new.argtypes = [ctypes.c_void_p, ctypes.POINTER(Config), EVENT_FN, ctypes.c_void_p]
        L.lamd_gossipd_free.argtypes = [ctypes.c_void_p]
        L.lamd_gossipd_set_backend.argtypes = [ctypes.c_void_p, SIGCHECK_FN, KEYPARSE_FN, ctypes.c_void_p]
        L.lamd_gossipd_push.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        L.lamd_gossipd_push_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.lamd_gossipd_txout_reply_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        L.lamd_gossipd_process.restype = ctypes.c_long
        L.lamd_gossipd_process.argtypes = [ctypes.c_void_p]
        L.lamd_gossipd_txout_reply.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t]
        L.lamd_gossipd_new_block.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.lamd_gossipd_channel_spent.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64]
        L.lamd_gossipd_prune.restype = ctypes.c_long
        L.lamd_gossipd_prune.argtypes = [ctypes.c_void_p]
        L.lamd_gossipd_store_image.restype = ctypes.c_size_t
        L.lamd_gossipd_store_image.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.lamd_gossipd_set_time.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.lamd_gossipd_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(Stats)]
        _lib = L
    return _lib


class GossipIngest:
    """engine: a lightning_amd.Engine (verdicts from the GPU), or None with `backend=(sigcheck, keyparse)` -- Python callables
    sigcheck(msgs_blob: bytes, off: list[int], ids: bytes) -> list[int], keyparse(keys: bytes) -> list[int] -- which tests use to
    run the host logic against a CPU checker on a machine without a GPU."""

    def __init__(self, engine, chain_hash, our_id, blockheight, now, prune_interval=0, backend=None, collect_events=True, store_version=0,
                 store_uuid=bytes(32), emit_store_writes=False):
        self._L = _load()
        cfg = Config()
        cfg.chain_hash[:] = chain_hash
        cfg.our_id[:] = our_id
        cfg.blockheight, cfg.now, cfg.prune_interval = blockheight, now, prune_interval
        cfg.store_version, cfg.emit_store_writes = store_version, 1 if emit_store_writes else 0
        cfg.store_uuid[:] = store_uuid
        self.events = []
        self.writes = []     # (offset, bytes) of every LAMD_GEV_STORE_WRITE event (emit_store_writes=True)
        self._cb = EVENT_FN(self._on_event) if collect_events else ctypes.cast(None, EVENT_FN)
        self._engine = engine
        self._g = self._L.lamd_gossipd_new(engine._ctx if engine is not None else None, ctypes.byref(cfg), self._cb, None)
        if not self._g:
            raise MemoryError("lamd_gossipd_new")
        if backend is not None:
            sig, key = backend

            def c_sig(_user, n, msgs, off, ids, verdict):
                offs = (ctypes.c_uint64 * (n + 1)).from_address(off)
                blob = ctypes.string_at(msgs, offs[n])
                out = sig(blob, list(offs), ctypes.string_at(ids, 33 * n) if ids else bytes(33 * n))   # NULL for announcement-only late verifies
                (ctypes.c_int8 * n).from_address(verdict)[:] = out
                return 0

            def c_key(_user, n, pub, ok):
                (ctypes.c_ubyte * n).from_address(ok)[:] = key(ctypes.string_at(pub, 33 * n))
                return 0
            self._be = (SIGCHECK_FN(c_sig), KEYPARSE_FN(c_key))
            self._L.lamd_gossipd_set_backend(self._g, self._be[0], self._be[1], None)

    def _on_event(self, _user, evp):
        e = evp.contents
        k = KINDS.get(e.kind, e.kind)
        peer = bytes(e.peer).hex() if e.has_peer else None
        if k in ("WARNING", "TRACE"):
            self.events.append((k, peer, e.text.decode()))
        elif k in ("GET_TXOUT", "TXOUT_FAILED"):
            self.events.append((k, e.scid))
        elif k == "STORE_ADD":
            self.events.append((k, e.index, e.type, e.timestamp, ctypes.string_at(e.data, e.len).hex()))
        elif k == "STORE_DEL":
            self.events.append((k, e.index, e.type))
        elif k == "STORE_SET_TS":
            self.events.append((k, e.index, e.timestamp))
        elif k == "STORE_FLAG":
            self.events.append((k, e.index, e.type, int(e.values[1])))
        elif k == "STORE_WRITE":
            self.writes.append((int(e.values[0]), ctypes.string_at(e.data, e.len)))
        elif k == "PEER_UPDATE":
            self.events.append((k, peer, e.scid) + tuple(e.values))
        elif k == "QUERY_CHANNEL":
            self.events.append((k, peer, e.scid))
        elif k == "QUERY_NODE":
            self.events.append((k, peer, ctypes.string_at(e.data, e.len).hex()))
        elif k == "GOOD_GOSSIP":
            self.events.append((k, peer))

    def push(self, peer, msg):
        rc = self._L.lamd_gossipd_push(self._g, peer, msg, len(msg))
        if rc != 0:
            raise RuntimeError("lamd_gossipd_push: %d" % rc)

    def push_batch(self, peer, msgs, off):
        """msgs: numpy uint8 blob, off: numpy uint64 [n+1]; every message from `peer` (33 bytes or None)"""
        rc = self._L.lamd_gossipd_push_batch(self._g, len(off) - 1, peer, 0, msgs.ctypes.data, off.ctypes.data)
        if rc != 0:
            raise RuntimeError("lamd_gossipd_push_batch: %d" % rc)

    def txout_reply_batch(self, scids, sats, scripts, script_off):
        applied = ctypes.c_size_t(0)
        rc = self._L.lamd_gossipd_txout_reply_batch(self._g, len(scids), scids.ctypes.data, sats.ctypes.data, scripts.ctypes.data, script_off.ctypes.data, ctypes.byref(applied))
        if rc != 0:
            raise RuntimeError("lamd_gossipd_txout_reply_batch: %d after %d replies" % (rc, applied.value))

    def process(self):
        n = self._L.lamd_gossipd_process(self._g)
        if n < 0:
            raise RuntimeError("lamd_gossipd_process: %d" % n)
        return n

    def txout_reply(self, scid, sat, script):
        rc = self._L.lamd_gossipd_txout_reply(self._g, scid, sat, script, len(script))
        if rc != 0:
            raise RuntimeError("lamd_gossipd_txout_reply: %d" % rc)

    def new_block(self, height):
        self._L.lamd_gossipd_new_block(self._g, height)

    def set_time(self, now):
        self._L.lamd_gossipd_set_time(self._g, now)

    def channel_spent(self, blockheight, scid):
        rc = self._L.lamd_gossipd_channel_spent(self._g, blockheight, scid)
        if rc != 0:
            raise RuntimeError("lamd_gossipd_channel_spent: %d" % rc)

    def prune(self):
        n = self._L.lamd_gossipd_prune(self._g)
        if n < 0:
            raise RuntimeError("lamd_gossipd_prune: %d" % n)
        return n

    def store_image(self):
        """the gossip_store file the reference would hold after the same operations (bytes)"""
        p = ctypes.c_void_p()
        n = self._L.lamd_gossipd_store_image(self._g, ctypes.byref(p))
        return ctypes.string_at(p, n)

    def store_size(self):
        """size of that file in bytes (without copying it)"""
        p = ctypes.c_void_p()
        return int(self._L.lamd_gossipd_store_image(self._g, ctypes.byref(p)))

    def stats(self):
        s = Stats()
        self._L.lamd_gossipd_get_stats(self._g, ctypes.byref(s))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    def close(self):
        if self._g:
            self._L.lamd_gossipd_free(self._g)
            self._g = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
