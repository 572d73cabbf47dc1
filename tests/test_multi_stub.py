"""lamd_multi_* (several GPUs behind one host process, include/lightning_amd.h) over a STUB device layer: the "devices" are host memory, the
verification is the CPU oracle (test infrastructure), the all-gather is memcpy -- everything else is the product's code: group-aligned
sharding (lamd_shard_bounds), one host thread per device, piecewise copies, padded shards, the gather layout, the scatter into the caller's
verdict vector.  Runs at 8 "devices" on a machine without a GPU; tests/test_gpu_parity.py runs the engine back end (HIP + RCCL) at n = 1."""
import ctypes
import threading

import numpy as np
import pytest

from lightning_amd import _ffi, sharding

H = bytes.fromhex
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [ctypes.c_void_p]
VP, SZ = ctypes.c_void_p, ctypes.c_size_t
PVP = ctypes.POINTER(ctypes.c_void_p)


class Backend(ctypes.Structure):
    _fields_ = [("user", VP),
                ("dev_open", ctypes.CFUNCTYPE(ctypes.c_int, VP, ctypes.c_int, PVP)),
                ("dev_close", ctypes.CFUNCTYPE(None, VP, VP)),
                ("dev_alloc", ctypes.CFUNCTYPE(VP, VP, VP, SZ)),
                ("dev_free", ctypes.CFUNCTYPE(None, VP, VP, VP)),
                ("h2d", ctypes.CFUNCTYPE(ctypes.c_int, VP, VP, VP, VP, SZ)),
                ("d2h", ctypes.CFUNCTYPE(ctypes.c_int, VP, VP, VP, VP, SZ)),
                ("verify_ecdsa", ctypes.CFUNCTYPE(ctypes.c_int, VP, VP, SZ, VP, VP, VP, SZ, SZ, VP)),
                ("verify_schnorr", ctypes.CFUNCTYPE(ctypes.c_int, VP, VP, SZ, VP, VP, VP, VP)),
                ("sigcheck_gossip", ctypes.CFUNCTYPE(ctypes.c_int, VP, VP, SZ, VP, VP, VP, VP, SZ, VP)),
                ("gather_open", ctypes.CFUNCTYPE(ctypes.c_int, VP, PVP, ctypes.c_int)),
                ("all_gather", ctypes.CFUNCTYPE(ctypes.c_int, VP, PVP, ctypes.c_int, PVP, PVP, SZ)),
                ("gather_close", ctypes.CFUNCTYPE(None, VP, PVP, ctypes.c_int)),
                ("error", ctypes.CFUNCTYPE(ctypes.c_char_p, VP)),
                ("engine_ctx", ctypes.CFUNCTYPE(VP, VP, VP))]


def view(ptr, nbytes):
    return np.ctypeslib.as_array((ctypes.c_ubyte * nbytes).from_address(ptr))


class Stub:
    """host-memory devices; records what every device was asked to do"""

    def __init__(self, orc):
        self.orc, self.calls, self.lock, self.threads = orc, [], threading.Lock(), set()
        self.live = {}
        f = dict(Backend._fields_)
        self.be = Backend()
        self.be.user = None
        self.keep = []

        def reg(name, fn):
            cb = f[name](fn)
            self.keep.append(cb)
            setattr(self.be, name, cb)
        reg("dev_open", self.dev_open); reg("dev_close", lambda u, h: None); reg("dev_alloc", self.alloc); reg("dev_free", self.free)
        reg("h2d", self.copy); reg("d2h", self.copy); reg("verify_ecdsa", self.ecdsa); reg("verify_schnorr", self.schnorr)
        reg("sigcheck_gossip", self.gossip); reg("gather_open", lambda u, h, n: 0); reg("all_gather", self.all_gather)
        reg("gather_close", lambda u, h, n: None); reg("error", lambda u: b"stub"); reg("engine_ctx", lambda u, h: None)

    def dev_open(self, user, device, out):
        out[0] = 1000 + device          # the handle is the device number in disguise
        return 0

    def alloc(self, user, handle, nbytes):
        p = libc.malloc(nbytes)
        with self.lock:
            self.live[p] = nbytes
        return p

    def free(self, user, handle, p):
        with self.lock:
            del self.live[p]
        libc.free(p)

    def copy(self, user, handle, dst, src, nbytes):
        ctypes.memmove(dst, src, nbytes)
        return 0

    def note(self, handle, kind, n):
        with self.lock:
            self.calls.append((handle - 1000, kind, n))
            self.threads.add(threading.get_ident())

    def ecdsa(self, user, handle, n, h, s, p, publen, stride, ok):
        assert stride == publen
        self.note(handle, "ecdsa", n)
        v = self.orc.ecdsa_verify_batch(view(h, 32 * n).reshape(n, 32), view(s, 64 * n).reshape(n, 64), view(p, publen * n).reshape(n, publen), publen, 1)
        ctypes.memmove(ok, v.ctypes.data, n)
        return 0

    def schnorr(self, user, handle, n, m, x, s, ok):
        self.note(handle, "schnorr", n)
        v = self.orc.schnorr_verify_batch(view(m, 32 * n).reshape(n, 32), view(x, 32 * n).reshape(n, 32), view(s, 64 * n).reshape(n, 64), 1)
        ctypes.memmove(ok, v.ctypes.data, n)
        return 0

    def gossip(self, user, handle, n, msgs, off, ids, rowbase, rows, verdict):
        self.note(handle, "gossip", n)
        o = np.frombuffer(view(off, 8 * (n + 1)).tobytes(), dtype=np.uint64)
        rb = np.frombuffer(view(rowbase, 8 * (n + 1)).tobytes(), dtype=np.uint64)
        assert o[0] == 0 and rb[0] == 0 and rb[-1] == rows
        blob = np.concatenate([view(msgs, int(o[-1])), np.zeros(1, np.uint8)])
        idarr = view(ids, 33 * n).reshape(n, 33).copy() if ids else np.zeros((n, 33), np.uint8)
        v = self.orc.sigcheck_gossip_batch(blob, o.copy(), idarr, 1)
        # the engine's contract: rowbase counts 4 signatures per channel_announcement, 1 otherwise
        kinds = [bytes(blob[int(o[i]):int(o[i]) + 2]) for i in range(n)]
        assert [int(rb[i + 1] - rb[i]) for i in range(n)] == [4 if k == b"\x01\x00" else 1 for k in kinds]
        ctypes.memmove(verdict, v.ctypes.data, n)
        return 0

    def all_gather(self, user, handles, n, send, recv, nbytes):
        for i in range(n):
            for k in range(n):
                ctypes.memmove(recv[i] + k * nbytes, send[k], nbytes)
        return 0


@pytest.fixture()
def multi(orc):
    lib = _ffi.load()

    def make(ndev):
        st = Stub(orc)
        m = ctypes.c_void_p()
        rc = lib.lamd_multi_init_backend(ctypes.byref(m), None, ndev, ctypes.cast(ctypes.pointer(st.be), ctypes.c_void_p))
        assert rc == 0, lib.lamd_multi_last_error(m)
        made.append((m, st))
        return m, st
    made = []
    yield lib, make
    for m, st in made:
        lib.lamd_multi_shutdown(m)
        assert not st.live, "device buffers leaked"


def test_shard_bounds_equal_the_python_partition():
    """lamd_shard_bounds (C, what a sidecar calls) cuts exactly where lightning_amd.sharding.shard_bounds (Python, what bench.py --gpus N and the
    gloo tests use) cuts: whole groups, balanced by rows, shards may be empty"""
    lib = _ffi.load()
    rnd = np.random.default_rng(5)
    for _ in range(300):
        ng = int(rnd.integers(1, 60))
        sizes = rnd.integers(1, 9, ng).astype(np.uint32) if rnd.random() < 0.7 else np.full(ng, 484, np.uint32)
        world = int(rnd.integers(1, 10))
        bg, br = np.zeros(world + 1, np.uint64), np.zeros(world + 1, np.uint64)
        assert lib.lamd_shard_bounds(ng, sizes.ctypes.data, world, bg.ctypes.data, br.ctypes.data) == 0
        want = sharding.shard_bounds(int(sizes.sum()), world, sizes)
        assert list(br) == list(want), (list(sizes), world)
        ends = np.concatenate([[0], np.cumsum(sizes)])
        assert all(int(ends[int(g)]) == int(r) for g, r in zip(bg, br))


@pytest.mark.parametrize("ndev", [1, 3, 8])
def test_multi_ecdsa_schnorr_over_stub_devices_equal_the_single_call(multi, orc, ndev):
    lib, make = multi
    m, st = make(ndev)
    for n, group, publen in ((10_007, 484, 33), (4_096, 1, 65), (12, 4, 33), (1, 484, 65)):
        h, s, p, c, e = orc.gen_ecdsa_edge_batch(0xA11CE + n, n, publen, 4)
        ok = np.full(n, 7, np.uint8)
        st.calls.clear()
        rc = lib.lamd_multi_verify_ecdsa_batch(m, n, h.ctypes.data, s.ctypes.data, p.ctypes.data, publen, publen, group, ok.ctypes.data)
        assert rc == 0, lib.lamd_multi_last_error(m)
        assert np.array_equal(ok, e), (n, group, publen)
        per_dev = {}
        for d, kind, rows in st.calls:
            per_dev[d] = per_dev.get(d, 0) + rows
        assert sum(per_dev.values()) == n
        # every device's range is whole groups (the last range takes the remainder)
        order = sorted(per_dev)
        assert all(per_dev[d] % group == 0 for d in order[:-1]), per_dev
        if n >= 8 * group * 2:
            assert len(per_dev) == ndev and max(per_dev.values()) - min(per_dev.values()) <= group
    m_, x, sg, c, e = orc.gen_schnorr_edge_batch(0xB0B, 5_000, 4)
    ok = np.full(5_000, 7, np.uint8)
    assert lib.lamd_multi_verify_schnorr_batch(m, 5_000, m_.ctypes.data, x.ctypes.data, sg.ctypes.data, 1, ok.ctypes.data) == 0
    assert np.array_equal(ok, e)
    # strided keys (a 65-byte key inside a 72-byte record) are packed for the device
    h, s, p, c, e = orc.gen_ecdsa_edge_batch(77, 999, 65, 4)
    wide = np.zeros((999, 72), np.uint8)
    wide[:, :65] = p
    ok = np.zeros(999, np.uint8)
    assert lib.lamd_multi_verify_ecdsa_batch(m, 999, h.ctypes.data, s.ctypes.data, wide.ctypes.data, 65, 72, 1, ok.ctypes.data) == 0
    assert np.array_equal(ok, e)
    if ndev > 1:
        assert len(st.threads) >= 2, "the devices' work must run on their own host threads"


def test_multi_gossip_over_8_stub_devices(multi, orc, kat):
    """raw wire messages sharded by message, balanced by signatures; a channel_announcement's four signatures never straddle two devices"""
    lib, make = multi
    m, st = make(8)
    vs = kat["gossip"]
    msgs = [H(v["msg"]) for v in vs] * 3
    ids = [H(v["node_id"]) if "node_id" in v else bytes(33) for v in vs] * 3
    blob = np.frombuffer(b"".join(msgs) + b"\x00", dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum([len(x) for x in msgs])]).astype(np.uint64)
    idarr = np.frombuffer(b"".join(ids), dtype=np.uint8).reshape(len(ids), 33).copy()
    want = orc.sigcheck_gossip_batch(blob, off, idarr, 4)
    got = np.full(len(msgs), 99, np.int8)
    rc = lib.lamd_multi_sigcheck_gossip_batch(m, len(msgs), blob.ctypes.data, off.ctypes.data, idarr.ctypes.data, got.ctypes.data)
    assert rc == 0, lib.lamd_multi_last_error(m)
    assert np.array_equal(got, want)
    assert sum(n for _, k, n in st.calls if k == "gossip") == len(msgs) and len({d for d, _, _ in st.calls}) == 8
    assert list(want[:len(vs)]) == [v["expect"] for v in vs]
    # the shards are the ones lightning_amd.sharding cuts for the same batch (message boundaries, balanced by cost: GOSSIP_WEIGHT_*)
    from lightning_amd import sharding
    b = sharding.shard_bounds(len(msgs), 8, None, sharding.gossip_weights(blob, off))
    per_dev = {}
    for d, k, n in st.calls:
        if k == "gossip":
            per_dev[d] = per_dev.get(d, 0) + n
    assert sorted(per_dev.values()) == sorted(int(b[i + 1] - b[i]) for i in range(8))
    # a channel_update without node ids is refused up front (as lamd_sigcheck_gossip_batch does), nothing reaches a device
    before = len(st.calls)
    rc = lib.lamd_multi_sigcheck_gossip_batch(m, len(msgs), blob.ctypes.data, off.ctypes.data, None, got.ctypes.data)
    assert rc == -3 and b"node_ids33 is NULL" in lib.lamd_multi_last_error(m) and len(st.calls) == before


def test_multi_gossip_replay_is_cut_kind_by_kind(multi, orc, kat):
    """a replay -- every channel_announcement first, then everything else -- is cut run by run: device i takes range i of the announcements AND range
    i of the rest in ONE engine call (its ranges back to back in its buffers), the verdicts come back in job order, and the cuts are the ones
    lightning_amd.sharding.segment_bounds makes for the same batch"""
    from lightning_amd import sharding
    lib, make = multi
    m, st = make(8)
    vs = sorted(kat["gossip"] * 5, key=lambda v: 0 if v["msg"][:4] == "0100" else 1)    # stable: announcements | the rest
    msgs = [H(v["msg"]) for v in vs]
    n_cann = sum(1 for x in msgs if x[:2] == b"\x01\x00")
    assert 0 < n_cann < len(msgs)
    ids = [H(v["node_id"]) if "node_id" in v else bytes(33) for v in vs]
    blob = np.frombuffer(b"".join(msgs) + b"\x00", dtype=np.uint8).copy()
    off = np.concatenate([[0], np.cumsum([len(x) for x in msgs])]).astype(np.uint64)
    idarr = np.frombuffer(b"".join(ids), dtype=np.uint8).reshape(len(ids), 33).copy()
    want = orc.sigcheck_gossip_batch(blob, off, idarr, 4)
    assert list(want) == [v["expect"] for v in vs] and len(set(want.tolist())) > 1
    got = np.full(len(msgs), 99, np.int8)
    rc = lib.lamd_multi_sigcheck_gossip_batch(m, len(msgs), blob.ctypes.data, off.ctypes.data, idarr.ctypes.data, got.ctypes.data)
    assert rc == 0, lib.lamd_multi_last_error(m)
    assert np.array_equal(got, want)
    sb = sharding.segment_bounds([0, n_cann, len(msgs)], 8, sharding.gossip_weights(blob, off))
    calls = [(d, n) for d, k, n in st.calls if k == "gossip"]
    assert len(calls) == len({d for d, _ in calls}), "one engine call per device"
    assert sorted(n for _, n in calls) == sorted(int(sb[0, i + 1] - sb[0, i] + sb[1, i + 1] - sb[1, i]) for i in range(8) if sb[0, i + 1] - sb[0, i] + sb[1, i + 1] - sb[1, i])
    # a batch of ONE kind, and one with more runs than the host cuts by (interleaved: uniform already), still verify -- one cut
    for sub in (slice(0, n_cann), slice(n_cann, len(msgs))):
        o2 = (off[sub.start:sub.stop + 1] - off[sub.start]).astype(np.uint64)
        b2 = np.concatenate([blob[int(off[sub.start]):int(off[sub.stop])], np.zeros(1, np.uint8)])
        g2 = np.full(sub.stop - sub.start, 99, np.int8)
        i2 = idarr[sub].copy()
        assert lib.lamd_multi_sigcheck_gossip_batch(m, len(g2), b2.ctypes.data, o2.ctypes.data, i2.ctypes.data, g2.ctypes.data) == 0
        assert np.array_equal(g2, want[sub])
