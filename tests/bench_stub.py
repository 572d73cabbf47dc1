"""A stand-in for the MI355X under bench.py (LAMD_BENCH_STUB=<this file>): host memory plays the device, gloo plays RCCL, and a verdict is a
fixed function of the row's bytes -- so that everything bench.py does AROUND the engine (argument handling, the late-gather protocol, the
sharded configs, the ragged all-gathers, the parity accounting over ranks, the one stdout line) runs in the CPU suite at world 1, 2 and 8
(tests/test_bench_line.py).  Test infrastructure: nothing here verifies a signature, and nothing in the product imports it."""
import types

import numpy as np
import torch

SEED_CFG2, SEED_CFG3, SEED_CFG4, SEED_CFG5 = 0xC1A00002, 0xC1A00003, 0xC1A00004, 0xC1A00005
G_WINDOWS = 11


def verdict_rows(a, b, c):
    """the stub's "verification": numpy / torch uint8 [n, w] columns -> uint8 [n] (about 3 in 4 rows pass)"""
    return (((a[:, 0].to(torch.int32) + b[:, 0].to(torch.int32) + c[:, -1].to(torch.int32)) & 3) != 0).to(torch.uint8)


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


class Workload:
    def __init__(self, kind, cols):
        self.kind, self.cols = kind, cols
        self.expect = verdict_rows(*[_t(c) for c in cols]).numpy().astype(bool)
        self.dev = [torch.from_numpy(c) for c in cols]
        self.d_ok = torch.zeros(self.n, dtype=torch.uint8)

    @property
    def n(self):
        return self.cols[0].shape[0]


def _rows(seed, n, widths):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [rng.integers(0, 256, (n, w), dtype=np.uint8) for w in widths]


def make_ecdsa(engine, n, seed=SEED_CFG2, nkeys=65536, publen=65, invalid_frac=0.10, device="cpu", group=0):
    return Workload("ecdsa", _rows(seed, n, (32, 64, publen)))


def make_schnorr(engine, n, seed=SEED_CFG3, nkeys=65536, invalid_frac=0.10, device="cpu"):
    return Workload("schnorr", _rows(seed, n, (32, 32, 64)))


CANN_LEN, CUPD_LEN = 12, 6


def gossip_verdicts(msgs, off):
    """int8 per message: 0 = every signature good, k = first bad one (announcements 0..4, everything else 0..1)"""
    off = np.asarray(off, dtype=np.int64)
    m = np.asarray(msgs)
    is_cann = (m[off[:-1]] == 1) & (m[off[:-1] + 1] == 0)
    x = m[off[:-1] + 2].astype(np.int64)
    return np.where(is_cann, np.where(x % 7 == 0, 1 + x % 4, 0), (x % 11 == 0).astype(np.int64)).astype(np.int8)


def make_gossip(engine, n_cann, n_cupd, n_nodes=16, seed=SEED_CFG4, corrupt_frac=0.01, device="cpu"):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_cann + n_cupd
    msgs = rng.integers(0, 256, n_cann * CANN_LEN + n_cupd * CUPD_LEN + 64, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:n_cann + 1] = np.arange(1, n_cann + 1, dtype=np.uint64) * CANN_LEN
    off[n_cann + 1:] = n_cann * CANN_LEN + np.arange(1, n_cupd + 1, dtype=np.uint64) * CUPD_LEN
    o = off[:-1].astype(np.int64)
    msgs[o[:n_cann]], msgs[o[:n_cann] + 1] = 1, 0          # type 256
    msgs[o[n_cann:]], msgs[o[n_cann:] + 1] = 1, 2          # type 258
    rowbase = np.zeros(n + 1, dtype=np.uint64)
    rowbase[1:n_cann + 1] = np.arange(1, n_cann + 1, dtype=np.uint64) * 4
    rowbase[n_cann + 1:] = n_cann * 4 + np.arange(1, n_cupd + 1, dtype=np.uint64)
    w = types.SimpleNamespace()
    w.n, w.n_cann, w.n_cupd, w.rows = n, n_cann, n_cupd, int(rowbase[-1])
    w.msgs, w.off, w.rowbase, w.ids = msgs, off, rowbase, rng.integers(0, 256, (n, 33), dtype=np.uint8)
    w.expect = gossip_verdicts(msgs, off)
    w.d_msgs = torch.from_numpy(msgs)
    w.d_off = torch.from_numpy(off.view(np.int64))
    w.d_rowbase = torch.from_numpy(rowbase.view(np.int64))
    w.d_ids = torch.from_numpy(w.ids)
    w.d_verdict = torch.zeros(n, dtype=torch.int8)
    return w


def make_commit_storm(engine, n_channels, htlcs=483, seed=SEED_CFG5, bip340_every=4, corrupt_frac=0.001, device="cpu"):
    per = htlcs + 1
    kinds = np.array([1 if (bip340_every and c % bip340_every == bip340_every - 1) else 0 for c in range(n_channels)], dtype=np.int8)
    ne, ns = int((kinds == 0).sum()), int((kinds == 1).sum())
    return {"per": per, "kinds": kinds, "ecdsa": Workload("ecdsa", _rows(seed, ne * per, (32, 64, 33))),
            "schnorr": Workload("schnorr", _rows(seed ^ 0xAAAA, ns * per, (32, 32, 64)))}


class StubEngine:
    LANES, SETS = 2, 9

    def __init__(self, device=0):
        self.device, self.auto_order = device, True
        self.calls, self.marks, self.waits, self.pins = [], [], [], []
        self._ms = [[0.0, 0], [0.0, 0]]
        self._last = {0: (0, 0), 1: (0, 0)}
        self._queued, self._flushed, self._reserved = [], [], None
        self._sets = {}
        self._next_set = 0

    # ---- the device-pointer calls
    def _note(self, mode, n):
        self.calls.append((mode, n))
        self._ms[mode][0] += 1.0
        self._ms[mode][1] += 1
        self._last[mode] = (n, min(n, 65536))

    def verify_ecdsa_device(self, d_hash, d_sig, d_pub, d_ok):
        d_ok.copy_(verdict_rows(d_hash, d_sig, d_pub))
        self._note(0, d_ok.numel())

    def verify_schnorr_device(self, d_msg, d_xonly, d_sig, d_ok):
        d_ok.copy_(verdict_rows(d_msg, d_xonly, d_sig))
        self._note(1, d_ok.numel())

    def sigcheck_gossip_device(self, n, d_msgs, d_off, d_ids, d_rowbase, rows, d_verdict):
        assert d_off.numel() == n + 1 and d_rowbase.numel() == n + 1 and int(d_rowbase[-1] - d_rowbase[0]) == rows
        d_verdict.copy_(torch.from_numpy(gossip_verdicts(d_msgs.numpy(), d_off.numpy())))
        self._note(0, rows)

    def sigcheck_gossip_spans_device(self, n, d_msgs, d_start, d_len, d_ids, d_rowbase, rows, d_verdict):
        assert d_start.numel() == n and d_len.numel() == n and d_rowbase.numel() == n + 1 and int(d_rowbase[-1]) == rows and d_ids.shape[0] == n
        m, st = d_msgs.numpy(), d_start.numpy().astype(np.int64)
        assert (d_len.numpy() >= 3).all()
        is_cann = (m[st] == 1) & (m[st + 1] == 0)
        x = m[st + 2].astype(np.int64)
        d_verdict.copy_(torch.from_numpy(np.where(is_cann, np.where(x % 7 == 0, 1 + x % 4, 0), (x % 11 == 0).astype(np.int64)).astype(np.int8)))
        self._note(0, rows)

    # ---- the streaming queue
    def queue_ecdsa_batch(self, h, s, p):
        self._queued.append(verdict_rows(_t(h), _t(s), _t(p)).numpy().astype(bool))

    def queue_schnorr_batch(self, m, k, s):
        self._queued.append(verdict_rows(_t(m), _t(k), _t(s)).numpy().astype(bool))

    # (host memory is the stub's device: "pinning" always succeeds and in-place rows are read when they are queued)
    def host_register(self, arr):
        self.pins.append(arr.nbytes)
        return True

    def host_unregister(self, arr):
        return True

    queue_ecdsa_batch_inplace = queue_ecdsa_batch
    queue_schnorr_batch_inplace = queue_schnorr_batch

    def queue_reserve(self, n, keylen):
        key = (self._next_set % self.SETS, keylen)
        self._next_set += 1
        if key not in self._sets:
            self._sets[key] = (np.zeros((n, 32), np.uint8), np.zeros((n, 64), np.uint8), np.zeros((n, keylen), np.uint8))
        a, b, c = self._sets[key]
        self._reserved = (a, b, c, keylen)
        return 0, a, b, c

    def flush(self):
        if self._reserved is not None:
            a, b, c, keylen = self._reserved
            cols = (a, b, c) if keylen != 32 else (a, c, b)      # BIP-340 columns are (msg, x-only key, signature)
            self._queued.append(verdict_rows(*[_t(x) for x in cols]).numpy().astype(bool))
            self._reserved = None
        if self._queued:
            self._flushed.append(np.concatenate(self._queued))
            self._queued = []

    def wait(self, cap=1 << 20):
        return self._flushed.pop(0)

    # ---- ordering and bookkeeping: recorded, never waited for (host memory is always "done")
    def wait_stream(self, stream_ptr):
        self.waits.append(("stream", stream_ptr))

    def wait_event(self, event_ptr):
        self.waits.append(("event", event_ptr))

    def results_mark(self, slot):
        self.marks.append(slot)

    results_mark_last = results_mark

    def stream_wait_mark(self, slot, stream_ptr):
        assert slot in self.marks, "a gather joined a mark nobody set"

    def stream_wait_results(self, stream_ptr):
        pass

    def synchronize(self):
        pass

    def set_timing(self, on=True):
        self._ms = [[0.0, 0], [0.0, 0]]

    def set_ecmult_chain(self, on):
        pass

    def mul32_peak(self, waves_per_simd=3, min_ms=4.0, launches=5):
        return 3.4e13, 4.2, 22.0

    def info(self, lane=None):
        mode = 0 if lane is None else lane % 2
        rows, keys = self._last[mode]
        sums = [self._ms[0][0] if mode == 0 else 0.0, self._ms[1][0] if mode == 1 else 0.0]
        cnts = [self._ms[0][1] if mode == 0 else 0, self._ms[1][1] if mode == 1 else 0]
        return dict(device=self.device, compute_units=256, arch="stub", gtable_bytes=G_WINDOWS * (64 << 24), last_kernel_ms=[0.5, 1.9, 2.6, 0.1],
                    last_unique_keys=keys, last_hot_rows=rows, last_keyed=7, last_mode=mode, lanes=self.LANES, last_cache_hits=0, last_cold_rows=0,
                    last_new_tables=keys, last_suspect_rows=0, cache_enabled=False, cache_entries=0, cache_capacity=0, cache_resets=0,
                    keyed_ecmult_ms_sum=sums, keyed_ecmult_launches=cnts, hw_queues_env=16, queue_sets=self.SETS)

    def close(self):
        pass


class _Event:
    cuda_event = 0


class Platform:
    backend, is_stub = "gloo", True

    def __init__(self, local_rank):
        self.device = "cpu"
        self.Engine = StubEngine
        self.workload = types.SimpleNamespace(make_ecdsa=make_ecdsa, make_schnorr=make_schnorr, make_gossip=make_gossip, make_commit_storm=make_commit_storm,
                                              SEED_CFG2=SEED_CFG2, SEED_CFG3=SEED_CFG3)

    def synchronize(self):
        pass

    def stream_ptr(self):
        return 0

    def new_event(self):
        return _Event()
