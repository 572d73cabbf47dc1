"""Synthetic workloads of SURVEY.md 8(d) / BASELINE.json `configs`, generated ON THE GPU by the
library's own signer kernels (lamd_gen_*_device; the role devtools/mkgossip.c plays in the
reference) and then corrupted on the host with a fixed-seed mix, so every row's expected
verdict is known by construction.

cfg2: N ECDSA, 65-byte keys, K = 65 536 distinct keys, 90 % valid / 10 % invalid spread over
      {flip hash bit, flip r bit, flip s bit, high-S twin, wrong key, r = 0, s = 0, off-curve key}
cfg3: N BIP-340, x-only keys, 90/10 over {flip msg, flip r, flip s, r >= p, s >= n,
      x not liftable, negated s}
Seeds 0xC1A00002 / 0xC1A00003 (+ rank for multi-GPU weak scaling).
"""
import numpy as np
import torch

P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141

SEED_CFG2 = 0xC1A00002
SEED_CFG3 = 0xC1A00003

ECDSA_CLASSES = ("flip_hash", "flip_r", "flip_s", "high_s", "wrong_key", "r_zero", "s_zero", "offcurve_key")
SCHNORR_CLASSES = ("flip_msg", "flip_r", "flip_s", "r_ge_p", "s_ge_n", "pk_nolift", "neg_s")


def _int(row):
    return int.from_bytes(row.tobytes(), "big")


def _put(row, v):
    row[:] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)


def _liftable(x):
    if x >= P:
        return False
    c = (pow(x, 3, P) + 7) % P
    return pow(pow(c, (P + 1) // 4, P), 2, P) == c


class Workload:
    """host numpy copies + device tensors + expected verdicts"""

    def __init__(self, kind, cols, expect, classes):
        self.kind = kind
        self.cols = cols            # list of numpy uint8 [n, w]
        self.expect = expect        # numpy bool [n]
        self.classes = classes      # numpy int8 [n]: -1 valid, else index into *_CLASSES
        self.dev = None

    @property
    def n(self):
        return self.expect.shape[0]

    def to_device(self, device):
        self.dev = [torch.from_numpy(c).to(device) for c in self.cols]
        self.d_ok = torch.zeros(self.n, dtype=torch.uint8, device=device)
        torch.cuda.synchronize()   # uploads complete before any engine lane (own streams) may read them
        return self


def make_ecdsa(engine, n, seed=SEED_CFG2, nkeys=65536, publen=65, invalid_frac=0.10, device="cuda:0", group=0):
    d_hash = torch.empty((n, 32), dtype=torch.uint8, device=device)
    d_sig = torch.empty((n, 64), dtype=torch.uint8, device=device)
    d_pub = torch.empty((n, publen), dtype=torch.uint8, device=device)
    engine.gen_ecdsa_device(seed, nkeys, d_hash, d_sig, d_pub, group=group)
    engine.synchronize()
    h, s, p = d_hash.cpu().numpy(), d_sig.cpu().numpy(), d_pub.cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(seed))
    ninv = int(n * invalid_frac)
    idx = rng.choice(n, ninv, replace=False) if ninv else np.zeros(0, dtype=np.int64)
    classes = np.full(n, -1, dtype=np.int8)
    for j, i in enumerate(idx):
        c = j % len(ECDSA_CLASSES)
        name = ECDSA_CLASSES[c]
        bit = int(rng.integers(0, 256))
        if name == "flip_hash":
            h[i, bit >> 3] ^= 1 << (bit & 7)
        elif name == "flip_r":
            s[i, bit >> 3] ^= 1 << (bit & 7)
        elif name == "flip_s":
            s[i, 32 + (bit >> 3)] ^= 1 << (bit & 7)
        elif name == "high_s":
            _put(s[i, 32:], N - _int(s[i, 32:]))
        elif name == "wrong_key":
            k, tries = (int(i) + 1) % n, 0
            while np.array_equal(p[k], p[i]) and tries < 256:
                k, tries = (k + 1) % n, tries + 1
            if tries < 256:
                p[i] = p[k]
            else:                           # every row carries the same key (nkeys = 1): the signature itself is swapped instead
                k = (int(i) + 1) % n
                s[i] = s[k]
                h[i, 0] ^= 1                # (and the hash is touched, in case the neighbour signed the same message)
        elif name == "r_zero":
            s[i, :32] = 0
        elif name == "s_zero":
            s[i, 32:] = 0
        elif name == "offcurve_key":
            p[i, publen - 1 - (bit >> 3) % 32] ^= 1 << (bit & 7)  # a bit of Y (65 B) / of X (33 B)
        classes[i] = c
    w = Workload("ecdsa", [h, s, p], classes < 0, classes)
    return w.to_device(device)


def make_schnorr(engine, n, seed=SEED_CFG3, nkeys=65536, invalid_frac=0.10, device="cuda:0"):
    d_msg = torch.empty((n, 32), dtype=torch.uint8, device=device)
    d_pk = torch.empty((n, 32), dtype=torch.uint8, device=device)
    d_sig = torch.empty((n, 64), dtype=torch.uint8, device=device)
    engine.gen_schnorr_device(seed, nkeys, d_msg, d_pk, d_sig)
    engine.synchronize()
    m, k, s = d_msg.cpu().numpy(), d_pk.cpu().numpy(), d_sig.cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(seed))
    ninv = int(n * invalid_frac)
    idx = rng.choice(n, ninv, replace=False) if ninv else np.zeros(0, dtype=np.int64)
    classes = np.full(n, -1, dtype=np.int8)
    for j, i in enumerate(idx):
        c = j % len(SCHNORR_CLASSES)
        name = SCHNORR_CLASSES[c]
        bit = int(rng.integers(0, 256))
        if name == "flip_msg":
            m[i, bit >> 3] ^= 1 << (bit & 7)
        elif name == "flip_r":
            s[i, bit >> 3] ^= 1 << (bit & 7)
        elif name == "flip_s":
            s[i, 32 + (bit >> 3)] ^= 1 << (bit & 7)
        elif name == "r_ge_p":
            _put(s[i, :32], P + bit)
        elif name == "s_ge_n":
            _put(s[i, 32:], N + bit)
        elif name == "pk_nolift":
            x = _int(k[i])
            while _liftable(x):
                x = (x + 1) % (1 << 256)
            _put(k[i], x)
        elif name == "neg_s":
            _put(s[i, 32:], (N - _int(s[i, 32:])) % N)
        classes[i] = c
    w = Workload("schnorr", [m, k, s], classes < 0, classes)
    return w.to_device(device)


# ------------------------------------------------------------------------------------------------
# cfg4: gossip replay -- channel_announcements (4 signatures each) + channel_updates, on the device
CANN_LEN, CUPD_LEN = 432, 138
SEED_CFG4 = 0xC1A00004
SEED_CFG5 = 0xC1A00005


class GossipWorkload:
    pass


def make_gossip(engine, n_cann, n_cupd, n_nodes=15000, seed=SEED_CFG4, corrupt_frac=0.01, device="cuda:0"):
    """-> object with device tensors msgs/off/ids/rowbase/verdict, host copies, and the expected verdicts
    (0 = OK, k = first bad signature) known by construction."""
    n = n_cann + n_cupd
    total = n_cann * CANN_LEN + n_cupd * CUPD_LEN
    d_msgs = torch.zeros(total + 64, dtype=torch.uint8, device=device)
    d_ids = torch.zeros((n, 33), dtype=torch.uint8, device=device)
    engine.gen_gossip_device(seed, n_cann, n_cupd, n_nodes, d_msgs, d_ids)
    engine.synchronize()
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:n_cann + 1] = np.arange(1, n_cann + 1, dtype=np.uint64) * CANN_LEN
    off[n_cann + 1:] = n_cann * CANN_LEN + np.arange(1, n_cupd + 1, dtype=np.uint64) * CUPD_LEN
    rowbase = np.zeros(n + 1, dtype=np.uint64)
    rowbase[1:n_cann + 1] = np.arange(1, n_cann + 1, dtype=np.uint64) * 4
    rowbase[n_cann + 1:] = n_cann * 4 + np.arange(1, n_cupd + 1, dtype=np.uint64)
    msgs = d_msgs.cpu().numpy()
    expect = np.zeros(n, dtype=np.int8)
    rng = np.random.Generator(np.random.PCG64(seed))
    nbad = int(n * corrupt_frac)
    for i in (rng.choice(n, nbad, replace=False) if nbad else []):
        o = int(off[i])
        bit = int(rng.integers(0, 512))
        if i < n_cann:
            j = int(rng.integers(0, 4))
            msgs[o + 2 + 64 * j + (bit >> 3)] ^= 1 << (bit & 7)      # corrupt signature j -> "Bad ..._signature_j"
            expect[i] = j + 1
        else:
            msgs[o + 66 + 40 + (bit >> 3) % 32] ^= 1 << (bit & 7)      # corrupt the signed body
            expect[i] = 1
    w = GossipWorkload()
    w.n, w.n_cann, w.n_cupd, w.rows = n, n_cann, n_cupd, int(rowbase[-1])
    w.msgs, w.off, w.rowbase, w.ids, w.expect = msgs, off, rowbase, d_ids.cpu().numpy(), expect
    w.d_msgs = torch.from_numpy(msgs).to(device)
    w.d_off = torch.from_numpy(off.view(np.int64)).to(device)
    w.d_rowbase = torch.from_numpy(rowbase.view(np.int64)).to(device)
    w.d_ids = d_ids
    w.d_verdict = torch.zeros(n, dtype=torch.int8, device=device)
    torch.cuda.synchronize()   # the uploads above must have LANDED before an engine lane (its own stream) reads them
    return w


# ------------------------------------------------------------------------------------------------
# cfg5: commit_tx storm -- per channel one commitment signature under the funding key + `htlcs` HTLC
# signatures under one shared htlc key; every `bip340_every`-th channel carries BIP-340 triples instead
def make_commit_storm(engine, n_channels, htlcs=483, seed=SEED_CFG5, bip340_every=4, corrupt_frac=0.001, device="cuda:0"):
    per = htlcs + 1
    kinds = np.array([1 if (bip340_every and c % bip340_every == bip340_every - 1) else 0 for c in range(n_channels)], dtype=np.int8)
    ne, ns = int((kinds == 0).sum()), int((kinds == 1).sum())
    out = {"per": per, "kinds": kinds}
    rng = np.random.Generator(np.random.PCG64(seed))
    if ne:
        h = torch.empty((ne * per, 32), dtype=torch.uint8, device=device)
        s = torch.empty((ne * per, 64), dtype=torch.uint8, device=device)
        p = torch.empty((ne * per, 33), dtype=torch.uint8, device=device)
        engine.gen_ecdsa_device(seed, 1 << 40, h, s, p, group=per)           # htlc key: one per channel
        ch = torch.empty((ne, 32), dtype=torch.uint8, device=device)
        cs = torch.empty((ne, 64), dtype=torch.uint8, device=device)
        cp = torch.empty((ne, 33), dtype=torch.uint8, device=device)
        engine.gen_ecdsa_device(seed ^ 0x5555, 1 << 40, ch, cs, cp, group=1)  # funding key: commitment signature
        engine.synchronize()
        h.view(ne, per, 32)[:, 0, :] = ch
        s.view(ne, per, 64)[:, 0, :] = cs
        p.view(ne, per, 33)[:, 0, :] = cp
        cols = [h.cpu().numpy(), s.cpu().numpy(), p.cpu().numpy()]
        exp = np.ones(ne * per, dtype=bool)
        for i in rng.choice(ne * per, int(ne * per * corrupt_frac), replace=False):
            cols[0][i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
            exp[i] = False
        out["ecdsa"] = Workload("ecdsa", cols, exp, np.where(exp, -1, 0).astype(np.int8)).to_device(device)
    if ns:
        m = torch.empty((ns * per, 32), dtype=torch.uint8, device=device)
        k = torch.empty((ns * per, 32), dtype=torch.uint8, device=device)
        s = torch.empty((ns * per, 64), dtype=torch.uint8, device=device)
        engine.gen_schnorr_device(seed ^ 0xAAAA, 1 << 40, m, k, s, group=per)
        engine.synchronize()
        cols = [m.cpu().numpy(), k.cpu().numpy(), s.cpu().numpy()]
        exp = np.ones(ns * per, dtype=bool)
        for i in rng.choice(ns * per, int(ns * per * corrupt_frac), replace=False):
            cols[0][i, int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
            exp[i] = False
        out["schnorr"] = Workload("schnorr", cols, exp, np.where(exp, -1, 0).astype(np.int8)).to_device(device)
    return out
