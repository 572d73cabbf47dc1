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
        return self


def make_ecdsa(engine, n, seed=SEED_CFG2, nkeys=65536, publen=65, invalid_frac=0.10, device="cuda:0"):
    d_hash = torch.empty((n, 32), dtype=torch.uint8, device=device)
    d_sig = torch.empty((n, 64), dtype=torch.uint8, device=device)
    d_pub = torch.empty((n, publen), dtype=torch.uint8, device=device)
    engine.gen_ecdsa_device(seed, nkeys, d_hash, d_sig, d_pub)
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
            k = (int(i) + 1) % n
            while np.array_equal(p[k], p[i]):
                k = (k + 1) % n
            p[i] = p[k]
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
