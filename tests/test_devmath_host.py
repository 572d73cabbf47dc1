"""Drives the DEVICE arithmetic compiled for the host (tests/devmath_host.cpp: same headers the
HIP kernels include, magnitude assertions on) against Python big-ints, the golden vectors and
the oracle.  CPU only; this is how the kernels' math is debugged before it reaches a GPU."""
import ctypes
import hashlib
import os
import random
import subprocess

import numpy as np
import pytest

import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
P, N = pyref.P, pyref.N
H = bytes.fromhex


@pytest.fixture(scope="session")
def dm():
    so = os.path.join(HERE, "libdevmath_host.so")
    srcs = [os.path.join(HERE, "devmath_host.cpp")] + [os.path.join(ROOT, "lightning_amd", "csrc", f)
                                                       for f in ("lamd_common.h", "fe.h", "scalar.h", "group.h", "sha256.h", "verify_core.h", "fuzz.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, srcs[0]])
    L = ctypes.CDLL(so)
    L.dm_fe_op.restype = ctypes.c_int
    L.dm_fe_sqrt.restype = ctypes.c_int
    L.dm_words_ge_p.restype = ctypes.c_int
    L.dm_sc_from.restype = ctypes.c_int
    L.dm_sc_is_high.restype = ctypes.c_int
    L.dm_parse_pubkey.restype = ctypes.c_int
    L.dm_init()
    return L


U9 = ctypes.c_uint32 * 9


def limbs_val(l):
    return sum(int(v) << (29 * i) for i, v in enumerate(l))


def rand_fe(rnd, mag):
    lim29, lim24 = (1 << 29) + (1 << 13), (1 << 24) + (1 << 13)
    mode = rnd.random()
    if mode < 0.2:
        return [mag * lim29] * 8 + [mag * lim24]          # every limb at its bound
    if mode < 0.3:
        return [0] * 9
    return [rnd.randrange(mag * lim29 + 1) for _ in range(8)] + [rnd.randrange(mag * lim24 + 1)]


def fe_op(dm, op, a, amag, b=None, bmag=0):
    out = U9()
    om = ctypes.c_int()
    rc = dm.dm_fe_op(op, U9(*a), amag, U9(*(b or [0] * 9)), bmag, out, ctypes.byref(om))
    assert rc == 0
    return list(out), om.value


def test_fe_ops_at_magnitude_bounds(dm):
    rnd = random.Random(1)
    for _ in range(3000):
        ma, mb = rnd.randrange(1, 8), rnd.randrange(1, 8)
        a, b = rand_fe(rnd, ma), rand_fe(rnd, mb)
        va, vb = limbs_val(a) % P, limbs_val(b) % P
        if ma + mb <= 7:
            r, m = fe_op(dm, 0, a, ma, b, mb)
            assert limbs_val(r) % P == (va + vb) % P and m == ma + mb
        if ma + 1 <= 7:
            r, m = fe_op(dm, 1, a, ma)
            assert limbs_val(r) % P == (-va) % P and m == ma + 1
        if ma * mb <= 7:
            r, m = fe_op(dm, 2, a, ma, b, mb)
            assert limbs_val(r) % P == va * vb % P and m == 1
        if ma <= 2:
            r, m = fe_op(dm, 3, a, ma)
            assert limbs_val(r) % P == va * va % P
        for op in (4, 5):
            r, m = fe_op(dm, op, a, ma)
            assert limbs_val(r) % P == va
        r, m = fe_op(dm, 6, a, ma)
        assert limbs_val(r) == va  # canonical
        assert all(x < (1 << 29) for x in r[:8]) and r[8] < (1 << 24)
        k = rnd.randrange(1, 8)
        if ma * k <= 7:
            r, m = fe_op(dm, 7, a, ma, None, k)
            assert limbs_val(r) % P == va * k % P
        r, _ = fe_op(dm, 8, a, ma)
        assert r[0] == (va == 0)


def test_fe_zero_detection_on_multiples_of_p(dm):
    # every representation of 0 reachable with lazy limbs: k*p spread over the limbs in different ways
    rnd = random.Random(2)
    pl = [0x1FFFFC2F, 0x1FFFFFF7] + [0x1FFFFFFF] * 6 + [0xFFFFFF]
    for k in range(0, 8):
        a = [k * x for x in pl]
        r, _ = fe_op(dm, 8, a, max(k, 1))
        assert r[0] == 1
        r, _ = fe_op(dm, 6, a, max(k, 1))
        assert limbs_val(r) == 0
        # perturb: +1 must be non-zero
        a2 = list(a); a2[0] += 1
        r, _ = fe_op(dm, 8, a2, min(7, k + 1))
        assert r[0] == 0
    # values just below / above p and 2^256
    for v in (P - 1, P, P + 1, 2**256 - 1, 1, 0, P - 977, 2**256 - 2**32 - 978):
        a = [(v >> (29 * i)) & 0x1FFFFFFF for i in range(8)] + [v >> 232]
        r, _ = fe_op(dm, 6, a, 1)
        assert limbs_val(r) == v % P


def test_fe_conversions_inv_sqrt(dm):
    rnd = random.Random(3)
    for i in range(200):
        v = rnd.randrange(P) if i > 8 else [0, 1, P - 1, P - 2, 2**255, 977, 2**32 + 977, 2**232, 2**29 - 1][i]
        b = v.to_bytes(32, "big")
        l = U9()
        dm.dm_fe_from_be(b, l)
        assert limbs_val(l) == v
        o = ctypes.create_string_buffer(32)
        dm.dm_fe_to_be(l, 1, o)
        assert o.raw == b
        if v:
            dm.dm_fe_inv(b, o)
            assert int.from_bytes(o.raw, "big") == pow(v, -1, P)
        ok = dm.dm_fe_sqrt(b, o)
        is_qr = pow(v, (P - 1) // 2, P) in (0, 1)
        assert bool(ok) == is_qr
        if is_qr:
            assert pow(int.from_bytes(o.raw, "big"), 2, P) == v
    for v in (P - 1, P, P + 1, 2**256 - 1, 0, P - 2**32, (P | (1 << 32)) & (2**256 - 1)):
        assert bool(dm.dm_words_ge_p(v.to_bytes(32, "big"))) == (v >= P)


def test_scalar_ops(dm):
    rnd = random.Random(4)
    o = ctypes.create_string_buffer(32)
    specials = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, (N + 1) // 2, 2**128, 2**255, N >> 1]
    for i in range(300):
        a = specials[i % len(specials)] if i < 40 else rnd.randrange(N)
        b = specials[(i // len(specials)) % len(specials)] if i < 40 else rnd.randrange(N)
        dm.dm_sc_mul(a.to_bytes(32, "big"), b.to_bytes(32, "big"), o)
        assert int.from_bytes(o.raw, "big") == a * b % N
        dm.dm_sc_neg(a.to_bytes(32, "big"), o)
        assert int.from_bytes(o.raw, "big") == (-a) % N
        assert bool(dm.dm_sc_is_high(a.to_bytes(32, "big"))) == (a > N // 2)
        raw = rnd.randrange(2**256) if i % 3 else N + rnd.randrange(2**256 - N)
        of = dm.dm_sc_from(raw.to_bytes(32, "big"), o)
        assert int.from_bytes(o.raw, "big") == raw % N and bool(of) == (raw >= N)
    for i in range(25):
        a = [1, 2, N - 1, 3][i] if i < 4 else rnd.randrange(1, N)
        dm.dm_sc_inv(a.to_bytes(32, "big"), o)
        assert int.from_bytes(o.raw, "big") == pow(a, -1, N)


LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def test_glv_split(dm):
    rnd = random.Random(5)
    out = (ctypes.c_uint32 * 12)()
    bias = int("8" * 32, 16)
    for i in range(3000):
        k = [0, 1, N - 1, LAM, N - LAM, 2**128, 2**128 - 1, (N - 1) // 2][i] if i < 8 else rnd.randrange(N)
        dm.dm_glv_split(k.to_bytes(32, "big"), out)
        vals = []
        for off in (0, 6):
            mag = sum(int(out[off + j]) << (32 * j) for j in range(4)) + (int(out[off + 4]) << 128) - bias
            assert 0 <= mag < 2**128
            # digits as the ladder reads them
            digs = [((int(out[off + (w >> 3)]) >> ((w & 7) * 4)) & 15) - 8 for w in range(32)] + [int(out[off + 4])]
            assert sum(d * 16**w for w, d in enumerate(digs)) == mag and all(-8 <= d <= 8 for d in digs)
            vals.append(-mag if out[off + 5] else mag)
        assert (vals[0] + vals[1] * LAM - k) % N == 0


def test_bip340_challenge(dm):
    rnd = random.Random(6)
    o = ctypes.create_string_buffer(32)
    for _ in range(50):
        r, pk, m = rnd.randbytes(32), rnd.randbytes(32), rnd.randbytes(32)
        dm.dm_bip340_challenge(r, pk, m, o)
        assert o.raw == pyref.tagged_hash("BIP0340/challenge", r + pk + m)


def test_parse_pubkey_and_gtable(dm, kat):
    o = ctypes.create_string_buffer(64)
    for v in kat["pubkey"]:
        pub = H(v["pub"])
        if len(pub) not in (33, 65):
            continue
        ok = dm.dm_parse_pubkey(pub, len(pub), o)
        assert bool(ok) == (v["expect"] is not None), v["pub"]
        if ok:
            assert o.raw == H(v["expect"])
    rnd = random.Random(7)
    for _ in range(40):
        w, d = rnd.randrange(32), rnd.randrange(1, 256)
        dm.dm_gtable_entry(w, d, o)
        pt = pyref.pmul(d << (8 * w), pyref.G)
        assert o.raw == pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def _run_ecdsa(dm, rows, threads):
    n = len(rows)
    publen = len(rows[0][2])
    hs = b"".join(r[0] for r in rows)
    sg = b"".join(r[1] for r in rows)
    pk = b"".join(r[2] for r in rows)
    out = ctypes.create_string_buffer(n)
    dm.dm_ecdsa_verify_batch(ctypes.c_size_t(n), hs, sg, pk, publen, publen, out, ctypes.c_size_t(threads))
    return [bool(b) for b in out.raw]


def test_pipeline_golden_ecdsa(dm, kat):
    for publen in (33, 65):
        rows = [(H(v["hash"]), H(v["sig"]), H(v["pub"]), v["expect"], v["name"]) for v in kat["ecdsa"] if len(v["pub"]) == 2 * publen]
        assert len(rows) > 100
        for threads in (1, 7, len(rows)):
            got = _run_ecdsa(dm, rows, threads)
            bad = [r[4] for r, g in zip(rows, got) if g != r[3]]
            assert not bad, (publen, threads, bad[:10])


def test_pipeline_golden_schnorr(dm, kat):
    rows = kat["schnorr"]
    n = len(rows)
    out = ctypes.create_string_buffer(n)
    dm.dm_schnorr_verify_batch(ctypes.c_size_t(n), b"".join(H(v["msg"]) for v in rows), b"".join(H(v["pk"]) for v in rows),
                               b"".join(H(v["sig"]) for v in rows), out)
    bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
    assert not bad, bad[:10]
    # the two-stage form the kernels use (stage 1 in the ecmult kernel, shared inversion in k_schnorr_final)
    for threads in (1, 5, n):
        out2 = ctypes.create_string_buffer(n)
        dm.dm_schnorr_verify_batch2(ctypes.c_size_t(n), b"".join(H(v["msg"]) for v in rows), b"".join(H(v["pk"]) for v in rows),
                                    b"".join(H(v["sig"]) for v in rows), out2, ctypes.c_size_t(threads))
        assert out2.raw == out.raw, threads


def test_pipeline_random_vs_oracle(dm, orc):
    rnd = random.Random(8)
    rows = []
    for i in range(120):
        d = rnd.randrange(1, N).to_bytes(32, "big")
        h = rnd.randbytes(32)
        sig = orc.ecdsa_sign(h, d, rnd.randrange(1, N).to_bytes(32, "big"))
        pub = orc.pubkey_create(d)
        pub33 = bytes([2 + (pub[64] & 1)]) + pub[1:33]
        c = i % 4
        if c == 1:
            h = bytes([h[0] ^ 0x40]) + h[1:]
        elif c == 2:
            sig = sig[:40] + bytes([sig[40] ^ 1]) + sig[41:]
        rows.append((h, sig, pub33))
    got = _run_ecdsa(dm, rows, 5)
    exp = [orc.ecdsa_verify(*r) for r in rows]
    assert got == exp and sum(exp) >= 50


def test_keyed_path_tables_and_verdicts(dm, kat):
    """per-key comb tables (T teeth, spacing D) hold 2^((T-1)D)*Q + sum_i +-2^(iD)*Q as affine points of the key's isomorphic
    curve, plus Q itself, and the table-driven ecmult gives the golden verdicts for ECDSA (33/65-byte keys) and BIP-340"""
    BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    d = 0x1F2E3D4C5B6A79881726354453627180AABBCCDDEEFF00112233445566778899
    Q = pyref.pubkey_create(d)
    o = ctypes.create_string_buffer(96)
    for T in (7, 8, 9, 10):
        D = dm.dm_comb_spacing(T)
        assert T * D >= 129
        ne = 1 << (T - 1)
        for idx in sorted({0, 1, 2, 3, 15, 16, 17, 31, ne // 2 - 1, ne // 2, ne - 2, ne - 1, ne, 0x2A % ne, 0x55 % ne}):
            dm.dm_keytable_entry(pyref.ser33(Q), T, idx, o)
            if idx == ne:
                k = 1
            else:
                k = (1 << ((T - 1) * D)) + sum((1 if (idx >> i) & 1 else -1) << (i * D) for i in range(T - 1))
            pt = pyref.pmul(k % N, Q)
            assert o.raw[:32] == pt[0].to_bytes(32, "big") and o.raw[32:64] == pt[1].to_bytes(32, "big"), (T, idx)
            assert int.from_bytes(o.raw[64:], "big") == BETA * pt[0] % P
    for T in (7, 8, 9, 10):
        for publen in (33, 65):
            rows = [v for v in kat["ecdsa"] if len(v["pub"]) == 2 * publen]
            rows = rows[:80] + rows[-40:]   # reference KATs + edge classes + special keys / key-less / x(R) = r + n
            out = ctypes.create_string_buffer(len(rows))
            dm.dm_verify_keyed(0, T, ctypes.c_size_t(len(rows)), b"".join(H(v["hash"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                               b"".join(H(v["pub"]) for v in rows), publen, out)
            bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
            assert not bad, (T, bad[:10])
        rows = kat["schnorr"][:60]
        out = ctypes.create_string_buffer(len(rows))
        dm.dm_verify_keyed(1, T, ctypes.c_size_t(len(rows)), b"".join(H(v["msg"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                           b"".join(H(v["pk"]) for v in rows), 32, out)
        bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
        assert not bad, (T, bad[:10])


@pytest.mark.parametrize("nk", [1, 2, 4])
def test_tree_builder_gives_the_same_tables_and_verdicts(dm, kat, nk):
    """round 6, kc_tree_affine: the key tables as a doubling tree of affine additions with the inversions shared by the lane's nk keys -- every entry the
    same POINT as the Gray-code chains make (checked against pyref like them), Zc = Zb, and the table-driven ecmult over such tables gives the golden
    verdicts (ECDSA 33/65-byte keys, BIP-340).  The key under test shares its inversions with nk - 1 other keys (2Q, 4Q, ..)."""
    dm.dm_set_table_builder(nk)
    try:
        test_keyed_path_tables_and_verdicts(dm, kat)
        test_keyed_ecmult_special_scalars(dm)
    finally:
        dm.dm_set_table_builder(0)


def test_pairs_first_ecmult_gives_the_golden_verdicts(dm, kat):
    """verify_core.h "Pairs first": half of a verification's mixed additions replaced by affine + affine additions whose inverses
    come from ONE inversion per batch of rows (Montgomery's trick; the inversion by division steps).  Every ECDSA golden (reference
    KATs, all edge classes, special keys) and every BIP-340 golden, comb shapes 7 / 10, batches of 1 .. PAIRS_BMAX rows -- the rows
    with degenerate sums (R = infinity, u1*G = +-u2*Q) must come back as suspects and be decided by the complete formulas."""
    dm.dm_verify_pairs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    dm.dm_suspects(1)
    for T in (7, 10):
        for batch in (1, 2, 5, 6):
            for publen in (33, 65):
                rows = [v for v in kat["ecdsa"] if len(v["pub"]) == 2 * publen]
                if batch not in (1, 6):
                    rows = rows[:40] + rows[-40:]
                out = ctypes.create_string_buffer(len(rows))
                dm.dm_verify_pairs(0, T, len(rows), batch, b"".join(H(v["hash"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                                   b"".join(H(v["pub"]) for v in rows), publen, out)
                bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
                assert not bad, (T, batch, bad[:10])
            rows = kat["schnorr"][:60]
            out = ctypes.create_string_buffer(len(rows))
            dm.dm_verify_pairs(1, T, len(rows), batch, b"".join(H(v["msg"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                               b"".join(H(v["pk"]) for v in rows), 32, out)
            bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
            assert not bad, (T, batch, bad[:10])
    assert dm.dm_suspects(0) > 0   # the edge classes reached the complete formulas


def test_task_split_of_the_latency_path_gives_the_golden_verdicts(dm, kat):
    """k_small_verify's arithmetic on the host: one verification cut into five independent partial sums (u1*G; the comb's two
    GLV halves, each split at a column -- or the two ladder halves for a key without a table) merged with complete Jacobian
    additions (gej_add_var).  Every ECDSA golden (reference KATs, all edge classes: R = infinity, u1*G = +-u2*Q, Q = G, Q = lambda*G,
    lattice-vector scalars, x(R) = r + n) and every BIP-340 golden through all three shapes."""
    for T in (7, 10, 0):
        for publen in (33, 65):
            rows = [v for v in kat["ecdsa"] if len(v["pub"]) == 2 * publen]
            out = ctypes.create_string_buffer(len(rows))
            dm.dm_verify_split(0, T, ctypes.c_size_t(len(rows)), b"".join(H(v["hash"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                               b"".join(H(v["pub"]) for v in rows), publen, out)
            bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
            assert not bad, (T, publen, bad[:10])
        rows = kat["schnorr"]
        out = ctypes.create_string_buffer(len(rows))
        dm.dm_verify_split(1, T, ctypes.c_size_t(len(rows)), b"".join(H(v["msg"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                           b"".join(H(v["pk"]) for v in rows), 32, out)
        bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
        assert not bad, (T, bad[:10])


def test_keyed_ecmult_special_scalars(dm):
    """u1*G + u2*Q through the comb for scalars that stress the recoding: zero / even / tiny GLV halves, halves that cancel,
    u2 = +-lambda^i (one half exactly +-1), the partial sums that meet -u1*G, and seeded random ones"""
    LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    d = 0x6C1F00D5A3E2B4C7918D7E6F5A4B3C2D1E0F99887766554433221100FFEEDDCC
    Q = pyref.pubkey_create(d)
    rng = random.Random(77)
    u2s = [0, 1, 2, 3, 4, N - 1, N - 2, LAM, N - LAM, LAM + 1, LAM - 1, 2 * LAM % N, (LAM + 2) % N, LAM * LAM % N, (N - LAM * LAM) % N,
           1 << 127, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, ((1 << 127) * LAM + 2) % N, (N + 1) // 2]
    u2s += [rng.randrange(N) for _ in range(12)]
    o = ctypes.create_string_buffer(64)
    dm.dm_suspects.restype = ctypes.c_size_t
    for T in (7, 10):
        dm.dm_suspects(1)
        ninf = 0
        for u2 in u2s:
            for u1 in (0, 1, rng.randrange(N), (-u2 * d) % N, (-u2 * d + 1) % N):
                # dm_ecmult_keyed runs the bare-formula form (both halves made odd by a lattice vector, no repair additions, one
                # Z == 0 test at the end) AND the complete form, and returns -1 if they describe different points
                got = dm.dm_ecmult_keyed(T, pyref.ser33(Q), u1.to_bytes(32, "big"), u2.to_bytes(32, "big"), o)
                exp = pyref.padd(pyref.pmul(u1, pyref.G), pyref.pmul(u2, Q))
                if exp is None:
                    assert got == 0, (T, hex(u1), hex(u2))
                    ninf += 1
                else:
                    assert got == 1 and o.raw == exp[0].to_bytes(32, "big") + exp[1].to_bytes(32, "big"), (T, hex(u1), hex(u2))
        # every infinity result (and every crafted collision) went through the complete form; honest random scalars never do
        assert dm.dm_suspects(1) >= ninf > 0
        for _ in range(40):
            u1, u2 = rng.randrange(N), rng.randrange(N)
            assert dm.dm_ecmult_keyed(T, pyref.ser33(Q), u1.to_bytes(32, "big"), u2.to_bytes(32, "big"), o) == 1
            exp = pyref.padd(pyref.pmul(u1, pyref.G), pyref.pmul(u2, Q))
            assert o.raw == exp[0].to_bytes(32, "big") + exp[1].to_bytes(32, "big")
        assert dm.dm_suspects(1) == 0


def test_ladder_hot_form_special_scalars_and_its_table(dm):
    """the per-signature ladder in its hot form (ecmult_lane_fast: both halves odd, signed odd 4-bit digits, the table 1Q, 3Q .. 15Q, bare
    additions, one Z == 0 test) against the complete ladder and the model: the table's entries are (2e+1)Q on the curve its Z names; scalars that
    stress the recoding (zero / even / tiny halves, halves that cancel, +-lambda^i, every small u2, partial sums that meet -u1*G) and random ones"""
    LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    rng = random.Random(1515)
    dm.dm_ladder_suspects.restype = ctypes.c_size_t
    ent, zg = ctypes.create_string_buffer(8 * 96), ctypes.create_string_buffer(32)
    for d in (1, 2, 3, 0x6C1F00D5A3E2B4C7918D7E6F5A4B3C2D1E0F99887766554433221100FFEEDDCC, rng.randrange(1, N), N - 1):
        Q = pyref.pubkey_create(d)
        dm.dm_odd_table(pyref.ser33(Q), ent, zg)
        z = int.from_bytes(zg.raw, "big")
        zi = pow(z, P - 2, P)
        for e in range(8):
            x, bx, y = (int.from_bytes(ent.raw[96 * e + 32 * c:96 * e + 32 * c + 32], "big") for c in range(3))
            want = pyref.pmul(2 * e + 1, Q)
            assert (x * zi * zi % P, y * pow(zi, 3, P) % P) == want, (hex(d), e)       # affine on y^2 = x^3 + 7 z^6 -> secp256k1
            assert bx == x * BETA % P
    d = 0x6C1F00D5A3E2B4C7918D7E6F5A4B3C2D1E0F99887766554433221100FFEEDDCC
    Q = pyref.pubkey_create(d)
    u2s = list(range(0, 40)) + [N - 1, N - 2, N - 3, LAM, N - LAM, LAM + 1, LAM - 1, 2 * LAM % N, (LAM + 2) % N, 3 * LAM % N, (15 * LAM) % N, (16 * LAM + 1) % N,
                                LAM * LAM % N, (N - LAM * LAM) % N, 1 << 127, (1 << 128) - 1, 1 << 128, (1 << 128) + 1, ((1 << 127) * LAM + 2) % N, (N + 1) // 2,
                                (1 << 129) - 1, ((1 << 128) - 1) * LAM % N, (0x88888888888888888888888888888888 * (LAM + 1)) % N]
    u2s += [rng.randrange(N) for _ in range(40)]
    o = ctypes.create_string_buffer(64)
    dm.dm_ladder_suspects(1)
    ninf = 0
    for u2 in u2s:
        for u1 in (0, 1, rng.randrange(N), (-u2 * d) % N, (-u2 * d + 1) % N):
            got = dm.dm_ecmult_ladder(pyref.ser33(Q), u1.to_bytes(32, "big"), u2.to_bytes(32, "big"), o)      # -1: the two forms disagree
            exp = pyref.padd(pyref.pmul(u1, pyref.G), pyref.pmul(u2, Q))
            if exp is None:
                assert got == 0, (hex(u1), hex(u2))
                ninf += 1
            else:
                assert got == 1 and o.raw == exp[0].to_bytes(32, "big") + exp[1].to_bytes(32, "big"), (hex(u1), hex(u2))
    assert dm.dm_ladder_suspects(1) >= ninf > 0          # every result at infinity went through the complete form
    for _ in range(300):                                 # honest random scalars and keys never do
        Qr = pyref.pubkey_create(rng.randrange(1, N))
        u1, u2 = rng.randrange(N), rng.randrange(N)
        assert dm.dm_ecmult_ladder(pyref.ser33(Qr), u1.to_bytes(32, "big"), u2.to_bytes(32, "big"), o) == 1
        exp = pyref.padd(pyref.pmul(u1, pyref.G), pyref.pmul(u2, Qr))
        assert o.raw == exp[0].to_bytes(32, "big") + exp[1].to_bytes(32, "big")
    assert dm.dm_ladder_suspects(1) == 0


def test_gtable_windows_that_straddle_words(kat):
    """the shipped G table uses 24-bit windows (11 windows, the last one runs past bit 255, two of three straddle a 32-bit word);
    the same digit extraction and table code built for the host with 11-bit windows must give the golden verdicts"""
    so = os.path.join(HERE, "libdevmath_host_w11.so")
    src = os.path.join(HERE, "devmath_host.cpp")
    hdr = os.path.join(ROOT, "lightning_amd", "csrc", "verify_core.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DLAMD_GTABLE_WINDOW_BITS=11", "-DDM_NO_KEYED", "-o", so, src])
    L = ctypes.CDLL(so)
    L.dm_init()
    o = ctypes.create_string_buffer(64)
    for w, d in ((0, 1), (2, 2047), (5, 1000), (23, 7)):          # window 23 holds bits 253..263
        L.dm_gtable_entry(w, d, o)
        pt = pyref.pmul(d << (11 * w), pyref.G)
        assert o.raw == pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big"), (w, d)
    for publen in (33, 65):
        rows = [v for v in kat["ecdsa"] if len(v["pub"]) == 2 * publen][:150]
        out = ctypes.create_string_buffer(len(rows))
        L.dm_ecdsa_verify_batch(ctypes.c_size_t(len(rows)), b"".join(H(v["hash"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                                b"".join(H(v["pub"]) for v in rows), publen, publen, out, ctypes.c_size_t(4))
        bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
        assert not bad, bad[:10]
    rows = kat["schnorr"][:40]
    out = ctypes.create_string_buffer(len(rows))
    L.dm_schnorr_verify_batch(ctypes.c_size_t(len(rows)), b"".join(H(v["msg"]) for v in rows), b"".join(H(v["pk"]) for v in rows),
                              b"".join(H(v["sig"]) for v in rows), out)
    bad = [v["name"] for v, g in zip(rows, out.raw) if bool(g) != v["expect"]]
    assert not bad, bad[:10]


def _recover_cases(orc, rnd, n):
    """rows (hash, sig, recid, expected key or None) covering valid recoveries under both parities, r/s range failures,
    recid 2/3 (r + n), x off the curve, high-S signatures (fine on this path) and the infinity result"""
    rows = []
    for i in range(n):
        sk = rnd.randrange(1, N)
        h = bytes(rnd.randrange(256) for _ in range(32))
        sig = orc.ecdsa_sign(h, sk.to_bytes(32, "big"), bytes(rnd.randrange(256) for _ in range(32)))
        recid = rnd.randrange(2)
        c = i % 10
        if c == 5:                                   # high S: same R, s -> n - s flips which recid gives the signer
            s = N - int.from_bytes(sig[32:], "big")
            sig = sig[:32] + s.to_bytes(32, "big")
        elif c == 6:
            recid = rnd.randrange(2, 4)              # r + n >= p almost always -> failure
        elif c == 7:
            sig = rnd.choice((bytes(32) + sig[32:], sig[:32] + bytes(32), N.to_bytes(32, "big") + sig[32:], sig[:32] + (N + 1).to_bytes(32, "big")))
        elif c == 8:
            recid = rnd.choice((4, 7, 255))
        elif c == 9:
            sig = rnd.randrange(1, N).to_bytes(32, "big") + sig[32:]      # random r: half of them have no point
        rows.append((h, sig, recid))
    # tiny r so that recid 2/3 (x = r + n < p) can succeed, and the infinity outcome: Q = (s R - z G)/r = inf when s R = z G
    for r in range(1, 40):
        for recid in range(4):
            rows.append((bytes(rnd.randrange(256) for _ in range(32)), r.to_bytes(32, "big") + rnd.randrange(1, N).to_bytes(32, "big"), recid))
    k = rnd.randrange(1, N)
    R = pyref.pmul(k, pyref.G)
    r, s = R[0] % N, rnd.randrange(1, N)
    z = s * k % N                                     # s*R = (s k) G = z G
    rows.append((z.to_bytes(32, "big"), r.to_bytes(32, "big") + s.to_bytes(32, "big"), R[1] & 1))
    return rows


def test_recover_pipeline_vs_pyref(dm, orc):
    rnd = random.Random(4711)
    rows = _recover_cases(orc, rnd, 120)
    n = len(rows)
    for threads in (1, 7):
        pub = ctypes.create_string_buffer(33 * n)
        ok = ctypes.create_string_buffer(n)
        dm.dm_recover_batch(ctypes.c_size_t(n), b"".join(r[0] for r in rows), b"".join(r[1] for r in rows), bytes(r[2] & 0xFF for r in rows),
                            pub, ok, ctypes.c_size_t(threads))
        good = 0
        for i, (h, sig, recid) in enumerate(rows):
            exp = pyref.ecdsa_recover(h, sig, recid)
            got = pub.raw[33 * i:33 * i + 33]
            if exp is None:
                assert ok.raw[i] == 0 and got == bytes(33), (i, recid, sig.hex())
            else:
                assert ok.raw[i] == 1 and got == pyref.ser33(exp), (i, recid, sig.hex())
                good += 1
                if int.from_bytes(sig[32:], "big") <= N // 2:
                    assert orc.ecdsa_verify(h, sig, got)      # a recovered key verifies its (low-S) signature
        assert good >= 70 and good < n


def test_recover_goldens_host(dm, kat):
    rows = kat["recover"]
    n = len(rows)
    pub = ctypes.create_string_buffer(33 * n)
    ok = ctypes.create_string_buffer(n)
    dm.dm_recover_batch(ctypes.c_size_t(n), b"".join(H(v["hash"]) for v in rows), b"".join(H(v["sig"]) for v in rows),
                        bytes(v["recid"] & 0xFF for v in rows), pub, ok, ctypes.c_size_t(3))
    for i, v in enumerate(rows):
        got = pub.raw[33 * i:33 * i + 33].hex() if ok.raw[i] else None
        assert got == v["expect"], v["name"]


def _host_grind(dm, pre, outputs, input_sat, weight, lo, hi, sig, stype, wit, pub):
    rate, fee = ctypes.c_uint32(0), ctypes.c_uint64(0)
    dm.dm_grind.restype = ctypes.c_int
    rc = dm.dm_grind(bytes(pre), ctypes.c_size_t(len(pre)), bytes(outputs), ctypes.c_size_t(len(outputs)), ctypes.c_uint64(input_sat),
                     ctypes.c_uint64(weight), ctypes.c_uint32(lo), ctypes.c_uint32(hi), bytes(sig), int(stype), int(wit), bytes(pub),
                     ctypes.byref(rate), ctypes.byref(fee))
    return (rate.value, fee.value) if rc == 1 else None


def test_fee_grind_host_build_vs_reference_kat_and_restated_loop(dm, kat, orc):
    """the device code of the fee grind (grind_prepare / grind_candidate, compiled for the host) against the reference's own
    known answer (onchaind/test/run-grind_feerate.c: fee 165 750 at feerate 250 000) and against the restated loop on seeded cases"""
    ko = next(v for v in kat["der"] if v["name"] == "KAT-O")
    sig = H(ko["expect_sig"])
    key = H("038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de054432eb7f")
    pre = H(next(v for v in kat["bip143"] if v["name"] == "KAT-O/fee=0")["preimage"])
    spk = H("002082e03c5a9cb79c82cd5a0572dc175290bc044609aabe9cc852d6192743604179")
    outputs = (700000).to_bytes(8, "little") + bytes([len(spk)]) + spk
    assert _host_grind(dm, pre, outputs, 700000, 663, 249800, 250000, sig, 1, True, key) == (250000, 165750)
    assert _host_grind(dm, pre, outputs, 700000, 663, 250001, 250001, sig, 1, True, key) == (250001, 165750)
    assert _host_grind(dm, pre, outputs, 700000, 663, 249800, 249998, sig, 1, True, key) is None
    assert _host_grind(dm, pre, outputs, 700000, 663, 249800, 250000, sig, 0x83, False, key) is None
    rnd = random.Random(99)
    ver = lambda h, s, k: orc.ecdsa_verify(h, s, k)
    for case in range(8):
        sk = rnd.randrange(1, N).to_bytes(32, "big")
        pub = pyref.ser33(pyref.pubkey_create(int.from_bytes(sk, "big")))
        script = bytes(rnd.randrange(256) for _ in range(rnd.choice((1, 25, 133, 200))))
        spk = bytes([0, 32]) + bytes(rnd.randrange(256) for _ in range(32))
        input_sat = rnd.randrange(50_000, 5_000_000)
        weight = rnd.choice((663, 703, 1000, 1))
        lo = rnd.randrange(0, 30_000)
        hi = lo + rnd.randrange(0, 300)
        hidden = rnd.randrange(max(0, lo - 20), hi + 20)
        fee = hidden * weight // 1000
        amount = max(0, input_sat - fee)
        stype = 0x83 if case % 3 == 0 else 1
        sighash, pre = pyref.bip143_sighash(2, [(bytes(rnd.randrange(256) for _ in range(32)), 1, 0)], [(amount, spk)], 500000 + case, 0, script, input_sat, stype)
        sig = orc.ecdsa_sign(sighash, sk, bytes(rnd.randrange(256) for _ in range(32)))
        outputs = amount.to_bytes(8, "little") + bytes([len(spk)]) + spk
        exp = pyref.grind_htlc_tx_fee(pre, outputs, input_sat, weight, lo, hi, sig, stype, True, pub, verify=ver)
        assert _host_grind(dm, pre, outputs, input_sat, weight, lo, hi, sig, stype, True, pub) == exp, case


def test_gossip_framing_host_build_vs_goldens_and_oracle(dm, kat, orc):
    """the device's framing rules (what fromwire_* rejects) on the host: every gossip golden, plus random truncations and
    byte flips in the framing fields, against the C oracle's verdict == -1"""
    dm.dm_gossip_frame.restype = ctypes.c_int

    def frame_bad(m):
        t, so, ko = ctypes.c_uint32(), ctypes.c_size_t(), ctypes.c_size_t()
        return bool(dm.dm_gossip_frame(m, len(m), ctypes.byref(t), ctypes.byref(so), ctypes.byref(ko)))

    def oracle_verdict(v, m):
        if v["kind"] == "channel_announcement":
            return orc.sigcheck_channel_announcement(m)
        if v["kind"] == "channel_update":
            return orc.sigcheck_channel_update(m, H(v["node_id"]))
        return orc.sigcheck_node_announcement(m)

    nbad = 0
    for v in kat["gossip"]:
        m = H(v["msg"])
        assert oracle_verdict(v, m) == v["expect"], v["name"]
        if v["expect"] != -1:
            assert not frame_bad(m), v["name"]
        elif "framing" in v["source"]:
            assert frame_bad(m), v["name"]
            nbad += 1
    assert nbad >= 20
    rnd = random.Random(77)
    base = [v for v in kat["gossip"] if v["expect"] == 0]
    for it in range(3000):
        v = rnd.choice(base)
        m = bytearray(H(v["msg"]))
        if rnd.random() < 0.5:
            m = m[:rnd.randrange(len(m) + 1)]
        else:
            # flip a byte outside the signatures: lengths, tlv bytes, addrlen ... (keys may stop parsing: not framing)
            lo = 258 if v["kind"] == "channel_announcement" else 66
            m[rnd.randrange(lo, len(m))] ^= 1 << rnd.randrange(8)
        m = bytes(m)
        # framing-bad must imply oracle -1; oracle -1 beyond framing (key / sig range) is decided by other kernels
        if frame_bad(m):
            assert oracle_verdict(v, m) == -1, (v["name"], m.hex())
        elif v["kind"] != "channel_announcement":
            assert oracle_verdict(v, m) != -1, (v["name"], m.hex())


def test_fuzz_lane_respects_magnitude_bounds(dm):
    """the GPU fuzzer's lane function (csrc/fuzz.h) under the host build's magnitude assertions: every operand it feeds the
    primitives is inside the bounds the group law guarantees, so a device/host checksum difference can only be a device fault"""
    dm.dm_fuzz_lane.restype = ctypes.c_uint64
    dm.dm_fuzz_lane.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
    seen = {dm.dm_fuzz_lane(1, lane, 8) for lane in range(300)}
    assert len(seen) == 300
    assert dm.dm_fuzz_lane(1, 5, 8) == dm.dm_fuzz_lane(1, 5, 8) != dm.dm_fuzz_lane(2, 5, 8)


def _rand_tx(rnd):
    n_in, n_out = rnd.choice([1, 1, 1, 2, 3, 7]), rnd.choice([0, 1, 1, 2, 3, 5])
    inputs = [(bytes(rnd.randrange(256) for _ in range(32)), rnd.randrange(1 << 32), rnd.randrange(1 << 32)) for _ in range(n_in)]
    outputs = [(rnd.randrange(1 << 51), bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 22, 34, 34, 100, 253, 300])))) for _ in range(n_out)]
    script = bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 25, 71, 133, 133, 252, 253, 254, 400])))
    return rnd.choice([1, 2, 2, 0xFFFFFFFF]), inputs, outputs, rnd.randrange(1 << 32), script, rnd.randrange(1 << 51)


def _tx_flat(inputs, outputs):
    ib = b"".join(t + v.to_bytes(4, "little") + s.to_bytes(4, "little") for t, v, s in inputs)
    ob = b"".join(a.to_bytes(8, "little") + pyref._varint(len(spk)) + spk for a, spk in outputs)
    return ib, ob


def test_host_side_front_ends_of_the_latency_paths(dm, kat, orc):
    """what lamd_sigcheck_gossip_batch / lamd_check_tx_sig_batch do ON THE HOST for a call of a few rows -- the kernels' own inline functions
    (gossip_expand_one, gossip_reduce_one, txsig_hash_one) compiled for the host: every gossip golden (the reference's KAT-G verdicts, its
    gossip_store messages, the framing classes) expanded into rows, the rows decided by the C oracle, reduced -- the golden verdict must come
    out; the sighash-type gate and the double SHA-256 of check_tx_sig on the reference's preimages"""
    dm.dm_gossip_expand.restype = ctypes.c_int
    dm.dm_gossip_expand.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    dm.dm_gossip_reduce.restype = ctypes.c_int
    dm.dm_gossip_reduce.argtypes = [ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    seen = set()
    for v in kat["gossip"]:
        m = H(v["msg"])
        typ = int.from_bytes(m[:2], "big") if len(m) >= 2 else 0
        nrows = 4 if typ == 256 else 1
        nid = H(v["node_id"]) if "node_id" in v else bytes(33)
        hs, sg, pk = ctypes.create_string_buffer(32 * nrows), ctypes.create_string_buffer(64 * nrows), ctypes.create_string_buffer(33 * nrows)
        bad = dm.dm_gossip_expand(m, len(m), nid, nrows, hs, sg, pk)
        ok = bytes(1 if orc.ecdsa_verify(hs.raw[32 * k:32 * k + 32], sg.raw[64 * k:64 * k + 64], pk.raw[33 * k:33 * k + 33]) else 0 for k in range(nrows))
        keyok = bytes(1 if pyref.pubkey_parse(pk.raw[33 * k:33 * k + 33]) is not None else 0 for k in range(nrows))
        got = dm.dm_gossip_reduce(nrows, ok, keyok, bad)
        assert got == v["expect"], v["name"]
        seen.add(got)
    assert {-1, 0, 1, 2}.issubset(seen)
    dm.dm_txsig_hash.restype = ctypes.c_int
    dm.dm_txsig_hash.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    o = ctypes.create_string_buffer(32)
    for v in kat["bip143"]:
        pre = H(v["preimage"])
        assert dm.dm_txsig_hash(pre, len(pre), 1, 0, o) == 1 and o.raw == H(v["expect"]), v["name"]
        assert dm.dm_txsig_hash(pre, len(pre), 0x83, 1, o) == 1 and o.raw == H(v["expect"])          # the gate only looks at the type; the hash is of the bytes given
        for t, w in ((0x83, 0), (2, 1), (3, 1), (0x81, 1), (0, 1)):
            assert dm.dm_txsig_hash(pre, len(pre), t, w, o) == 0 and o.raw == bytes(32)                # outside the gate (bitcoin/signature.c:206-211)


def test_bip143_sighash_host_build_vs_pyref_and_reference_kat(dm, kat):
    """the device's BIP143 code (streaming SHA-256 over the template's pieces) on the host: the reference's own transaction
    (onchaind/test/run-grind_feerate.c: 290-byte preimage, sighash 45fa7ea1... at fee 165 750) and 100 000 random templates
    -- every sighash type the format defines, inputs/outputs/script sizes across the CompactSize boundaries -- against the
    spec-level model"""
    dm.dm_bip143.restype = ctypes.c_int
    dm.dm_bip143.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
    o = ctypes.create_string_buffer(32)
    tx = H("0200000001e1ebca08cf1c301ac563580a1126d5c8fcb0e5e2043230b852c726553caf1e1d0000000000000000000160ae0a0000000000"
           "22002082e03c5a9cb79c82cd5a0572dc175290bc044609aabe9cc852d61927436041796d000000")
    ws = H("76a914a8c40c334351dbe8e5908544f1c98fbcfb8719fc8763ac6721038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de05"
           "4432eb7f7c820120876475527c2103cf8e2f193a6aed60db80af75f3c8d59c2de735b299b7c7083527be9bd23b77a852ae67a914b8bcd51e"
           "fa35be1e50ae2d5f72f4500acb005c9c88ac6868")
    spk = tx[56:90]
    for v in kat["bip143"]:
        fee = int(v["name"].split("=")[1])
        ib, ob = _tx_flat([(tx[5:37], 0, 0)], [(700000 - fee, spk)])
        assert dm.dm_bip143(2, 109, ib, 1, ob, len(ob), 1, 0, ws, len(ws), 700000, 1, o) == 1
        assert o.raw == H(v["expect"]), v["name"]
    assert any(v["expect"].startswith("45fa7ea15e62277f") for v in kat["bip143"])
    for v in kat["txsig"]:   # the reference-held BOLT #3 HTLC transactions (channeld/test/run-full_channel.c): template -> the sighash their signatures verify under
        ib, ob = _tx_flat([(H(t), vout, seq) for t, vout, seq in v["inputs"]], [(a, H(spk)) for a, spk in v["outputs"]])
        wsx = H(v["script"])
        assert dm.dm_bip143(v["version"], v["locktime"], ib, len(v["inputs"]), ob, len(ob), len(v["outputs"]), v["input_num"], wsx, len(wsx), v["amount"],
                            v["sighash_type"], o) == 1
        assert o.raw == H(v["sighash"]), v["name"]
    rnd = random.Random(143)
    for it in range(100_000):
        version, inputs, outputs, lock, script, amount = _rand_tx(rnd)
        ib, ob = _tx_flat(inputs, outputs)
        sht = rnd.choice([1, 1, 0x83, 0x83, 2, 3, 0x81, 0x82])
        idx = rnd.randrange(len(inputs))
        ok = dm.dm_bip143(version, lock, ib, len(inputs), ob, len(ob), len(outputs), idx, script, len(script), amount, sht, o)
        assert ok == 1 and o.raw == pyref.bip143_sighash(version, inputs, outputs, lock, idx, script, amount, sht)[0], (it, sht)
    # inconsistent templates are refused, not hashed
    version, inputs, outputs, lock, script, amount = 2, [(bytes(32), 0, 0)], [(5, b"\x51")], 0, b"\x51", 9
    ib, ob = _tx_flat(inputs, outputs)
    assert dm.dm_bip143(version, lock, ib, 1, ob, len(ob), 1, 1, script, 1, amount, 1, o) == 0          # input index out of range
    assert dm.dm_bip143(version, lock, ib, 1, ob, len(ob) - 1, 1, 0, script, 1, amount, 1, o) == 0      # truncated outputs
    assert dm.dm_bip143(version, lock, ib, 1, ob, len(ob), 2, 0, script, 1, amount, 1, o) == 0          # fewer outputs than claimed


def test_host_sha256_with_and_without_the_sha_extensions():
    """sha256.h on the host: the x86 SHA-extension compression (what txsig_pack hashes a commitment transaction's 20 KB of outputs with) and the
    portable rounds (-DLAMD_NO_SHA_NI), and verify_core.h's whole-block path of the streaming form: both builds against hashlib on every length
    0 .. 300 and on long inputs fed in pieces of 1 .. 5 000 bytes, and against the BIP143 model on transactions with up to 600 outputs"""
    import hashlib
    src = os.path.join(HERE, "c", "sha_host_paths.cpp")
    libs = []
    for tag, flags in (("ni", []), ("portable", ["-DLAMD_NO_SHA_NI"])):
        so = os.path.join(HERE, "libsha_host_%s.so" % tag)
        deps = [src] + [os.path.join(ROOT, "lightning_amd", "csrc", f) for f in ("sha256.h", "verify_core.h", "lamd_common.h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas"] + flags + ["-o", so, src])
        L = ctypes.CDLL(so)
        L.h_sha256d.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.h_sha256d_stream.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p]
        L.h_bip143.restype = ctypes.c_int
        L.h_bip143.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32,
                               ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_char_p]
        libs.append(L)
    assert libs[1].h_sha_ni() == 0
    has_ni = "sha_ni" in open("/proc/cpuinfo").read()
    assert libs[0].h_sha_ni() == (1 if has_ni else 0)          # the fast path is what runs where the CPU has it
    rnd = random.Random(256)
    o = ctypes.create_string_buffer(32)
    d2 = lambda b: hashlib.sha256(hashlib.sha256(b).digest()).digest()
    for L in libs:
        for n in list(range(0, 301)) + [447, 448, 449, 511, 512, 513, 4095, 4096, 20855, 65537]:
            b = bytes(rnd.randrange(256) for _ in range(n))
            L.h_sha256d(b, n, o)
            assert o.raw == d2(b), n
            for piece in (1, 7, 63, 64, 65, 127, 128, 129, 200, 1000, 5000):
                if n and (n <= 600 or piece >= 63):
                    L.h_sha256d_stream(b, n, piece, o)
                    assert o.raw == d2(b), (n, piece)
        for n_out in (0, 1, 2, 3, 4, 5, 30, 485, 600):
            inputs = [(bytes(rnd.randrange(256) for _ in range(32)), rnd.randrange(1 << 32), rnd.randrange(1 << 32)) for _ in range(rnd.choice([1, 2, 9]))]
            outputs = [(rnd.randrange(1 << 44), bytes(rnd.randrange(256) for _ in range(rnd.choice([22, 34])))) for _ in range(n_out)]
            script = bytes(rnd.randrange(256) for _ in range(rnd.choice([71, 133, 500])))
            ib, ob = _tx_flat(inputs, outputs)
            for sht in (1, 0x83):
                assert L.h_bip143(2, 7, ib, len(inputs), ob, len(ob), n_out, 0, script, len(script), 12345, sht, o) == 1
                assert o.raw == pyref.bip143_sighash(2, inputs, outputs, 7, 0, script, 12345, sht)[0], (n_out, sht)


def _tlv(t, v):
    return pyref.bigsize(t) + pyref.bigsize(len(v)) + v


# BOLT #12 signature-test vectors (bolt12/signature-test.json of the specification, the file common/test/run-bolt12_merkle.c prints):
# recalled offline; the three roots below are reproduced digit for digit by the restatement, which no wrong algorithm would do
BOLT12_N1 = [((1, (1000).to_bytes(2, "big")),),
             ((1, (1000).to_bytes(2, "big")), (2, ((1 << 40) | (2 << 16) | 3).to_bytes(8, "big"))),
             ((1, (1000).to_bytes(2, "big")), (2, ((1 << 40) | (2 << 16) | 3).to_bytes(8, "big")),
              (3, H("0266e4598d1d3c415f572a8488830b60f7e744ed9235eb0b1ba93283b315c03518") + (1).to_bytes(8, "big") + (2).to_bytes(8, "big")))]
BOLT12_N1_ROOTS = ["b013756c8fee86503a0b4abdab4cddeb1af5d344ca6fc2fa8b6c08938caa6f93", "c3774abbf4815aa54ccaa026bff6581f01f3be5fe814c620a252534f434bc0d1",
                   "ab2e79b1283b0b31e0b035258de23782df6b89a38cfa7237bde69aed1a658c5d"]


def test_bolt12_merkle_host_build_vs_pyref_and_spec_vectors(dm):
    """the device's BOLT #12 front end on the host: the specification's n1 vectors (the trees common/test/run-bolt12_merkle.c builds
    by hand, :150-330), the explicit construction that test asserts against merkle_tlv(), and 6 000 random TLV streams -- 1..70
    fields, values across the BigSize boundaries, signature fields (240..1000) in between, big types -- against the restatement;
    malformed streams are refused"""
    dm.dm_bolt12.restype = ctypes.c_int
    dm.dm_bolt12.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    root, sh = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    assert _tlv(*BOLT12_N1[0][0]) == H("010203e8")
    for fields, want in zip(BOLT12_N1, BOLT12_N1_ROOTS):
        st = b"".join(_tlv(t, v) for t, v in fields)
        assert pyref.bolt12_merkle(list(fields)).hex() == want
        assert dm.dm_bolt12(st, len(st), b"invoice_request", b"signature", root, sh) == 1 and root.raw.hex() == want
        assert sh.raw == pyref.bolt12_sighash(b"invoice_request", b"signature", root.raw)
    # the three-leaf tree spelled out as the reference's test does (:259-263): H(LnBranch, ordered(H(LnBranch, ordered(l0, l1)), l2))
    f = BOLT12_N1[2]
    first = _tlv(*f[0])
    Hh, order = pyref.bolt12_H, lambda a, b: min(a, b) + max(a, b)
    leaf = [Hh(b"LnBranch", order(Hh(b"LnLeaf", _tlv(t, v)), Hh(b"LnNonce" + first, pyref.bigsize(t)))) for t, v in f]
    assert Hh(b"LnBranch", order(Hh(b"LnBranch", order(leaf[0], leaf[1])), leaf[2])).hex() == BOLT12_N1_ROOTS[2]
    rnd = random.Random(1212)
    for it in range(6_000):
        nf = rnd.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 70])
        types = sorted(rnd.sample(list(range(0, 240)) + list(range(240, 1001, 7)) + list(range(1001, 1100)) + [0x10000, 0x10001, 1 << 32, (1 << 40) + 5, (1 << 64) - 1], nf))
        fields = [(t, bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 8, 33, 64, 100, 252, 253, 300])))) for t in types]
        st = b"".join(_tlv(t, v) for t, v in fields)
        want = pyref.bolt12_merkle(fields)
        assert pyref.tlv_stream_parse(st) == fields
        got = dm.dm_bolt12(st, len(st), b"invoice", b"signature", root, sh)
        if want is None:
            assert got == 0
        else:
            assert got == 1 and root.raw == want and sh.raw == pyref.bolt12_sighash(b"invoice", b"signature", want), it
    ok = _tlv(1, b"ab") + _tlv(5, b"c")
    for bad in (ok[:-1], _tlv(5, b"c") + _tlv(1, b"ab"), _tlv(1, b"a") + _tlv(1, b"b"), b"\xfd\x00\x01\x00", b"\x01\xfd\x00\x01a", b"\x01", _tlv(240, bytes(64)), b""):
        assert dm.dm_bolt12(bad, len(bad), b"invoice", b"signature", root, sh) == 0, bad.hex()
        assert pyref.tlv_stream_parse(bad) is None or pyref.bolt12_merkle(pyref.tlv_stream_parse(bad)) is None
    assert dm.dm_bolt12(ok, len(ok), b"invoice", b"signature", root, sh) == 1


def test_bolt12_reference_held_strings_host_build(dm, kat):
    """the device's BOLT #12 front end (host build) + its BIP-340 pipeline on the reference-held lni1 / lnr1 strings (kat.json "bolt12")"""
    dm.dm_bolt12.restype = ctypes.c_int
    dm.dm_bolt12.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    root, sh = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    msgs, pks, sigs, exp = [], [], [], []
    for v in kat["bolt12"]:
        st = H(v["stream"])
        ok = dm.dm_bolt12(st, len(st), v["messagename"].encode(), b"signature", root, sh)
        fields = pyref.tlv_stream_parse(st)
        m = pyref.bolt12_merkle(fields) if fields is not None else None
        assert (ok == 1) == (m is not None), v["name"]
        if not ok:
            assert not v["expect"]
            continue
        assert root.raw == m and sh.raw == pyref.bolt12_sighash(v["messagename"].encode(), b"signature", m), v["name"]
        msgs.append(sh.raw); pks.append(H(v["key"])[1:]); sigs.append(H(v["sig"])); exp.append(v["expect"])
    n = len(msgs)
    out = ctypes.create_string_buffer(n)
    dm.dm_schnorr_verify_batch(ctypes.c_size_t(n), b"".join(msgs), b"".join(pks), b"".join(sigs), out)
    assert [bool(x) for x in out.raw] == exp and sum(exp) >= 9


def test_sc29_scalar_arithmetic_vs_integers(dm):
    """the 9x29-limb arithmetic mod n that the ECDSA preparation runs in (prefix products, shared inversion, u1, u2): products of
    arbitrary 256-bit values (not only residues), chains of lazy values, inversion -- against Python integers; edge operands"""
    Nn = pyref.N
    out = ctypes.create_string_buffer(32)
    rnd = random.Random(2929)
    edge = [0, 1, 2, Nn - 1, Nn, Nn + 1, (1 << 256) - 1, (1 << 256) - Nn, 1 << 255, (1 << 128) - 1, (1 << 261) % Nn, Nn - 2, (Nn - 1) // 2]
    vals = edge + [rnd.getrandbits(256) for _ in range(300)]
    for a in vals:
        for b in rnd.sample(vals, 12) + edge[:6]:
            dm.dm_sc29_mul(a.to_bytes(32, "big"), b.to_bytes(32, "big"), out)
            assert int.from_bytes(out.raw, "big") == a * b % Nn, (hex(a), hex(b))
    for _ in range(200):
        a, b, steps = rnd.getrandbits(256), rnd.getrandbits(256), rnd.randrange(1, 40)
        dm.dm_sc29_chain(a.to_bytes(32, "big"), b.to_bytes(32, "big"), steps, out)
        p, q = a, b
        for _ in range(steps):
            p, q = q, p * q % Nn
        assert int.from_bytes(out.raw, "big") == q % Nn
    for a in [1, 2, Nn - 1, Nn + 5, (1 << 256) - 1] + [rnd.getrandbits(256) for _ in range(40)]:
        dm.dm_sc29_inv(a.to_bytes(32, "big"), out)
        assert int.from_bytes(out.raw, "big") == pow(a % Nn, -1, Nn), hex(a)


def test_sc_inv_var_division_steps_vs_integers(dm):
    """the division-step (safegcd) inversion mod n that the scalar preparation uses: edge values, small and huge operands, 20 000 random
    residues -- against pow(a, -1, n); 0 maps to 0"""
    Nn = pyref.N
    out = ctypes.create_string_buffer(32)
    rnd = random.Random(590)
    vals = [1, 2, 3, Nn - 1, Nn - 2, (Nn - 1) // 2, (Nn + 1) // 2, 1 << 255, (1 << 255) - 1, (1 << 128) - 1, 1 << 128, 0x14551231950B75FC4402DA1732FC9BEBF,
            (1 << 256) - Nn, pow(2, -1, Nn), pow(3, -1, Nn)] + [1 << k for k in range(0, 256, 7)] + [Nn - (1 << k) for k in range(0, 255, 11)]
    vals += [rnd.randrange(1, Nn) for _ in range(20_000)] + [rnd.randrange(1, 1 << rnd.randrange(1, 256)) for _ in range(2000)]
    for a in vals:
        dm.dm_sc_inv_var(a.to_bytes(32, "big"), out)
        assert int.from_bytes(out.raw, "big") == pow(a, -1, Nn), hex(a)
    dm.dm_sc_inv_var(bytes(32), out)
    assert out.raw == bytes(32)


def test_fe_inv_var_division_steps_vs_integers(dm):
    """the division-step inversion mod p (BIP-340 parity stage, key recovery, the G table build): edge values and 20 000 random
    elements, non-canonical inputs (>= p) included -- against pow(a, -1, p)"""
    Pp = pyref.P
    out = ctypes.create_string_buffer(32)
    rnd = random.Random(977)
    vals = [1, 2, 3, Pp - 1, Pp - 2, (Pp - 1) // 2, 1 << 255, (1 << 255) - 1, 977, (1 << 32) + 977, Pp + 1, Pp + 5, (1 << 256) - 1]
    vals += [1 << k for k in range(0, 256, 5)] + [Pp - (1 << k) for k in range(0, 255, 13)] + [rnd.getrandbits(256) for _ in range(20_000)]
    for a in vals:
        dm.dm_fe_inv_var(a.to_bytes(32, "big"), out)
        assert int.from_bytes(out.raw, "big") == pow(a % Pp, -1, Pp), hex(a)
    dm.dm_fe_inv_var(bytes(32), out)
    assert out.raw == bytes(32)
    dm.dm_fe_inv_var(Pp.to_bytes(32, "big"), out)
    assert out.raw == bytes(32)
