"""N3 as a function: ONE commitment_signed = ONE call (lamd_check_commitment_signed, the mirror's check_commit_sigs) against the reference's loop
(channeld/channeld.c:2171-2232) restated over the oracle: check_tx_sig on the commitment transaction under the funding key, then on HTLC transaction i
under the htlc key, first failure wins.  Needs an MI355X: -m gpu."""
import ctypes
import os
import random

import numpy as np
import pytest

import pyref

pytestmark = pytest.mark.gpu
H = bytes.fromhex


@pytest.fixture(scope="module")
def eng():
    from lightning_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def _tx(v):
    return dict(version=v["version"], locktime=v["locktime"], inputs=[(H(t), vout, seq) for t, vout, seq in v["inputs"]],
                outputs=[(a, H(spk)) for a, spk in v["outputs"]], input_num=v["input_num"], amount=v["amount"], script=H(v["script"]))


def _reference_loop(orc, commit_tx, fund33, commit_sig, commit_type, htlc_txs, htlc33, htlc_sigs, htlc_types):
    """channeld.c:2171 then :2209-2232, one check_tx_sig() at a time: the sighash-type gate (bitcoin/signature.c:206-211), BIP143 (pyref), ECDSA (C oracle).
    Returns (first_bad, every row's verdict)."""
    rows = []
    for t, key, sig, typ in [(commit_tx, fund33, commit_sig, commit_type)] + [(t, htlc33, s, ty) for t, s, ty in zip(htlc_txs, htlc_sigs, htlc_types)]:
        ok = typ == 1 or typ == 0x83   # a witness script is always there on this path
        if ok:
            sighash, _ = pyref.bip143_sighash(t["version"], t["inputs"], t["outputs"], t["locktime"], t["input_num"], t["script"], t["amount"], typ)
            ok = bool(orc.ecdsa_verify(sighash, sig, key))
        rows.append(ok)
    return next((i for i, ok in enumerate(rows) if not ok), -1), rows


def _bolt3(kat):
    """the BOLT #3 appendix C commitment the reference tree holds signed: the commitment transaction of wallet/test/run-wallet.c:1520 under the remote
    funding key, the five HTLC transactions of channeld/test/run-full_channel.c:635-673 under the remote htlc key"""
    by = {v["name"]: v for v in kat["txsig"]}
    c = by["KAT-BOLT3/commit/line1520/key2"]
    hs = [by["KAT-BOLT3/line%d/remote" % ln] for ln in (637, 646, 655, 664, 673)]
    assert len({v["pub"] for v in hs}) == 1
    return _tx(c), H(c["pub"]), H(c["sig"]), [_tx(v) for v in hs], H(hs[0]["pub"]), [H(v["sig"]) for v in hs], c, hs


def _flip(sig, byte=40):
    return sig[:byte] + bytes([sig[byte] ^ 1]) + sig[byte + 1:]


def test_bolt3_commitment_as_one_call(eng, kat, orc):
    ctx, fund, csig, htxs, hkey, hsigs, c, hs = _bolt3(kat)
    # the templates hash to the sighashes the goldens pin (pyref = the reference-held signatures' message)
    assert pyref.bip143_sighash(ctx["version"], ctx["inputs"], ctx["outputs"], ctx["locktime"], 0, ctx["script"], ctx["amount"], 1)[0] == H(c["sighash"])
    types = [1] * 5

    def both(cs, ct, sigs, tys, txs=htxs, key=hkey, fkey=fund):
        got = eng.check_commitment_signed(ctx, fkey, cs, ct, txs, key, sigs, tys)
        exp = _reference_loop(orc, ctx, fkey, cs, ct, txs, key, sigs, tys)
        assert got[0] == exp[0] and list(got[1]) == exp[1], (got, exp)
        return got[0]
    for _ in range(3):   # first sight (ladder), learning call (tables built), cache hits: the same verdicts on every path
        assert both(csig, 1, hsigs, types) == -1
    for k in range(5):   # corrupt HTLC k -> 1 + k
        assert both(csig, 1, hsigs[:k] + [_flip(hsigs[k])] + hsigs[k + 1:], types) == 1 + k
    assert both(_flip(csig), 1, hsigs[:2] + [_flip(hsigs[2])] + hsigs[3:], types) == 0          # commit sig AND HTLC 2 -> 0 (the reference stops at :2171)
    assert both(csig, 1, [_flip(hsigs[0])] + hsigs[1:3] + [_flip(hsigs[3])] + hsigs[4:], types) == 1
    # the sighash-type gate (bitcoin/signature.c:206-211) is part of the verdict: SIGHASH_NONE on HTLC 1, an unknown type on the commitment
    assert both(csig, 1, hsigs, [1, 2, 1, 1, 1]) == 2
    assert both(csig, 0x81, hsigs, types) == 0
    # SINGLE|ANYONECANPAY hashes differently: a signature made for SIGHASH_ALL does not verify under it (option_anchors peers send 0x83)
    assert both(csig, 1, hsigs, [1, 1, 1, 0x83, 1]) == 4
    # wrong keys
    assert both(csig, 1, hsigs, types, fkey=hkey) == 0
    assert both(csig, 1, hsigs, types, key=fund) == 1
    # an HTLC transaction whose spent amount is off by one (the goldens' "amount+1" twins)
    t2 = dict(htxs[2], amount=htxs[2]["amount"] + 1)
    assert both(csig, 1, hsigs, types, txs=htxs[:2] + [t2] + htxs[3:]) == 3
    # no HTLCs at all: one signature
    got = eng.check_commitment_signed(ctx, fund, csig, 1, [], hkey, [], [])
    assert got[0] == -1 and list(got[1]) == [True]
    got = eng.check_commitment_signed(ctx, fund, _flip(csig), 1, [], hkey, [], [])
    assert got[0] == 0


def _synthetic_commitment(orc, rnd, n_htlc, anchors):
    """a commitment of n_htlc HTLC transactions signed with the C oracle's signer over pyref's BIP143 hashes"""
    fsk, hsk = (rnd.randrange(1, pyref.N).to_bytes(32, "big") for _ in range(2))
    fund, hkey = (pyref.ser33(pyref.pubkey_create(int.from_bytes(sk, "big"))) for sk in (fsk, hsk))
    rb = lambda n: bytes(rnd.randrange(256) for _ in range(n))
    fw = b"\x52\x21" + fund + b"\x21" + hkey + b"\x52\xae"
    outs = [(rnd.randrange(330, 10**7), b"\x00\x20" + rb(32)) for _ in range(n_htlc + 2)]
    ctx = dict(version=2, locktime=0x20000000 | rnd.randrange(1 << 24), inputs=[(rb(32), rnd.randrange(4), 0x80000000 | rnd.randrange(1 << 24))], outputs=outs,
               input_num=0, amount=sum(a for a, _ in outs) + rnd.randrange(1000, 50000), script=fw)
    sign = lambda t, sk, ty: orc.ecdsa_sign(pyref.bip143_sighash(t["version"], t["inputs"], t["outputs"], t["locktime"], 0, t["script"], t["amount"], ty)[0], sk, rb(32))
    csig = sign(ctx, fsk, 1)
    ctxid = rb(32)
    ty = 0x83 if anchors else 1
    htxs, hsigs = [], []
    for i in range(n_htlc):
        ws = rb(rnd.choice((133, 136, 139, 140)))
        t = dict(version=2, locktime=rnd.choice((0, 500000 + i)), inputs=[(ctxid, i, 1 if anchors else 0)], outputs=[(outs[i][0] - rnd.randrange(0, 300), b"\x00\x20" + rb(32))],
                 input_num=0, amount=outs[i][0], script=ws)
        htxs.append(t)
        hsigs.append(sign(t, hsk, ty))
    return ctx, fund, csig, htxs, hkey, hsigs, [ty] * n_htlc


@pytest.mark.parametrize("n_htlc,anchors", [(483, False), (483, True), (63, False), (1, True), (5000, False)])
def test_synthetic_commitments_equal_the_reference_loop(eng, orc, n_htlc, anchors):
    """the product's largest natural batch (1 + 483 signatures, 483 under one key) -- and one past the 4096-row limit of the two-launch path, which takes the
    batch machinery -- all good, one bad HTLC, several bad rows: first_bad and EVERY row's verdict equal the reference's loop over the oracle"""
    rnd = random.Random(0xC0FFEE + n_htlc + anchors)
    ctx, fund, csig, htxs, hkey, hsigs, tys = _synthetic_commitment(orc, rnd, n_htlc, anchors)
    cases = [(csig, hsigs)]
    k = rnd.randrange(n_htlc)
    cases.append((csig, hsigs[:k] + [_flip(hsigs[k], 7)] + hsigs[k + 1:]))
    bad = sorted(rnd.sample(range(n_htlc), min(n_htlc, 3)))
    cases.append((csig, [_flip(s, 50) if i in bad else s for i, s in enumerate(hsigs)]))
    cases.append((_flip(csig), hsigs))
    for cs, sigs in cases:
        exp = _reference_loop(orc, ctx, fund, cs, 1, htxs, hkey, sigs, tys) if n_htlc <= 483 else None
        for rep in range(2 if n_htlc <= 483 else 1):   # (the second call of a kind finds the htlc key's table)
            got = eng.check_commitment_signed(ctx, fund, cs, 1, htxs, hkey, sigs, tys)
            if exp is not None:
                assert got[0] == exp[0] and list(got[1]) == exp[1]
            else:   # 5001 rows: the oracle on a sample + construction
                want = [cs == csig] + [s == o for s, o in zip(sigs, hsigs)]
                assert list(got[1]) == want and got[0] == next((i for i, w in enumerate(want) if not w), -1)
                for i in [0] + rnd.sample(range(1, n_htlc + 1), 40):
                    t, key, sg = (ctx, fund, cs) if i == 0 else (htxs[i - 1], hkey, sigs[i - 1])
                    sh = pyref.bip143_sighash(t["version"], t["inputs"], t["outputs"], t["locktime"], 0, t["script"], t["amount"], 1 if i == 0 else tys[i - 1])[0]
                    assert bool(orc.ecdsa_verify(sh, sg, key)) == bool(got[1][i])


def test_mirror_check_commit_sigs_prints_the_reference_warnings(kat, orc):
    """include/cln_shim.h check_commit_sigs(): NULL / the reference's text for the first failing check (channeld.c:2171-2232), on the BOLT #3 commitment"""
    from lightning_amd import _build
    from test_cln_shim import BitcoinSig, BitcoinTx, Pubkey, make_tx
    _build.build()
    shim = ctypes.CDLL(_build.build_shim())
    for n in ("lamd_shim_setup", "pubkey_from_der", "fromwire_secp256k1_ecdsa_signature"):
        getattr(shim, n).restype = ctypes.c_bool
    shim.check_commit_sigs.restype = ctypes.c_char_p
    shim.lamd_shim_last_error.restype = ctypes.c_char_p
    shim.shim_tal_dup.restype = ctypes.c_void_p
    shim.shim_tal_dup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    shim.check_commit_sigs.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()
    ctx, fund, csig, htxs, hkey, hsigs, c, hs = _bolt3(kat)
    keep = []

    def mk(t):
        tx, k = make_tx(shim, t["version"], t["locktime"], [(i[0], i[1], i[2], t["amount"]) for i in t["inputs"]], t["outputs"])
        keep.append((tx, k))
        return tx
    txs = [mk(ctx)] + [mk(t) for t in htxs]
    ptrs = (ctypes.c_void_p * len(txs))(*[ctypes.addressof(t) for t in txs])
    tal_txs = shim.shim_tal_dup(None, ctypes.string_at(ptrs, ctypes.sizeof(ptrs)), ctypes.sizeof(ptrs))
    fw = shim.shim_tal_dup(None, ctx["script"], len(ctx["script"]))
    ws = (ctypes.c_void_p * 5)(*[shim.shim_tal_dup(None, t["script"], len(t["script"])) for t in htxs])
    fk, hk = Pubkey(), Pubkey()
    assert shim.pubkey_from_der(fund, 33, ctypes.byref(fk)) and shim.pubkey_from_der(hkey, 33, ctypes.byref(hk))

    def bsig(raw, ty=1):
        s = BitcoinSig()
        assert shim.fromwire_secp256k1_ecdsa_signature(raw, ctypes.byref(s.s))
        s.sighash_type = ty
        return s

    def call(cs, sigs, n_sigs=None):
        arr = (BitcoinSig * len(sigs))(*sigs)
        blob = ctypes.string_at(arr, ctypes.sizeof(BitcoinSig) * (len(sigs) if n_sigs is None else n_sigs))
        tal_sigs = shim.shim_tal_dup(None, blob, len(blob))
        return shim.check_commit_sigs(None, 42, tal_txs, fw, ctypes.byref(fk), ctypes.byref(cs), ws, ctypes.byref(hk), tal_sigs, 15000,
                                      b". Outpoint 00:0, funding_sats: 10000000sat, funding_txid: N/A, inflight splice count: 0")
    from test_cln_shim import _der_hex
    good = [bsig(s) for s in hsigs]
    assert call(bsig(csig), good) is None
    # HTLC 3 bad: the second warning, naming the signature passed, the HTLC transaction, its wscript and the htlc key
    bad3 = _flip(hsigs[3])
    err = call(bsig(csig), good[:3] + [bsig(bad3)] + good[4:])
    t = htxs[3]
    lin = (t["version"].to_bytes(4, "little") + b"\x01" + t["inputs"][0][0] + t["inputs"][0][1].to_bytes(4, "little") + b"\x00" + t["inputs"][0][2].to_bytes(4, "little") +
           b"\x01" + t["outputs"][0][0].to_bytes(8, "little") + bytes([len(t["outputs"][0][1])]) + t["outputs"][0][1] + t["locktime"].to_bytes(4, "little"))
    assert err == (b"Bad commit_sig signature " + _der_hex(bad3) + b"01 for htlc " + lin.hex().encode() + b" wscript " + t["script"].hex().encode() + b" key " +
                   hkey.hex().encode()), err
    # commit sig bad AND an HTLC bad: the first warning (the reference never reaches the HTLC loop)
    err = call(bsig(_flip(csig)), good[:1] + [bsig(_flip(hsigs[1]))] + good[2:])
    assert err.startswith(b"Bad commit_sig signature 42 " + _der_hex(_flip(csig)) + b"01 for tx 02000000") and b" wscript " + ctx["script"].hex().encode() + b" key " + fund.hex().encode() + \
        b" feerate 15000. Outpoint 00:0, funding_sats: 10000000sat, funding_txid: N/A, inflight splice count: 0" in err
    # the count check sits between the two (:2203-2206)
    assert call(bsig(csig), good, n_sigs=4) == b"Expected 5 htlc sigs, not 4"
    assert call(bsig(_flip(csig)), good, n_sigs=4).startswith(b"Bad commit_sig signature 42 ")
    # a sighash type that only LOOKS like SIGHASH_ALL in its low byte is the reference gate's business (bitcoin/signature.c:206-211 compares the int):
    # 0x101 on HTLC 2 makes that row the first bad one, 0x101 on the commitment signature the commitment's (the mirror used to truncate to u8)
    err = call(bsig(csig), good[:2] + [bsig(hsigs[2], 0x101)] + good[3:])
    assert err is not None and err.startswith(b"Bad commit_sig signature " + _der_hex(hsigs[2])) and b" for htlc " in err, err
    err = call(bsig(csig, 0x101), good)
    assert err is not None and err.startswith(b"Bad commit_sig signature 42 "), err
    assert call(bsig(csig), good) is None
