"""The C++ host mirror of the reference's prototypes (lightning_amd/csrc/cln_shim.*).
CPU part: host framing (DER / compact parsing, SHA256d, sighash-type gate) against the goldens.
GPU part (-m gpu): the reference's own unit-test expectations through the reference's own names."""
import ctypes
import hashlib
import os

import pytest

H = bytes.fromhex


class Sig(ctypes.Structure):
    _fields_ = [("data", ctypes.c_ubyte * 64)]


class BitcoinSig(ctypes.Structure):
    _fields_ = [("s", Sig), ("sighash_type", ctypes.c_int)]


class Pubkey(ctypes.Structure):
    _fields_ = [("data", ctypes.c_ubyte * 64)]


class NodeId(ctypes.Structure):
    _fields_ = [("k", ctypes.c_ubyte * 33)]


class Sha256d(ctypes.Structure):
    _fields_ = [("u8", ctypes.c_ubyte * 32)]


class TxInput(ctypes.Structure):
    _fields_ = [("txid", ctypes.c_ubyte * 32), ("index", ctypes.c_uint32), ("sequence", ctypes.c_uint32), ("amount_sat", ctypes.c_uint64)]


class TxOutput(ctypes.Structure):
    _fields_ = [("amount_sat", ctypes.c_uint64), ("script", ctypes.c_void_p)]


class BitcoinTx(ctypes.Structure):
    _fields_ = [("version", ctypes.c_uint32), ("locktime", ctypes.c_uint32), ("num_inputs", ctypes.c_size_t), ("num_outputs", ctypes.c_size_t),
                ("inputs", ctypes.POINTER(TxInput)), ("outputs", ctypes.POINTER(TxOutput))]


def make_tx(shim, version, locktime, inputs, outputs):
    """inputs: [(txid32, vout, sequence, amount)], outputs: [(amount, spk)] -> (BitcoinTx, keepalive)"""
    ins = (TxInput * len(inputs))()
    for k, (txid, vout, seq, amt) in enumerate(inputs):
        ins[k].txid[:] = txid
        ins[k].index, ins[k].sequence, ins[k].amount_sat = vout, seq, amt
    outs = (TxOutput * max(1, len(outputs)))()
    for k, (amt, spk) in enumerate(outputs):
        outs[k].amount_sat = amt
        outs[k].script = shim.shim_tal_dup(None, spk, len(spk))
    tx = BitcoinTx(version, locktime, len(inputs), len(outputs), ins, outs)
    return tx, (ins, outs)


@pytest.fixture(scope="module")
def shim():
    from lightning_amd import _build
    _build.build()
    L = ctypes.CDLL(_build.build_shim())
    for n in ("pubkey_from_der", "pubkey_from_node_id", "fromwire_secp256k1_ecdsa_signature", "signature_from_der", "check_signed_hash",
              "check_signed_hash_nodeid", "check_schnorr_sig", "check_tx_sig", "lamd_shim_setup"):
        getattr(L, n).restype = ctypes.c_bool
    for n in ("sigcheck_channel_update", "sigcheck_channel_announcement", "sigcheck_node_announcement", "sigcheck_channel_update_len",
              "sigcheck_channel_announcement_len", "sigcheck_node_announcement_len", "lamd_shim_last_error"):
        getattr(L, n).restype = ctypes.c_char_p  # leaks the shim_tal_dup()ed string; fine in a test
    # (argtypes matter here: the eleventh argument of sigcheck_channel_announcement_len is a size_t that travels on the stack)
    vp = ctypes.c_void_p
    L.sigcheck_channel_announcement_len.argtypes = [vp] * 10 + [ctypes.c_size_t]
    L.sigcheck_channel_announcement.argtypes = [vp] * 10
    L.sigcheck_channel_update_len.argtypes = L.sigcheck_node_announcement_len.argtypes = [vp] * 4 + [ctypes.c_size_t]
    L.sigcheck_channel_update.argtypes = L.sigcheck_node_announcement.argtypes = [vp] * 4
    for n in ("lamd_secp256k1_ecdsa_verify", "lamd_secp256k1_ecdsa_recoverable_signature_convert"):
        getattr(L, n).restype = ctypes.c_int
    L.check_tx_sig_preimage.restype = ctypes.c_bool
    L.shim_tal_dup.restype = ctypes.c_void_p
    L.shim_tal_dup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.shim_tal_bytelen.restype = ctypes.c_size_t
    L.shim_tal_bytelen.argtypes = [ctypes.c_void_p]
    L.check_tx_sig.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return L


def test_signature_from_der_goldens(shim, kat):
    for v in kat["der"]:
        der = H(v["der"])
        if not (v.get("full") or v["name"] == "KAT-O"):
            der = der + b"\x01"  # signature_from_der wants the trailing sighash byte
        sig = BitcoinSig()
        ok = shim.signature_from_der(der, len(der), ctypes.byref(sig))
        if v["expect_sig"] is None:
            assert not ok, v["name"]
        else:
            assert ok, v["name"]
            assert bytes(sig.s.data) == H(v["expect_sig"]), v["name"]
            assert sig.sighash_type == (v["expect_sighash"] or 1)


def test_fromwire_compact_and_sha256_double(shim, kat):
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    sig = Sig()
    for r, s, exp in ((1, 1, True), (N - 1, N - 1, True), (N, 1, False), (1, N, False), (2**256 - 1, 5, False), (0, 0, True)):
        assert shim.fromwire_secp256k1_ecdsa_signature(r.to_bytes(32, "big") + s.to_bytes(32, "big"), ctypes.byref(sig)) == exp
    for ln in (0, 1, 55, 56, 64, 72, 290, 1000):
        data = bytes(range(256)) * 4
        out = Sha256d()
        shim.sha256_double(ctypes.byref(out), data[:ln], ln)
        assert bytes(out.u8) == hashlib.sha256(hashlib.sha256(data[:ln]).digest()).digest()
    for v in kat["bip143"]:
        out = Sha256d()
        pre = H(v["preimage"])
        shim.sha256_double(ctypes.byref(out), pre, len(pre))
        assert bytes(out.u8) == H(v["expect"])


def test_check_tx_sig_sighash_gate_needs_no_device(shim):
    """bitcoin/signature.c:206-211 rejects before any curve work -- through the reference's own prototype
    (struct bitcoin_tx *, input_num, subscript, witness_script, key, sig) and through the preimage form"""
    sig = BitcoinSig()
    key = Pubkey()
    tx, keep = make_tx(shim, 2, 0, [(bytes(32), 0, 0, 1000)], [(900, b"\x00\x14" + bytes(20))])
    sub = shim.shim_tal_dup(None, b"\x76\xa9", 2)
    assert shim.shim_tal_bytelen(sub) == 2
    assert not hasattr(shim, "tal_bytelen")                 # the mirror must not define ccan/tal's symbol (it links next to it in-tree)
    foreign = ctypes.create_string_buffer(64)                # a pointer shim_tal_dup() did not make: fails closed, no abort()
    assert shim.shim_tal_bytelen(ctypes.byref(foreign, 32)) == ctypes.c_size_t(-1).value
    for t, wit, reaches_verify in ((2, b"\x51", False), (0x83, None, False), (0x81, b"\x51", False), (3, b"\x51", False)):
        sig.sighash_type = t
        assert shim.check_tx_sig_preimage(b"\x00" * 10, 10, wit, ctypes.byref(key), ctypes.byref(sig)) is False
        w = shim.shim_tal_dup(None, wit, len(wit)) if wit is not None else None
        assert shim.check_tx_sig(ctypes.byref(tx), 0, sub, w, ctypes.byref(key), ctypes.byref(sig)) is False


def _cann_args(shim, m):
    """what gossmap_manage.c:620-687 hands sigcheck_channel_announcement(): the fromwire_channel_announcement() fields of `m`"""
    flen = int.from_bytes(m[258:260], "big")
    koff = 260 + flen + 40
    sigs = [Sig() for _ in range(4)]
    for i in range(4):
        assert shim.fromwire_secp256k1_ecdsa_signature(m[2 + 64 * i:66 + 64 * i], ctypes.byref(sigs[i]))
    ids = [NodeId.from_buffer_copy(m[koff + 33 * i:koff + 33 * i + 33]) for i in range(2)]
    keys = [Pubkey() for _ in range(2)]
    for i in range(2):
        assert shim.pubkey_from_der(m[koff + 66 + 33 * i:koff + 99 + 33 * i], 33, ctypes.byref(keys[i]))
    return sigs, ids, keys


def _nann_args(shim, m):
    flen = int.from_bytes(m[66:68], "big")
    sg = Sig()
    assert shim.fromwire_secp256k1_ecdsa_signature(m[2:66], ctypes.byref(sg))
    return sg, NodeId.from_buffer_copy(m[68 + flen + 4:68 + flen + 4 + 33])


def _cann_call(shim, sigs, ids, keys, m):
    return shim.sigcheck_channel_announcement_len(None, ctypes.byref(ids[0]), ctypes.byref(ids[1]), ctypes.byref(keys[0]), ctypes.byref(keys[1]),
                                                  ctypes.byref(sigs[0]), ctypes.byref(sigs[1]), ctypes.byref(sigs[2]), ctypes.byref(sigs[3]), m, len(m))


def _der_hex(sig64):
    """fmt_secp256k1_ecdsa_signature (bitcoin/signature.c:325-335): hex of the DER serialisation"""
    def enc(v):
        v = v.lstrip(b"\x00") or b"\x00"
        if v[0] & 0x80:
            v = b"\x00" + v
        return b"\x02" + bytes([len(v)]) + v
    body = enc(sig64[:32]) + enc(sig64[32:])
    return (b"\x30" + bytes([len(body)]) + body).hex().encode()


@pytest.mark.gpu
def test_sigcheck_verifies_its_arguments_not_the_message_bytes(shim, kat, orc):
    """gossipd/sigcheck.c:35,78,87,96,105,144: the three functions hash the message tail and then decide check_signed_hash[_nodeid]() on the
    signature and key ARGUMENTS, in order, first failure wins; they never parse the message and never call it malformed.  Every verdict below is
    the C oracle's on (SHA256d(tail), passed signature, passed key), row by row in the reference's order."""
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()
    names = [b"Bad node_signature_1", b"Bad node_signature_2", b"Bad bitcoin_signature_1", b"Bad bitcoin_signature_2"]
    good = [H(v["msg"]) for v in kat["gossip"] if v["kind"] == "channel_announcement" and v["expect"] == 0 and v["name"].startswith(("ref-store", "cann/ok"))]
    assert len(good) >= 8

    def oracle_cann(sigs, ids, keys, m):
        h = hashlib.sha256(hashlib.sha256(m[258:]).digest()).digest()
        k33 = [bytes(ids[0].k), bytes(ids[1].k)]
        for k in keys:
            out = (ctypes.c_ubyte * 33)()
            shim.pubkey_to_der(out, ctypes.byref(k))
            k33.append(bytes(out))
        for i in range(4):
            if not orc.ecdsa_verify(h, bytes(sigs[i].data), k33[i]):
                return i, h
        return -1, h

    def check(sigs, ids, keys, m):
        err = _cann_call(shim, sigs, ids, keys, m)
        bad, h = oracle_cann(sigs, ids, keys, m)
        if bad < 0:
            assert err is None, err
        else:
            want = names[bad] + b" " + _der_hex(bytes(sigs[bad].data)) + b" hash " + h.hex().encode() + b" on channel_announcement " + m.hex().encode()
            assert err == want, (err, want)
        return bad

    a, b = good[0], good[1]
    sa, ia, ka = _cann_args(shim, a)
    sb, ib, kb = _cann_args(shim, b)
    assert check(sa, ia, ka, a) == -1
    # (a) one signature ARGUMENT swapped for another message's: the string names the position and prints the PASSED signature
    for pos in range(4):
        swapped = list(sa)
        swapped[pos] = sb[pos]
        assert check(swapped, ia, ka, a) == pos
    # two bad arguments: the first in the reference's order is reported (early return)
    assert check([sa[0], sb[1], sa[2], sb[3]], ia, ka, a) == 1
    # a key argument that is not the message's
    assert check(sa, [ia[0], ib[1]], ka, a) == 1
    assert check(sa, ia, [kb[0], ka[1]], a) == 2
    # (b) the signatures EMBEDDED in the message are garbage (not even in range: fromwire would have refused them), the arguments are
    # good: NULL -- the function hashes msg + 258 and looks at nothing before it
    garbled = a[:2] + b"\xff" * 256 + a[258:]
    assert check(sa, ia, ka, garbled) == -1
    # embedded keys are part of the signed tail: changing one changes the hash, every argument then fails from the first on
    koff = 260 + int.from_bytes(a[258:260], "big") + 40
    assert check(sa, ia, ka, a[:koff + 5] + bytes([a[koff + 5] ^ 1]) + a[koff + 6:]) == 0
    # never "malformed": a truncated / over-long message is hashed as it stands
    for m2 in (a[:300], a + b"\x00", a[:258]):
        err = _cann_call(shim, sa, ia, ka, m2)
        assert err is not None and err.startswith(b"Bad node_signature_1 ") and b"malformed" not in err
        assert oracle_cann(sa, ia, ka, m2)[0] == 0
    # an invalid node_id argument "fails here" (sigcheck.c:142, common/node_id.c:72-80)
    assert check(sa, [NodeId.from_buffer_copy(b"\x05" + bytes(ia[0].k)[1:]), ia[1]], ka, a) == 0

    # ---- channel_update: the signer is the node_id ARGUMENT, the signature the ARGUMENT
    ups = [v for v in kat["gossip"] if v["kind"] == "channel_update" and v["expect"] == 0 and "node_id" in v]
    u0, u1 = ups[0], ups[1]
    m0, m1 = H(u0["msg"]), H(u1["msg"])
    s0, s1 = Sig(), Sig()
    assert shim.fromwire_secp256k1_ecdsa_signature(m0[2:66], ctypes.byref(s0)) and shim.fromwire_secp256k1_ecdsa_signature(m1[2:66], ctypes.byref(s1))
    n0, n1 = NodeId.from_buffer_copy(H(u0["node_id"])), NodeId.from_buffer_copy(H(u1["node_id"]))
    h0 = hashlib.sha256(hashlib.sha256(m0[66:]).digest()).digest()
    for sg, nid, mm in ((s0, n0, m0), (s1, n0, m0), (s0, n1, m0), (s0, n0, m0[:2] + b"\xee" * 64 + m0[66:])):
        err = shim.sigcheck_channel_update_len(None, ctypes.byref(nid), ctypes.byref(sg), mm, len(mm))
        if orc.ecdsa_verify(h0, bytes(sg.data), bytes(nid.k)):
            assert err is None
        else:
            assert err == b"Bad signature for " + _der_hex(bytes(sg.data)) + b" hash " + h0.hex().encode() + b" on channel_update " + mm.hex().encode()
    assert shim.sigcheck_channel_update_len(None, ctypes.byref(n0), ctypes.byref(s0), m0[:2] + b"\xee" * 64 + m0[66:], len(m0)) is None
    assert shim.sigcheck_channel_update_len(None, ctypes.byref(n0), ctypes.byref(s1), m0, len(m0)) is not None
    assert u0["node_id"] == u1["node_id"] or shim.sigcheck_channel_update_len(None, ctypes.byref(n1), ctypes.byref(s0), m0, len(m0)) is not None

    # ---- node_announcement
    nn = [H(v["msg"]) for v in kat["gossip"] if v["kind"] == "node_announcement" and v["expect"] == 0][:2]
    (g0, i0), (g1, i1) = _nann_args(shim, nn[0]), _nann_args(shim, nn[1])
    hn = hashlib.sha256(hashlib.sha256(nn[0][66:]).digest()).digest()
    for sg, nid, mm in ((g0, i0, nn[0]), (g1, i0, nn[0]), (g0, i1, nn[0]), (g0, i0, nn[0][:2] + bytes(64) + nn[0][66:])):
        err = shim.sigcheck_node_announcement_len(None, ctypes.byref(nid), ctypes.byref(sg), mm, len(mm))
        if orc.ecdsa_verify(hn, bytes(sg.data), bytes(nid.k)):
            assert err is None
        else:
            assert err == b"Bad signature for " + _der_hex(bytes(sg.data)) + b" hash " + hn.hex().encode() + b" on node_announcement " + mm.hex().encode()
    assert shim.sigcheck_node_announcement_len(None, ctypes.byref(i0), ctypes.byref(g0), nn[0][:2] + bytes(64) + nn[0][66:], len(nn[0])) is None


@pytest.mark.gpu
def test_sigcheck_arguments_from_c(kat, tmp_path):
    """the same two properties from C through the reference's prototype (tests/c/run_sigcheck_arguments.c): (a) a valid message with the
    node_signature_2 ARGUMENT taken from another message -> "Bad node_signature_2 <DER of the passed signature>"; (b) a message whose embedded
    signatures are garbage with good arguments -> NULL"""
    import subprocess
    from lightning_amd import _build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _build.build_shim()
    exe = tmp_path / "rsa"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), "-o", str(exe),
                           os.path.join(root, "tests", "c", "run_sigcheck_arguments.c"), "-L" + os.path.join(root, "lightning_amd"),
                           "-llightning_amd_cln", "-llightning_amd", "-Wl,-rpath," + os.path.join(root, "lightning_amd")])
    good = [v["msg"] for v in kat["gossip"] if v["kind"] == "channel_announcement" and v["expect"] == 0 and v["name"].startswith(("ref-store", "cann/ok"))]
    r = subprocess.run([str(exe), good[0], good[1]], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    b = H(good[1])
    assert "Bad node_signature_2 " + _der_hex(b[66:130]).decode() + " hash " in r.stdout


@pytest.mark.gpu
def test_reference_unit_test_expectations(shim, kat):
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()
    # ---- gossipd/test/run-check_channel_announcement.c:62-108
    for name, needle in (("KAT-G/orig", b"Bad node_signature_1"), ("KAT-G/features-stripped", b"Bad node_signature_2")):
        m = H(next(v for v in kat["gossip"] if v["name"] == name)["msg"])
        flen = int.from_bytes(m[258:260], "big")
        koff = 260 + flen + 40
        sigs = [Sig() for _ in range(4)]
        for i in range(4):
            assert shim.fromwire_secp256k1_ecdsa_signature(m[2 + 64 * i:66 + 64 * i], ctypes.byref(sigs[i]))
        ids = [NodeId.from_buffer_copy(m[koff + 33 * i:koff + 33 * i + 33]) for i in range(2)]
        keys = [Pubkey() for _ in range(2)]
        for i in range(2):
            assert shim.pubkey_from_der(m[koff + 66 + 33 * i:koff + 99 + 33 * i], 33, ctypes.byref(keys[i]))
        err = shim.sigcheck_channel_announcement_len(None, ctypes.byref(ids[0]), ctypes.byref(ids[1]), ctypes.byref(keys[0]), ctypes.byref(keys[1]),
                                                     ctypes.byref(sigs[0]), ctypes.byref(sigs[1]), ctypes.byref(sigs[2]), ctypes.byref(sigs[3]), m, len(m))
        assert err is not None and needle in err
        # the reference's own prototype (gossipd/sigcheck.h:15-24): the message is a tal array, no length argument
        tm = ctypes.c_void_p(shim.shim_tal_dup(None, m, len(m)))
        err2 = shim.sigcheck_channel_announcement(None, ctypes.byref(ids[0]), ctypes.byref(ids[1]), ctypes.byref(keys[0]), ctypes.byref(keys[1]),
                                                  ctypes.byref(sigs[0]), ctypes.byref(sigs[1]), ctypes.byref(sigs[2]), ctypes.byref(sigs[3]), tm)
        assert err2 == err
        # a pointer that is not a tal array fails closed -- never "OK"
        foreign = ctypes.create_string_buffer(bytes(32) + m)
        err3 = shim.sigcheck_channel_announcement(None, ctypes.byref(ids[0]), ctypes.byref(ids[1]), ctypes.byref(keys[0]), ctypes.byref(keys[1]),
                                                  ctypes.byref(sigs[0]), ctypes.byref(sigs[1]), ctypes.byref(sigs[2]), ctypes.byref(sigs[3]),
                                                  ctypes.byref(foreign, 32))
        assert err3 is not None and b"not a tal array" in err3
        if name == "KAT-G/orig":
            # the exact text quoted at the top of the reference's test file
            assert err.startswith(b"Bad node_signature_1 3044022011effc9ed10fceccfae5f9e3fef20d983b06eed030e968fd8d1e6c5905e18f9f02202df6a43f00d7c0ddf52e0467ab1e32394051b72ea6343fb008a4117c265f3d7b "
                                  b"hash bb92b8f45b48e65ad2f2cfff2242fa921b4cf46f709a372ca7788537e89d9de1 on channel_announcement 010011effc9ed10f")
    # ---- onchaind/test/run-grind_feerate.c:120-150: one fee matches
    ko = next(v for v in kat["der"] if v["name"] == "KAT-O")
    sig = BitcoinSig()
    der = H(ko["der"])
    assert shim.signature_from_der(der, len(der), ctypes.byref(sig))
    key = Pubkey()
    assert shim.pubkey_from_der(H("038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de054432eb7f"), 33, ctypes.byref(key))
    verdicts = {}
    for v in kat["bip143"]:
        pre = H(v["preimage"])
        verdicts[v["name"]] = shim.check_tx_sig_preimage(pre, len(pre), b"\x76", ctypes.byref(key), ctypes.byref(sig))
    assert verdicts["KAT-O/fee=165750"] is True and sum(verdicts.values()) == 1
    # ... and as onchaind calls it (onchaind.c:430-432): check_tx_sig(tx, 0, NULL, wscript, &keyset->other_htlc_key, remotesig) on the
    # transaction itself -- the BIP143 hash is built on the device from the template
    rawtx = H("0200000001e1ebca08cf1c301ac563580a1126d5c8fcb0e5e2043230b852c726553caf1e1d0000000000000000000160ae0a0000000000"
              "22002082e03c5a9cb79c82cd5a0572dc175290bc044609aabe9cc852d61927436041796d000000")
    wscript = H("76a914a8c40c334351dbe8e5908544f1c98fbcfb8719fc8763ac6721038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de05"
                "4432eb7f7c820120876475527c2103cf8e2f193a6aed60db80af75f3c8d59c2de735b299b7c7083527be9bd23b77a852ae67a914b8bcd51e"
                "fa35be1e50ae2d5f72f4500acb005c9c88ac6868")
    ws = shim.shim_tal_dup(None, wscript, len(wscript))
    got = {}
    for fee in (165749, 165750, 165751, 0):
        tx, keep = make_tx(shim, 2, 109, [(rawtx[5:37], 0, 0, 700000)], [(700000 - fee, rawtx[56:90])])
        got[fee] = shim.check_tx_sig(ctypes.byref(tx), 0, None, ws, ctypes.byref(key), ctypes.byref(sig))
    assert got == {165749: False, 165750: True, 165751: False, 0: False}
    # ... the signed BOLT #3 transactions the reference tree holds (channeld/test/run-full_channel.c HTLC transactions, wallet/test/run-wallet.c
    # commitment transaction) through the same prototype: check_tx_sig(tx, 0, NULL, wscript, &key, &sig)
    for v in kat["txsig"]:
        txv, keepv = make_tx(shim, v["version"], v["locktime"], [(H(t), vout, seq, v["amount"]) for t, vout, seq in v["inputs"]],
                             [(a, H(spk)) for a, spk in v["outputs"]])
        wsv = shim.shim_tal_dup(None, H(v["script"]), len(v["script"]) // 2)
        kv, sv = Pubkey(), BitcoinSig()
        assert shim.pubkey_from_der(H(v["pub"]), 33, ctypes.byref(kv))
        assert shim.fromwire_secp256k1_ecdsa_signature(H(v["sig"]), ctypes.byref(sv.s))
        sv.sighash_type = v["sighash_type"]
        assert shim.check_tx_sig(ctypes.byref(txv), v["input_num"], None, wsv, ctypes.byref(kv), ctypes.byref(sv)) is v["expect"], v["name"]
    # ... and the grind itself as the test runs it: weight 663, max_possible_feerate 250 000, 1000 iterations
    pre = H(next(v for v in kat["bip143"] if v["name"] == "KAT-O/fee=0")["preimage"])
    spk = H("002082e03c5a9cb79c82cd5a0572dc175290bc044609aabe9cc852d6192743604179")
    outputs = (700000).to_bytes(8, "little") + bytes([len(spk)]) + spk
    fee = ctypes.c_uint64(0)
    shim.grind_htlc_tx_fee.restype = ctypes.c_bool
    shim.grind_htlc_tx_fee.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64,
                                       ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    assert shim.grind_htlc_tx_fee(ctypes.byref(fee), pre, len(pre), outputs, len(outputs), 700000, ctypes.byref(sig), b"\x76", 663, 249001, 250000,
                                  ctypes.byref(key)) is True
    assert fee.value == 165750
    assert shim.grind_htlc_tx_fee(ctypes.byref(fee), pre, len(pre), outputs, len(outputs), 700000, ctypes.byref(sig), b"\x76", 663, 0, 249000,
                                  ctypes.byref(key)) is False
    # ---- check_signed_hash / _nodeid / schnorr through the reference's names
    b11 = next(v for v in kat["ecdsa"] if v["name"] == "KAT-B11")
    h = Sha256d.from_buffer_copy(H(b11["hash"]))
    s = Sig()
    assert shim.fromwire_secp256k1_ecdsa_signature(H(b11["sig"]), ctypes.byref(s))
    nid = NodeId.from_buffer_copy(H(b11["pub"]))
    assert shim.check_signed_hash_nodeid(ctypes.byref(h), ctypes.byref(s), ctypes.byref(nid)) is True
    pk = Pubkey()
    assert shim.pubkey_from_node_id(ctypes.byref(pk), ctypes.byref(nid))
    assert shim.check_signed_hash(ctypes.byref(h), ctypes.byref(s), ctypes.byref(pk)) is True
    out = (ctypes.c_ubyte * 33)()
    shim.pubkey_to_der(out, ctypes.byref(pk))
    assert bytes(out) == H(b11["pub"])
    h2 = Sha256d.from_buffer_copy(bytes([H(b11["hash"])[0] ^ 1]) + H(b11["hash"])[1:])
    assert shim.check_signed_hash(ctypes.byref(h2), ctypes.byref(s), ctypes.byref(pk)) is False
    bad = NodeId.from_buffer_copy(b"\x05" + H(b11["pub"])[1:])
    assert shim.check_signed_hash_nodeid(ctypes.byref(h), ctypes.byref(s), ctypes.byref(bad)) is False  # "If node_id is invalid, it fails here"
    for v in kat["schnorr"][:15]:
        # check_schnorr_sig gets a full key; both liftings of the x-only key must give the same verdict
        pkx = Pubkey()
        if not shim.pubkey_from_der(b"\x02" + H(v["pk"]), 33, ctypes.byref(pkx)):
            continue  # not a valid x: the reference could not even construct the pubkey argument
        msg = Sha256d.from_buffer_copy(H(v["msg"]))
        sg = Sig.from_buffer_copy(H(v["sig"]))
        assert shim.check_schnorr_sig(ctypes.byref(msg), ctypes.byref(pkx), ctypes.byref(sg)) == v["expect"], v["name"]
    # channel_update / node_announcement wording
    cu = next(v for v in kat["gossip"] if v["name"] == "cupd/lastnibble/0")
    m = H(cu["msg"])
    sg = Sig()
    assert shim.fromwire_secp256k1_ecdsa_signature(m[2:66], ctypes.byref(sg))
    err = shim.sigcheck_channel_update_len(None, ctypes.byref(NodeId.from_buffer_copy(H(cu["node_id"]))), ctypes.byref(sg), m, len(m))
    assert err.startswith(b"Bad signature for 30") and b" on channel_update 0102" in err  # tests/test_gossip.py:2001 greps "Bad signature"
    tm = ctypes.c_void_p(shim.shim_tal_dup(None, m, len(m)))
    assert shim.sigcheck_channel_update(None, ctypes.byref(NodeId.from_buffer_copy(H(cu["node_id"]))), ctypes.byref(sg), tm) == err
    ok = next(v for v in kat["gossip"] if v["name"] == "nann/ok/0")
    m = H(ok["msg"])
    nsg, nid = _nann_args(shim, m)
    assert shim.sigcheck_node_announcement_len(None, ctypes.byref(nid), ctypes.byref(nsg), m, len(m)) is None
    assert shim.sigcheck_node_announcement(None, ctypes.byref(nid), ctypes.byref(nsg), ctypes.c_void_p(shim.shim_tal_dup(None, m, len(m)))) is None
    # the signature of ANOTHER message (the channel_update's) as the argument: the reference verifies what it is handed
    err = shim.sigcheck_node_announcement_len(None, ctypes.byref(nid), ctypes.byref(sg), m, len(m))
    assert err.startswith(b"Bad signature for 30") and b" on node_announcement 0101" in err


@pytest.mark.gpu
def test_reference_unit_test_calls_compile_unchanged_and_give_the_reference_strings(kat, tmp_path):
    """gossipd/test/run-check_channel_announcement.c:77-85,100-108 -- the two call statements as they stand there, compiled against
    include/cln_shim.h (tests/c/run_check_channel_announcement.c) and linked with the mirror: "Bad node_signature_1" for the message as
    received, "Bad node_signature_2" once the features are stripped and the message re-serialised (the reference's assertions)"""
    import subprocess
    from lightning_amd import _build
    _build.build_shim()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "rcca"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), "-o", str(exe),
                           os.path.join(root, "tests", "c", "run_check_channel_announcement.c"), "-L" + os.path.join(root, "lightning_amd"),
                           "-llightning_amd_cln", "-llightning_amd", "-Wl,-rpath," + os.path.join(root, "lightning_amd")])
    msg = next(v for v in kat["gossip"] if v["name"] == "KAT-G/orig")["msg"]
    r = subprocess.run([str(exe), msg], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("Bad node_signature_1 3044022011effc9ed10fceccfae5f9e3fef20d983b06eed030e968fd8d1e6c5905e18f9f02202df6a43f00d7c0ddf52e0467ab1e32394051b72ea6343fb008a4117c265f3d7b "
                               "hash bb92b8f45b48e65ad2f2cfff2242fa921b4cf46f709a372ca7788537e89d9de1 on channel_announcement 010011effc9ed10f")
    assert lines[1].startswith("Bad node_signature_2 30440220732bab7df4ee404ac926aef6610f4eb33e31baabfd9afdbf897c8a80057efa14022068362b4d2cc0a5482013e1058c8205717f85c3bc82c3ea89f17cfeac21e2cb2a "
                               "hash 59e255d34a96fa25dc666ecee7f2d70aefd34cb39170fe4246ba6328fabcc85b on channel_announcement 0100")


@pytest.mark.gpu
def test_libsecp_names_of_the_bolt11_n_field_path(shim, kat):
    """common/bolt11.c:1021-1057 with an `n` field: parse_compact -> convert -> secp256k1_ecdsa_verify(ctx, &sig, hash, &key) -- the
    reference's own invoice vectors (common/test/run-bolt11.c): the converted signature verifies under the key recovery yields, and
    neither under another key nor with one hash bit flipped; lightningd/dual_open_control.c:2254 makes the same raw call"""
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()

    class RecSig(ctypes.Structure):
        _fields_ = [("data", ctypes.c_ubyte * 65)]
    HALF_N = 0x7FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF5D576E7357A4501DDFE92F46681B20A0
    distinct, n_checked = {}, 0
    for v in kat["recover"]:
        if not v["expect"] or not (v["name"].startswith("KAT-B11R") or v["name"].startswith("KAT-SIGNMSG")):
            continue
        rs, sg, pk = RecSig(), Sig(), Pubkey()
        assert shim.lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(None, ctypes.byref(rs), H(v["sig"]), v["recid"]) == 1
        assert shim.lamd_secp256k1_ecdsa_recoverable_signature_convert(None, ctypes.byref(sg), ctypes.byref(rs)) == 1
        assert bytes(sg.data) == H(v["sig"])
        assert shim.pubkey_from_der(H(v["expect"]), 33, ctypes.byref(pk))
        h = H(v["hash"])
        # secp256k1_ecdsa_verify accepts low-S signatures only (bitcoin/signature.c:185-187); recovery has no such rule
        low_s = int.from_bytes(H(v["sig"])[32:], "big") <= HALF_N
        assert shim.lamd_secp256k1_ecdsa_verify(None, ctypes.byref(sg), h, ctypes.byref(pk)) == (1 if low_s else 0), v["name"]
        assert shim.lamd_secp256k1_ecdsa_verify(None, ctypes.byref(sg), bytes([h[0] ^ 1]) + h[1:], ctypes.byref(pk)) == 0, v["name"]
        n_checked += low_s
        if "/" not in v["name"].split("/", 1)[1]:          # the vector itself, not its other-parity / modified-message twin
            distinct.setdefault(bytes(pk.data), (sg, h, pk))
    assert n_checked >= 12
    ds = list(distinct.values())
    assert len(ds) >= 4
    for a, b in zip(ds, ds[1:]):      # under another vector's signer nothing verifies (both recovery ids of ONE signature do: those twins are skipped)
        assert shim.lamd_secp256k1_ecdsa_verify(None, ctypes.byref(a[0]), a[1], ctypes.byref(b[2])) == 0


@pytest.mark.gpu
def test_bolt11_recovery_through_libsecp_names(shim, kat):
    """common/bolt11.c:1021-1046 as written there: parse the 64+1 byte signature, recover, node_id_from_pubkey -- for the
    reference's own test invoices (common/test/run-bolt11.c) the receiver id must be the key the test pins at :310"""
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()

    class RecSig(ctypes.Structure):
        _fields_ = [("data", ctypes.c_ubyte * 65)]
    for v in kat["recover"]:
        if not v["name"].startswith("KAT-B11R/") and v["recid"] > 3:
            rs = RecSig()
            assert shim.lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(None, ctypes.byref(rs), H(v["sig"]), v["recid"]) == 0
            continue
        rs, pk, nid = RecSig(), Pubkey(), NodeId()
        parsed = shim.lamd_secp256k1_ecdsa_recoverable_signature_parse_compact(None, ctypes.byref(rs), H(v["sig"]), v["recid"])
        got = None
        if parsed and shim.lamd_secp256k1_ecdsa_recover(None, ctypes.byref(pk), ctypes.byref(rs), H(v["hash"])):
            shim.node_id_from_pubkey(ctypes.byref(nid), ctypes.byref(pk))
            got = bytes(nid.k).hex()
        assert got == v["expect"], v["name"]


def _cfg1_rows():
    blob = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1.bin"), "rb").read()
    assert len(blob) == 1024 * 130
    return [(blob[o:o + 32], blob[o + 32:o + 96], blob[o + 96:o + 129], bool(blob[o + 129])) for o in range(0, len(blob), 130)]


def test_cfg1_fixture_on_the_cpu_oracle(kat, orc):
    """BASELINE configs[0]: 1 024 ECDSA triples, seed 0xC1A00001, one by one through the CPU oracle's
    check_signed_hash equivalent; the reference-held KAT-G / KAT-O / KAT-B11 rows are rows 0..13; exact ok[]"""
    import pyref
    rows = _cfg1_rows()
    kats = [v for v in kat["ecdsa"] if v["name"].startswith("KAT")]
    assert len(kats) == 14 and {v["name"].split("/")[0] for v in kats} == {"KAT-G", "KAT-O", "KAT-B11"}
    for (h, s, p, e), v in zip(rows, kats):
        assert (h.hex(), s.hex(), p.hex(), e) == (v["hash"], v["sig"], v["pub"], v["expect"])
    got = [bool(orc.ecdsa_verify(h, s, p)) for h, s, p, _ in rows]
    assert got == [e for _, _, _, e in rows]
    assert 880 <= sum(got) <= 960
    for h, s, p, e in rows[:14] + rows[14::37]:
        assert pyref.ecdsa_verify(h, s, p) == e


@pytest.mark.gpu
def test_cfg1_one_by_one_through_check_signed_hash(shim):
    """the same 1 024 triples pushed ONE BY ONE through the shim's check_signed_hash() (bitcoin/signature.c:174-192 shape:
    parsed signature + parsed key), and through check_signed_hash_nodeid(); exact ok[]"""
    assert shim.lamd_shim_setup(), shim.lamd_shim_last_error()
    rows = _cfg1_rows()
    got, got_id = [], []
    for h, s, p, e in rows:
        hh, sg, pk = Sha256d.from_buffer_copy(h), Sig(), Pubkey()
        sig_ok = shim.fromwire_secp256k1_ecdsa_signature(s, ctypes.byref(sg))
        key_ok = shim.pubkey_from_der(p, 33, ctypes.byref(pk))
        # the reference cannot even call check_signed_hash when either parse fails: that IS the combined verdict
        got.append(bool(sig_ok and key_ok and shim.check_signed_hash(ctypes.byref(hh), ctypes.byref(sg), ctypes.byref(pk))))
        nid = NodeId.from_buffer_copy(p)
        got_id.append(bool(sig_ok and shim.check_signed_hash_nodeid(ctypes.byref(hh), ctypes.byref(sg), ctypes.byref(nid))))
    exp = [e for _, _, _, e in rows]
    assert got == exp, [i for i in range(1024) if got[i] != exp[i]][:10]
    assert got_id == exp


class TlvField(ctypes.Structure):
    _fields_ = [("meta", ctypes.c_void_p), ("numtype", ctypes.c_uint64), ("length", ctypes.c_size_t), ("value", ctypes.c_void_p)]


@pytest.mark.gpu
def test_bolt12_check_signature_through_the_reference_prototype(shim):
    """bolt12_check_signature(fields, messagename, fieldname, key, sig) / merkle_tlv / sighash_from_merkle (common/bolt12.c:80-92,
    common/bolt12_merkle.h) on the invoice_request of common/test/run-bolt12_merkle.c:332-361 and the specification's n1 root"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref
    shim.shim_tal_dup.restype = ctypes.c_void_p
    shim.shim_tal_dup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    shim.bolt12_check_signature.restype = ctypes.c_bool
    shim.bolt12_check_signature.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    shim.merkle_tlv.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    shim.sighash_from_merkle.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]

    def tal_fields(fields):
        arr = (TlvField * len(fields))()
        keep = []
        for i, (t, v) in enumerate(fields):
            buf = ctypes.create_string_buffer(bytes(v), max(1, len(v)))
            keep.append(buf)
            arr[i].numtype, arr[i].length, arr[i].value = t, len(v), ctypes.cast(buf, ctypes.c_void_p).value
        return shim.shim_tal_dup(None, bytes(arr), ctypes.sizeof(arr)), keep
    out = (ctypes.c_ubyte * 32)()
    f1, keep1 = tal_fields([(1, (1000).to_bytes(2, "big"))])
    shim.merkle_tlv(f1, out)
    assert bytes(out).hex() == "b013756c8fee86503a0b4abdab4cddeb1af5d344ca6fc2fa8b6c08938caa6f93"
    bob_d = int.from_bytes(b"B" * 32, "big")
    alice, bob = pyref.pubkey_create(int.from_bytes(b"A" * 32, "big")), pyref.pubkey_create(bob_d)
    fields = [(0, bytes(8)), (6, b"USD"), (8, b"\x64"), (10, b"A Mathematical Treatise"), (22, pyref.ser33(alice)), (88, pyref.ser33(bob))]
    root = pyref.bolt12_merkle(fields)
    sh = (ctypes.c_ubyte * 32)()
    shim.sighash_from_merkle(b"invoice_request", b"signature", root, sh)
    assert bytes(sh) == pyref.bolt12_sighash(b"invoice_request", b"signature", root)
    sig = pyref.schnorr_sign(bytes(sh), bob_d)
    fa, keep2 = tal_fields(fields + [(240, sig)])
    shim.merkle_tlv(fa, out)
    assert bytes(out) == root
    key = Pubkey()
    assert shim.pubkey_from_der(pyref.ser33(bob), 33, ctypes.byref(key))
    assert shim.bolt12_check_signature(fa, b"invoice_request", b"signature", ctypes.byref(key), sig) is True
    assert shim.bolt12_check_signature(fa, b"invoice", b"signature", ctypes.byref(key), sig) is False
    bad = sig[:10] + bytes([sig[10] ^ 1]) + sig[11:]
    assert shim.bolt12_check_signature(fa, b"invoice_request", b"signature", ctypes.byref(key), bad) is False
    akey = Pubkey()
    assert shim.pubkey_from_der(pyref.ser33(alice), 33, ctypes.byref(akey))
    assert shim.bolt12_check_signature(fa, b"invoice_request", b"signature", ctypes.byref(akey), sig) is False


@pytest.mark.gpu
def test_bolt12_reference_held_strings_through_the_mirror(shim, kat):
    """bolt12_check_signature() of the mirror (common/bolt12.c:80-92) on the lni1 / lnr1 literals the reference tree holds
    (kat.json "bolt12"): the strings signed by the reference's libsecp256k1 verify, their damaged twins do not"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyref
    shim.shim_tal_dup.restype = ctypes.c_void_p
    shim.shim_tal_dup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    shim.bolt12_check_signature.restype = ctypes.c_bool
    shim.bolt12_check_signature.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    n_true = 0
    for v in kat["bolt12"]:
        fields = pyref.tlv_stream_parse(bytes.fromhex(v["stream"]))
        if fields is None:
            continue                                   # fromwire_tlv already failed in the caller: the prototype takes parsed fields
        arr = (TlvField * len(fields))()
        keep = []
        for i, (t, val) in enumerate(fields):
            buf = ctypes.create_string_buffer(bytes(val), max(1, len(val)))
            keep.append(buf)
            arr[i].numtype, arr[i].length, arr[i].value = t, len(val), ctypes.cast(buf, ctypes.c_void_p).value
        fa = shim.shim_tal_dup(None, bytes(arr), ctypes.sizeof(arr))
        key = Pubkey()
        if not shim.pubkey_from_der(bytes.fromhex(v["key"]), 33, ctypes.byref(key)):
            assert not v["expect"]
            continue
        got = shim.bolt12_check_signature(fa, v["messagename"].encode(), b"signature", ctypes.byref(key), bytes.fromhex(v["sig"]))
        assert got is v["expect"], v["name"]
        n_true += got
    assert n_true >= 9
