"""ctypes binding of the C oracle (oracle/secp256k1_oracle.c).  TEST INFRASTRUCTURE ONLY:
import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else."""
import ctypes
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "liblnamd_oracle.so")
_OSSL = os.path.join(_DIR, "libossl_xcheck.so")
_SECPDL = os.path.join(_DIR, "libsecp_dl.so")


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in ("secp256k1_oracle.c", "secp256k1_oracle.h", "edge_gen.c", "openssl_xcheck.c", "libsecp_dl.c", "Makefile")]
    stale = force or not (os.path.exists(_LIB) and os.path.exists(_OSSL) and os.path.exists(_SECPDL))
    if not stale:
        t = min(os.path.getmtime(_LIB), os.path.getmtime(_OSSL), os.path.getmtime(_SECPDL))
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _DIR, "-s"])


_lib = None
_ossl = None
_u8p = ctypes.c_char_p


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.orc_init.restype = None
        for name, args in {
            "orc_pubkey_parse": [_u8p, ctypes.c_size_t, _u8p],
            "orc_sig_parse_compact": [_u8p],
            "orc_sig_parse_der": [_u8p, ctypes.c_size_t, _u8p],
            "orc_signature_from_der": [_u8p, ctypes.c_size_t, _u8p, ctypes.POINTER(ctypes.c_int)],
            "orc_ecdsa_verify": [_u8p, _u8p, _u8p, ctypes.c_size_t],
            "orc_schnorr_verify": [_u8p, _u8p, _u8p],
            "orc_sigcheck_channel_announcement": [_u8p, ctypes.c_size_t],
            "orc_sigcheck_channel_update": [_u8p, ctypes.c_size_t, _u8p],
            "orc_sigcheck_node_announcement": [_u8p, ctypes.c_size_t],
            "orc_ecdsa_recover": [_u8p, _u8p, ctypes.c_int, _u8p],
            "orc_pubkey_create": [_u8p, _u8p],
            "orc_ecdsa_sign": [_u8p, _u8p, _u8p, _u8p],
            "orc_schnorr_sign": [_u8p, _u8p, _u8p, _u8p],
        }.items():
            f = getattr(L, name)
            f.argtypes = args
            f.restype = ctypes.c_int
        for name in ("orc_sha256", "orc_sha256d"):
            f = getattr(L, name)
            f.argtypes = [_u8p, ctypes.c_size_t, _u8p]
            f.restype = None
        L.orc_ecdsa_verify_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
        L.orc_ecdsa_verify_batch.restype = None
        L.orc_schnorr_verify_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_int]
        L.orc_schnorr_verify_batch.restype = None
        L.orc_sigcheck_gossip_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_sigcheck_gossip_batch.restype = None
        L.orc_ecdsa_recover_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int]
        L.orc_ecdsa_recover_batch.restype = None
        L.orc_init()
        _lib = L
    return _lib


def ossl():
    global _ossl
    if _ossl is None:
        build()
        L = ctypes.CDLL(_OSSL)
        L.ossl_ecdsa_verify.argtypes = [_u8p, _u8p, _u8p, ctypes.c_size_t]
        L.ossl_ecdsa_verify.restype = ctypes.c_int
        L.ossl_schnorr_verify.argtypes = [_u8p, _u8p, _u8p]
        L.ossl_schnorr_verify.restype = ctypes.c_int
        L.ossl_ecdsa_recover.argtypes = [_u8p, _u8p, ctypes.c_int, _u8p]
        L.ossl_ecdsa_recover.restype = ctypes.c_int
        L.ossl_ecdsa_verify_rules_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                    ctypes.c_void_p]
        L.ossl_ecdsa_verify_rules_batch.restype = None
        _ossl = L
    return _ossl


_secpdl = None


def secpdl():
    """oracle/libsecp_dl.c: a real libsecp256k1 through dlopen, if this machine has one (leg C0)"""
    global _secpdl
    if _secpdl is None:
        build()
        L = ctypes.CDLL(_SECPDL)
        L.secpdl_open.argtypes = [ctypes.c_char_p]
        L.secpdl_path.restype = ctypes.c_char_p
        L.secpdl_ecdsa_verify_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                ctypes.c_void_p]
        L.secpdl_schnorr_verify_batch.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _secpdl = L
    return _secpdl


def bundled_libsecp_candidates():
    """shared objects under the interpreter's package directories that may BE a libsecp256k1: the wheels of coincurve, python-bitcoinlib's
    friends, electrum, secp256k1, ... bundle one (coincurve/_libsecp256k1*.so, secp256k1/_libsecp256k1*.so, *.libs/libsecp256k1-*.so.*).
    Only names are matched here; secpdl_open() decides by the symbols it finds."""
    import glob
    import site
    import sysconfig
    roots = []
    for r in list(getattr(site, "getsitepackages", lambda: [])()) + [getattr(site, "getusersitepackages", lambda: "")(), sysconfig.get_paths().get("purelib", ""),
                                                                    sysconfig.get_paths().get("platlib", "")]:
        if r and os.path.isdir(r) and r not in roots:
            roots.append(r)
    out = []
    for r in roots:
        for pat in ("*secp256k1*.so*", "*/*secp256k1*.so*", "*.libs/*secp256k1*", "*/.libs/*secp256k1*", "*/*/*secp256k1*.so*"):
            for f in sorted(glob.glob(os.path.join(r, pat))):
                if os.path.isfile(f) and f not in out:
                    out.append(f)
    return out


def libsecp_available(path=None):
    """path of the libsecp256k1 shared object that could be loaded, or None: `path`, $LAMD_LIBSECP256K1 and the usual sonames first (libsecp_dl.c),
    then whatever a Python package of this interpreter bundles -- so that BASELINE.md's leg C0 appears the day a node has the library in ANY form"""
    L = secpdl()
    if L.secpdl_open(path.encode() if path else None):
        return L.secpdl_path().decode()
    if path is None and not os.environ.get("LAMD_LIBSECP256K1"):
        for cand in bundled_libsecp_candidates():
            if L.secpdl_open(cand.encode()):
                return L.secpdl_path().decode()
    return None


def libsecp_ecdsa_verify_batch(hashes, sigs, pubs, publen):
    """uint8 [n] verdicts of the real library, or None when there is none"""
    import numpy as np
    out = np.zeros(hashes.shape[0], dtype=np.uint8)
    rc = secpdl().secpdl_ecdsa_verify_batch(hashes.shape[0], hashes.ctypes.data, sigs.ctypes.data, pubs.ctypes.data, publen, pubs.strides[0], out.ctypes.data)
    return out if rc == 0 else None


def libsecp_schnorr_verify_batch(msgs, xonly, sigs):
    import numpy as np
    out = np.zeros(msgs.shape[0], dtype=np.uint8)
    rc = secpdl().secpdl_schnorr_verify_batch(msgs.shape[0], msgs.ctypes.data, xonly.ctypes.data, sigs.ctypes.data, out.ctypes.data)
    return out if rc == 0 else None


def ossl_ecdsa_verify_rules_batch(hashes, sigs, pubs, publen, nthreads=1):
    """OpenSSL ECDSA_do_verify + libsecp256k1's range / low-S rules (leg C2: one thread; tests use more)"""
    import numpy as np
    out = np.zeros(hashes.shape[0], dtype=np.uint8)
    L = ossl()
    L.ossl_ecdsa_verify_rules_batch_mt.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                                   ctypes.c_void_p, ctypes.c_int]
    L.ossl_ecdsa_verify_rules_batch_mt.restype = None
    L.ossl_ecdsa_verify_rules_batch_mt(hashes.shape[0], hashes.ctypes.data, sigs.ctypes.data, pubs.ctypes.data, publen, pubs.strides[0], out.ctypes.data, nthreads)
    return out


def ossl_schnorr_verify_batch(msgs, xonly, sigs, nthreads=1):
    import numpy as np
    out = np.zeros(msgs.shape[0], dtype=np.uint8)
    L = ossl()
    L.ossl_schnorr_verify_batch_mt.argtypes = [ctypes.c_size_t] + [ctypes.c_void_p] * 4 + [ctypes.c_int]
    L.ossl_schnorr_verify_batch_mt.restype = None
    L.ossl_schnorr_verify_batch_mt(msgs.shape[0], msgs.ctypes.data, xonly.ctypes.data, sigs.ctypes.data, out.ctypes.data, nthreads)
    return out


def sha256(b):
    out = ctypes.create_string_buffer(32)
    lib().orc_sha256(bytes(b), len(b), out)
    return out.raw


def sha256d(b):
    out = ctypes.create_string_buffer(32)
    lib().orc_sha256d(bytes(b), len(b), out)
    return out.raw


def pubkey_parse(pub):
    out = ctypes.create_string_buffer(64)
    return out.raw if lib().orc_pubkey_parse(bytes(pub), len(pub), out) else None


def sig_parse_compact(sig64):
    return bool(lib().orc_sig_parse_compact(bytes(sig64)))


def sig_parse_der(der):
    out = ctypes.create_string_buffer(64)
    return out.raw if lib().orc_sig_parse_der(bytes(der), len(der), out) else None


def signature_from_der(der):
    out = ctypes.create_string_buffer(64)
    t = ctypes.c_int(0)
    if not lib().orc_signature_from_der(bytes(der), len(der), out, ctypes.byref(t)):
        return None
    return out.raw, t.value


def ecdsa_verify(hash32, sig64, pub):
    return bool(lib().orc_ecdsa_verify(bytes(hash32), bytes(sig64), bytes(pub), len(pub)))


def schnorr_verify(msg32, xonly32, sig64):
    return bool(lib().orc_schnorr_verify(bytes(msg32), bytes(xonly32), bytes(sig64)))


def sigcheck_channel_announcement(msg):
    return lib().orc_sigcheck_channel_announcement(bytes(msg), len(msg))


def sigcheck_channel_update(msg, node_id33):
    return lib().orc_sigcheck_channel_update(bytes(msg), len(msg), bytes(node_id33))


def sigcheck_node_announcement(msg):
    return lib().orc_sigcheck_node_announcement(bytes(msg), len(msg))


def pubkey_create(seckey32):
    out = ctypes.create_string_buffer(65)
    return out.raw if lib().orc_pubkey_create(bytes(seckey32), out) else None


def ecdsa_sign(hash32, seckey32, nonce32):
    out = ctypes.create_string_buffer(64)
    return out.raw if lib().orc_ecdsa_sign(bytes(hash32), bytes(seckey32), bytes(nonce32), out) else None


def schnorr_sign(msg32, seckey32, aux32=b"\x00" * 32):
    out = ctypes.create_string_buffer(64)
    return out.raw if lib().orc_schnorr_sign(bytes(msg32), bytes(seckey32), bytes(aux32), out) else None


def ecdsa_verify_batch(hashes, sigs, pubs, publen, nthreads=1):
    """numpy uint8 arrays [n,32], [n,64], [n,publen] (C-contiguous) -> uint8 [n]"""
    import numpy as np
    n = hashes.shape[0]
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_ecdsa_verify_batch(n, hashes.ctypes.data, sigs.ctypes.data, pubs.ctypes.data, publen,
                                 out.ctypes.data, nthreads)
    return out


def schnorr_verify_batch(msgs, xonly, sigs, nthreads=1):
    import numpy as np
    n = msgs.shape[0]
    out = np.zeros(n, dtype=np.uint8)
    lib().orc_schnorr_verify_batch(n, msgs.ctypes.data, xonly.ctypes.data, sigs.ctypes.data, out.ctypes.data, nthreads)
    return out


def sigcheck_gossip_batch(msgs, off, ids, nthreads=1):
    """msgs: uint8 blob, off: uint64 [n+1], ids: uint8 [n,33] -> int8 [n]"""
    import numpy as np
    n = off.shape[0] - 1
    out = np.zeros(n, dtype=np.int8)
    lib().orc_sigcheck_gossip_batch(n, msgs.ctypes.data, off.ctypes.data, ids.ctypes.data, out.ctypes.data, nthreads)
    return out


def ossl_ecdsa_verify(hash32, sig64, pub):
    return ossl().ossl_ecdsa_verify(bytes(hash32), bytes(sig64), bytes(pub), len(pub))


def ossl_schnorr_verify(msg32, xonly32, sig64):
    return ossl().ossl_schnorr_verify(bytes(msg32), bytes(xonly32), bytes(sig64))


def ossl_ecdsa_recover(hash32, sig64, recid):
    """compressed key, None where the library calls would fail"""
    out = ctypes.create_string_buffer(33)
    rc = ossl().ossl_ecdsa_recover(bytes(hash32), bytes(sig64), int(recid), out)
    assert rc >= 0
    return out.raw if rc == 1 else None


def ecdsa_recover(hash32, sig64, recid):
    """compressed key or None (C oracle)"""
    out = ctypes.create_string_buffer(33)
    return out.raw if lib().orc_ecdsa_recover(bytes(hash32), bytes(sig64), int(recid), out) else None


def ecdsa_recover_batch(hashes, sigs, recids, nthreads=1):
    """numpy uint8 [n,32], [n,64], [n] -> (keys uint8 [n,33], ok uint8 [n])"""
    import numpy as np
    n = hashes.shape[0]
    keys = np.zeros((n, 33), dtype=np.uint8)
    ok = np.zeros(n, dtype=np.uint8)
    recids = np.ascontiguousarray(recids, dtype=np.uint8)
    lib().orc_ecdsa_recover_batch(n, hashes.ctypes.data, sigs.ctypes.data, recids.ctypes.data, keys.ctypes.data, ok.ctypes.data, nthreads)
    return keys, ok


def gen_ecdsa_edge_batch(seed, n, publen, nthreads=1):
    """oracle/edge_gen.c: n signed rows covering every synthesised ECDSA edge class -> (hash [n,32], sig [n,64], pub [n,publen], cls [n], expect [n])"""
    import numpy as np
    L = lib()
    L.orc_gen_ecdsa_edge_batch.argtypes = [ctypes.c_uint64, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_void_p] * 5 + [ctypes.c_int]
    L.orc_gen_ecdsa_edge_batch.restype = None
    h, s, p = np.zeros((n, 32), np.uint8), np.zeros((n, 64), np.uint8), np.zeros((n, publen), np.uint8)
    c, e = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    L.orc_gen_ecdsa_edge_batch(seed, n, publen, h.ctypes.data, s.ctypes.data, p.ctypes.data, c.ctypes.data, e.ctypes.data, nthreads)
    return h, s, p, c, e


def gen_schnorr_edge_batch(seed, n, nthreads=1):
    """-> (msg [n,32], xonly [n,32], sig [n,64], cls [n], expect [n])"""
    import numpy as np
    L = lib()
    L.orc_gen_schnorr_edge_batch.argtypes = [ctypes.c_uint64, ctypes.c_size_t] + [ctypes.c_void_p] * 5 + [ctypes.c_int]
    L.orc_gen_schnorr_edge_batch.restype = None
    m, x, s = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), np.zeros((n, 64), np.uint8)
    c, e = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    L.orc_gen_schnorr_edge_batch(seed, n, m.ctypes.data, x.ctypes.data, s.ctypes.data, c.ctypes.data, e.ctypes.data, nthreads)
    return m, x, s, c, e


def edge_classes():
    L = lib()
    return L.orc_edge_nclasses_ecdsa(), L.orc_edge_nclasses_schnorr()
