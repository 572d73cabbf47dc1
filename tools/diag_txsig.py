#!/usr/bin/env python3
"""Diagnostic: the transaction-template check_tx_sig path, first through the C ABI (Engine), then through the reference-named mirror."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
from lightning_amd import Engine, _build  # noqa: E402

H = bytes.fromhex
kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))
rawtx = H("0200000001e1ebca08cf1c301ac563580a1126d5c8fcb0e5e2043230b852c726553caf1e1d0000000000000000000160ae0a0000000000"
          "22002082e03c5a9cb79c82cd5a0572dc175290bc044609aabe9cc852d61927436041796d000000")
wscript = H("76a914a8c40c334351dbe8e5908544f1c98fbcfb8719fc8763ac6721038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de05"
            "4432eb7f7c820120876475527c2103cf8e2f193a6aed60db80af75f3c8d59c2de735b299b7c7083527be9bd23b77a852ae67a914b8bcd51e"
            "fa35be1e50ae2d5f72f4500acb005c9c88ac6868")
ko = next(v for v in kat["der"] if v["name"] == "KAT-O")
print("step 1: engine path", flush=True)
with Engine(0) as eng:
    txs = [dict(version=2, locktime=109, inputs=[(rawtx[5:37], 0, 0)], outputs=[(700000 - fee, rawtx[56:90])], input_num=0, amount=700000,
                script=wscript, sighash_type=1, has_witness=True) for fee in (165749, 165750, 165751, 0)]
    sig = np.frombuffer(H(ko["sig64"]) if "sig64" in ko else H(ko["expect_sig"]) if "expect_sig" in ko else bytes(64), dtype=np.uint8)
    print("kat keys", list(ko.keys()), flush=True)
    pub = np.frombuffer(H("038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de054432eb7f"), dtype=np.uint8)
    got = eng.check_tx_sig_tx_batch(txs, np.tile(sig, (4, 1)), np.tile(pub, (4, 1)))
    print("engine verdicts", got, flush=True)
print("step 2: shim path", flush=True)
import test_cln_shim as T  # noqa: E402
L = ctypes.CDLL(_build.build_shim())
for n in ("check_tx_sig", "signature_from_der", "pubkey_from_der"):
    getattr(L, n).restype = ctypes.c_bool
L.shim_tal_dup.restype = ctypes.c_void_p
L.shim_tal_dup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
L.check_tx_sig.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
sig = T.BitcoinSig()
der = H(ko["der"])
assert L.signature_from_der(der, len(der), ctypes.byref(sig))
key = T.Pubkey()
assert L.pubkey_from_der(H("038ffd2621647812011960152bfb79c5a2787dfe6c4f37e2222547de054432eb7f"), 33, ctypes.byref(key))
ws = L.shim_tal_dup(None, wscript, len(wscript))
print("ws", hex(ws), flush=True)
for fee in (165749, 165750, 165751, 0):
    tx, keep = T.make_tx(L, 2, 109, [(rawtx[5:37], 0, 0, 700000)], [(700000 - fee, rawtx[56:90])])
    print("calling check_tx_sig fee", fee, flush=True)
    print(L.check_tx_sig(ctypes.byref(tx), 0, None, ws, ctypes.byref(key), ctypes.byref(sig)), flush=True)
