"""Python front end over the C ABI (include/lightning_amd.h).  Device memory and streams come
from PyTorch-ROCm when the *_device methods are used; the numpy methods go through the
library's own staging.  Nothing here computes: every verdict comes from the HIP kernels."""
import ctypes

import numpy as np

from . import _ffi

ERRORS = {0: "OK", -1: "no usable gfx950 device", -2: "HIP error", -3: "bad argument", -4: "out of memory", -5: "bad state"}


class LamdError(RuntimeError):
    pass


def _u8(a, shape_tail):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1:
        a = a.reshape(-1, shape_tail)
    if a.shape[1] != shape_tail:
        raise ValueError("expected rows of %d bytes, got %r" % (shape_tail, a.shape))
    return a


class Engine:
    """One context = one GPU, one stream.  Mirrors lamd_init()/lamd_shutdown()."""

    def __init__(self, device=0):
        self._lib = _ffi.load()
        self._ctx = ctypes.c_void_p()
        rc = self._lib.lamd_init(ctypes.byref(self._ctx), device)
        if rc != 0:
            msg = self._lib.lamd_last_error(self._ctx).decode() if self._ctx else ""
            if self._ctx:
                self._lib.lamd_shutdown(self._ctx)
                self._ctx = ctypes.c_void_p()
            raise LamdError("lamd_init failed: %s %s" % (ERRORS.get(rc, rc), msg))
        self.device = device

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.lamd_shutdown(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise LamdError("%s: %s" % (ERRORS.get(rc, rc), self._lib.lamd_last_error(self._ctx).decode()))
        return rc

    # ---- host-buffer batches (numpy uint8 in, numpy bool out)
    def verify_ecdsa(self, hash32, sig64, pub):
        pub = np.ascontiguousarray(pub, dtype=np.uint8)
        if pub.ndim != 2 or pub.shape[1] not in (33, 65):
            raise ValueError("pub must be [n,33] or [n,65]")
        hash32, sig64 = _u8(hash32, 32), _u8(sig64, 64)
        n = hash32.shape[0]
        if not (sig64.shape[0] == n == pub.shape[0]):
            raise ValueError("row counts differ")
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_verify_ecdsa_batch(self._ctx, n, hash32.ctypes.data, sig64.ctypes.data, pub.ctypes.data,
                                                    pub.shape[1], pub.shape[1], ok.ctypes.data))
        return ok.astype(bool)

    def verify_schnorr(self, msg32, xonly32, sig64):
        msg32, xonly32, sig64 = _u8(msg32, 32), _u8(xonly32, 32), _u8(sig64, 64)
        n = msg32.shape[0]
        if not (xonly32.shape[0] == n == sig64.shape[0]):
            raise ValueError("row counts differ")
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_verify_schnorr_batch(self._ctx, n, msg32.ctypes.data, xonly32.ctypes.data, sig64.ctypes.data,
                                                      ok.ctypes.data))
        return ok.astype(bool)

    def check_tx_sig_batch(self, preimages, sighash_types, has_witness, sig64, pub):
        """preimages: list of bytes (BIP143 preimages); returns bool verdicts (gate + SHA256d + verify, all on the device)"""
        n = len(preimages)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(p) for p in preimages], dtype=np.uint64)
        blob = np.frombuffer(b"".join(preimages) + b"\x00", dtype=np.uint8)
        types = np.ascontiguousarray(sighash_types, dtype=np.uint8)
        wit = np.ascontiguousarray(has_witness, dtype=np.uint8)
        sig64 = _u8(sig64, 64)
        pub = np.ascontiguousarray(pub, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_check_tx_sig_batch(self._ctx, n, blob.ctypes.data, off.ctypes.data, types.ctypes.data, wit.ctypes.data,
                                                    sig64.ctypes.data, pub.ctypes.data, pub.shape[1], pub.shape[1], ok.ctypes.data))
        return ok.astype(bool)

    def check_tx_sig_tx_batch(self, txs, sig64, pub):
        """txs: list of dicts {version, locktime, inputs: [(txid32, vout, sequence)], outputs: [(amount, spk)], input_num, amount, script,
        sighash_type, has_witness}; the BIP143 sighash is computed on the device (lamd_check_tx_sig_tx_batch).  Returns bool verdicts."""
        n = len(txs)
        u32 = lambda k: np.array([t[k] for t in txs], dtype=np.uint32)
        inputs = b"".join(b"".join(bytes(i[0]) + int(i[1]).to_bytes(4, "little") + int(i[2]).to_bytes(4, "little") for i in t["inputs"]) for t in txs)

        def varint(v):
            return bytes([v]) if v < 0xfd else (b"\xfd" + v.to_bytes(2, "little") if v <= 0xffff else b"\xfe" + v.to_bytes(4, "little"))
        outs = [b"".join(int(a).to_bytes(8, "little") + varint(len(spk)) + bytes(spk) for a, spk in t["outputs"]) for t in txs]
        scripts = [bytes(t["script"]) for t in txs]
        off = lambda lens: np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        in_off, out_off, sc_off = off([len(t["inputs"]) for t in txs]), off([len(o) for o in outs]), off([len(x) for x in scripts])
        blob = lambda b: np.frombuffer(b + b"\x00", dtype=np.uint8)
        ib, ob, sb = blob(inputs), blob(b"".join(outs)), blob(b"".join(scripts))
        ver, lock, inum, nout = u32("version"), u32("locktime"), u32("input_num"), np.array([len(t["outputs"]) for t in txs], dtype=np.uint32)
        amt = np.array([t["amount"] for t in txs], dtype=np.uint64)
        types = np.array([t["sighash_type"] for t in txs], dtype=np.uint8)
        wit = np.array([1 if t["has_witness"] else 0 for t in txs], dtype=np.uint8)
        sig64 = _u8(sig64, 64)
        pub = np.ascontiguousarray(pub, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_check_tx_sig_tx_batch(self._ctx, n, ver.ctypes.data, lock.ctypes.data, ib.ctypes.data, in_off.ctypes.data, inum.ctypes.data,
                                                       amt.ctypes.data, ob.ctypes.data, out_off.ctypes.data, nout.ctypes.data, sb.ctypes.data,
                                                       sc_off.ctypes.data, types.ctypes.data, wit.ctypes.data, sig64.ctypes.data, pub.ctypes.data,
                                                       pub.shape[1], pub.shape[1], ok.ctypes.data))
        return ok.astype(bool)

    def commitment_call(self, commit_tx, remote_funding33, commit_sig64, commit_sighash_type, htlc_txs, remote_htlckey33, htlc_sigs64, htlc_sighash_types):
        """the arguments of check_commitment_signed() marshalled ONCE; returns a function that makes the C call and nothing else -> (first_bad, ok_rows).
        (bench.py times this: flattening 484 templates in Python costs more than the call.)"""
        def varint(v):
            return bytes([v]) if v < 0xfd else (b"\xfd" + v.to_bytes(2, "little") if v <= 0xffff else b"\xfe" + v.to_bytes(4, "little"))
        keep = []

        def tmpl(t):
            ins = b"".join(bytes(i[0]) + int(i[1]).to_bytes(4, "little") + int(i[2]).to_bytes(4, "little") for i in t["inputs"])
            outs = b"".join(int(a).to_bytes(8, "little") + varint(len(spk)) + bytes(spk) for a, spk in t["outputs"])
            bufs = [ctypes.create_string_buffer(x, len(x) + 1) for x in (ins, outs, bytes(t["script"]))]
            keep.extend(bufs)
            return _ffi.LamdTxTemplate(t["version"], t["locktime"], ctypes.addressof(bufs[0]), len(t["inputs"]), t.get("input_num", 0), t["amount"],
                                       ctypes.addressof(bufs[1]), len(outs), len(t["outputs"]), ctypes.addressof(bufs[2]), len(t["script"]))
        ct = tmpl(commit_tx)
        n = len(htlc_txs)
        arr = (_ffi.LamdTxTemplate * max(1, n))(*[tmpl(t) for t in htlc_txs])
        sigs = np.ascontiguousarray(np.frombuffer(b"".join(bytes(x) for x in htlc_sigs64) + b"\x00", dtype=np.uint8))
        types = np.array(list(htlc_sighash_types) + [0], dtype=np.uint8)
        first_bad = ctypes.c_int64(0)
        ok = np.zeros(1 + n, dtype=np.uint8)
        fk, hk, cs = bytes(remote_funding33), bytes(remote_htlckey33), bytes(commit_sig64)
        keep += [ct, arr, sigs, types, fk, hk, cs]

        def call():
            self._chk(self._lib.lamd_check_commitment_signed(self._ctx, ctypes.addressof(ct), fk, cs, commit_sighash_type, n, ctypes.addressof(arr), hk,
                                                             sigs.ctypes.data, types.ctypes.data, ctypes.byref(first_bad), ok.ctypes.data))
            return int(first_bad.value), ok.astype(bool)
        call.keep = keep
        return call

    def check_commitment_signed(self, commit_tx, remote_funding33, commit_sig64, commit_sighash_type, htlc_txs, remote_htlckey33, htlc_sigs64,
                                htlc_sighash_types):
        """ONE commitment_signed (channeld/channeld.c:2171-2232) through lamd_check_commitment_signed.  commit_tx / htlc_txs[i]: dicts {version,
        locktime, inputs: [(txid32, vout, sequence)], outputs: [(amount, spk)], input_num, amount, script}.  Returns (first_bad, ok_rows):
        first_bad = -1 all good, 0 the commitment signature, 1 + i htlc_sigs[i] -- the first failure in the reference's order."""
        return self.commitment_call(commit_tx, remote_funding33, commit_sig64, commit_sighash_type, htlc_txs, remote_htlckey33, htlc_sigs64,
                                    htlc_sighash_types)()

    @staticmethod
    def _streams(streams):
        off = np.concatenate([[0], np.cumsum([len(x) for x in streams])]).astype(np.uint64)
        return np.frombuffer(b"".join(bytes(x) for x in streams) + b"\x00", dtype=np.uint8), off

    def bolt12_check_signature_batch(self, streams, messagename, fieldname, key33, sig64):
        """streams: list of serialised TLV streams; key33 uint8 [n,33]; sig64 uint8 [n,64] -> bool verdicts (lamd_bolt12_check_signature_batch)"""
        blob, off = self._streams(streams)
        key33, sig64 = _u8(key33, 33), _u8(sig64, 64)
        ok = np.zeros(len(streams), dtype=np.uint8)
        self._chk(self._lib.lamd_bolt12_check_signature_batch(self._ctx, len(streams), blob.ctypes.data, off.ctypes.data, messagename, fieldname,
                                                              key33.ctypes.data, 33, sig64.ctypes.data, ok.ctypes.data))
        return ok.astype(bool)

    def bolt12_merkle_batch(self, streams, messagename, fieldname):
        """-> (merkle roots uint8 [n,32], signature hashes uint8 [n,32], ok bool [n])"""
        blob, off = self._streams(streams)
        n = len(streams)
        mk, sh, ok = np.zeros((n, 32), dtype=np.uint8), np.zeros((n, 32), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_bolt12_merkle_batch(self._ctx, n, blob.ctypes.data, off.ctypes.data, messagename, fieldname, mk.ctypes.data,
                                                     sh.ctypes.data, ok.ctypes.data))
        return mk, sh, ok.astype(bool)

    def ecdsa_recover(self, hash32, sig64, recid):
        """numpy uint8 [n,32], [n,64], [n] -> (keys uint8 [n,33], ok bool [n]); secp256k1_ecdsa_recover semantics"""
        hash32, sig64 = _u8(hash32, 32), _u8(sig64, 64)
        recid = np.ascontiguousarray(recid, dtype=np.uint8)
        n = hash32.shape[0]
        keys = np.zeros((n, 33), dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        if n:
            self._chk(self._lib.lamd_ecdsa_recover_batch(self._ctx, n, hash32.ctypes.data, sig64.ctypes.data, recid.ctypes.data,
                                                         keys.ctypes.data, ok.ctypes.data))
        return keys, ok.astype(bool)

    def _after_torch(self):
        """order the next submission after what torch's current stream holds (the tensors handed to the *_device methods are
        usually produced there); a device-side event, no host synchronisation.  The caller still has to keep the tensors alive
        until the results are complete."""
        if not getattr(self, "auto_order", True):
            return
        try:
            import torch
            self._chk(self._lib.lamd_wait_stream(self._ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        except ImportError:
            pass

    def ecdsa_recover_device(self, d_hash32, d_sig64, d_recid, d_pub33, d_ok):
        self._after_torch()
        self._chk(self._lib.lamd_ecdsa_recover_batch_device(self._ctx, d_hash32.shape[0], d_hash32.data_ptr(), d_sig64.data_ptr(),
                                                            d_recid.data_ptr(), d_pub33.data_ptr(), d_ok.data_ptr()))

    def grind_htlc_tx_fee(self, preimage, outputs, input_sat, weight, min_feerate, max_feerate, sig64, sighash_type, has_witness, pub33):
        """grind_htlc_tx_fee (onchaind/onchaind.c:388-438) on the device: (feerate, fee) of the lowest matching feerate, or None"""
        rate, fee = ctypes.c_uint32(0), ctypes.c_uint64(0)
        pre, outs, sig, key = bytes(preimage), bytes(outputs), bytes(sig64), bytes(pub33)
        rc = self._chk(self._lib.lamd_grind_htlc_tx_fee(self._ctx, pre, len(pre), outs, len(outs), int(input_sat), int(weight), int(min_feerate),
                                                         int(max_feerate), sig, int(sighash_type), int(bool(has_witness)), key,
                                                         ctypes.byref(rate), ctypes.byref(fee)))
        return (rate.value, fee.value) if rc == 1 else None

    def pubkey_parse(self, pub):
        pub = np.ascontiguousarray(pub, dtype=np.uint8)
        n, ln = pub.shape
        out = np.zeros((n, 64), dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(self._lib.lamd_pubkey_parse_batch(self._ctx, n, pub.ctypes.data, ln, ln, out.ctypes.data, ok.ctypes.data))
        return out, ok.astype(bool)

    def sigcheck_gossip(self, msgs, node_ids=None):
        """msgs: list of bytes (raw wire messages).  node_ids: list of 33-byte ids (or None entries), needed for
        channel_update.  Returns int8 verdicts (see lamd_sigcheck_gossip_batch)."""
        n = len(msgs)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
        blob = np.frombuffer(b"".join(msgs) + b"\x00", dtype=np.uint8)
        ids_ptr = None
        if node_ids is not None:
            ids = np.zeros((n, 33), dtype=np.uint8)
            for i, k in enumerate(node_ids):
                if k is not None:
                    ids[i] = np.frombuffer(k, dtype=np.uint8)
            ids_ptr = ids.ctypes.data
        verdict = np.zeros(n, dtype=np.int8)
        self._chk(self._lib.lamd_sigcheck_gossip_batch(self._ctx, n, blob.ctypes.data, off.ctypes.data, ids_ptr, verdict.ctypes.data))
        return verdict

    # ---- single-item veneers (reference semantics)
    def check_signed_hash(self, hash32, sig64, pubkey):
        return bool(self._chk(self._lib.lamd_check_signed_hash(self._ctx, bytes(hash32), bytes(sig64), bytes(pubkey), len(pubkey))))

    def check_signed_hash_nodeid(self, hash32, sig64, node_id33):
        return bool(self._chk(self._lib.lamd_check_signed_hash_nodeid(self._ctx, bytes(hash32), bytes(sig64), bytes(node_id33))))

    def check_schnorr_sig(self, hash32, pubkey33, sig64):
        return bool(self._chk(self._lib.lamd_check_schnorr_sig(self._ctx, bytes(hash32), bytes(pubkey33), bytes(sig64))))

    # ---- streaming
    def queue_ecdsa(self, hash32, sig64, pubkey):
        return self._chk(self._lib.lamd_queue_ecdsa(self._ctx, bytes(hash32), bytes(sig64), bytes(pubkey), len(pubkey)))

    def queue_schnorr(self, msg32, xonly32, sig64):
        return self._chk(self._lib.lamd_queue_schnorr(self._ctx, bytes(msg32), bytes(xonly32), bytes(sig64)))

    def queue_ecdsa_batch(self, hash32, sig64, pub):
        """numpy uint8 [n,32], [n,64], [n,33|65]: first ticket"""
        hash32, sig64 = _u8(hash32, 32), _u8(sig64, 64)
        pub = np.ascontiguousarray(pub, dtype=np.uint8)
        return self._chk(self._lib.lamd_queue_ecdsa_batch(self._ctx, hash32.shape[0], hash32.ctypes.data, sig64.ctypes.data, pub.ctypes.data,
                                                          pub.shape[1], pub.shape[1]))

    def queue_schnorr_batch(self, msg32, xonly32, sig64):
        msg32, xonly32, sig64 = _u8(msg32, 32), _u8(xonly32, 32), _u8(sig64, 64)
        return self._chk(self._lib.lamd_queue_schnorr_batch(self._ctx, msg32.shape[0], msg32.ctypes.data, xonly32.ctypes.data, sig64.ctypes.data))

    def queue_reserve(self, n, keylen):
        """zero-copy producer form: (first ticket, hash/msg [n,32], sig [n,64], key [n,keylen]) -- numpy views of the pinned staging
        set, to be filled before flush() and not kept beyond the next queue_* / flush call"""
        ptr = [ctypes.c_void_p() for _ in range(3)]
        first = self._chk(self._lib.lamd_queue_reserve(self._ctx, int(n), int(keylen), *[ctypes.byref(p) for p in ptr]))
        views = [np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(n), w)) for p, w in zip(ptr, (32, 64, int(keylen)))]
        return (first, *views)

    def queue_ecdsa_batch_inplace(self, hash32, sig64, pub):
        """the rows are NOT copied: C-contiguous numpy uint8 [n,32], [n,64], [n,33|65] that the caller keeps alive and unchanged until the flush
        carrying them has been collected (wait / poll); pin them with host_register() for DMA.  First ticket."""
        for a, w in ((hash32, 32), (sig64, 64)):
            if a.dtype != np.uint8 or not a.flags.c_contiguous or a.ndim != 2 or a.shape[1] != w:
                raise ValueError("in-place rows must be C-contiguous uint8 [n,%d]" % w)
        if pub.dtype != np.uint8 or not pub.flags.c_contiguous or pub.ndim != 2 or pub.shape[0] != hash32.shape[0] or sig64.shape[0] != hash32.shape[0]:
            raise ValueError("in-place keys must be C-contiguous uint8 [n,33|65]")
        return self._chk(self._lib.lamd_queue_ecdsa_batch_inplace(self._ctx, hash32.shape[0], hash32.ctypes.data, sig64.ctypes.data, pub.ctypes.data, pub.shape[1]))

    def queue_schnorr_batch_inplace(self, msg32, xonly32, sig64):
        for a, w in ((msg32, 32), (xonly32, 32), (sig64, 64)):
            if a.dtype != np.uint8 or not a.flags.c_contiguous or a.ndim != 2 or a.shape[1] != w or a.shape[0] != msg32.shape[0]:
                raise ValueError("in-place rows must be C-contiguous uint8 [n,%d]" % w)
        return self._chk(self._lib.lamd_queue_schnorr_batch_inplace(self._ctx, msg32.shape[0], msg32.ctypes.data, xonly32.ctypes.data, sig64.ctypes.data))

    def host_register(self, arr):
        """pins a numpy array's memory for every device (rows queued in place leave it by DMA); False when the runtime refuses the range"""
        return self._lib.lamd_host_register(self._ctx, arr.ctypes.data, arr.nbytes) == 0

    def host_unregister(self, arr):
        return self._lib.lamd_host_unregister(self._ctx, arr.ctypes.data) == 0

    def flush(self):
        self._chk(self._lib.lamd_flush(self._ctx))

    def wait(self, cap=1 << 20):
        ok = np.zeros(cap, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        self._chk(self._lib.lamd_wait(self._ctx, ok.ctypes.data, cap, ctypes.byref(n)))
        return ok[:n.value].astype(bool)

    def poll(self, cap=1 << 20):
        ok = np.zeros(cap, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        rc = self._chk(self._lib.lamd_poll(self._ctx, ok.ctypes.data, cap, ctypes.byref(n)))
        return ok[:n.value].astype(bool) if rc == 1 else None

    # ---- device-resident API (torch uint8 CUDA tensors; asynchronous on self.stream_ptr)
    def verify_ecdsa_device(self, d_hash, d_sig, d_pub, d_ok):
        n, publen = d_pub.shape
        self._after_torch()
        self._chk(self._lib.lamd_verify_ecdsa_batch_device(self._ctx, n, d_hash.data_ptr(), d_sig.data_ptr(), d_pub.data_ptr(),
                                                           publen, publen, d_ok.data_ptr()))

    def verify_schnorr_device(self, d_msg, d_xonly, d_sig, d_ok):
        n = d_msg.shape[0]
        self._after_torch()
        self._chk(self._lib.lamd_verify_schnorr_batch_device(self._ctx, n, d_msg.data_ptr(), d_xonly.data_ptr(), d_sig.data_ptr(),
                                                             d_ok.data_ptr()))

    def gen_ecdsa_device(self, seed, nkeys, d_hash, d_sig, d_pub, group=0):
        n, publen = d_pub.shape
        self._after_torch()   # the output tensors may still be being filled on torch's stream (torch.zeros)
        self._chk(_ffi.load_testgen().lamd_gen_ecdsa_device(self._ctx, n, seed, nkeys, group, publen, d_hash.data_ptr(), d_sig.data_ptr(), d_pub.data_ptr()))

    def gen_schnorr_device(self, seed, nkeys, d_msg, d_xonly, d_sig, group=0):
        n = d_msg.shape[0]
        self._after_torch()
        self._chk(_ffi.load_testgen().lamd_gen_schnorr_device(self._ctx, n, seed, nkeys, group, d_msg.data_ptr(), d_xonly.data_ptr(), d_sig.data_ptr()))

    def gen_gossip_device(self, seed, n_cann, n_cupd, n_nodes, d_msgs, d_ids):
        self._after_torch()
        self._chk(_ffi.load_testgen().lamd_gen_gossip_device(self._ctx, n_cann, n_cupd, seed, n_nodes, d_msgs.data_ptr(), d_ids.data_ptr()))

    def sigcheck_gossip_device(self, n, d_msgs, d_off, d_ids, d_rowbase, rows, d_verdict):
        self._after_torch()
        self._chk(self._lib.lamd_sigcheck_gossip_batch_device(self._ctx, n, d_msgs.data_ptr(), d_off.data_ptr(),
                                                              d_ids.data_ptr() if d_ids is not None else None, d_rowbase.data_ptr(), rows,
                                                              d_verdict.data_ptr()))

    def sigcheck_gossip_spans_device(self, n, d_msgs, d_start, d_len, d_ids, d_rowbase, rows, d_verdict):
        """lamd_sigcheck_gossip_spans_device: message i = d_msgs[d_start[i] : d_start[i] + d_len[i]]; ids / rowbase / verdicts index the selection"""
        self._after_torch()
        self._chk(self._lib.lamd_sigcheck_gossip_spans_device(self._ctx, n, d_msgs.data_ptr(), d_start.data_ptr(), d_len.data_ptr(),
                                                              d_ids.data_ptr() if d_ids is not None else None, d_rowbase.data_ptr(), rows,
                                                              d_verdict.data_ptr()))

    def selftest(self, hash32, sig64, pub33):
        buf = ctypes.create_string_buffer(4096)
        rc = self._chk(self._lib.lamd_selftest(self._ctx, bytes(hash32), bytes(sig64), bytes(pub33), buf, 4096))
        return rc, buf.value.decode()

    def chain_debug(self, use_mul):
        buf = ctypes.create_string_buffer(8192)
        rc = self._chk(self._lib.lamd_chain_debug(self._ctx, int(use_mul), buf, 8192))
        return rc, buf.value.decode()

    def inv_debug(self):
        buf = ctypes.create_string_buffer(8192)
        rc = self._chk(self._lib.lamd_inv_debug(self._ctx, buf, 8192))
        return rc, buf.value.decode()

    def fuzz_field(self, lanes=16384, iters=64, seed=1):
        """(mismatching lanes, field operations executed on the device, report)"""
        buf = ctypes.create_string_buffer(4096)
        ops = ctypes.c_uint64(0)
        rc = self._chk(self._lib.lamd_fuzz_field(self._ctx, lanes, iters, seed, ctypes.byref(ops), buf, 4096))
        return rc, ops.value, buf.value.decode()

    def synchronize(self):
        self._chk(self._lib.lamd_synchronize(self._ctx))

    @property
    def stream_ptr(self):
        return self._lib.lamd_stream(self._ctx)

    def set_timing(self, on=True):
        self._chk(self._lib.lamd_set_timing(self._ctx, int(on)))

    def set_ecmult_chain(self, on):
        """large table-driven ecmult launches one after the other (True) or overlapping (False, default): lamd_set_ecmult_chain"""
        self._chk(self._lib.lamd_set_ecmult_chain(self._ctx, int(on)))

    def stream_wait_results(self, stream_ptr):
        """make a caller's HIP stream wait (on the device) for every verification submitted so far"""
        self._chk(self._lib.lamd_stream_wait_results(self._ctx, ctypes.c_void_p(stream_ptr)))

    def wait_event(self, event_ptr):
        """verification submitted from now on waits (on the device) for a hipEvent_t the caller recorded"""
        self._chk(self._lib.lamd_wait_event(self._ctx, ctypes.c_void_p(event_ptr)))

    def results_mark(self, slot):
        """remember "everything submitted so far" in slot 0..3 (no waiting); see stream_wait_mark"""
        self._chk(self._lib.lamd_results_mark(self._ctx, int(slot)))

    def results_mark_last(self, slot):
        """lamd_results_mark_last: one event, on the lane of the call submitted last"""
        self._chk(self._lib.lamd_results_mark_last(self._ctx, slot))

    def stream_wait_mark(self, slot, stream_ptr):
        """make a caller's HIP stream wait (on the device) for the work remembered by results_mark(slot)"""
        self._chk(self._lib.lamd_stream_wait_mark(self._ctx, int(slot), ctypes.c_void_p(stream_ptr)))

    def wait_stream(self, stream_ptr):
        """verification submitted from now on waits (on the device) for what the caller's stream holds now"""
        self._chk(self._lib.lamd_wait_stream(self._ctx, ctypes.c_void_p(stream_ptr)))

    def info(self, lane=None):
        inf = _ffi.LamdInfo()
        if lane is None:
            self._chk(self._lib.lamd_get_info(self._ctx, ctypes.byref(inf)))
        else:
            self._chk(self._lib.lamd_get_lane_info(self._ctx, int(lane), ctypes.byref(inf)))
        return dict(device=inf.device, compute_units=inf.compute_units, arch=inf.arch.decode(), gtable_bytes=inf.gtable_bytes,
                    last_kernel_ms=list(inf.last_kernel_ms), last_unique_keys=inf.last_unique_keys, last_hot_rows=inf.last_hot_rows, last_keyed=int(inf.last_keyed), last_mode=int(inf.last_mode), lanes=int(inf.lanes),
                    last_cache_hits=inf.last_cache_hits, last_cold_rows=inf.last_cold_rows, last_new_tables=inf.last_new_tables,
                    last_suspect_rows=inf.last_suspect_rows, cache_enabled=bool(inf.cache_enabled), cache_entries=inf.cache_entries,
                    cache_capacity=inf.cache_capacity, cache_resets=inf.cache_resets,
                    keyed_ecmult_ms_sum=list(inf.keyed_ecmult_ms_sum), keyed_ecmult_launches=list(inf.keyed_ecmult_launches), hw_queues_env=int(inf.hw_queues_env), queue_sets=int(inf.queue_sets))

    def set_chunk_rows(self, rows):
        """lamd_set_chunk_rows: rows per launch sequence of the calls that follow (0 = default 2^22)"""
        self._chk(self._lib.lamd_set_chunk_rows(self._ctx, rows))

    def mul32_peak(self, waves_per_simd=3, min_ms=4.0, launches=5):
        """lamd_debug_mul32_peak: sustained v_mad_u64_u32 rate of the chip (lane-ops/s), average launch ms, s_memtime / s_memrealtime ratio"""
        r, ms, ck = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._chk(self._lib.lamd_debug_mul32_peak(self._ctx, waves_per_simd, min_ms, launches, ctypes.byref(r), ctypes.byref(ms), ctypes.byref(ck)))
        return r.value, ms.value, ck.value

    def cache_clear(self):
        """empty the key-table cache (cold-path measurements)"""
        self._chk(self._lib.lamd_cache_clear(self._ctx))
