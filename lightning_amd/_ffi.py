"""ctypes binding of the C ABI declared in include/lightning_amd.h (one line per exported symbol)."""
import ctypes
import os

from . import _build

c_u8p = ctypes.c_void_p
c_sz = ctypes.c_size_t


class LamdTxTemplate(ctypes.Structure):
    """lamd_tx_template (include/lightning_amd.h): what check_tx_sig() reads of one (transaction, input)"""
    _fields_ = [("version", ctypes.c_uint32), ("locktime", ctypes.c_uint32), ("inputs40", ctypes.c_void_p), ("n_inputs", ctypes.c_uint32),
                ("input_num", ctypes.c_uint32), ("amount_sat", ctypes.c_uint64), ("outputs", ctypes.c_void_p), ("outputs_len", ctypes.c_uint64),
                ("n_outputs", ctypes.c_uint32), ("script", ctypes.c_void_p), ("script_len", ctypes.c_uint64)]


class LamdInfo(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("compute_units", ctypes.c_int), ("arch", ctypes.c_char * 64),
                ("gtable_bytes", ctypes.c_size_t), ("last_kernel_ms", ctypes.c_double * 4), ("last_unique_keys", ctypes.c_size_t),
                ("last_hot_rows", ctypes.c_size_t), ("last_keyed", ctypes.c_int), ("last_mode", ctypes.c_int), ("lanes", ctypes.c_int),
                ("last_cache_hits", ctypes.c_size_t), ("last_cold_rows", ctypes.c_size_t), ("last_new_tables", ctypes.c_size_t),
                ("last_suspect_rows", ctypes.c_size_t), ("cache_enabled", ctypes.c_int), ("cache_entries", ctypes.c_size_t),
                ("cache_capacity", ctypes.c_size_t), ("cache_resets", ctypes.c_size_t),
                ("keyed_ecmult_ms_sum", ctypes.c_double * 2), ("keyed_ecmult_launches", ctypes.c_size_t * 2), ("hw_queues_env", ctypes.c_int), ("queue_sets", ctypes.c_int)]


# name -> (restype, argtypes); every symbol of include/lightning_amd.h and include/lightning_amd_debug.h
SYMBOLS = {
    "lamd_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "lamd_shutdown": (None, [ctypes.c_void_p]),
    "lamd_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "lamd_version": (ctypes.c_char_p, []),
    "lamd_verify_ecdsa_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz, c_sz, c_u8p]),
    "lamd_verify_schnorr_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_verify_ecdsa_batch_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz, c_sz, c_u8p]),
    "lamd_verify_schnorr_batch_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_stream": (ctypes.c_void_p, [ctypes.c_void_p]),
    "lamd_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "lamd_check_signed_hash": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p, c_sz]),
    "lamd_check_signed_hash_nodeid": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p]),
    "lamd_check_schnorr_sig": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p]),
    "lamd_check_tx_sig_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_sz, c_sz, c_u8p]),
    "lamd_check_tx_sig_tx_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz] + [c_u8p] * 15 + [c_sz, c_sz, c_u8p]),
    "lamd_check_commitment_signed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_u8p, c_u8p, ctypes.c_uint8, c_sz, ctypes.c_void_p, c_u8p, c_u8p, c_u8p,
                                                    ctypes.POINTER(ctypes.c_int64), c_u8p]),
    "lamd_bolt12_check_signature_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, ctypes.c_char_p, ctypes.c_char_p, c_u8p, c_sz, c_u8p, c_u8p]),
    "lamd_bolt12_merkle_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, ctypes.c_char_p, ctypes.c_char_p, c_u8p, c_u8p, c_u8p]),
    "lamd_ecdsa_recover_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_ecdsa_recover_batch_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_grind_htlc_tx_fee": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_sz, c_u8p, c_sz, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                              ctypes.c_uint32, c_u8p, ctypes.c_uint8, ctypes.c_int, c_u8p, ctypes.POINTER(ctypes.c_uint32),
                                              ctypes.POINTER(ctypes.c_uint64)]),
    "lamd_pubkey_parse_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_sz, c_sz, c_u8p, c_u8p]),
    "lamd_sigcheck_gossip_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_queue_ecdsa": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p, c_sz]),
    "lamd_queue_schnorr": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p]),
    "lamd_queue_ecdsa_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz, c_sz]),
    "lamd_queue_schnorr_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p]),
    "lamd_flush": (ctypes.c_int, [ctypes.c_void_p]),
    "lamd_poll": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_sz, ctypes.POINTER(c_sz)]),
    "lamd_wait": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_sz, ctypes.POINTER(c_sz)]),
    "lamd_sigcheck_gossip_batch_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p, c_sz, c_u8p]),
    "lamd_sigcheck_gossip_spans_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p, c_u8p, c_sz, c_u8p]),
    "lamd_selftest": (ctypes.c_int, [ctypes.c_void_p, c_u8p, c_u8p, c_u8p, ctypes.c_char_p, c_sz]),
    "lamd_chain_debug": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, c_sz]),
    "lamd_inv_debug": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_sz]),
    "lamd_x2_debug": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_sz]),
    "lamd_debug_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_sz, c_sz, c_u8p]),
    "lamd_debug_mul32_peak": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int] + [ctypes.POINTER(ctypes.c_double)] * 3),
    "lamd_debug_gtable": (ctypes.c_void_p, [ctypes.c_void_p]),
    "lamd_fuzz_field": (ctypes.c_int, [ctypes.c_void_p, c_sz, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, c_sz]),
    "lamd_get_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(LamdInfo)]),
    "lamd_set_timing": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "lamd_set_ecmult_chain": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "lamd_set_chunk_rows": (ctypes.c_int, [ctypes.c_void_p, c_sz]),
    "lamd_cache_clear": (ctypes.c_int, [ctypes.c_void_p]),
    "lamd_get_lane_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(LamdInfo)]),
    "lamd_stream_wait_results": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "lamd_wait_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "lamd_wait_event": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "lamd_queue_reserve": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_sz] + [ctypes.POINTER(ctypes.c_void_p)] * 3),
    "lamd_queue_ecdsa_batch_inplace": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz]),
    "lamd_queue_schnorr_batch_inplace": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p]),
    "lamd_host_register": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_sz]),
    "lamd_host_unregister": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "lamd_device_numa_node": (ctypes.c_int, [ctypes.c_int]),
    "lamd_results_mark": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "lamd_results_mark_last": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "lamd_stream_wait_mark": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    # several devices behind one process (lamd_multi.cpp)
    "lamd_multi_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "lamd_multi_init_backend": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_void_p]),
    "lamd_multi_shutdown": (None, [ctypes.c_void_p]),
    "lamd_multi_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "lamd_multi_devices": (ctypes.c_int, [ctypes.c_void_p]),
    "lamd_multi_ctx": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "lamd_multi_verify_ecdsa_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz, c_sz, c_sz, c_u8p]),
    "lamd_multi_verify_schnorr_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_sz, c_u8p]),
    "lamd_multi_sigcheck_gossip_batch": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "lamd_shard_bounds": (ctypes.c_int, [c_sz, c_u8p, ctypes.c_int, c_u8p, c_u8p]),
}

# include/lightning_amd_testgen.h -> liblightning_amd_testgen.so (test / bench infrastructure: the signer kernels)
TESTGEN_SYMBOLS = {
    "lamd_gen_ecdsa_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, ctypes.c_uint64, c_sz, c_sz, c_sz, c_u8p, c_u8p, c_u8p]),
    "lamd_gen_schnorr_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, ctypes.c_uint64, c_sz, c_sz, c_u8p, c_u8p, c_u8p]),
    "lamd_gen_gossip_device": (ctypes.c_int, [ctypes.c_void_p, c_sz, c_sz, ctypes.c_uint64, c_sz, c_u8p, c_u8p]),
}

_lib = None
_testgen = None


def load(build_if_needed=True):
    """dlopen liblightning_amd.so (building it first if the sources are newer).  Raises if it is
    missing: there is no fallback implementation."""
    global _lib
    if _lib is None:
        alt = os.environ.get("LAMD_LIB_PATH")   # experiments only (tools/variants/*.so built with other constants): same ABI, no fallback either
        if alt:
            L = ctypes.CDLL(alt)
            for name, (res, args) in SYMBOLS.items():
                f = getattr(L, name)
                f.restype = res
                f.argtypes = args
            _lib = L
            return _lib
        if build_if_needed and os.path.exists("/opt/rocm/bin/hipcc"):
            _build.build()
        if not os.path.exists(_build.LIB):
            raise RuntimeError("liblightning_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "lightning_amd has no CPU fallback")
        L = ctypes.CDLL(_build.LIB)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def load_testgen():
    """dlopen liblightning_amd_testgen.so (synthetic signed workloads for tests/ and bench.py; not part of the product)"""
    global _testgen
    if _testgen is None:
        load()
        alt = os.environ.get("LAMD_TESTGEN_LIB_PATH")   # experiments only: a variant engine built with another table layout needs its own signer library
        path = alt if alt else (_build.build_testgen() if os.path.exists("/opt/rocm/bin/hipcc") else _build.TESTGEN)
        L = ctypes.CDLL(path)
        for name, (res, args) in TESTGEN_SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _testgen = L
    return _testgen
