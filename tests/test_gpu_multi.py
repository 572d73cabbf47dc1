"""lamd_multi_* with the ENGINE back end (HIP + RCCL) on the devices this box has: one engine context + one host thread per device, the shards'
copies and kernels, the RCCL all-gather of the verdict bytes on the devices (with ONE device: a one-rank communicator -- the same code path,
collective included), one D2H.  Verdicts must equal the oracle's and the single-device calls'.  tests/test_multi_stub.py runs the same layer at
8 stub devices on the CPU."""
import ctypes

import numpy as np
import pytest

H = bytes.fromhex


@pytest.mark.gpu
def test_multi_engine_backend_on_the_local_devices(orc, kat):
    import torch
    from lightning_amd import _ffi
    lib = _ffi.load()
    ndev = min(torch.cuda.device_count(), 8)
    m = ctypes.c_void_p()
    rc = lib.lamd_multi_init(ctypes.byref(m), None, ndev)
    assert rc == 0, lib.lamd_multi_last_error(m)
    try:
        assert lib.lamd_multi_devices(m) == ndev and lib.lamd_multi_ctx(m, 0)
        for n, group, publen in ((200_003, 484, 33), (65_536, 1, 65), (484, 484, 33), (7, 4, 65)):
            h, s, p, c, e = orc.gen_ecdsa_edge_batch(0xD0D0 + n, n, publen, 8)
            ok = np.full(n, 7, np.uint8)
            rc = lib.lamd_multi_verify_ecdsa_batch(m, n, h.ctypes.data, s.ctypes.data, p.ctypes.data, publen, publen, group, ok.ctypes.data)
            assert rc == 0, lib.lamd_multi_last_error(m)
            assert np.array_equal(ok, e), (n, group, publen, int((ok != e).sum()))
        mm, x, sg, c, e = orc.gen_schnorr_edge_batch(0xFEED, 150_001, 8)
        ok = np.full(150_001, 7, np.uint8)
        assert lib.lamd_multi_verify_schnorr_batch(m, 150_001, mm.ctypes.data, x.ctypes.data, sg.ctypes.data, 1, ok.ctypes.data) == 0, lib.lamd_multi_last_error(m)
        assert np.array_equal(ok, e)
        vs = kat["gossip"]
        msgs = [H(v["msg"]) for v in vs] * 40
        ids = [H(v["node_id"]) if "node_id" in v else bytes(33) for v in vs] * 40
        blob = np.frombuffer(b"".join(msgs) + b"\x00", dtype=np.uint8).copy()
        off = np.concatenate([[0], np.cumsum([len(q) for q in msgs])]).astype(np.uint64)
        idarr = np.frombuffer(b"".join(ids), dtype=np.uint8).reshape(len(ids), 33).copy()
        got = np.full(len(msgs), 99, np.int8)
        assert lib.lamd_multi_sigcheck_gossip_batch(m, len(msgs), blob.ctypes.data, off.ctypes.data, idarr.ctypes.data, got.ctypes.data) == 0, lib.lamd_multi_last_error(m)
        assert list(got) == [v["expect"] for v in vs] * 40
    finally:
        lib.lamd_multi_shutdown(m)
