"""The >= 10 M-signature bit-exact soak of SURVEY 8(d) ("acceptance") and the million-row edge-class differential, under the driver's
`pytest -m gpu` (VERDICT r03 "Next" 7): every row is decided by the HIP engine AND by the CPU oracle; the verdict vectors must be
identical, and equal to what each row's construction fixes."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return n


@pytest.mark.gpu
def test_soak_10m_signatures_gpu_equals_oracle_equals_construction():
    """configs[1]..[4] at full size (1 M ECDSA-65, 1 M BIP-340, 500 k channel_announcements + 2 M channel_updates, 10 k commitments x 484):
    10.84 M signatures, GPU == C oracle == construction on every row (tests/soak_10m.py)"""
    import soak_10m
    res = soak_10m.run()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "soak_10M.json"), "w") as f:
        json.dump(res, f, indent=1)
    assert res["total_verifies"] >= 10_000_000, res
    assert res["total_mismatches"] == 0, res


@pytest.mark.gpu
def test_million_edge_class_rows_gpu_equals_oracle_equals_construction(orc):
    """the rows of tests/test_oracle_golden.py::test_million_row_differential_... (oracle/edge_gen.c: every synthesised edge class of SURVEY
    8(c), where the CPU suite has C oracle == OpenSSL == construction) through the HIP engine: 2 x 500 k ECDSA rows (33- and 65-byte keys,
    hybrid 06/07 keys, hash >= n, r / s = 0 and >= n, high S, damaged keys of every kind) + 100 k BIP-340 rows (r >= p, s >= n, unliftable
    keys, negated s, ...); host buffers in, verdicts out (the general path) and a slice through the one-launch latency path"""
    from lightning_amd import Engine
    cores = _cores()
    with Engine(0) as eng:
        for publen, seed in ((33, 0xC1A00006), (65, 0xC1A00016)):
            h, s, p, c, e = orc.gen_ecdsa_edge_batch(seed, 500_000, publen, cores)
            got = eng.verify_ecdsa(h, s, p).astype(np.uint8)
            cpu = orc.ecdsa_verify_batch(h, s, p, publen, cores)
            bad = np.nonzero((got != cpu) | (got != e))[0]
            assert bad.size == 0, ("publen %d: %d rows differ; first %d class %d gpu %d oracle %d expected %d" %
                                   (publen, bad.size, bad[0], c[bad[0]], got[bad[0]], cpu[bad[0]], e[bad[0]]))
            small = eng.verify_ecdsa(np.ascontiguousarray(h[:3000]), np.ascontiguousarray(s[:3000]), np.ascontiguousarray(p[:3000])).astype(np.uint8)
            assert np.array_equal(small, e[:3000]), "latency path, publen %d" % publen
        m, x, sg, c, e = orc.gen_schnorr_edge_batch(0xC1A00007, 100_000, cores)
        got = eng.verify_schnorr(m, x, sg).astype(np.uint8)
        cpu = orc.schnorr_verify_batch(m, x, sg, cores)
        bad = np.nonzero((got != cpu) | (got != e))[0]
        assert bad.size == 0, ("BIP-340: %d rows differ; first %d class %d gpu %d oracle %d expected %d" % (bad.size, bad[0], c[bad[0]], got[bad[0]], cpu[bad[0]], e[bad[0]]))
        small = eng.verify_schnorr(np.ascontiguousarray(m[:3000]), np.ascontiguousarray(x[:3000]), np.ascontiguousarray(sg[:3000])).astype(np.uint8)
        assert np.array_equal(small, e[:3000])

