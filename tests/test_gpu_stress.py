"""Randomised stress of every entry point, interleaved without host synchronisation between the asynchronous ones, on
batch sizes from 1 row to several chunks -- results compared with the CPU oracle.  Guards the cross-stream ordering of the
engine's lanes (front end of one call under the ecmult kernels of the previous ones, prep / cold-row side streams, workspace
re-use and growth): verdicts must not depend on how many hardware queues the runtime hands out (tools/queue_sweep.sh runs
this file under GPU_MAX_HW_QUEUES = 4, 16 and 32)."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H = bytes.fromhex


def _rows(items, w):
    return np.frombuffer(b"".join(items), dtype=np.uint8).reshape(len(items), w).copy()


def _run(eng, orc, kat, seed, rounds, max_rows):
    import torch
    from lightning_amd import workload
    rnd = random.Random(seed)
    # pools of device-generated, partly corrupted workloads (expected verdicts known by construction AND checked vs the oracle)
    pools = []
    for r in range(3):
        pools.append(("e", workload.make_ecdsa(eng, max_rows, seed=7000 + seed * 10 + r, nkeys=[40, 3000, 1 << 40][r], publen=33 if r & 1 else 65)))
        pools.append(("s", workload.make_schnorr(eng, max_rows, seed=7100 + seed * 10 + r, nkeys=[1 << 40, 25, 2000][r])))
    g = workload.make_gossip(eng, max_rows // 10, max_rows // 3, n_nodes=80, corrupt_frac=0.04)
    for kind, w in pools[:2]:  # oracle pin of the construction on a sample
        m = min(w.n, 1500)
        exp = (orc.ecdsa_verify_batch(w.cols[0][:m], w.cols[1][:m], w.cols[2][:m], w.cols[2].shape[1], 4) if kind == "e"
               else orc.schnorr_verify_batch(w.cols[0][:m], w.cols[1][:m], w.cols[2][:m], 4)).astype(bool)
        assert np.array_equal(exp, w.expect[:m])
    gmsgs = [g.msgs[int(g.off[i]):int(g.off[i + 1])].tobytes() for i in range(g.n)]
    gids = [g.ids[i].tobytes() if i >= g.n_cann else None for i in range(g.n)]
    rec = [v for v in kat["recover"]]
    pending_flushes = []
    for rd in range(rounds):
        checks = []          # (description, thunk returning (got, expect)) evaluated after ONE synchronize
        nops = rnd.randrange(3, 9)
        for _ in range(nops):
            op = rnd.choice(["dev", "dev", "dev", "gossip_dev", "host", "gossip_host", "queue", "recover", "parse", "single"])
            if op == "dev":
                kind, w = rnd.choice(pools)
                n = rnd.choice([1, 2, 63, 64, 65, 255, 257, rnd.randrange(1, max_rows), rnd.randrange(1, max_rows), max_rows])
                o = rnd.randrange(0, w.n - n + 1)
                d_ok = torch.full((n,), 7, dtype=torch.uint8, device="cuda:0")      # poisoned: a launch that writes nothing is caught
                a, b, c = (t[o:o + n] for t in w.dev)
                (eng.verify_ecdsa_device if kind == "e" else eng.verify_schnorr_device)(a, b, c, d_ok)
                checks.append(("dev %s n=%d o=%d" % (kind, n, o), lambda d_ok=d_ok, w=w, o=o, n=n: (d_ok.cpu().numpy(), w.expect[o:o + n].astype(np.uint8))))
            elif op == "gossip_dev":
                d_v = torch.full((g.n,), 99, dtype=torch.int8, device="cuda:0")
                eng.sigcheck_gossip_device(g.n, g.d_msgs, g.d_off, g.d_ids, g.d_rowbase, g.rows, d_v)
                checks.append(("gossip dev", lambda d_v=d_v: (d_v.cpu().numpy(), g.expect)))
            elif op == "host":
                kind, w = rnd.choice(pools)
                n = rnd.choice([1, 7, 484, rnd.randrange(1, max_rows)])
                o = rnd.randrange(0, w.n - n + 1)
                f = eng.verify_ecdsa if kind == "e" else eng.verify_schnorr
                got = f(w.cols[0][o:o + n], w.cols[1][o:o + n], w.cols[2][o:o + n])
                assert np.array_equal(got, w.expect[o:o + n]), ("host", kind, n, o)
            elif op == "gossip_host":
                k = rnd.randrange(1, min(g.n, 4000))
                o = rnd.randrange(0, g.n - k + 1)
                got = eng.sigcheck_gossip(gmsgs[o:o + k], gids[o:o + k])
                assert np.array_equal(got, g.expect[o:o + k]), ("gossip host", k, o)
            elif op == "queue" and len(pending_flushes) < 3:
                exp = []
                for _ in range(rnd.randrange(1, 4)):
                    kind, w = rnd.choice(pools)
                    n = rnd.choice([1, 484, rnd.randrange(1, 3000)])
                    o = rnd.randrange(0, w.n - n + 1)
                    if kind == "e":
                        t = eng.queue_ecdsa_batch(w.cols[0][o:o + n], w.cols[1][o:o + n], w.cols[2][o:o + n])
                    else:
                        t = eng.queue_schnorr_batch(w.cols[0][o:o + n], w.cols[1][o:o + n], w.cols[2][o:o + n])
                    assert t == len(exp)
                    exp.extend(w.expect[o:o + n])
                eng.flush()
                pending_flushes.append(np.array(exp, dtype=bool))
            elif op == "recover":
                vs = [rnd.choice(rec) for _ in range(rnd.randrange(1, 40))]
                keys, ok = eng.ecdsa_recover(_rows([H(v["hash"]) for v in vs], 32), _rows([H(v["sig"]) for v in vs], 64),
                                             np.array([v["recid"] for v in vs], dtype=np.uint8))
                for v, k, o in zip(vs, keys, ok):
                    assert (k.tobytes().hex() if o else None) == v["expect"], v["name"]
            elif op == "parse":
                vs = [rnd.choice(kat["pubkey"]) for _ in range(rnd.randrange(1, 30))]
                for ln in (33, 65):
                    sel = [v for v in vs if len(v["pub"]) == 2 * ln]
                    if sel:
                        _, ok = eng.pubkey_parse(_rows([H(v["pub"]) for v in sel], ln))
                        assert [bool(x) for x in ok] == [v["expect"] is not None for v in sel]
            elif op == "single":
                v = rnd.choice([v for v in kat["ecdsa"] if len(v["pub"]) == 66])
                assert eng.check_signed_hash(H(v["hash"]), H(v["sig"]), H(v["pub"])) == v["expect"], v["name"]
            if pending_flushes and rnd.random() < 0.4:
                got = eng.poll()
                if got is not None:
                    assert np.array_equal(got, pending_flushes.pop(0)), "flush (poll)"
        eng.synchronize()
        for desc, thunk in checks:
            got, exp = thunk()
            assert np.array_equal(got, exp), (seed, rd, desc, np.nonzero(got != exp)[0][:8], got[:8])
    while pending_flushes:
        assert np.array_equal(eng.wait(), pending_flushes.pop(0)), "flush (wait)"


def test_stress_interleaved_entry_points(orc, kat):
    from lightning_amd import Engine
    with Engine(0) as eng:
        _run(eng, orc, kat, seed=int(os.environ.get("LAMD_STRESS_SEED", "1")), rounds=int(os.environ.get("LAMD_STRESS_ROUNDS", "12")), max_rows=30000)


def test_stress_small_chunks_alternate_lanes(orc, kat):
    """the same with 4096-row chunks: calls larger than a chunk alternate between a lane and its peer, workspaces are small
    and re-used at once"""
    from lightning_amd import Engine
    old = os.environ.get("LAMD_CHUNK_ROWS")
    os.environ["LAMD_CHUNK_ROWS"] = "4096"
    os.environ["LAMD_KEYED_MIN_ROWS"] = "1024"
    try:
        with Engine(0) as eng:
            _run(eng, orc, kat, seed=2 + int(os.environ.get("LAMD_STRESS_SEED", "1")), rounds=6, max_rows=20000)
    finally:
        os.environ.pop("LAMD_KEYED_MIN_ROWS", None)
        if old is None:
            os.environ.pop("LAMD_CHUNK_ROWS", None)
        else:
            os.environ["LAMD_CHUNK_ROWS"] = old


def test_device_field_fuzz_one_million_ops():
    """>= 10^6 randomised field/group operations at the magnitude limits on the GPU (generated-asm fe_mul / fe_sqr under the
    register pressure of live Jacobian points) against the host build of the same functions"""
    from lightning_amd import Engine
    with Engine(0) as eng:
        bad, ops, rep = eng.fuzz_field(lanes=16384, iters=64, seed=0xF00D)
        assert bad == 0, rep
        assert ops >= 30_000_000
        bad, ops, rep = eng.fuzz_field(lanes=64, iters=2000, seed=0xBEEF)   # one latency-bound wave
        assert bad == 0, rep


@pytest.mark.gpu
def test_calls_in_flight_may_share_one_verdict_buffer(monkeypatch):
    """Regression (round 2): the kernels of a call used to keep their inter-kernel state (SCHNORR_PENDING / VERDICT_SUSPECT) in the
    CALLER's verdict buffer; two calls in flight on different lanes with the same buffer (bench.py's steps) then disturbed each
    other's BIP-340 parity stage and valid signatures came back rejected -- visible once the scalar preparation was slow enough
    (few, long-running threads) for calls to overlap widely.  Verdicts now reach the caller's buffer in one final copy."""
    import numpy as np
    import torch
    from lightning_amd import Engine, workload
    monkeypatch.setenv("LAMD_CACHE", "0")
    monkeypatch.setenv("LAMD_PREP_BATCH", "64")
    monkeypatch.setenv("LAMD_PREP_MIN_THREADS", "4096")
    n = 300_000
    with Engine(0) as eng:
        we = workload.make_ecdsa(eng, n, seed=91, nkeys=20000, publen=65)
        ws = workload.make_schnorr(eng, n, seed=92, nkeys=20000)
        eng.auto_order = False
        for rep in range(3):
            for k in range(10):
                eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
                eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok)
            eng.synchronize()
            torch.cuda.synchronize()
            assert int((we.d_ok.cpu().numpy().astype(bool) != we.expect).sum()) == 0, rep
            assert int((ws.d_ok.cpu().numpy().astype(bool) != ws.expect).sum()) == 0, rep
