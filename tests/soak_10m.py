#!/usr/bin/env python3
"""10 M-signature soak (SURVEY.md 8(d) acceptance): every row of configs[1]..[4] at full size is verified on the GPU and
by the CPU oracle; verdict vectors must be bit-identical.  Writes a JSON summary (stdout).  ~1-2 minutes on a 16-core host.
tests/test_gpu_soak.py runs the same function under pytest (-m gpu)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orc
from lightning_amd import Engine, workload



def run(eng=None):
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    own = eng is None
    eng = eng or Engine(0)
    out = {"cores": cores, "configs": {}}
    total = mism = 0
    t00 = time.time()


    def rec(name, gpu, cpu, by_construction):
        nonlocal total, mism
        m = int((gpu != cpu).sum())
        c = int((gpu != by_construction).sum())
        out["configs"][name] = {"rows": int(gpu.shape[0]), "gpu_vs_oracle_mismatches": m, "gpu_vs_construction_mismatches": c,
                                "accepts": int((gpu == (1 if gpu.dtype == np.bool_ else 0)).sum()) if gpu.dtype != np.bool_ else int(gpu.sum())}
        total += gpu.shape[0]
        mism += m + c


    w = workload.make_ecdsa(eng, 1_000_000, publen=65)
    eng.verify_ecdsa_device(w.dev[0], w.dev[1], w.dev[2], w.d_ok); eng.synchronize()
    rec("configs[1] 1M ECDSA-65", w.d_ok.cpu().numpy().astype(bool), orc.ecdsa_verify_batch(w.cols[0], w.cols[1], w.cols[2], 65, cores).astype(bool), w.expect)
    w = workload.make_schnorr(eng, 1_000_000)
    eng.verify_schnorr_device(w.dev[0], w.dev[1], w.dev[2], w.d_ok); eng.synchronize()
    rec("configs[2] 1M BIP-340", w.d_ok.cpu().numpy().astype(bool), orc.schnorr_verify_batch(w.cols[0], w.cols[1], w.cols[2], cores).astype(bool), w.expect)
    g = workload.make_gossip(eng, 500_000, 2_000_000)
    eng.sigcheck_gossip_device(g.n, g.d_msgs, g.d_off, g.d_ids, g.d_rowbase, g.rows, g.d_verdict); eng.synchronize()
    gv = g.d_verdict.cpu().numpy()
    cv = orc.sigcheck_gossip_batch(g.msgs, g.off, np.ascontiguousarray(g.ids), cores)
    m = int((gv != cv).sum()); c = int((gv != g.expect).sum())
    out["configs"]["configs[3] gossip replay 500k cann + 2M cupd"] = {"messages": g.n, "rows": g.rows, "gpu_vs_oracle_mismatches": m, "gpu_vs_construction_mismatches": c,
                                                                     "ok_messages": int((gv == 0).sum())}
    total += g.rows; mism += m + c
    del g
    st = workload.make_commit_storm(eng, 10_000)
    we, ws = st["ecdsa"], st["schnorr"]
    eng.verify_ecdsa_device(we.dev[0], we.dev[1], we.dev[2], we.d_ok)
    eng.verify_schnorr_device(ws.dev[0], ws.dev[1], ws.dev[2], ws.d_ok); eng.synchronize()
    rec("configs[4] commit storm ECDSA part", we.d_ok.cpu().numpy().astype(bool), orc.ecdsa_verify_batch(we.cols[0], we.cols[1], we.cols[2], 33, cores).astype(bool), we.expect)
    rec("configs[4] commit storm BIP-340 part", ws.d_ok.cpu().numpy().astype(bool), orc.schnorr_verify_batch(ws.cols[0], ws.cols[1], ws.cols[2], cores).astype(bool), ws.expect)
    out["total_verifies"] = total
    out["total_mismatches"] = mism
    out["wall_s"] = time.time() - t00
    if own:
        eng.close()
    return out


if __name__ == "__main__":
    res = run()
    print(json.dumps(res, indent=1))
    sys.exit(1 if res["total_mismatches"] else 0)
