#!/usr/bin/env python3
"""configs[4] (commit storm) strong-scaling shards through the streaming queue with the first flushes of a shard ramped up (LAMD_BENCH_RAMP = q: grp/q, 2 grp/q, ...
rows) against flushes of one size: the whole job (W = 1) and every 1/8 shard, best of 4, alternating.  The all-gather of the verdict bytes is left out (0.04 ms).
usage: python tools/cfg5_ramp_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from lightning_amd import Engine, sharding, workload


def main():
    device = "cuda:0"
    torch.cuda.set_device(0)
    eng = Engine(0)
    st = workload.make_commit_storm(eng, 10_000, device=device)
    per, grp = st["per"], 256 * st["per"]
    depth = min(8, eng.info()["queue_sets"] - 1)

    def best(fn, reps):
        ts = []
        for _ in range(reps):
            eng.synchronize()
            t1 = time.perf_counter()
            fn()
            eng.synchronize()
            ts.append(time.perf_counter() - t1)
        return min(ts[1:])
    for ramp, cpf in ((0, 256), (0, 320), (0, 416), (0, 512), (0, 640), (0, 256), (0, 320), (0, 416), (0, 512), (0, 640)) if os.environ.get("PROBE_FLUSH_SIZES") else \
            tuple((r, 256) for r in (0, 4, 8, 2, 0, 4, 8, 2)):
        os.environ["LAMD_BENCH_RAMP"] = str(ramp)
        grp = cpf * per
        res = {}
        for W in (1, 8):
            bb = {kind: sharding.shard_bounds(st[kind].n, W, [per] * (st[kind].n // per)) for kind in ("ecdsa", "schnorr")}
            ms, bad = [], 0
            for k in range(W):
                keep = {}
                ms.append(best(lambda: keep.update(bench.stream_shard(eng, st, bb, k, grp, depth)), 4) * 1e3)
                for kind, got in keep.items():
                    bad += int((got.astype(bool) != st[kind].expect[int(bb[kind][k]):int(bb[kind][k + 1])]).sum())
            res[W] = (max(ms), bad)
        print("%d commitments per flush, ramp %d: whole job %.2f ms, slowest 1/8 shard %.2f ms -> predicted speed-up %.2f; mismatches %d"
              % (cpf, ramp, res[1][0], res[8][0], res[1][0] / res[8][0], res[1][1] + res[8][1]), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
