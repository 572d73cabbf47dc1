#!/usr/bin/env python3
"""Where a commitment_signed's latency goes on the one-launch path: the whole 484-row batch, the 483 htlc rows alone, subsets of them,
rows under many cached keys for comparison (submit -> verdicts in host memory, p50 of 100 calls)."""
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightning_amd import Engine, workload

eng = Engine(0)
cs = workload.make_commit_storm(eng, 2, device="cuda:0")["ecdsa"]


def p50(fn, reps=100):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    ts = np.sort(np.array(ts[reps // 10:])) * 1e3
    return ts[len(ts) // 2]


def leg(name, sl):
    hh, ss, pp = [np.ascontiguousarray(x[sl]) for x in cs.cols]
    for _ in range(3):
        assert (eng.verify_ecdsa(hh, ss, pp) == cs.expect[sl]).all()
    inf = eng.info()
    print("%-44s %4d rows  p50 %.3f ms   hits %d ladder %d teeth %s" % (name, len(hh), p50(lambda: eng.verify_ecdsa(hh, ss, pp)), inf["last_cache_hits"], inf["last_cold_rows"], inf["last_keyed"]))


leg("commitment (funding + 483 htlc rows)", slice(0, 484))
leg("the 483 htlc rows", slice(1, 484))
leg("64 htlc rows", slice(1, 65))
leg("128 htlc rows", slice(1, 129))
leg("256 htlc rows", slice(1, 257))
leg("funding row alone", slice(0, 1))
leg("second commitment (other keys)", slice(484, 968))
eng.close()
