#!/usr/bin/env python3
"""Timing of the BOLT #12 front end on the device: N invoice-shaped TLV streams (a dozen fields each, 300-400 bytes) through lamd_bolt12_merkle_batch
(merkle root + tagged signature hash per stream, k_bolt12_hash) and the kernel's own duration when run under rocprofv3 --kernel-trace.
usage: python tools/bolt12_probe.py [N]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lightning_amd import Engine


def bigsize(v):
    if v < 0xfd:
        return bytes([v])
    if v <= 0xffff:
        return b"\xfd" + v.to_bytes(2, "big")
    return b"\xfe" + v.to_bytes(4, "big")


def stream(rnd):
    types = sorted(rnd.sample([0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 80, 82, 84, 88, 160, 162, 164, 168, 170, 176], rnd.randrange(8, 16)))
    out = b""
    for t in types:
        ln = rnd.choice([1, 4, 8, 32, 33, 33, 20, 60])
        out += bigsize(t) + bigsize(ln) + bytes(rnd.randrange(256) for _ in range(ln))
    return out + bigsize(240) + bigsize(64) + bytes(64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rnd = random.Random(12)
    base = [stream(rnd) for _ in range(2000)]
    streams = [base[i % len(base)] for i in range(n)]
    eng = Engine(0)
    mk, sh, ok = eng.bolt12_merkle_batch(streams, b"invoice", b"signature")
    assert ok.all()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        eng.bolt12_merkle_batch(streams, b"invoice", b"signature")
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("bolt12_merkle_batch: %d streams (%.0f bytes each on average): p50 %.3f ms per call = %.2f M streams/s; checksum %s"
          % (n, sum(map(len, streams)) / n, 1e3 * ts[len(ts) // 2], n / ts[len(ts) // 2] / 1e6, bytes(np.bitwise_xor.reduce(sh, axis=0)).hex()[:16]))
    eng.close()


if __name__ == "__main__":
    main()
