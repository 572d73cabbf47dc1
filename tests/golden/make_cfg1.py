#!/usr/bin/env python3
"""BASELINE.json configs[0] / SURVEY.md 8(d) cfg1: 1 024 ECDSA triples over precomputed sighashes, seed 0xC1A00001, to be
pushed ONE BY ONE through the check_signed_hash()-shaped entry (bitcoin/signature.c:174-192).  Rows 0..13 are the
reference-held known answers (KAT-G: gossipd/test/run-check_channel_announcement.c:62-108, KAT-O:
onchaind/test/run-grind_feerate.c:120-150, KAT-B11: common/test/run-bolt11.c:465-467); the rest are signed here with the C
oracle's signer (vector generation only) under 64 splitmix64-derived keys, 90 % valid / 10 % invalid over the classes of
cfg2.  Output: tests/golden/cfg1.bin = 1024 rows of hash32 | sig64 | pub33 | expect(1 byte).

usage: python tests/golden/make_cfg1.py     (needs oracle/liblnamd_oracle.so; the product never reads oracle/)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
import pyref  # noqa: E402

SEED = 0xC1A00001
N_ROWS = 1024
M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def rand32(stream, idx):
    return b"".join(splitmix64(SEED ^ splitmix64(idx * 4 + j + (stream << 56))).to_bytes(8, "little") for j in range(4))


def main():
    kat = json.load(open(os.path.join(HERE, "kat.json")))
    H = bytes.fromhex
    rows = []
    for v in kat["ecdsa"]:
        if v["name"].startswith("KAT"):
            rows.append((H(v["hash"]), H(v["sig"]), H(v["pub"]), v["expect"]))
    assert len(rows) == 14
    keys = []
    for k in range(64):
        d = rand32(1, k)
        pub = orc.pubkey_create(d)
        keys.append((d, pyref.ser33((int.from_bytes(pub[1:33], "big"), int.from_bytes(pub[33:], "big")))))
    classes = ("flip_hash", "flip_r", "flip_s", "high_s", "wrong_key", "r_zero", "s_zero", "offcurve_key")
    i = 0
    while len(rows) < N_ROWS:
        d, pub = keys[splitmix64(SEED ^ splitmix64(i + (7 << 56))) % 64]
        h = rand32(3, i)
        sig = orc.ecdsa_sign(h, d, rand32(2, i))
        sel = splitmix64(SEED ^ splitmix64(i + (9 << 56)))
        expect = True
        if sel % 10 == 0:
            c = classes[(sel >> 8) % 8]
            bit = (sel >> 16) % 256
            hb, sb, pb = bytearray(h), bytearray(sig), bytearray(pub)
            if c == "flip_hash":
                hb[bit >> 3] ^= 1 << (bit & 7)
            elif c == "flip_r":
                sb[bit >> 3] ^= 1 << (bit & 7)
            elif c == "flip_s":
                sb[32 + (bit >> 3)] ^= 1 << (bit & 7)
            elif c == "high_s":
                sb[32:] = (pyref.N - int.from_bytes(sig[32:], "big")).to_bytes(32, "big")
            elif c == "wrong_key":
                pb[:] = keys[(keys.index((d, pub)) + 1) % 64][1]
            elif c == "r_zero":
                sb[:32] = bytes(32)
            elif c == "s_zero":
                sb[32:] = bytes(32)
            elif c == "offcurve_key":
                pb[1 + (bit >> 3)] ^= 1 << (bit & 7)
            h, sig, pub = bytes(hb), bytes(sb), bytes(pb)
            expect = pyref.ecdsa_verify(h, sig, pub)      # a flipped bit can, very rarely, still verify / the x may still lift
        rows.append((h, sig, pub, expect))
        i += 1
    blob = b"".join(h + s + p + bytes([1 if e else 0]) for h, s, p, e in rows)
    assert len(blob) == N_ROWS * 130
    with open(os.path.join(HERE, "cfg1.bin"), "wb") as f:
        f.write(blob)
    print("cfg1.bin:", N_ROWS, "rows,", sum(1 for r in rows if r[3]), "accept")


if __name__ == "__main__":
    main()
