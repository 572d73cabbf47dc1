#!/usr/bin/env python3
"""Harvests the prototypes of the reference's signature-check boundary from the Core Lightning tree and writes
them, whitespace-normalised, to tests/golden/ref_prototypes.json.  tests/test_abi.py requires include/cln_shim.h to hold each of
them token for token; when /root/reference is present the test also re-runs this harvest and requires the fixture to be current.

    python tests/golden/make_ref_prototypes.py [/root/reference]
"""
import json
import os
import re
import sys

WANT = [  # (header relative to the reference tree, function)
    ("bitcoin/signature.h", "check_signed_hash"),
    ("bitcoin/signature.h", "check_tx_sig"),
    ("bitcoin/signature.h", "check_schnorr_sig"),
    ("bitcoin/signature.h", "signature_from_der"),
    ("common/node_id.h", "check_signed_hash_nodeid"),
    ("common/node_id.h", "pubkey_from_node_id"),
    ("gossipd/sigcheck.h", "sigcheck_channel_update"),
    ("gossipd/sigcheck.h", "sigcheck_channel_announcement"),
    ("gossipd/sigcheck.h", "sigcheck_node_announcement"),
    ("bitcoin/shadouble.h", "sha256_double"),
    ("bitcoin/pubkey.h", "pubkey_from_der"),
    ("bitcoin/pubkey.h", "pubkey_to_der"),
]


def strip_comments(src):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))


def normalise(proto):
    return re.sub(r"\s+", " ", proto).replace("( ", "(").replace(" )", ")").strip()


def find(src, name):
    """the declaration `<type> name(...);` in a comment-free header"""
    m = re.search(r"(?:^|\n)((?:const\s+)?[A-Za-z_][A-Za-z_0-9 ]*?[\s\*]+%s\s*\([^;{]*\))\s*;" % re.escape(name), src)
    return normalise(m.group(1)) if m else None


def harvest(ref):
    out = {}
    for hdr, fn in WANT:
        proto = find(strip_comments(open(os.path.join(ref, hdr)).read()), fn)
        assert proto, (hdr, fn)
        out[fn] = {"header": hdr, "prototype": proto}
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "ref_prototypes.json"), "w") as f:
        json.dump(harvest(ref), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", os.path.join(here, "ref_prototypes.json"))
