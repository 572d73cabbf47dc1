"""CPU-only checks of the drop-in boundary: the library builds for gfx950, loads, and exports
every symbol include/lightning_amd.h declares; without a GPU it fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="lightning_amd.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lamd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from lightning_amd import _ffi
    lib = _ffi.load()
    product, debug = _declared(), _declared("lightning_amd_debug.h")
    assert len(product) >= 20 and not set(product) & set(debug)
    # the drop-in boundary carries no diagnostics
    assert not [n for n in product if "debug" in n or "selftest" in n or "fuzz" in n]
    names = sorted(product + debug)
    for n in names:
        assert hasattr(lib, n), "liblightning_amd.so does not export %s" % n
        assert n in _ffi.SYMBOLS, "ctypes binding missing for %s" % n
    assert set(_ffi.SYMBOLS) == set(names)
    assert b"gfx950" in lib.lamd_version()
    # no signing code in the product library: the synthetic-workload signer kernels live in liblightning_amd_testgen.so
    assert not [n for n in names if n.startswith("lamd_gen_")]
    assert not [n for n in ("lamd_gen_ecdsa_device", "lamd_gen_schnorr_device", "lamd_gen_gossip_device") if hasattr(lib, n)]


def test_struct_layouts_match_the_headers(tmp_path):
    """the ctypes mirrors of the structures the C ABI hands over (lamd_info; the gossip ingest's event, config and statistics) have the size and
    the field offsets a C compiler gives the declarations in include/: a field added on one side only would shift everything behind it"""
    import subprocess
    from lightning_amd import _ffi
    fields = [f[0] for f in _ffi.LamdInfo._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lightning_amd.h"\nint main(void) {\n  printf("%zu\\n", sizeof(lamd_info));\n'
                   + "".join('  printf("%%zu\\n", offsetof(lamd_info, %s));\n' % f for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_ffi.LamdInfo), (out[0], ctypes.sizeof(_ffi.LamdInfo))
    assert out[1:] == [getattr(_ffi.LamdInfo, f).offset for f in fields]
    # ... and the three structures of the gossip ingest's ABI (include/lightning_amd_gossipd.h <-> lightning_amd/gossipd.py)
    from lightning_amd import gossipd
    gsrc = tmp_path / "layout_g.c"
    pairs = [("lamd_gossipd_event", gossipd.Event), ("lamd_gossipd_config", gossipd.Config), ("lamd_gossipd_stats", gossipd.Stats)]
    lines = []
    for cname, cls in pairs:
        lines.append('  printf("%%zu\\n", sizeof(%s));\n' % cname)
        lines += ['  printf("%%zu\\n", offsetof(%s, %s));\n' % (cname, f[0]) for f in cls._fields_]
    gsrc.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lightning_amd_gossipd.h"\nint main(void) {\n' + "".join(lines) + "  return 0;\n}\n")
    gexe = tmp_path / "layout_g"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), "-o", str(gexe), str(gsrc)])
    gout = [int(x) for x in subprocess.check_output([str(gexe)]).split()]
    want = []
    for cname, cls in pairs:
        want.append(ctypes.sizeof(cls))
        want += [getattr(cls, f[0]).offset for f in cls._fields_]
    assert gout == want, (gout, want)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lightning_amd.h")).read(), flags=re.S)
    body = re.search(r"typedef struct[^{;]*\{([^}]*)\} lamd_info;", hdr, re.S).group(1)
    declared = re.findall(r"\b([a-z_0-9]+)\s*(?:\[[^\]]*\])?\s*[,;]", body)
    assert declared == fields, (declared, fields)


def test_testgen_header_symbols_exported():
    """include/lightning_amd_testgen.h (test / bench infrastructure) -> liblightning_amd_testgen.so"""
    from lightning_amd import _ffi
    names = _declared("lightning_amd_testgen.h")
    assert names == sorted(_ffi.TESTGEN_SYMBOLS) and len(names) == 3
    tg = _ffi.load_testgen()
    for n in names:
        assert hasattr(tg, n)


def test_mirror_header_is_installed_and_exported():
    """include/cln_shim.h: every prototype with one of the reference's names is exported by liblightning_amd_cln.so"""
    from lightning_amd import _build
    src = open(os.path.join(ROOT, "include", "cln_shim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    lib = ctypes.CDLL(_build.build_shim())
    want = ["check_signed_hash", "check_signed_hash_nodeid", "check_schnorr_sig", "check_tx_sig", "sigcheck_channel_update",
            "sigcheck_channel_announcement", "sigcheck_node_announcement", "signature_from_der", "pubkey_from_der",
            "fromwire_secp256k1_ecdsa_signature", "sha256_double", "bolt12_check_signature", "merkle_tlv", "sighash_from_merkle",
            "shim_tal_dup", "shim_tal_bytelen", "sigcheck_channel_update_len", "sigcheck_channel_announcement_len", "sigcheck_node_announcement_len",
            "lamd_secp256k1_ecdsa_verify", "lamd_secp256k1_ecdsa_recoverable_signature_convert",
            "lamd_secp256k1_ecdsa_recoverable_signature_parse_compact", "lamd_secp256k1_ecdsa_recover"]
    for n in want:
        assert re.search(r"\b%s\s*\(" % n, src), n
        assert hasattr(lib, n), n
    assert not hasattr(lib, "tal_bytelen")
    # lightningd, libwally and bitcoin/signature.c link the real libsecp256k1, whose opaque types have another layout: the mirror must not
    # define (and so interpose) any of that library's symbols, nor ccan's -- its libsecp-shaped entry points carry the lamd_ prefix
    import subprocess
    dyn = subprocess.check_output(["nm", "-D", "--defined-only", _build.build_shim()], text=True).split("\n")
    exported = [ln.split()[-1] for ln in dyn if ln.strip()]
    clash = [n for n in exported if n.startswith(("secp256k1_", "tal_", "wally_"))]
    assert clash == [], clash


def test_mirror_prototypes_equal_the_reference_headers_token_for_token(tmp_path):
    """the prototypes of the boundary as the reference declares them (harvested into tests/golden/ref_prototypes.json by
    tests/golden/make_ref_prototypes.py) appear in include/cln_shim.h verbatim modulo white space -- same types, same argument names, no
    extra argument; with the reference tree present the fixture itself is re-harvested and must be current.  The reference's own
    unit-test calls compile against the header (tests/c/run_check_channel_announcement.c) and link with the mirror."""
    import importlib.util
    import json
    import subprocess
    gold = os.path.join(ROOT, "tests", "golden")
    want = json.load(open(os.path.join(gold, "ref_prototypes.json")))
    spec = importlib.util.spec_from_file_location("make_ref_prototypes", os.path.join(gold, "make_ref_prototypes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hdr = mod.strip_comments(open(os.path.join(ROOT, "include", "cln_shim.h")).read())
    for fn in ("check_signed_hash", "check_tx_sig", "check_schnorr_sig", "check_signed_hash_nodeid", "sigcheck_channel_update",
               "sigcheck_channel_announcement", "sigcheck_node_announcement", "signature_from_der", "pubkey_from_node_id", "sha256_double",
               "pubkey_from_der", "pubkey_to_der"):
        assert mod.find(hdr, fn) == want[fn]["prototype"], (fn, mod.find(hdr, fn), want[fn]["prototype"])
    if os.path.isdir("/root/reference/gossipd"):
        assert mod.harvest("/root/reference") == want, "tests/golden/ref_prototypes.json is stale: run tests/golden/make_ref_prototypes.py"
    from lightning_amd import _build
    _build.build_shim()
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", str(tmp_path / "rcca"),
                           os.path.join(ROOT, "tests", "c", "run_check_channel_announcement.c"), "-L" + os.path.join(ROOT, "lightning_amd"),
                           "-llightning_amd_cln", "-llightning_amd", "-Wl,-rpath," + os.path.join(ROOT, "lightning_amd")])
    if not _have_gpu():   # no device: the mirror fails closed, loudly (exit 3 = lamd_shim_setup() refused), it never says "OK"
        r = subprocess.run([str(tmp_path / "rcca"), "0100"], capture_output=True, text=True)
        assert r.returncode == 3 and "no engine" in r.stdout


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_gossipd_header_symbols_exported():
    """include/lightning_amd_gossipd.h (the batched gossip ingest) lives in the host-side library next to the reference-named mirror"""
    from lightning_amd import _build, gossipd
    names = _declared("lightning_amd_gossipd.h")
    assert len(names) >= 9
    lib = ctypes.CDLL(_build.build_shim())
    for n in names:
        assert hasattr(lib, n), "liblightning_amd_cln.so does not export %s" % n
    assert gossipd._load() is not None


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lightning_amd import Engine, LamdError
    with pytest.raises(LamdError):
        Engine(0)
    # and the raw ABI reports it, leaving no usable context behind
    from lightning_amd import _ffi
    lib = _ffi.load()
    ctx = ctypes.c_void_p()
    assert lib.lamd_init(ctypes.byref(ctx), 0) == -1


def test_product_never_touches_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may reach oracle/"""
    pkg = os.path.join(ROOT, "lightning_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("isomorphic", ""), os.path.join(dp, f)
                assert "pyref" not in txt


def test_generated_fe_asm_is_current():
    """lightning_amd/csrc/fe_asm.inc is generated: it must be exactly what tools/gen_fe_asm.py writes, and must hold the
    multiply-add counts fe.h documents (99 per multiplication, 63 per squaring, the fused forms)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "gen_fe_asm.py")]).decode()
    cur = open(os.path.join(root, "lightning_amd", "csrc", "fe_asm.inc")).read()
    assert out == cur
    import re
    secs = dict(re.findall(r"#define (LAMD_FE_\w+_ASM) \\\n((?:  \".*\n)+)", cur))
    count = {k: v.count("v_mad_u64_u32 v[") for k, v in secs.items()}
    # 81 + 15 fold + 2 for what column 8 holds before the low chain + 1 at the end (45 + 18 for a square); + 9 for an addend; two products share one fold
    assert count == {"LAMD_FE_MUL_ASM": 99, "LAMD_FE_SQR_ASM": 63, "LAMD_FE_MULADD_ASM": 108, "LAMD_FE_SQRADD_ASM": 72, "LAMD_FE_MUL2_ASM": 180}, count
