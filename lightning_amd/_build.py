"""Builds liblightning_amd.so (+ the host mirror and the test-traffic library) in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit under lightning_amd/build/, then one link: a change to the multi-device layer (lamd_multi.cpp, seconds) does
not recompile the kernels (lamd_engine.hip, minutes).  Staleness is decided by CONTENT: every artefact has a stamp file holding the sha256 of
its sources, its flags and its inputs' stamps; file times play no part (a snapshot copied to the GPU box keeps the stamps next to the
libraries, so nothing is rebuilt there)."""
import hashlib
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "liblightning_amd.so")
INC = os.path.join(ROOT, "include")
_H = lambda *names: [os.path.join(CSRC, f) for f in names]
_I = lambda *names: [os.path.join(INC, f) for f in names]

# -amdgpu-codegenprepare-mul24=false: ROCm 7.2's AMDGPUCodeGenPrepare mul24 rewrite miscompiles the fused
# 64-bit column arithmetic of fe.h (a^2*a came out wrong on gfx950; reproduced instruction-for-instruction by
# an ISA emulator, so it is a code-generation bug, not a hardware hazard -- DESIGN.md "toolchain notes").
# lamd_selftest() re-checks every primitive on the device against the host evaluation of the same code.
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wno-unknown-pragmas", "-mllvm", "-amdgpu-codegenprepare-mul24=false"]
EXTRA = os.environ.get("LAMD_BUILD_FLAGS", "").split()

# translation units of liblightning_amd.so: (source, everything it includes from this tree, flags)
ENGINE_TUS = [
    ("lamd_engine.hip", _H("verify_core.h", "group.h", "fe.h", "fe_asm.inc", "fuzz.h", "scalar.h", "sha256.h", "lamd_common.h", "bolt12.h") +
     _I("lightning_amd.h", "lightning_amd_debug.h"), HIPFLAGS + EXTRA),
    ("lamd_multi.cpp", _H("numa_cpus.h") + _I("lightning_amd.h", "lightning_amd_debug.h"), ["-O2", "-std=c++17", "-fPIC", "-pthread", "-Wall"]),
]
SOURCES = sorted({os.path.join(CSRC, tu[0]) for tu in ENGINE_TUS} | {h for tu in ENGINE_TUS for h in tu[1]})

# test / bench infrastructure: the synthetic-workload signer kernels (include/lightning_amd_testgen.h)
TESTGEN = os.path.join(PKG, "liblightning_amd_testgen.so")
TESTGEN_SOURCES = _H("lamd_testgen.hip", "verify_core.h", "group.h", "fe.h", "fe_asm.inc", "scalar.h", "sha256.h", "lamd_common.h") + _I("lightning_amd_testgen.h")

SHIM = os.path.join(PKG, "liblightning_amd_cln.so")
SHIM_CPP = _H("cln_shim.cpp", "gossip_ingest.cpp")
SHIM_SOURCES = SHIM_CPP + _I("cln_shim.h", "lightning_amd_gossipd.h", "lightning_amd.h") + _H("verify_core.h")


def _digest(files, extra=()):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _fresh(artefact, digest):
    try:
        return os.path.exists(artefact) and open(artefact + ".stamp").read().strip() == digest
    except OSError:
        return False


def _stamp(artefact, digest):
    with open(artefact + ".stamp", "w") as f:
        f.write(digest + "\n")


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _engine_digests():
    return [(src, _digest([os.path.join(CSRC, src)] + deps, flags), flags) for src, deps, flags in ENGINE_TUS]


def is_stale():
    tus = _engine_digests()
    return not _fresh(LIB, _digest([], [d for _, d, _ in tus]))


def build_shim(force=False):
    """host-side C++ mirror of the reference prototypes + the gossip ingest (links against liblightning_amd.so)"""
    digest = _digest(SHIM_SOURCES, ["shim-v2"])
    if not force and _fresh(SHIM, digest) and os.path.exists(LIB):
        return SHIM
    cxx = os.environ.get("CXX", "g++")
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function", "-o", SHIM + ".tmp"] + SHIM_CPP +
                          ["-L" + PKG, "-llightning_amd", "-Wl,-rpath,$ORIGIN"])
    os.replace(SHIM + ".tmp", SHIM)
    _stamp(SHIM, digest)
    return SHIM


def build_testgen(force=False):
    digest = _digest(TESTGEN_SOURCES, HIPFLAGS)
    if not force and _fresh(TESTGEN, digest) and os.path.exists(LIB):
        return TESTGEN
    subprocess.check_call([_hipcc()] + [f for f in HIPFLAGS if f != "-pthread"] + ["-shared", "-o", TESTGEN + ".tmp", os.path.join(CSRC, "lamd_testgen.hip"),
                           "-L" + PKG, "-llightning_amd", "-Wl,-rpath,$ORIGIN"])
    os.replace(TESTGEN + ".tmp", TESTGEN)
    _stamp(TESTGEN, digest)
    return TESTGEN


# the shared-service front (include/lightning_amd_served.h): host code only -- the server binds the engine through dlopen, the client speaks
# to the server; liblightning_amd_cln_client.so is the mirror linked against the CLIENT instead of the engine
SERVED = os.path.join(PKG, "lamd_served")
CLIENT = os.path.join(PKG, "liblightning_amd_client.so")
SHIM_CLIENT = os.path.join(PKG, "liblightning_amd_cln_client.so")
SERVED_SOURCES = _H("lamd_served.cpp", "served_common.h", "numa_cpus.h") + _I("lightning_amd_served.h", "lightning_amd.h")
CLIENT_SOURCES = _H("lamd_client.cpp", "served_common.h") + _I("lightning_amd_served.h", "lightning_amd.h")


def build_served(force=False):
    """lamd_served + liblightning_amd_client.so + liblightning_amd_cln_client.so (g++, seconds; no device code)"""
    cxx = os.environ.get("CXX", "g++")
    flags = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function"]
    d = _digest(SERVED_SOURCES, ["served-v1"])
    if force or not _fresh(SERVED, d):
        subprocess.check_call([cxx] + flags + ["-o", SERVED + ".tmp", os.path.join(CSRC, "lamd_served.cpp"), "-ldl"])
        os.replace(SERVED + ".tmp", SERVED)
        _stamp(SERVED, d)
    d = _digest(CLIENT_SOURCES, ["client-v2"])
    if force or not _fresh(CLIENT, d):
        # -Bsymbolic-functions: the library's own calls (lamd_check_signed_hash -> lamd_verify_ecdsa_batch) bind inside it even in a process that has the
        # engine library's symbols of the same names in its global scope
        subprocess.check_call([cxx] + flags + ["-shared", "-Wl,-Bsymbolic-functions", "-o", CLIENT + ".tmp", os.path.join(CSRC, "lamd_client.cpp")])
        os.replace(CLIENT + ".tmp", CLIENT)
        _stamp(CLIENT, d)
    d = _digest(SHIM_SOURCES + CLIENT_SOURCES, ["shim-client-v1"])
    if force or not _fresh(SHIM_CLIENT, d):
        subprocess.check_call([cxx] + flags + ["-shared", "-o", SHIM_CLIENT + ".tmp"] + SHIM_CPP + ["-L" + PKG, "-llightning_amd_client", "-Wl,-rpath,$ORIGIN"])
        os.replace(SHIM_CLIENT + ".tmp", SHIM_CLIENT)
        _stamp(SHIM_CLIENT, d)
    return SERVED, CLIENT, SHIM_CLIENT


def build(force=False, verbose=False):
    tus = _engine_digests()
    link_digest = _digest([], [d for _, d, _ in tus])
    if not force and _fresh(LIB, link_digest):
        build_shim()
        build_testgen()
        build_served()
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    for src, digest, flags in tus:
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if not force and _fresh(obj, digest):
            continue
        cmd = [_hipcc()] + flags + ["-c", "-o", obj + ".tmp", os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)
        _stamp(obj, digest)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB + ".tmp"] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    _stamp(LIB, link_digest)
    build_shim(force=True)
    build_testgen()
    build_served()
    return LIB
