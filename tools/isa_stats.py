#!/usr/bin/env python3
"""Static instruction statistics of the device code in liblightning_amd.so (or a variant): per kernel the number of instructions,
VGPR/SGPR/scratch from the metadata notes, and a histogram of mnemonics.  usage: tools/isa_stats.py [lib.so] [kernel-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lightning_amd", "liblightning_amd.so")
pat = [a for a in sys.argv[1:] if not a.endswith(".so")]
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co]).decode()
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
meta = {}
for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
    meta[m.group(1)] = (int(m.group(4)), int(m.group(3)), int(m.group(2)))
cur, kern = None, collections.OrderedDict()
for line in asm.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        cur = m.group(1)
        kern[cur] = collections.Counter()
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if m and cur:
        kern[cur][m.group(1)] += 1
for k, c in kern.items():
    if pat and not any(p in k for p in pat):
        continue
    tot = sum(c.values())
    v = meta.get(k, ("?", "?", "?"))
    valu = sum(n for i, n in c.items() if i.startswith("v_"))
    print("%-70s instr %6d  valu %6d  mad64 %5d  vgpr %s sgpr %s scratch %s" % (k[:70], tot, valu, c["v_mad_u64_u32"], v[0], v[1], v[2]))
    if pat:
        for i, n in c.most_common(28):
            print("     %-28s %6d" % (i, n))
