#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of a built liblightning_amd.so (from the gfx950 code object's metadata).
usage: tools/kernel_resources.py [path/to/lib.so]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lightning_amd", "liblightning_amd.so")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    # llvm-objcopy with no output operand rewrites its INPUT in place (round 4: running this tool changed the shipped library's bytes while its
    # build stamp still said "fresh"): work on a copy, send the rewritten object to the bit bucket
    import shutil
    tmp_so = os.path.join(d, "lib.so")
    shutil.copyfile(so, tmp_so)
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, tmp_so, os.path.join(d, "discard.so")], stderr=subprocess.DEVNULL)
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([LLVM + "llvm-readelf", "--notes", co]).decode()
    if len(sys.argv) > 2:
        subprocess.check_call([LLVM + "llvm-objdump", "-d", co], stdout=open(sys.argv[2], "w"))
rows = []
for blk in notes.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = subprocess.check_output(["c++filt", g("name").group(1)]).decode().strip().split("(")[0]
    v, a = int(g("vgpr_count").group(1)), int(blk.split()[0])
    rows.append((name, v, a, int(g("sgpr_count").group(1)), int(g("private_segment_fixed_size").group(1)), int(g("group_segment_fixed_size").group(1))))
print("%-44s %5s %5s %5s %8s %6s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "waves"))
for r in sorted(rows):
    print("%-44s %5d %5d %5d %8d %6d %6d" % (r[0][:44], r[1], r[2], r[3], r[4], r[5], min(8, 512 // max(1, r[1]))))
