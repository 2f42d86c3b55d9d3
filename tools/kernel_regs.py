#!/usr/bin/env python
"""Registers / LDS / scratch of every kernel in libbyolo.so (code-object metadata), optionally filtered by a regex on the
demangled name:   python tools/kernel_regs.py [regex]"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.environ.get("BYOLO_LIB") or os.path.join(REPO, "bayesian-yolov3_amd", "byolo", "libbyolo.so")


def main():
    pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
    for k, a in enumerate(starts):
        part, co = os.path.join(tmp, "b%d.bin" % k), os.path.join(tmp, "d%d.co" % k)
        open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if pat and not pat.search(name):
                continue
            agpr = blk.strip().split()[0]
            print("vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s  %s" % (g("vgpr_count"), agpr, g("sgpr_count"), g("group_segment_fixed_size"),
                                                                         g("private_segment_fixed_size"), name[:150]))


if __name__ == "__main__":
    main()
