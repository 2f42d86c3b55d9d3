#!/usr/bin/env python
"""Registers / scratch / LDS of the kernels in libbyolo.so (code-object metadata): python tools/kernel_regs.py [pattern] [lib]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-yolov3_amd", "byolo", "libbyolo.so")
with tempfile.TemporaryDirectory() as tmp:
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
    for k, a in enumerate(starts):
        part, co = os.path.join(tmp, "b%d" % k), os.path.join(tmp, "d%d.co" % k)
        open(part, "wb").write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+(\S+)", "agpr_count:" + blk))
            name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True, text=True).stdout.strip()
            if pat.search(name):
                print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %5s lds %6s" % (name[:90], f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"),
                                                                             f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
