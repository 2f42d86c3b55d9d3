"""Build libbyolo.so (gfx950 only) in-tree: bayesian-yolov3_amd/byolo/libbyolo.so.

    python bayesian-yolov3_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "byolo", "libbyolo.so")
SRCS = ["byolo_api.hip", "byolo_pack.hip", "byolo_plan.hip", "conv_igemm.hip", "conv_kernels.hip", "winograd.hip", "gemm_stream.hip", "wino_fused.hip", "wino_split.hip", "tail_kernels.hip",
        "train_kernels.hip", "host_io.cpp"]
HDRS = ["byolo_kernels.h", "byolo_internal.h", "byolo_rng.h", "mfma_pipe.h", "epilogue.h", os.path.join("..", "..", "include", "byolo.h")]
DEPS = SRCS + HDRS
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result"]
LIBS = ["-lz"]          # host_io.cpp: the PNG decoder inflates with zlib
# train_kernels.hip restates float32 arithmetic operation by operation (bit-identical masks against the float32 oracle): no
# fused multiply-add, correctly rounded division
FILE_FLAGS = {"train_kernels.hip": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]}


def build(force=False, verbose=False, ablate=0, ablate_wf=0, ablate_ws=0, jobs=None):
    """One object per source, compiled in parallel (hipcc is a process per file anyway), linked into the shared library.
    Objects are reused when newer than their source and every header (force: nothing is reused)."""
    # ablate: timing-ablation build of the convolution K loop (conv_igemm.hip), written next to the product
    # library as libbyolo_abl<N>.so and loaded with BYOLO_LIB=<path>; never the default
    tag = ""
    out = os.path.abspath(OUT)
    if ablate:
        tag = "abl%d" % ablate
    if ablate_wf:       # same for the fused Winograd kernel (wino_fused.hip): libbyolo_wf<N>.so
        tag = "wf%d" % ablate_wf
    if ablate_ws:       # and for the split-arithmetic Winograd kernel (wino_split.hip): libbyolo_ws<N>.so
        tag = "ws%d" % ablate_ws
    if tag:
        out = os.path.abspath(OUT.replace("libbyolo.so", "libbyolo_%s.so" % tag))
    newest = max(os.path.getmtime(os.path.join(HERE, d)) for d in DEPS)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    defs = (["-DBYOLO_CONV_ABLATE=%d" % ablate] if ablate else []) + (["-DBYOLO_WF_ABLATE=%d" % ablate_wf] if ablate_wf else []) + \
           (["-DBYOLO_WS_ABLATE=%d" % ablate_ws] if ablate_ws else [])
    objdir = os.path.join(HERE, "build", tag or "main")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(HERE, d)) for d in HDRS + ["build.py"])
    todo, objs = [], []
    for s in SRCS:
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_time, os.path.getmtime(os.path.join(HERE, s))):
            todo.append([hipcc] + FLAGS + FILE_FLAGS.get(s, []) + defs + ["-c", os.path.join(HERE, s), "-o", obj])
    jobs = jobs or int(os.environ.get("BYOLO_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 1)
    running, failed = [], None
    for cmd in todo + [None] * jobs:
        while len(running) >= jobs or (cmd is None and running):
            pr, c = running.pop(0)
            if pr.wait() != 0 and failed is None:
                failed = c
        if cmd is None or failed:
            continue
        if verbose:
            print(" ".join(cmd), flush=True)
        running.append((subprocess.Popen(cmd), cmd))
    for pr, c in running:
        if pr.wait() != 0 and failed is None:
            failed = c
    if failed:
        raise subprocess.CalledProcessError(1, failed)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + LIBS + ["-o", out]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    abl = int(sys.argv[sys.argv.index("--ablate") + 1]) if "--ablate" in sys.argv else 0
    wf = int(sys.argv[sys.argv.index("--ablate-wf") + 1]) if "--ablate-wf" in sys.argv else 0
    wsa = int(sys.argv[sys.argv.index("--ablate-ws") + 1]) if "--ablate-ws" in sys.argv else 0
    print(build(force="--force" in sys.argv, verbose=True, ablate=abl, ablate_wf=wf, ablate_ws=wsa))
