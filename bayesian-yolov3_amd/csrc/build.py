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
SRCS = ["byolo_api.hip", "conv_igemm.hip", "conv_kernels.hip", "winograd.hip", "gemm_stream.hip", "wino_fused.hip", "wino_split.hip", "tail_kernels.hip"]
DEPS = SRCS + ["byolo_kernels.h", "byolo_rng.h", "mfma_pipe.h", "epilogue.h", os.path.join("..", "..", "include", "byolo.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result"]


def build(force=False, verbose=False, ablate=0, ablate_wf=0, ablate_ws=0):
    # ablate: timing-ablation build of the convolution K loop (conv_igemm.hip), written next to the product
    # library as libbyolo_abl<N>.so and loaded with BYOLO_LIB=<path>; never the default
    out = os.path.abspath(OUT if not ablate else OUT.replace("libbyolo.so", "libbyolo_abl%d.so" % ablate))
    if ablate_wf:       # same for the fused Winograd kernel (wino_fused.hip): libbyolo_wf<N>.so
        out = os.path.abspath(OUT.replace("libbyolo.so", "libbyolo_wf%d.so" % ablate_wf))
    if ablate_ws:       # and for the split-arithmetic Winograd kernel (wino_split.hip): libbyolo_ws<N>.so
        out = os.path.abspath(OUT.replace("libbyolo.so", "libbyolo_ws%d.so" % ablate_ws))
    newest = max(os.path.getmtime(os.path.join(HERE, d)) for d in DEPS)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + (["-DBYOLO_CONV_ABLATE=%d" % ablate] if ablate else []) + (["-DBYOLO_WF_ABLATE=%d" % ablate_wf] if ablate_wf else []) + (["-DBYOLO_WS_ABLATE=%d" % ablate_ws] if ablate_ws else []) + [os.path.join(HERE, s) for s in SRCS] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    abl = int(sys.argv[sys.argv.index("--ablate") + 1]) if "--ablate" in sys.argv else 0
    wf = int(sys.argv[sys.argv.index("--ablate-wf") + 1]) if "--ablate-wf" in sys.argv else 0
    wsa = int(sys.argv[sys.argv.index("--ablate-ws") + 1]) if "--ablate-ws" in sys.argv else 0
    print(build(force="--force" in sys.argv, verbose=True, ablate=abl, ablate_wf=wf, ablate_ws=wsa))
