"""The arithmetic of the default precision (split-f16, csrc/mfma_pipe.h), emulated on the CPU around the oracle:

    every activation and weight as hi = RNE_f16(s x), lo = RNE_f16(s x - hi)   (s = a power of two: 4 for activations,
    per layer for weights: largest |w| in [2^13, 2^14)),  x * w := hi_x hi_w + hi_x lo_w + lo_x hi_w  in fp32

against the same network in float64 and in plain float32 (the reference's precision: TensorFlow float32 kernels,
lib_yolo/layers.py:550).  The emulation replaces `_conv2d` of oracle/cpu_ref.py (products of fp16 values are exact
in fp32, so three float32 convolutions on fp16-valued tensors reproduce the products; the device additionally sums the
16 products of one MFMA before rounding, which can only be more accurate) and rounds every activation to hi + lo after
the activation function, as the epilogue does.

Claim checked here (and quoted in DESIGN.md section 5): measured against the float64 run, the split-f16 network is as
accurate as the float32 one -- both sit at 0.5 .. 1.0 of the contract's bound 1e-4 * max(1, |ref|) (the worst values are
variances over few MC samples and exp(logvar) columns; the fewer samples, the closer to 1), so two implementations of
float32 grade may differ from EACH OTHER by about the bound at T <= 3 (1.05 at 320x320 T=2, 1.04 at 416x416 T=3,
0.67 at 608x608 T=4 in this emulation; on the device every parity test against the float32 oracle holds the bound --
the worst is 0.52 -- tests/test_gpu_*.py).  Neither a fourth product (lo * lo) nor an fp32 detection convolution
changes that figure: it is the distance between two float32-grade evaluations, not a defect of one of them.

    BYOLO_EMU_SIZE=608 BYOLO_EMU_T=4 python -m pytest tests/test_split_numerics.py -s     (the table under profiles/)
"""
import math
import os
import sys

import numpy as np
import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))

ACT_SCALE = 4.0


def _split(a, scale):
    import torch
    a = a * scale
    hi = a.half().float()
    lo = (a - hi).half().float()
    return hi / scale, lo / scale


def _wino_split_conv(x, w):
    """One 3x3 / stride-1 convolution as Winograd F(2x2,3x3) in split-f16 arithmetic, as a fused device kernel would run it:
    V = B^T d B in fp32 from the decoded hi + lo input, stored as hi/lo pairs (scale 1: |V| <= 4 |d|, the same fp16 range as the
    activations' 4 * value); U = G g G^T in double, rounded once, one power-of-two scale per output channel, hi/lo pairs; the 16
    transform-domain products x_hi u_hi + x_hi u_lo + x_lo u_hi accumulated in fp32 over the input channels; Y = A^T M A in fp32."""
    import torch
    import torch.nn.functional as F
    S, H, W, C = x.shape
    N = w.shape[3]
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1 + 2 * tw - W, 1, 1 + 2 * th - H))
    p = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [S, C, th, tw, 4, 4]
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    V = torch.einsum("ij,sctujk,lk->sctuil", Bt, p, Bt)                      # fp32 adds of exactly representable inputs
    U = torch.einsum("ij,jkcn,lk->ilcn", G, w.double(), G).float()           # [4, 4, C, N]
    ws = torch.exp2(13 - torch.floor(torch.log2(U.abs().amax(dim=(0, 1, 2)).clamp(min=1e-30))))
    Vh, Vl = _split(V, 1.0)
    Uh, Ul = _split(U, ws)
    M = (torch.einsum("sctuil,ilcn->sntuil", Vh, Uh) + (torch.einsum("sctuil,ilcn->sntuil", Vh, Ul) + torch.einsum("sctuil,ilcn->sntuil", Vl, Uh)))
    Y = torch.einsum("ai,sntuil,bl->sntaub", At, M, At)                      # [S, N, th, 2, tw, 2]
    return Y.reshape(S, N, 2 * th, 2 * tw)[:, :, :H, :W].permute(0, 2, 3, 1).contiguous()


def _wino1d_split_conv(x, w):
    """One 3x3 / stride-1 convolution as ONE-DIMENSIONAL Winograd F(2,3) along W with the three filter rows kept direct (VERDICT r4
    item 3, for the 128-channel 76x76 head convolutions where the 4x input expansion of F(2x2,3x3) does not pay): per output pair
    (x, x+1) of a row and per filter row ky, V[xi] = B^T d over the four input pixels x-1 .. x+2 of row y+ky-1 (fp32 adds of the
    decoded hi + lo values, stored as hi/lo pairs at scale 2: |V| <= 2 |d|), U[xi][ky] = G g[ky, :] in double, rounded once, one
    power of two per output channel; M[xi] = sum over (ky, c) of the three split products, fp32; Y = A^T M.  12 products per output
    pair and channel pair instead of 18."""
    import torch
    import torch.nn.functional as F
    S, H, W, C = x.shape
    N = w.shape[3]
    tw = (W + 1) // 2
    xp = F.pad(x.permute(0, 3, 1, 2), (1, 1 + 2 * tw - W, 1, 1))             # [S, C, H+2, 2 tw + 2]
    p = xp.unfold(3, 4, 2)                                                   # [S, C, H+2, tw, 4]
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    V = torch.einsum("ij,schuj->schui", Bt, p)                               # [S, C, H+2, tw, 4]: fp32 adds of exactly representable inputs
    U = torch.einsum("ij,kjcn->kicn", G, w.double()).float()                 # [ky, xi, C, N]
    ws = torch.exp2(13 - torch.floor(torch.log2(U.abs().amax(dim=(0, 1, 2)).clamp(min=1e-30))))
    Vh, Vl = _split(V, 2.0)
    Uh, Ul = _split(U, ws)
    M = 0
    for ky in range(3):                                                      # the filter rows: K = (ky, c)
        vh, vl = Vh[:, :, ky:ky + H], Vl[:, :, ky:ky + H]
        M = M + (torch.einsum("schui,icn->snhui", vh, Uh[ky]) + (torch.einsum("schui,icn->snhui", vh, Ul[ky]) + torch.einsum("schui,icn->snhui", vl, Uh[ky])))
    Y = torch.einsum("ai,snhui->snhua", At, M)                               # [S, N, H, tw, 2]
    return Y.reshape(S, N, H, 2 * tw)[:, :, :, :W].permute(0, 2, 3, 1).contiguous()


def _run(H, W, T, wino=False):
    import torch
    from oracle import cpu_ref
    from byolo import synth
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    variant = "bayesian_yolov3_aleatoric"
    params = synth.base_params(cpu_ref.variable_shapes(variant, 2), variant, 2, seed=7)
    imgs = synth.synthetic_images(1, H, W, seed=1234)
    tp = cpu_ref.to_torch_params(params)
    cpu_ref.forward(tp, imgs, variant, T=1, calibrate=True)
    tp64 = {k: v.double() for k, v in tp.items()}
    orig_conv, orig_leaky = cpu_ref._conv2d, cpu_ref._leaky
    mode = {"split": False}

    def conv(x, w, stride):
        if x.dtype != torch.float32 or not mode["split"]:
            return orig_conv(x, w, stride)
        if x.shape[3] == 3:
            return orig_conv(x, w, stride)                             # the stem reads the fp32 image as it is
        if wino == "1d" and w.shape[0] == 3 and stride == 1 and x.shape[3] == 128:
            return _wino1d_split_conv(x, w)                            # the stride-8 head convolutions (128 -> 256 channels)
        if wino == "1d" and w.shape[0] == 3 and stride == 1 and x.shape[3] >= 256 and w.shape[3] >= 256:
            return _wino_split_conv(x, w)                              # ... beside what the product's plan already transforms in 2-D
        if wino is True and w.shape[0] == 3 and stride == 1 and x.shape[3] >= 64:
            return _wino_split_conv(x, w)
        # one power of two per output channel (byolo_finalize): the channel's largest |w'| in [2^13, 2^14)
        ws = torch.exp2(13 - torch.floor(torch.log2(w.abs().amax(dim=(0, 1, 2)).clamp(min=1e-30))))
        xh, xl = _split(x, ACT_SCALE)
        wh, wl = _split(w, ws)
        return orig_conv(xh, wh, stride) + (orig_conv(xh, wl, stride) + orig_conv(xl, wh, stride))

    def leaky(x):
        y = orig_leaky(x)
        if x.dtype == torch.float32 and mode["split"]:
            hi, lo = _split(y, ACT_SCALE)
            y = hi + lo
        return y

    cpu_ref._conv2d, cpu_ref._leaky = conv, leaky
    try:
        with torch.no_grad():
            ref64, _ = cpu_ref.detect_boxes(tp64, imgs, variant, T=T, seed=1000, dtype=torch.float64)
            f32, _ = cpu_ref.detect_boxes(tp, imgs, variant, T=T, seed=1000)
            mode["split"] = True
            spl, _ = cpu_ref.detect_boxes(tp, imgs, variant, T=T, seed=1000)
    finally:
        cpu_ref._conv2d, cpu_ref._leaky = orig_conv, orig_leaky

    def worst(a, b):          # largest |a - b| in units of the bound 1e-4 * max(1, |b|), NaN / inf positions excluded
        a, b = a.double(), b.double()
        r = (a - b).abs() / (1e-4 * torch.clamp(b.abs(), min=1.0))
        return float(torch.where(torch.isfinite(r), r, torch.zeros_like(r)).max())
    return dict(f32_vs_f64=worst(f32, ref64), split_vs_f64=worst(spl, ref64), split_vs_f32=worst(spl, f32))


def test_split_f16_is_float32_grade():
    size = int(os.environ.get("BYOLO_EMU_SIZE", "320"))
    T = int(os.environ.get("BYOLO_EMU_T", "2"))
    r = _run(size, size, T)
    print("\n%dx%d, T=%d, worst value in units of the bound 1e-4*max(1,|ref|): float32 vs float64 %.3f | split-f16 vs float64 %.3f | "
          "split-f16 vs float32 %.3f" % (size, size, T, r["f32_vs_f64"], r["split_vs_f64"], r["split_vs_f32"]))
    assert r["split_vs_f64"] < 1.0                                         # inside the contract's bound of the exact result
    assert r["split_vs_f64"] < 1.3 * r["f32_vs_f64"] + 0.05               # and no further from it than float32 is
    assert r["split_vs_f32"] <= r["split_vs_f64"] + r["f32_vs_f64"] + 1e-6  # (the two float32-grade runs: triangle inequality)


def test_winograd_in_split_arithmetic_is_float32_grade_too():
    """VERDICT r2 item 4 asks for this BEFORE any kernel: F(2x2,3x3) on every 3x3 / stride-1 convolution with >= 64 input
    channels, in split-f16 arithmetic, against float64 and float32 -- only worth building if the rows stay inside the bound."""
    size = int(os.environ.get("BYOLO_EMU_SIZE", "320"))
    T = int(os.environ.get("BYOLO_EMU_T", "2"))
    r = _run(size, size, T, wino=True)
    print("\nWinograd F(2x2,3x3) in split-f16, %dx%d, T=%d, worst value in units of the bound: float32 vs float64 %.3f | split-f16 + Winograd vs "
          "float64 %.3f | vs float32 %.3f" % (size, size, T, r["f32_vs_f64"], r["split_vs_f64"], r["split_vs_f32"]))
    assert r["split_vs_f64"] < 1.0
    assert r["split_vs_f64"] < 1.3 * r["f32_vs_f64"] + 0.05


def test_one_dimensional_winograd_in_split_arithmetic():
    """1-D Winograd F(2,3) along W (three filter rows direct) on the 128-channel head convolutions, beside the
    2-D form on the 256- / 512-channel ones as the product's plan has them -- emulated before any kernel.  Kill criterion of the
    experiment: the rows must stay within 0.5 of the bound of float64 at 608x608, T = 30 (BYOLO_EMU_SIZE=608 BYOLO_EMU_T=30;
    measured, profiles/r5_wino1d.md); the default size of the suite checks float32 grade like the tests above."""
    import torch
    size = int(os.environ.get("BYOLO_EMU_SIZE", "320"))
    T = int(os.environ.get("BYOLO_EMU_T", "2"))
    # the algebra first: the emulated convolution equals the direct one in float64 up to the split rounding
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 9, 11, 128, generator=g)
    w = torch.randn(3, 3, 128, 32, generator=g) / 30
    from oracle import cpu_ref
    ref = cpu_ref._conv2d(x.double(), w.double(), 1)
    got = _wino1d_split_conv(x, w)
    assert float((got.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    r = _run(size, size, T, wino="1d")
    print("\n1-D Winograd F(2,3) on the 128-channel 3x3 layers + F(2x2,3x3) on the larger ones, split-f16, %dx%d, T=%d, worst value in units of the "
          "bound: float32 vs float64 %.3f | split-f16 + Winograd vs float64 %.3f | vs float32 %.3f" % (size, size, T, r["f32_vs_f64"], r["split_vs_f64"], r["split_vs_f32"]))
    assert r["split_vs_f64"] < 1.0
    assert r["split_vs_f64"] < 1.3 * r["f32_vs_f64"] + 0.05
