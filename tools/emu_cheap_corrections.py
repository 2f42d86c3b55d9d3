#!/usr/bin/env python
"""Numerics study for DESIGN.md section 9 (no kernel behind it): what if the two CORRECTION products of the split arithmetic
(hi_x lo_w + lo_x hi_w, each ~2^-11 of the main product) ran on the 8-bit matrix instructions (twice the fp16 rate on gfx950)?

    x * w ~= hi_x hi_w  [fp16 x fp16, as today]  +  q(hi_x) q(lo_w) + q(lo_x) q(hi_w)  [q = an 8-bit float format]

Emulated on the CPU around the oracle exactly like tests/test_split_numerics.py (whose helpers this imports): the worst pre-NMS value in
units of the contract's bound 1e-4 * max(1, |ref|) against the float64 run.  Matrix work per product would drop from 3 fp16 units to
1 + 2 x 0.5 = 2.

    python tools/emu_cheap_corrections.py [--size 320] [--T 2]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def q8(t, fmt, per_channel_dim=None):
    """Round to an 8-bit float format after a power-of-two scale that puts the largest magnitude (per tensor, or per slice of
    `per_channel_dim`) at the top of the format's range -- what a packed operand plane with one exponent per channel / tensor would hold."""
    import torch
    dt, top = {"e5m2": (torch.float8_e5m2, 2.0 ** 15), "e4m3": (torch.float8_e4m3fn, 2.0 ** 8)}[fmt]
    if per_channel_dim is None:
        m = t.abs().amax().clamp(min=1e-30)
    else:
        dims = [d for d in range(t.dim()) if d != per_channel_dim]
        m = t.abs().amax(dim=dims, keepdim=True).clamp(min=1e-30)
    s = torch.exp2(torch.floor(torch.log2(top / m)))
    return (t * s).to(dt).float() / s


def run(size, T, variant_name):
    import torch
    import test_split_numerics as tsn
    from oracle import cpu_ref
    from byolo import synth
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    variant = "bayesian_yolov3_aleatoric"
    params = synth.base_params(cpu_ref.variable_shapes(variant, 2), variant, 2, seed=7)
    imgs = synth.synthetic_images(1, size, size, seed=1234)
    tp = cpu_ref.to_torch_params(params)
    cpu_ref.forward(tp, imgs, variant, T=1, calibrate=True)
    tp64 = {k: v.double() for k, v in tp.items()}
    orig_conv, orig_leaky = cpu_ref._conv2d, cpu_ref._leaky
    mode = {"split": False}

    def conv(x, w, stride):
        if x.dtype != torch.float32 or not mode["split"] or x.shape[3] == 3:
            return orig_conv(x, w, stride)
        ws = torch.exp2(13 - torch.floor(torch.log2(w.abs().amax(dim=(0, 1, 2)).clamp(min=1e-30))))
        xh, xl = tsn._split(x, tsn.ACT_SCALE)
        wh, wl = tsn._split(w, ws)
        main = orig_conv(xh, wh, stride)
        if variant_name == "three fp16 products (the product today)":
            return main + (orig_conv(xh, wl, stride) + orig_conv(xl, wh, stride))
        if variant_name == "hi * hi only":
            return main
        fmt = "e5m2" if "e5m2" in variant_name else "e4m3"
        # activations: one exponent per tensor (a plane written by the producing epilogue); weights: one per output channel (packed on the host)
        return main + (orig_conv(q8(xh, fmt), q8(wl, fmt, 3), stride) + orig_conv(q8(xl, fmt), q8(wh, fmt, 3), stride))

    def leaky(x):
        y = orig_leaky(x)
        if x.dtype == torch.float32 and mode["split"]:
            hi, lo = tsn._split(y, tsn.ACT_SCALE)
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

    def worst(a, b):
        a, b = a.double(), b.double()
        r = (a - b).abs() / (1e-4 * torch.clamp(b.abs(), min=1.0))
        return float(torch.where(torch.isfinite(r), r, torch.zeros_like(r)).max())
    return worst(f32, ref64), worst(spl, ref64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--T", type=int, default=2)
    a = ap.parse_args()
    print("| arithmetic of a product (%dx%d, T=%d) | matrix work, fp16 units | worst value vs float64, units of the bound | float32 CPU run vs float64 |" % (a.size, a.size, a.T))
    print("|---|---|---|---|")
    for name, cost in (("three fp16 products (the product today)", "3"), ("hi * hi + the two corrections on e4m3 operands (scaled per tensor / per output channel)", "2"),
                       ("hi * hi + the two corrections on e5m2 operands", "2"), ("hi * hi only", "1")):
        f, s = run(a.size, a.T, name)
        print("| %s | %s | %.3f | %.3f |" % (name, cost, s, f), flush=True)


if __name__ == "__main__":
    main()
