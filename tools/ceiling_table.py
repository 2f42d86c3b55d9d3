#!/usr/bin/env python
"""The step's lower bound, kernel family by kernel family, beside what was measured.

    python tools/ceiling_table.py profiles/r6_f_per_launch.md --mfma-tflops 1778 --hbm-tbs 4.96 --write-tbs 6.84 [--ms-per-step 21.2]

Input: the per-launch table `bench.py --dump-steps` writes (one row per launch of the convolution stack: layer, variant, EXECUTED
M / N / K, ms).  For every launch

    bound = max( executed fp16 matrix FLOPs / sustained matrix rate ,  algorithmic bytes / sustained HBM rate )

* executed fp16 FLOPs = 3 x 2 M N K for a split-f16 launch (three fp16 products per fp32 product, mfma_pipe.h); the table's M / N / K are
  the executed extents (a Winograd launch: the 16 transform-domain GEMMs incl. tile padding, i.e. 1 / 2.25 of the direct count);
* sustained matrix rate = tools/mfma_f16_power_probe.hip on random operands under the socket's power cap (profiles/r6_ceiling_probe.md),
  NOT the nominal 2 500: no instruction mix on this part runs the fp16 matrix pipe faster than that for longer than a launch;
* bytes = input once + weights once + output once (4 bytes per element: hi/lo pairs), transforms: input + 4 x input written, the
  element-wise finish launches: what they read + write; HBM rate = tools/hbm_probe.py (copy: mixed read/write; fill: pure writes for
  the write-dominated transform).
"""
import argparse
import collections
import re


def family(layer, variant, K, backbone_end=75):
    if variant == 140:
        return "Winograd GEMM + output transform + epilogue (`wino_split_kernel`, 19x19 / 38x38 head 3x3)"
    if variant == -4:
        return "Winograd input transform (`wino_split_input2_kernel`)"
    if variant == 4256:
        return "76x76 head 3x3 + fused 1x1 follower (`conv_igemm_kernel<128,256,1,8,kx3>` + `fused_tail`)"
    if variant == -5:
        return "element-wise finish of the upsampled concat halves (`finish_upsampled_kernel`)"
    if layer < backbone_end:
        return "backbone (52 convolutions on 8 images: stem, 3x3 shared-tap, 1x1 loop, stride-2)"
    return "head 1x1 convolutions, concat halves, detection heads (1x1 loop / general loop)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("per_launch")
    ap.add_argument("--mfma-tflops", type=float, default=1778.0)
    ap.add_argument("--hbm-tbs", type=float, default=4.96)
    ap.add_argument("--write-tbs", type=float, default=6.84)
    ap.add_argument("--ms-per-step", type=float, default=None, help="bench.py's ms_per_step of the same run (wall, pipelined)")
    a = ap.parse_args()
    rows = []
    for line in open(a.per_launch):
        m = re.match(r"\|\s*(\d+)\s*\|\s*(-?\d+)\s*\|\s*(-?\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*(\d+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|", line)
        if m:
            rows.append(tuple(int(x) for x in m.groups()[:6]) + (float(m.group(7)), float(m.group(9)) * 1e12 * float(m.group(7)) * 1e-3))
    fam = collections.OrderedDict()
    for idx, layer, variant, M, N, K, ms, algo in rows:
        if variant == -4:                                   # M = output tiles, N = channels: reads the input (4 pixels per tile), writes V = 16 values per tile
            flops, bytes_, rate = 0.0, M * N * 4 * (4 + 16), a.write_tbs
        elif variant == -5:                                 # M = output pixels, N = channels: reads low (1/4) + partial (1/T) + writes
            flops, bytes_, rate = 0.0, M * N * 4 * 1.3, a.hbm_tbs
        elif variant == -1:                                 # stem (vector FMA): fp32 image in, hi/lo out
            flops, bytes_, rate = 0.0, M * (3 + N) * 4, a.hbm_tbs
        elif variant == 140:                                # M = 16 x P_pad rows, K = C
            flops = 3 * 2.0 * M * N * K
            bytes_, rate = (M * K + 16 * K * N + (M / 16) * 4 * N) * 4, a.hbm_tbs
        else:
            flops = 3 * max(2.0 * M * N * K, algo if variant == 4256 else 0.0)      # (a fused launch also carries its follower's FLOPs)
            cin = K / 9.0 if K % 9 == 0 and K >= 288 else K
            bytes_, rate = (M * cin + K * N + M * N) * 4, a.hbm_tbs
        t_mm = flops / (a.mfma_tflops * 1e12) * 1e3
        t_mem = bytes_ / (rate * 1e12) * 1e3
        f = family(layer, variant, K)
        e = fam.setdefault(f, dict(n=0, ms=0.0, mm=0.0, mem=0.0, bound=0.0, flops=0.0, bytes=0.0))
        e["n"] += 1; e["ms"] += ms; e["mm"] += t_mm; e["mem"] += t_mem; e["bound"] += max(t_mm, t_mem); e["flops"] += flops; e["bytes"] += bytes_
    tot = dict(ms=sum(e["ms"] for e in fam.values()), bound=sum(e["bound"] for e in fam.values()))
    print("| kernel family (launches per step) | executed fp16 TFLOP | bytes GB | matrix bound ms | HBM bound ms | **bound ms** | **measured ms** | gap ms | gap / step |")
    print("|---|---|---|---|---|---|---|---|---|")
    for f, e in fam.items():
        print("| %s (%d) | %.2f | %.2f | %.2f | %.2f | **%.2f** | **%.2f** | %.2f | %.1f %% |"
              % (f, e["n"], e["flops"] / 1e12, e["bytes"] / 1e9, e["mm"], e["mem"], e["bound"], e["ms"], e["ms"] - e["bound"], 100 * (e["ms"] - e["bound"]) / tot["ms"]))
    print("| **sum of the convolution stack's launches** | | | | | **%.2f** | **%.2f** | %.2f | %.1f %% |" % (tot["bound"], tot["ms"], tot["ms"] - tot["bound"], 100 * (tot["ms"] - tot["bound"]) / tot["ms"]))
    if a.ms_per_step:
        print("| wall - sum: the tail (0.6 ms) runs under the next step's convolutions, and unprofiled steps run their backbone beside the previous step's heads (the sum is of a profiled step, which runs alone) | | | | | 0 | %.2f | | |" % (a.ms_per_step - tot["ms"]))
        print("| **`ms_per_step`** (wall) | | | | | **%.2f** | **%.2f** | | |" % (tot["bound"], a.ms_per_step))


if __name__ == "__main__":
    main()
