#!/usr/bin/env python
"""soak_parity.py -- randomised parity sweep on the GPU box: N random configurations (variant, class count, image
size in multiples of 32, batch, T, NMS mode, kernel-selection knobs) through the product path, every one against the
CPU restatement (pre-NMS rows within 1e-4 abs / rel of its float64 run, or within twice the float32 restatement's own
loss where the quantity is ill-conditioned) and the oracle NMS (kept indices and rows bit-exact).

    python tools/soak_parity.py --cases 60 --seed 1 > gpurun_out/soak.md

The fixed cases live in tests/; this is the wide net (odd grids for the Winograd tile padding, split-K shapes,
channel counts of other class counts, batches that do not divide anything).  Output: one markdown row per case.
Test infrastructure: imports oracle/ as the checker."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

VARIANTS = ("yolov3", "yolov3_aleatoric", "bayesian_yolov3_aleatoric")


def one_case(rng, idx, a):
    import torch
    from conftest import build_model, assert_close
    from byolo import synth
    from oracle import cpu_ref
    variant = VARIANTS[rng.integers(0, 3)]
    cls_cnt = int(rng.choice([1, 2, 2, 3, 5, 20, 80]))
    bayes = variant.startswith("bayes")
    while True:                                 # bounded CPU work per case (the float64 restatement is the slow side)
        H, W = int(rng.integers(1, a.max_cells + 1)) * 32, int(rng.integers(1, a.max_cells + 1)) * 32
        T = int(rng.integers(1, a.max_T + 1)) if bayes else 1
        B = int(rng.integers(1, a.max_B + 1))
        if (H // 32) * (W // 32) * B * max(T, 3) <= a.budget:
            break
    nms_mode = int(rng.integers(0, 2)) if cls_cnt == 2 else 0      # the 2-class mode is defined for C = 2
    env = {"BYOLO_WINOGRAD": str(rng.choice(["", "0", "2"])), "BYOLO_WINO_FUSED": str(rng.choice(["", "0", "2"])),
           "BYOLO_KSPLIT": str(rng.choice(["", "", "0", "2", "3", "5"])),
           "BYOLO_WINO_CHUNK_MB": str(rng.choice(["", "", "1", "8", "64"])),       # small budgets: many chunks per layer
           "BYOLO_STREAMK": str(rng.choice(["", "", "0", "2"])),                   # stream-K never / on every launch
           "BYOLO_STREAM1X1": str(rng.choice(["", "", "0", "2"])),                 # row-streaming 1x1 kernel never / wherever expressible
           "BYOLO_PRECISION": str(rng.choice(["", "", "f32"])),                    # default (split-f16) twice as often as the fp32 mode
           # round 3, default precision: Winograd in split arithmetic never / planner / every eligible layer, its workgroup shapes and
           # chunk budget (many chunks, odd tile paddings), the 1x1 loop on / off
           "BYOLO_WINO_SPLIT": str(rng.choice(["", "0", "2", "2"])), "BYOLO_WINO_SPLIT_PERSIST": str(rng.choice(["", "0", "1", "2"])),
           "BYOLO_WINO_SPLIT_BN": str(rng.choice(["", "128", "256"])), "BYOLO_WINO_SPLIT_CHUNK_MB": str(rng.choice(["", "1", "4", "32"])),
           "BYOLO_P1": str(rng.choice(["", "", "0"]))}
    for k, v in env.items():
        if v:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    seed = int(rng.integers(0, 1 << 30))
    t0 = time.time()
    m = build_model(variant, H, W, T=T, cls_cnt=cls_cnt, engine_options={"nms_mode": nms_mode})[1]
    eng = m.engine
    eng.set_params(synth.base_params(eng.param_shapes(), variant, cls_cnt, seed=seed % 1000))
    eng.finalize()
    # BN statistics from at least 64 pixels at the coarsest stride (a 32x32 image has ONE there: statistics of 4 values
    # make a degenerate network, not a test case)
    n_cal = max(4, -(-64 // ((H // 32) * (W // 32))))
    eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(n_cal, H, W, seed=seed % 977)).cuda())
    imgs = synth.synthetic_images(B, H, W, seed=seed % 7919)
    out = eng.forward(torch.from_numpy(imgs).cuda(), T=T, seed=seed, want_boxes=True)
    torch.cuda.synchronize()
    params = eng.get_params()
    with torch.no_grad():
        ref32, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=seed, cls_cnt=cls_cnt)
        ref64, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, torch.float64), imgs, variant, T=T, seed=seed,
                                        cls_cnt=cls_cnt, dtype=torch.float64)
    boxes = out["boxes"].cpu().numpy()
    r32, r64 = ref32.numpy().astype(np.float64), ref64.numpy()
    # Errors in units of the tolerance (1e-4 abs, relative above 1) against the float64 run.  A value passes within one
    # unit -- or, where the quantity itself is ill-conditioned, within twice what the float32 CPU restatement loses in
    # the same column: with random weights the logits reach tens, and exp(logit) box sizes / variances, their products
    # and the one-pass covariance E[l l^T] - E[l]E[l]^T turn 1e-4 on a logit into far more on the row (any two
    # float32 evaluations of the graph differ by that much; trained networks keep the logits small).
    with np.errstate(all="ignore"):
        unit = 1e-4 + 1e-4 * np.abs(r64)
        u_gpu, u_cpu = np.abs(boxes - r64) / unit, np.abs(r32 - r64) / unit
    finite = np.isfinite(r64) & np.isfinite(r32)
    assert np.array_equal(np.isnan(boxes), np.isnan(r32)), "case %d: NaN pattern differs" % idx
    u_gpu, u_cpu = np.where(finite, u_gpu, 0.0), np.where(finite, u_cpu, 0.0)
    allowed = np.maximum(1.0, 2.0 * u_cpu.max(axis=(0, 1)))             # per column
    bad = u_gpu > allowed
    assert not bad.any(), "case %d rows: %d / %d out of tolerance; worst column %d: %.2f units (CPU float32: %.2f)" % (
        idx, int(bad.sum()), boxes.size, int((u_gpu / allowed).max(axis=(0, 1)).argmax()),
        float(u_gpu.max()), float(u_cpu.max()))
    err = "%.2f / %.2f" % (float(u_gpu.max()), float(u_cpu.max()))
    # tail in isolation, bit-exact on the GPU's own rows
    refn = cpu_ref.nms_batch(torch.from_numpy(boxes), variant, max_out=1000, two_class=bool(nms_mode), cls_cnt=cls_cnt)
    kept, count, rows = out["kept"].cpu().numpy(), out["count"].cpu().numpy(), out["rows"].cpu().numpy()
    for b in range(B):
        n = int(count[b, 0])
        assert n == len(refn[b][1]), "case %d image %d: kept %d vs oracle %d" % (idx, b, n, len(refn[b][1]))
        assert np.array_equal(kept[b, :n], refn[b][1]), "case %d image %d: kept indices differ" % (idx, b)
        assert np.array_equal(rows[b, :n].view(np.uint32), refn[b][0].view(np.uint32)), "case %d image %d: rows differ" % (idx, b)
    knobs = " ".join("%s=%s" % (k[6:], v) for k, v in env.items() if v) or "-"
    return "| %d | %s | %d | %dx%d | %d | %d | %d | %s | %d | %s | %.1f |" % (
        idx, variant, cls_cnt, H, W, B, T, nms_mode, knobs, boxes.shape[1], err, time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=int, default=10, help="image sides up to 32 * this")
    ap.add_argument("--max-T", type=int, default=8)
    ap.add_argument("--max-B", type=int, default=5)
    ap.add_argument("--budget", type=int, default=4000, help="cap on coarsest-grid cells * B * T per case")
    a = ap.parse_args()
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    rng = np.random.default_rng(a.seed)
    print("| # | variant | C | HxW | B | T | nms | knobs | boxes/img | worst error in tolerance units vs float64: GPU / CPU float32 | s |\n|---|---|---|---|---|---|---|---|---|---|---|")
    bad = 0
    for i in range(a.cases):
        try:
            print(one_case(rng, i, a), flush=True)
        except Exception as e:                      # keep sweeping: report every failing case
            bad += 1
            print("| %d | FAILED: %s |" % (i, str(e).replace("\n", " ")[:300]), flush=True)
    print("\n%d cases, %d failed" % (a.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
