#!/usr/bin/env python
"""soak_nms.py -- randomised sweep of the tail (score sort + greedy NMS + gather: tf.image.non_max_suppression and
the 2-class variant, inference_epistemic.py:99-128) on the GPU against the oracle NMS: kept indices, gathered rows
and counts bit-exact.  Random box counts 1 .. 130 000, row widths, batch sizes, max_out, IoU thresholds and box
populations built to hit every path of tail_kernels.hip (bit-matrix fast path on a 4096 prefix, radix-select overflow
on mass ties, prefix exhaustion under heavy clustering, the general fallback): spread / clustered / identical boxes,
tied scores, zero-area and flipped corners, NaN and +-inf scores and coordinates.

    python tools/soak_nms.py --cases 200 --seed 1 > gpurun_out/soak_nms.md

Test infrastructure (imports oracle/ as the checker)."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))


def make_rows(g, B, N, D, obj_idx, cls_start):
    rows = g.random((B, N, D)).astype(np.float32)
    kinds = []
    for b in range(B):
        pop = g.choice(["spread", "clustered", "tight", "identical", "big"])
        if pop == "spread":
            c = g.random((N, 2)).astype(np.float32); s = (g.random((N, 2)) * 0.02 + 0.002).astype(np.float32)
        elif pop == "clustered":
            k = int(g.integers(1, 300)); cen = g.random((k, 2)).astype(np.float32)
            c = cen[g.integers(0, k, N)] + (g.standard_normal((N, 2)) * 0.01).astype(np.float32)
            s = (g.random((N, 2)) * 0.05 + 0.01).astype(np.float32)
        elif pop == "tight":
            k = int(g.integers(1, 50)); cen = g.random((k, 2)).astype(np.float32)
            c = cen[g.integers(0, k, N)] + (g.standard_normal((N, 2)) * 0.002).astype(np.float32)
            s = np.full((N, 2), 0.05, np.float32)
        elif pop == "identical":
            c = np.full((N, 2), 0.5, np.float32); s = np.full((N, 2), 0.1, np.float32)
        else:
            c = g.random((N, 2)).astype(np.float32); s = (g.random((N, 2)) * 0.6).astype(np.float32)
        rows[b, :, 0:2] = c - s; rows[b, :, 2:4] = c + s
        sc = g.choice(["random", "ties", "few_levels", "some_nan", "some_inf", "all_bad", "flipped", "zero_area", "nan_coords"])
        if sc == "ties":
            rows[b, :, obj_idx] = np.float32(g.random())
        elif sc == "few_levels":
            rows[b, :, obj_idx] = g.integers(0, 4, N).astype(np.float32) / 4
        elif sc == "some_nan":
            rows[b, g.random(N) < 0.3, obj_idx] = np.nan
        elif sc == "some_inf":
            m = g.random(N)
            rows[b, m < 0.1, obj_idx] = np.inf; rows[b, m > 0.9, obj_idx] = -np.inf
        elif sc == "all_bad":
            rows[b, :, obj_idx] = g.choice([np.nan, -np.inf])
        elif sc == "flipped":
            m = g.random(N) < 0.5
            rows[b, m, 0:2], rows[b, m, 2:4] = rows[b, m, 2:4].copy(), rows[b, m, 0:2].copy()
        elif sc == "zero_area":
            m = g.random(N) < 0.5
            rows[b, m, 2:4] = rows[b, m, 0:2]
        elif sc == "nan_coords":
            rows[b, g.random(N) < 0.05, int(g.integers(0, 4))] = np.nan
        kinds.append("%s/%s" % (pop, sc))
    return rows, kinds


def one_case(g, idx):
    import torch
    from byolo import Engine
    from oracle import nms_ref
    B = int(g.integers(1, 4))
    N = int(g.choice([1, 2, 63, 64, 65, 1000, 4095, 4096, 4097, 8193, 22743, 64512, 120960, int(g.integers(1, 130000))]))
    D = int(g.choice([7, 16, 23]))
    obj_idx, cls_start = {7: (4, 5), 16: (9, 11), 23: (14, 17)}[D]
    two = bool(g.integers(0, 2))
    max_out = int(g.choice([1, 10, 100, 1000, 1000, 1000, 2048]))       # the ABI takes 1 .. 2048
    iou = float(g.choice([0.5, 0.5, 0.5, 0.3, 0.7, 0.0, 1.0]))
    rows, kinds = make_rows(g, B, N, D, obj_idx, cls_start)
    t0 = time.time()
    eng = Engine((64, 64, 3), 2, nms_mode=1 if two else 0, max_out=max_out, iou_thresh=iou)
    res = eng.sort_nms(torch.from_numpy(rows).cuda(), obj_idx=obj_idx, cls_start_idx=cls_start)
    torch.cuda.synchronize()
    kept, count, out = res["kept"].cpu().numpy(), res["count"].cpu().numpy(), res["rows"].cpu().numpy()
    tot = []
    for b in range(B):
        if two:
            r_rows, r_keep, n_ped = nms_ref.nms_two_class(rows[b], obj_idx, cls_start, max_out, iou)
        else:
            r_rows, r_keep = nms_ref.nms_agnostic(rows[b], obj_idx, max_out, iou); n_ped = len(r_keep)
        n = int(count[b, 0])
        assert n == len(r_keep), "image %d (%s): kept %d vs oracle %d" % (b, kinds[b], n, len(r_keep))
        assert np.array_equal(kept[b, :n], r_keep), "image %d (%s): kept indices differ" % (b, kinds[b])
        assert np.array_equal(out[b, :n].view(np.uint32), r_rows.view(np.uint32)), "image %d (%s): rows differ" % (b, kinds[b])
        assert int(count[b, 1]) == n_ped, "image %d (%s): first-class count" % (b, kinds[b])
        assert (kept[b, n:] == -1).all() and (out[b, n:] == 0).all()
        tot.append(n)
    eng.close()
    return "| %d | %d | %d | %d | %s | %d | %.1f | %s | %s | %.1f |" % (idx, B, N, D, "2-class" if two else "agnostic", max_out, iou,
                                                                 " ".join(kinds), tot, time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    g = np.random.default_rng(a.seed)
    print("| # | B | N | D | mode | max_out | IoU | boxes / scores per image | kept | s |\n|---|---|---|---|---|---|---|---|---|---|")
    bad = 0
    for i in range(a.cases):
        try:
            print(one_case(g, i), flush=True)
        except Exception as e:
            bad += 1
            print("| %d | FAILED: %s |" % (i, str(e).replace("\n", " ")[:300]), flush=True)
    print("\n%d cases, %d failed" % (a.cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
