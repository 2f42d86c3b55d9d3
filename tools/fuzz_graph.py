#!/usr/bin/env python
"""fuzz_graph.py -- random layer graphs through the C-ABI graph builder on the GPU, every retrievable layer output
against a plain PyTorch fp32 interpretation of the same graph (F.conv2d, explicit pads, the dropout masks of
oracle/rng.py): the lowering (views for route / upsample / stack, the two-source loader, fused residuals, the
T-invariant de-duplication, tile / split-K / Winograd planning) sees topologies YOLOv3 never produces.

    python tools/fuzz_graph.py --cases 100 --seed 1 > gpurun_out/fuzz_graph.md

Test infrastructure (imports oracle/rng.py and tests/test_gpu_layers.py's reference convolution)."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

BN, DROP = 1, 2
FILTERS = [8, 16, 24, 32, 40, 48, 64, 96, 128, 192, 256]


def one_case(rng, idx, max_cells=5):
    import torch
    from byolo import Engine, ByoloError
    from conftest import assert_close
    from test_gpu_layers import _ref_conv, _random_params
    H, W = int(rng.integers(1, max_cells + 1)) * 32, int(rng.integers(1, max_cells + 1)) * 32
    B, T = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    prob = float(rng.choice([0.1, 0.25, 0.5]))
    # (BYOLO_WINO_SPLIT / BYOLO_KX3_WIDE: the split-precision plans -- Winograd in split arithmetic and the 8-wave shared-tap
    #  tile -- forced on every eligible layer or off; they are read when a handle plans a (B, T))
    for k in ("BYOLO_WINOGRAD", "BYOLO_WINO_FUSED", "BYOLO_KSPLIT", "BYOLO_STREAMK", "BYOLO_STREAM1X1", "BYOLO_WINO_SPLIT", "BYOLO_KX3_WIDE"):
        v = str(rng.choice(["", "", "0", "2"] if k != "BYOLO_KSPLIT" else ["", "", "0", "2", "3"]))
        if v:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    IC = int(rng.choice([3, 3, 3, 1, 4]))          # image channels (the reference feeds RGB)
    eng = Engine((H, W, IC), 2, drop_prob=prob, keep_all_outputs=True)
    # python-side model of the graph: one record per layer
    L = []          # dict(op, h, w, c, stacked, args)
    desc = []

    def add(op, h, w, c, stacked, **args):
        L.append(dict(op=op, h=h, w=w, c=c, stacked=stacked, **args))
        return len(L) - 1

    n_conv = 0

    def conv(src_rec, filters, k, stride, flags):
        nonlocal n_conv
        scope = "c%d" % n_conv
        n_conv += 1
        i = eng.add_conv(scope, filters, k, stride, flags)
        assert i == len(L)
        add("conv", src_rec["h"] // stride, src_rec["w"] // stride, filters, src_rec["stacked"], scope=scope, k=k, stride=stride,
            flags=flags, src=len(L) - 1 if L else -1)
        desc.append("%s%dx%d/%d->%d%s" % ("C", k, k, stride, filters, "d" if flags & DROP else ""))

    img_rec = dict(h=H, w=W, c=IC, stacked=False)
    conv(img_rec, int(rng.choice([8, 16, 32, 64])), 3, 1, BN)
    stacked = False
    # several detection heads like the reference's models: all of one kind; the Bayesian kind sits on stacked layers
    bayes = T > 1 and rng.random() < 0.6
    kind = 2 if bayes else int(rng.integers(0, 2))
    pri = [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)]
    n_det = 0

    def plain(j):                                   # layers other layers may read
        return L[j]["op"] != "det"

    for _ in range(int(rng.integers(3, 14))):
        last = L[-1]
        op = rng.choice(["conv", "conv", "conv", "res", "up", "route", "stack", "det"])
        if op == "stack" and not bayes:
            op = "conv"
        if op == "det":
            if n_det < 2 and (stacked or not bayes) and len(L) >= 2:
                if last["op"] != "conv":
                    conv(last, int(rng.choice(FILTERS)), 1, 1, BN)
                assert eng.add_detection("d%d/detection" % n_det, kind, pri) == len(L)
                add("det", L[-1]["h"], L[-1]["w"], 3 * 7 * (1 if kind == 0 else 2), L[-1]["stacked"], scope="d%d/detection" % n_det, src=len(L) - 1)
                desc.append("det%d" % kind)
                n_det += 1
                back = [j for j in range(len(L)) if plain(j) and L[j]["stacked"] == L[-1]["stacked"]]      # route back (yolov3.py:258-260)
                j = int(rng.choice(back))
                assert eng.add_route([j]) == len(L)
                add("route", L[j]["h"], L[j]["w"], L[j]["c"], L[j]["stacked"], srcs=[j])
                desc.append("id(%d)" % j)
            continue
        if op == "conv":
            k = int(rng.choice([1, 3]))
            stride = 2 if (k == 3 and rng.random() < 0.3 and last["h"] % 2 == 0 and last["w"] % 2 == 0 and min(last["h"], last["w"]) >= 4) else 1
            conv(last, int(rng.choice(FILTERS)), k, stride, BN | (DROP if rng.random() < 0.35 else 0))
        elif op == "res":
            cands = [j for j, r in enumerate(L[:-1]) if plain(j) and (r["h"], r["w"], r["c"], r["stacked"]) == (last["h"], last["w"], last["c"], last["stacked"])]
            if cands:
                j = int(rng.choice(cands))
                assert eng.add_residual(j) == len(L)
                add("res", last["h"], last["w"], last["c"], last["stacked"], a=len(L) - 1, b=j)
                desc.append("R%d" % j)
        elif op == "up" and max(last["h"], last["w"]) <= 80:
            assert eng.add_upsample() == len(L)
            add("up", 2 * last["h"], 2 * last["w"], last["c"], last["stacked"], src=len(L) - 1)
            desc.append("U")
        elif op == "route":
            cands = [j for j, r in enumerate(L) if plain(j) and r["stacked"] == last["stacked"]]
            j = int(rng.choice(cands))
            same = [q for q in cands if q != j and (L[q]["h"], L[q]["w"]) == (L[j]["h"], L[j]["w"])]
            if same and rng.random() < 0.7:
                q = int(rng.choice(same))
                assert eng.add_route([j, q]) == len(L)
                add("route", L[j]["h"], L[j]["w"], L[j]["c"] + L[q]["c"], last["stacked"], srcs=[j, q])
                desc.append("cat(%d,%d)" % (j, q))
            else:
                assert eng.add_route([j]) == len(L)
                add("route", L[j]["h"], L[j]["w"], L[j]["c"], last["stacked"], srcs=[j])
                desc.append("id(%d)" % j)
        elif op == "stack" and T > 1 and (not stacked or rng.random() < 0.3):
            # the T-fold tile of ANY unstacked layer so far (the Bayesian YOLOv3 stacks three taps of its backbone)
            cands = [j for j, r in enumerate(L) if plain(j) and not r["stacked"]]
            j = len(L) - 1 if not stacked and rng.random() < 0.6 else int(rng.choice(cands))
            assert eng.add_stack(j) == len(L)
            add("stack", L[j]["h"], L[j]["w"], L[j]["c"], True, src=j)
            stacked = True
            desc.append("S%d" % j)
    if bayes and not stacked:                       # the Bayesian detection reduces over T samples
        j = len(L) - 1
        assert eng.add_stack(j) == len(L)
        add("stack", L[j]["h"], L[j]["w"], L[j]["c"], True, src=j)
        stacked = True
        desc.append("S%d" % j)
    if L[-1]["op"] != "conv":                       # detection reads a convolution's output in every reference model
        conv(L[-1], int(rng.choice(FILTERS)), 1, 1, BN)
    det_idx = eng.add_detection("d/detection", kind, pri)
    desc.append("det%d" % kind)
    run_T = T if stacked else 1
    print("case %d: %dx%d B=%d T=%d %s  env %s" % (idx, H, W, B, run_T, " ".join(desc), {k: os.environ.get(k) for k in (
        "BYOLO_WINOGRAD", "BYOLO_WINO_FUSED", "BYOLO_KSPLIT")}), file=sys.stderr, flush=True)    # survives a device fault
    try:
        eng.workspace_bytes(B, run_T)
    except ByoloError as e:                         # a graph the lowering refuses (documented limits): not a failure
        eng.close()
        return "| %d | %dx%d B=%d T=%d | %s | refused: %s |" % (idx, H, W, B, run_T, " ".join(desc), str(e)[:80]), None
    p = _random_params(eng, int(rng.integers(0, 1 << 30)))
    eng.set_params(p)
    eng.finalize()
    img = rng.random((B, H, W, IC)).astype(np.float32)
    seed = int(rng.integers(0, 1 << 30))
    out = eng.forward(torch.from_numpy(img).cuda(), T=run_T, seed=seed, want_boxes=True, want_nms=False)
    torch.cuda.synchronize()
    # reference interpretation
    ref = []
    ordinal = 0
    x0 = torch.from_numpy(img)
    for i, r in enumerate(L):
        if r["op"] == "conv":
            x = x0 if r["src"] < 0 else ref[r["src"]]
            drop = None
            if r["flags"] & DROP:
                drop = (seed, ordinal, prob)
                ordinal += 1
            ref.append(_ref_conv(x, p, r["scope"], r["k"], r["stride"], r["flags"], drop=drop))
        elif r["op"] == "res":
            ref.append(ref[r["a"]] + ref[r["b"]])
        elif r["op"] == "up":
            ref.append(ref[r["src"]].repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))
        elif r["op"] == "route":
            ref.append(torch.cat([ref[j] for j in r["srcs"]], dim=3) if len(r["srcs"]) == 2 else ref[r["srcs"][0]])
        elif r["op"] == "stack":
            ref.append(ref[r["src"]].repeat_interleave(run_T, dim=0))
        elif r["op"] == "det":
            wd = torch.from_numpy(p[r["scope"] + "/conv2d/kernel"]).permute(3, 2, 0, 1)
            ref.append(torch.nn.functional.conv2d(ref[r["src"]].permute(0, 3, 1, 2), wd).permute(0, 2, 3, 1) + torch.from_numpy(p[r["scope"] + "/conv2d/bias"]))
    wdet = torch.from_numpy(p["d/detection/conv2d/kernel"]).permute(3, 2, 0, 1)
    det = torch.nn.functional.conv2d(ref[-1].permute(0, 3, 1, 2), wdet).permute(0, 2, 3, 1) + torch.from_numpy(p["d/detection/conv2d/bias"])
    checked = 0
    worst = 0.0
    for i in list(range(len(L))) + [det_idx]:
        try:
            got = eng.layer_output(i).cpu().numpy()
        except ByoloError:
            continue                                # a view / fused layer without a tensor of its own
        want = (det if i == det_idx else ref[i]).numpy()
        # 3e-4 here, not the 1e-4 of tests/: a dozen random layers with dropout scales up to 2 and random BN is two
        # float32 evaluations drifting apart (1 value in a million reaches 1.3e-4); a lowering bug is an O(1) error
        worst = max(worst, assert_close(got, want, "case %d layer %d (%s)" % (idx, i, "det" if i == det_idx else L[i]["op"]),
                                        rtol=3e-4, atol=3e-4))
        checked += 1
    # decode + concat of the heads: the oracle's decode applied to the DEVICE's raw head outputs (so that this checks the
    # decode and the concat offsets of 1 .. 3 heads of arbitrary grids, not error propagation), corners / objectness /
    # layer and prior ids of every row, in concat_bbox order (head after head, prior-major inside a head)
    from oracle import cpu_ref
    heads = [i for i, r in enumerate(L) if r["op"] == "det"] + [det_idx]
    rows = []
    for lid, i in enumerate(heads):
        raw = eng.layer_output(i).cpu()
        if kind == 0:
            rows.append(cpu_ref.decode_standard(raw, pri, 2))
        elif kind == 1:
            rows.append(cpu_ref.decode_aleatoric(raw, pri, 2, lid))
        else:
            rows.append([cpu_ref.decode_epistemic(raw[b * run_T:(b + 1) * run_T], pri, 2, lid) for b in range(B)])
    if kind == 2:
        want_boxes = torch.stack([cpu_ref.concat_bbox([rows[k][b] for k in range(len(heads))], False) for b in range(B)]).numpy()
    else:
        want_boxes = cpu_ref.concat_bbox(rows, True).numpy()
    got_boxes = out["boxes"].cpu().numpy()
    assert got_boxes.shape == want_boxes.shape, "case %d: boxes %s vs %s" % (idx, got_boxes.shape, want_boxes.shape)
    obj_col = {0: 4, 1: 9, 2: 14}[kind]
    cols = [0, 1, 2, 3, obj_col] + ([] if kind == 0 else [got_boxes.shape[-1] - 2, got_boxes.shape[-1] - 1])
    with np.errstate(all="ignore"):
        sane = np.isfinite(want_boxes[..., :4]).all(-1) & (np.abs(want_boxes[..., :4]).max(-1) < 1e3)   # exp(t) of a wild logit
    assert_close(got_boxes[..., cols][sane], want_boxes[..., cols][sane], "case %d: decoded rows of %d head(s)" % (idx, len(heads)))
    # the same batch in two calls (first_image tells the second where it sits in the logical batch): equal to
    # the unsplit run -- what sub-batching and the one-shard-per-GPU mode rely on (dropout masks are indexed by position)
    if B >= 2:
        whole = eng.layer_output(det_idx).clone()
        k = int(rng.integers(1, B))
        S = run_T if stacked else 1
        parts = []
        for lo, hi in ((0, k), (k, B)):
            eng.forward(torch.from_numpy(img[lo:hi]).cuda(), T=run_T, seed=seed, want_boxes=True, want_nms=False, first_image=lo)
            parts.append(eng.layer_output(det_idx).clone())
        torch.cuda.synchronize()
        # (same masks; the launch plan -- tile shape, K slices -- may differ with the batch size, so the float32
        # summation order may too: a wrong mask is an O(1) difference, this is 1e-4)
        assert_close(torch.cat(parts, 0).cpu().numpy(), whole.cpu().numpy(), "case %d: split batch (%d + %d images) vs the whole" % (idx, k, B - k))
        assert whole.shape[0] == B * S
    eng.close()
    assert checked >= 2
    return "| %d | %dx%d B=%d T=%d | %s | %d of %d layers compared, max abs err %.1e |" % (
        idx, H, W, B, run_T, " ".join(desc), checked, len(L) + 1, worst), worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=int, default=5, help="image sides up to 32 * this")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    print("| # | input | graph | result |\n|---|---|---|---|")
    bad = refused = 0
    t0 = time.time()
    for i in range(a.cases):
        try:
            line, worst = one_case(rng, i, a.max_cells)
            refused += worst is None
            print(line, flush=True)
        except Exception as e:
            bad += 1
            print("| %d | | | FAILED: %s |" % (i, str(e).replace("\n", " ")[:400]), flush=True)
    print("\n%d graphs, %d refused by the lowering, %d failed, %.0f s" % (a.cases, refused, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
