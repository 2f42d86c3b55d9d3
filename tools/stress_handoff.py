#!/usr/bin/env python
"""Stress the slab / ticket hand-off of conv_igemm.hip (split-K slices and stream-K segments): the block that draws the
last ticket of a tile sums the other blocks' raw accumulators, which travel as sc1 (write-through) stores + sc1 loads
with a relaxed agent-scope ticket in between -- no release / acquire fence (conv_igemm.hip explains why).  If a
reducer ever saw a stale slab line, or a counter were left non-zero for the next launch, the bits would change.

    python tools/stress_handoff.py [iterations]        (default 1000; tests/test_gpu_layers.py runs 150)

Three handles with three different plans -- K slices forced on every launch (BYOLO_KSPLIT=3), stream-K forced on every
launch (BYOLO_STREAMK=2), the planner's own choice -- run the same small conv stack (3x3 / 1x1 / stride-2 layers at
head-like shapes: 64 .. 512 channels, K up to 4608, a few hundred tiles per launch, launches back to back so that slabs
and counters are reused immediately) CONCURRENTLY on three HIP streams, `iterations` times.  Every iteration of a handle
must reproduce the handle's first result bit for bit (the reduce order is fixed by the plan, not by arrival), and all
three must agree with each other within fp32 re-association."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))

BN = 1


def build(env, H=96, W=96, C=64):
    from byolo import Engine
    for k in ("BYOLO_KSPLIT", "BYOLO_STREAMK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["BYOLO_WINOGRAD"] = "0"                       # every 3x3 on the direct kernel: that is where the hand-off lives
    eng = Engine((H, W, C), 2)
    eng.add_conv("a", 256, 3, 1, BN)                          # K = 576
    eng.add_conv("b", 128, 1, 1, BN)
    eng.add_conv("c", 512, 3, 2, BN)                          # K = 1152, stride 2
    eng.add_conv("d", 256, 1, 1, BN)
    eng.add_conv("e", 512, 3, 1, BN)                          # K = 2304
    eng.add_residual(-3)
    eng.add_conv("f", 1024, 3, 2, BN)                         # K = 4608: few tiles, long K
    eng.add_detection("h/detection", 0, [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])
    g = np.random.default_rng(5)
    p = {}
    for name, shape in eng.param_shapes().items():
        if name.endswith("kernel"):
            p[name] = (g.standard_normal(shape) * np.sqrt(2.0 / np.prod(shape[:3]))).astype(np.float32)
        elif name.endswith("moving_variance") or name.endswith("gamma"):
            p[name] = (g.random(shape) + 0.5).astype(np.float32)
        else:
            p[name] = (g.standard_normal(shape) * 0.1).astype(np.float32)
    eng.set_params(p)
    eng.finalize()
    return eng


def main(iters=1000, B=6, verbose=True):
    import torch
    plans = [("K slices forced", {"BYOLO_KSPLIT": "3"}), ("stream-K forced", {"BYOLO_STREAMK": "2"}), ("planner", {})]
    x = torch.from_numpy(np.random.default_rng(1).random((B, 96, 96, 64), dtype=np.float32)).cuda()
    engs, first = [], []
    for name, env in plans:
        e = build(env)
        e.set_profiling(2)
        out = e.forward(x, T=1, seed=0, want_boxes=True, want_nms=False)     # the plan is made here, under this env
        torch.cuda.synchronize()
        prof = e.step_profile()
        e.set_profiling(0)
        engs.append(e)
        first.append(out["boxes"].clone())
        if verbose:
            print("%-16s launches: %s" % (name, [(s["M"], s["N"], s["K"], s["ksplit"]) for s in prof]))
    for k in ("BYOLO_KSPLIT", "BYOLO_STREAMK", "BYOLO_WINOGRAD"):
        os.environ.pop(k, None)
    ref = first[2].cpu().numpy()
    for f in first[:2]:
        d = np.abs(f.cpu().numpy() - ref)
        assert np.nanmax(d / np.maximum(1.0, np.abs(ref))) < 1e-4, "plans disagree beyond fp32 re-association"
    assert not torch.equal(first[0], first[2]) and not torch.equal(first[1], first[2]), "the forced plans did not change the reduce order"
    streams = [torch.cuda.Stream() for _ in engs]
    outs = [dict(boxes=torch.empty_like(f)) for f in first]
    bad = 0
    for it in range(iters):
        for k, (e, st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                e.forward(x, T=1, seed=0, want_boxes=True, want_nms=False, out=outs[k], slot=1 + k)
        torch.cuda.synchronize()
        for k in range(3):
            if not torch.equal(outs[k]["boxes"], first[k]):
                bad += 1
                print("iteration %d: %s differs from its first run" % (it, plans[k][0]))
    assert bad == 0, "%d mismatches in %d iterations" % (bad, iters)
    if verbose:
        print("%d iterations x 3 concurrent handles: every result bit-identical to its first run" % iters)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
