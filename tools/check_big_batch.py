#!/usr/bin/env python
"""Run BASELINE config 4 geometry with a batch whose largest activation exceeds 2 GiB (32-bit buffer offsets
with the top bit set) and compare every image with its own batch-1 run.

    python tools/check_big_batch.py [B]        (default 12: the 76x76x256 activation is 2.13 GB)

Beyond Engine.max_images(T) (18 at this geometry) byolo_forward runs consecutive pieces; B = 20 checks that.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "bayesian-yolov3_amd")]


def main():
    import torch
    from byolo import synth
    from lib_yolo import yolov3, model as lmodel   # noqa: F401
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from conftest import build_model
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    v = "bayesian_yolov3_aleatoric"
    m = build_model(v, 608, 608, T=30)[1]
    eng = m.engine
    eng.set_params(synth.base_params(eng.param_shapes(), v, 2, seed=7))
    eng.finalize()
    imgs = synth.synthetic_images(B, 608, 608, seed=1234)
    x = torch.from_numpy(imgs).cuda()
    eng.calibrate_bn(x[:2])
    print("max images per call at T=30: %d" % eng.max_images(30))
    # dropout off: an image's rows then do not depend on its position in the batch
    big = eng.forward(x, T=30, seed=42, dropout_on=False, want_boxes=True)["boxes"].cpu().numpy()
    worst = 0.0
    for i in (0, B // 2, B - 1):
        one = eng.forward(x[i:i + 1], T=30, seed=42, dropout_on=False, want_boxes=True)["boxes"].cpu().numpy()
        d = np.abs(big[i] - one[0])
        tol = 1e-4 + 1e-4 * np.abs(one[0])
        bad = int((d > tol).sum())
        worst = max(worst, float(np.nanmax(d)))
        print("image %d: max |diff| %.3e, %d of %d outside 1e-4" % (i, float(np.nanmax(d)), bad, d.size))
        assert bad == 0
    print("OK B=%d worst %.3e" % (B, worst))


if __name__ == "__main__":
    main()
