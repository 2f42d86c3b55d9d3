#!/usr/bin/env python
"""Digest of everything a forward hands out, for bit-for-bit A/B of two builds of libbyolo (BYOLO_LIB=...): SHA-256 of the pre-NMS
rows, the kept rows / indices / counts and the raw detection outputs, for each reference model at 64 x 96 (golden weights, dropout
on) and the Bayesian model at a benchmark-like shape (320 x 320, T = 6, 3 images: Winograd, fused pairs, 1x1 loop all in the plan).

    BYOLO_LIB=$PWD/bayesian-yolov3_amd/byolo/libbyolo_ref.so python tools/rows_digest.py > a.txt; python tools/rows_digest.py > b.txt; diff a.txt b.txt
"""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


def main():
    import torch
    from conftest import build_model, golden_params, golden_images
    from byolo import synth
    for v in ("yolov3", "yolov3_aleatoric", "bayesian_yolov3_aleatoric"):
        B = 2
        _, m = build_model(v, 64, 96, T=3, params=golden_params(v), engine_options={"keep_all_outputs": True})
        m.finalize()
        out = m.run(torch.from_numpy(golden_images(B)).cuda(), seed=42)
        torch.cuda.synchronize()
        raws = [dl.raw_output.cpu().numpy() for dl in m.det_layers]
        print(v, "64x96", digest(out["boxes"].cpu().numpy()), digest(out["rows"].cpu().numpy(), out["kept"].cpu().numpy(), out["count"].cpu().numpy()),
              digest(*raws), digest(m.engine.layer_output(36).cpu().numpy(), m.engine.layer_output(74).cpu().numpy()))
    v = "bayesian_yolov3_aleatoric"
    for (H, W, T, B) in ((320, 320, 6, 3), (416, 416, 4, 2)):
        _, m = build_model(v, H, W, T=T)
        eng = m.engine
        eng.set_params(synth.base_params(eng.param_shapes(), v, 2, seed=7))
        eng.finalize()
        x = torch.from_numpy(synth.synthetic_images(B, H, W, seed=1234)).cuda()
        eng.calibrate_bn(x[:2])
        out = eng.forward(x, T=T, seed=42, want_boxes=True, want_nms=True)
        torch.cuda.synchronize()
        print(v, "%dx%d T=%d B=%d" % (H, W, T, B), digest(out["boxes"].cpu().numpy()),
              digest(out["rows"].cpu().numpy(), out["kept"].cpu().numpy(), out["count"].cpu().numpy()), eng.precision)


if __name__ == "__main__":
    main()
