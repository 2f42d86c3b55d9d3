"""Oracle (test infrastructure): generate tests/golden/loss_gt.npz by IMPORTING THE REFERENCE (/root/reference) in this
container and running its OWN ground-truth encoding and loss -- `lib_yolo/tfdata.py:77-171` `encode_boxes`,
`lib_yolo/layers.py:11-84` `split_detection(_aleatoric)`, `lib_yolo/layers.py:126-188` `loss_tf` -- UNMODIFIED under
oracle/tf1_shim.py (TensorFlow primitives restated from their documented semantics: "parity unpinned" at that boundary).

    python -m oracle.make_golden_loss        # from the repo root; needs /root/reference

The fixture holds inputs (boxes, labels, raw detection tensors) and the reference's outputs (encoded ground truth per
detection layer, the three loss terms per layer in float32 and float64).  Data only.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from oracle import tf1_shim as shim                  # noqa: E402
from oracle import make_golden as mg                 # noqa: E402

IGN = 0.7           # pretraining.py:20, yolov3_training.py:20, uncertainty_training.py:20


def boxes_for(rng, n, lh, lw, priors_hw):
    """Ground-truth boxes of an image: prior-shaped boxes jittered around random centres (so that every detection layer
    gets objects), one tiny box, one box centred exactly on a cell corner (both neighbours satisfy 0 <= d <= 1), and a
    repeated box (the later one overwrites the earlier one's cell, lib_yolo/tfdata.py:139-146)."""
    bb, lab = [], []
    for i in range(n):
        ph, pw = priors_hw[rng.integers(len(priors_hw))]
        h = ph * rng.uniform(0.7, 1.4)
        w = pw * rng.uniform(0.7, 1.4)
        cy = rng.uniform(0.05, 0.95)
        cx = rng.uniform(0.05, 0.95)
        bb.append([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2])
        lab.append(int(rng.integers(0, 2)))
    bb.append([0.30, 0.40, 0.302, 0.401]); lab.append(1)                                   # tiny
    ph, pw = priors_hw[4]
    cy, cx = 2.0 / lh, 3.0 / lw                                                          # a corner of the stride-16 grid
    bb.append([cy - ph / 2, cx - pw / 2, cy + ph / 2, cx + pw / 2]); lab.append(0)
    bb.append(list(bb[0])); lab.append(1 - lab[0])                                         # same cell, other label
    return np.asarray(bb, np.float32), np.asarray(lab, np.int32)


def run_reference_encode(rdata, rtfdata, hw, priors, bboxes, labels, dtype):
    shim.install(dtype=dtype)
    layers = [rdata.DetLayerInfo(h=hw[0] // s, w=hw[1] // s, priors=priors[s]) for s in (32, 16, 8)]
    enc = rtfdata.encode_boxes(shim.Tensor(torch.as_tensor(bboxes).to(dtype)), shim.Tensor(torch.as_tensor(labels)), layers, ign_thresh=IGN)
    return [{k: v.numpy() for k, v in e.items()} for e in enc]


def run_reference_loss(rlayers, raw, gt, cls_cnt, aleatoric, aleatoric_loss, dtype):
    tf = shim.install(dtype=dtype)
    tf.losses.items = []
    x = shim.Tensor(torch.as_tensor(raw).to(dtype))
    det = (rlayers.split_detection_aleatoric if aleatoric else rlayers.split_detection)(x, boxes_per_cell=3, cls_cnt=cls_cnt)
    g = {"loc": shim.Tensor(torch.as_tensor(gt["loc"]).to(dtype)), "obj": shim.Tensor(torch.as_tensor(gt["obj"]).to(dtype)),
         "ign": shim.Tensor(torch.as_tensor(gt["ign"]).to(dtype)), "cls": shim.Tensor(torch.as_tensor(gt["cls"]))}
    l = rlayers.loss_tf(det, g, aleatoric_loss=aleatoric_loss)
    return np.asarray([float(l["loc"].numpy()), float(l["obj"].numpy()), float(l["cls"].numpy())], np.float64)


def import_training_side():
    shim.install()
    ryolo = mg.import_reference()[0]
    import lib_yolo.data as rdata
    import lib_yolo.tfdata as rtfdata
    import lib_yolo.layers as rlayers
    assert rtfdata.__file__.startswith(mg.REF)
    return ryolo, rdata, rtfdata, rlayers


def main():
    ryolo, rdata, rtfdata, rlayers = import_training_side()
    rng = np.random.default_rng(2024)
    out = {}
    priors = ryolo.ECP_9_PRIORS
    flat_priors = [(p.h, p.w) for s in (32, 16, 8) for p in priors[s]]
    cases = (("a", (64, 96), 5), ("b", (128, 128), 12), ("c", (64, 96), 0))
    for tag, hw, n in cases:
        if n:
            bb, lab = boxes_for(rng, n, hw[0] // 16, hw[1] // 16, flat_priors)
        else:
            bb, lab = np.zeros((0, 4), np.float32), np.zeros((0,), np.int32)                # an image without objects
        out["%s_hw" % tag] = np.asarray(hw, np.int32)
        out["%s_boxes" % tag] = bb
        out["%s_labels" % tag] = lab
        enc = run_reference_encode(rdata, rtfdata, hw, priors, bb, lab, torch.float32)
        for k, e in enumerate(enc):
            for name, v in e.items():
                out["%s_gt%d_%s" % (tag, k, name)] = v
    # losses: batch of two images (cases a and c share the size), raw detection tensors ~ N(0, 1.5)
    for variant, aleatoric, aleatoric_loss in (("std", False, False), ("ale", True, False), ("ale_loss", True, True)):
        F = 3 * ((10 + 4) if aleatoric else 7)
        for k, (lh, lw) in enumerate(((2, 3), (4, 6), (8, 12))):
            raw = (rng.standard_normal((2, lh, lw, F)) * 1.5).astype(np.float32)
            if aleatoric:
                r = raw.reshape(2, lh, lw, 3, 14)
                r[0, 0, 0, 0, 4] = 55.0          # log variance beyond the clip (lib_yolo/layers.py:151)
                r[1, -1, -1, 2, 7] = -47.5
            gt = {name: np.stack([out["a_gt%d_%s" % (k, name)], out["c_gt%d_%s" % (k, name)]]) for name in ("loc", "obj", "cls", "ign")}
            gt["loc"][1, 0, 0, 0] = [0.3, -0.2, 0.1, 0.4]; gt["obj"][1, 0, 0, 0] = 1; gt["cls"][1, 0, 0, 0] = 1   # an object in image 1 too
            out["loss_%s_raw%d" % (variant, k)] = raw
            if variant == "std":
                for name, v in gt.items():
                    out["loss_gt%d_%s" % (k, name)] = v
            out["loss_%s_f32_%d" % (variant, k)] = run_reference_loss(rlayers, raw, gt, 2, aleatoric, aleatoric_loss, torch.float32)
            out["loss_%s_f64_%d" % (variant, k)] = run_reference_loss(rlayers, raw, gt, 2, aleatoric, aleatoric_loss, torch.float64)
    np.savez_compressed(os.path.join(OUT, "loss_gt.npz"), **out)
    print("wrote", os.path.join(OUT, "loss_gt.npz"), os.path.getsize(os.path.join(OUT, "loss_gt.npz")), "bytes;",
          "objects per layer (case a):", [int(out["a_gt%d_obj" % k].sum()) for k in range(3)],
          "(case b):", [int(out["b_gt%d_obj" % k].sum()) for k in range(3)])
    mg.restore_environment()


if __name__ == "__main__":
    main()
