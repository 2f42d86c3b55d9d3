"""The CPU restatement against the REFERENCE'S OWN graph code, live: `lib_yolo/{yolov3,model,layers}.py` and the
`inference_*` helpers imported from /root/reference and executed unmodified under oracle/tf1_shim.py -- on weights,
images, image size, T and dropout seed that are NOT the ones the committed fixtures were generated with (the fixtures
pin one configuration; this keeps the restatement from being fitted to it).

Runs only where the reference is mounted (the build container); skipped on the GPU box."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")

VARIANTS = ("yolov3", "yolov3_aleatoric", "bayesian_yolov3_aleatoric")
H, W, T, SEED_W, SEED_DROP = 32, 64, 2, 21, 77


@pytest.fixture(autouse=True)
def _leave_no_reference_modules_behind():
    """oracle.make_golden.import_reference() points `lib_yolo` at the reference and installs stand-in `tensorflow` /
    `cv2` modules; undo that after every test so the rest of the session imports the build's own package."""
    yield
    from oracle import make_golden as mg
    mg.restore_environment()


def _params(variant):
    from oracle import cpu_ref
    from byolo import synth
    # BN statistics calibrated once with the restatement (oracle/make_golden.py:81-90 does the same for the fixtures)
    shapes = cpu_ref.variable_shapes("yolov3_aleatoric", 2)
    p = cpu_ref.to_torch_params(synth.base_params(shapes, "yolov3_aleatoric", 2, seed=SEED_W))
    cpu_ref.forward(p, synth.synthetic_images(32, H, W, seed=5), "yolov3_aleatoric", calibrate=True)
    stats = {k: v.numpy() for k, v in p.items() if k.endswith("moving_mean") or k.endswith("moving_variance")}
    out = synth.base_params(cpu_ref.variable_shapes(variant, 2), variant, 2, seed=SEED_W)
    for k, v in stats.items():
        out[k] = v.astype(np.float32)
    return out


@pytest.mark.parametrize("variant", VARIANTS)
def test_restatement_equals_reference_graph_code(variant, monkeypatch):
    from oracle import cpu_ref, make_golden as mg
    from byolo import synth
    monkeypatch.setattr(mg, "H", H)
    monkeypatch.setattr(mg, "W", W)
    params = _params(variant)
    B = 1 if variant.startswith("bayes") else 2            # the reference asserts batch 1 in epistemic mode
    imgs = synth.synthetic_images(B, H, W, seed=31)
    ref = mg.run_reference(variant, params, imgs, torch.float32, seed=SEED_DROP, T=T)
    with torch.no_grad():
        boxes, f = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=SEED_DROP)
    for k in range(3):
        assert_close(f["raw"][k].numpy(), ref["raw"][k], "%s raw detection output %d" % (variant, k), rtol=1e-5, atol=1e-5)
    rb = ref["bbox"] if ref["bbox"].ndim == 3 else ref["bbox"][None]
    assert_close(boxes.numpy(), rb, "%s pre-NMS rows" % variant, rtol=1e-5, atol=1e-5)
    # NMS of the reference's rows: the restated tail on the same rows keeps the same boxes
    mine = cpu_ref.nms_batch(torch.from_numpy(rb), variant)
    for b in range(rb.shape[0]):
        assert np.array_equal(mine[b][0], ref["nms_rows"][b]), "%s image %d: NMS rows differ" % (variant, b)
    # structure: variables in creation order, layer count, row layout
    assert ref["var_names"] == list(cpu_ref.variable_shapes(variant, 2))
    assert (ref["obj_idx"], ref["cls_start_idx"]) == cpu_ref.row_layout(variant, 2)[1:]


def test_injected_masks_mean_the_same_in_the_reference_graph_and_the_restatement(monkeypatch):
    """byolo_forward's d_mask_bits hands the product one keep-bit per element of every dropout input, in the order the
    reference CALLS tf.layers.dropout (lib_yolo/layers.py:521-524; yolov3.py:544-548, :575-579, :606-610).  The same
    numpy-drawn arrays given to the reference's own graph code (under the shim) and to the CPU restatement must give
    the same rows: that pins call order and element order of the injection, independently of oracle/rng.py."""
    from oracle import cpu_ref, make_golden as mg
    from byolo import synth
    monkeypatch.setattr(mg, "H", H)
    monkeypatch.setattr(mg, "W", W)
    variant = "bayesian_yolov3_aleatoric"
    params = _params(variant)
    imgs = synth.synthetic_images(1, H, W, seed=33)
    # shapes of the 15 dropout inputs: from a run of the reference itself
    probe = mg.run_reference(variant, params, imgs, torch.float32, seed=1, T=T)
    shapes = [s for _, s in probe["dropout_calls"]]
    assert len(shapes) == 15
    g = np.random.default_rng(2024)
    masks = [g.random(s) < 0.9 for s in shapes]
    ref = mg.run_reference(variant, params, imgs, torch.float32, seed=999, T=T, masks=masks)
    with torch.no_grad():
        boxes, f = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=555, masks=masks)
    for k in range(3):
        assert_close(f["raw"][k].numpy(), ref["raw"][k], "raw detection output %d under injected masks" % k, rtol=1e-5, atol=1e-5)
    assert_close(boxes.numpy(), ref["bbox"][None], "pre-NMS rows under injected masks", rtol=1e-5, atol=1e-5)
    # and the masks matter: another draw gives other rows
    other = [g.random(s) < 0.9 for s in shapes]
    with torch.no_grad():
        b2, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=555, masks=other)
    assert np.abs(b2.numpy() - boxes.numpy()).max() > 1e-3


def test_ground_truth_encoding_and_loss_restatement_equal_the_reference_functions():
    """Row f4: oracle/train_ref.py against the reference's `tfdata.encode_boxes` / `layers.loss_tf`, live, on another image size,
    another prior table, other boxes and raw tensors than the committed fixture (tests/golden/loss_gt.npz)."""
    from oracle import make_golden_loss as mgl, train_ref
    ryolo, rdata, rtfdata, rlayers = mgl.import_training_side()
    rng = np.random.default_rng(99)
    hw = (96, 160)
    priors = ryolo.CITY_PERSONS_9_PRIORS
    flat = [(p.h, p.w) for s in (32, 16, 8) for p in priors[s]]
    bb, lab = mgl.boxes_for(rng, 9, hw[0] // 16, hw[1] // 16, flat)
    ref = mgl.run_reference_encode(rdata, rtfdata, hw, priors, bb, lab, torch.float32)
    layers = [(hw[0] // s, hw[1] // s, [(p.h, p.w) for p in priors[s]]) for s in (32, 16, 8)]
    enc = train_ref.encode_boxes(bb, lab, layers, mgl.IGN)
    for k in range(3):
        for n in ("obj", "cls", "ign"):
            assert np.array_equal(enc[k][n], ref[k][n]), (k, n)
        assert_close(enc[k]["loc"], ref[k]["loc"], "loc %d" % k, rtol=1e-6, atol=1e-6)
    assert sum(int(e["obj"].sum()) for e in enc) >= 9
    for aleatoric, aleatoric_loss in ((False, False), (True, False), (True, True)):
        for k, (lh, lw, _) in enumerate(layers):
            raw = (rng.standard_normal((3, lh, lw, 3 * (14 if aleatoric else 7))) * 2.0).astype(np.float32)
            gt = {n: np.stack([enc[k][n]] * 3) for n in ("loc", "obj", "cls", "ign")}
            want = mgl.run_reference_loss(rlayers, raw, gt, 2, aleatoric, aleatoric_loss, torch.float64)
            l = train_ref.loss(raw, gt, 2, aleatoric, aleatoric_loss)
            assert_close(np.asarray([l["loc"], l["obj"], l["cls"]]), want, "loss layer %d" % k, rtol=1e-12, atol=1e-12)
