"""The default arithmetic (split-f16: hi + lo fp16 pairs, include/byolo.h) on weights that are NOT the friendly synthetic
recipe, and what happens where it cannot follow the reference's float32 (lib_yolo/layers.py:550; real checkpoints arrive
through tf.train.Saver.restore, inference_epistemic.py:27-38, :58):

  * adversarial-weights parity at 416x416 and 608x608 (B=2, T=3), both precisions, against the float64 AND the float32
    oracle per column group: per-channel filter scales over 2^+-10 inside a layer, BN gamma up to 50 and moving variances
    down to 1e-6, heavy-tailed (Student-t) filters, activations driven to 1e3 .. 1e4 (byolo.synth.adversarial_params);
  * per-OUTPUT-CHANNEL weight scales against one scale per layer (A/B through BYOLO_WSHIFT_PER_LAYER);
  * an activation beyond the split-f16 range is an ERROR (BYOLO_ERR_RANGE naming the layer), never a row of inf / NaN;
    lib_yolo.model.Model.run then re-runs the batch in the fp32 mode;
  * byolo_finalize refuses non-finite weights and falls back to fp32 for graphs split storage cannot express;
  * injected dropout masks (byolo_forward's d_mask_bits): parity of the dropout layers without oracle/rng.py.

Bound (north_star): |err| <= 1e-4 * max(1, |ref|) per value against the FLOAT64 oracle; on a column group where the
float32 oracle itself does not reach that on these ill-conditioned networks, the device may be no further from float64
than 1.1 x the float32 oracle is (conftest.assert_rows_close, `floor`) -- printed per group for both.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import (assert_rows_close, build_model, format_report, golden_images, golden_params, rows_report)
from oracle.report import allowance

VARIANT = "bayesian_yolov3_aleatoric"


# ------------------------------------------------------------------------------------------------------------------
# no GPU needed
# ------------------------------------------------------------------------------------------------------------------
def _tiny_engine(filters=32):
    from byolo import Engine, _lib
    e = Engine((32, 32, 3), 2)
    e.add_conv("a", filters, 3, 1, _lib.NORM_BN)
    e.add_conv("b", 32, 3, 2, _lib.NORM_BN | _lib.NORM_DROPOUT)
    e.add_detection("d", _lib.DET_STANDARD, [(0.1, 0.1), (0.2, 0.2), (0.3, 0.3)])
    return e


@pytest.mark.parametrize("name,bad", [("a/conv2d/kernel", np.nan), ("b/conv2d/kernel", np.inf), ("a/batch_normalization/gamma", np.inf),
                                      ("b/batch_normalization/moving_mean", np.nan), ("d/conv2d/bias", -np.inf)])
def test_finalize_refuses_non_finite_weights(name, bad):
    """The reference restores whatever the checkpoint holds (inference_epistemic.py:58); a NaN / inf weight would also
    poison the per-channel weight scales, so byolo_finalize names the variable instead (before it touches the device)."""
    from byolo import ByoloError, _lib
    e = _tiny_engine()
    shp = e.param_shapes()[name]
    v = np.full(shp, 0.01, dtype=np.float32)
    v.reshape(-1)[v.size // 2] = bad
    e.set_param(name, v)
    with pytest.raises(ByoloError) as ei:
        e.finalize()
    assert ei.value.code == _lib.ERR_ARG and name in str(ei.value) and "non-finite" in str(ei.value)


def test_finalize_refuses_a_variance_below_minus_eps():
    from byolo import ByoloError
    e = _tiny_engine()
    v = np.ones(e.param_shapes()["a/batch_normalization/moving_variance"], dtype=np.float32)
    v[3] = -1e-3                                            # rsqrt(var + 1e-5) of a negative number (layers.py:510-518)
    e.set_param("a/batch_normalization/moving_variance", v)
    with pytest.raises(ByoloError) as ei:
        e.finalize()
    assert "moving_variance" in str(ei.value)


def test_split_storage_falls_back_to_fp32_for_channel_counts_it_cannot_hold(monkeypatch):
    """A convolution with 30 output channels (no reference model has one) does not fit groups of 4: byolo_finalize
    chooses the fp32 mode and says so, instead of refusing the graph the fp32 mode and the reference accept."""
    from byolo import ByoloError
    monkeypatch.setenv("BYOLO_QUIET", "1")
    e = _tiny_engine(filters=30)
    assert e.precision == "split"
    try:
        e.finalize()
    except ByoloError as err:                              # no GPU here: the fall-back is decided before the first HIP call
        assert "hip" in str(err).lower()
    assert e.precision == "f32" and "groups of 4" in e.precision_note and "'a'" in e.precision_note
    e.set_precision("f32")
    assert e.precision_note == ""


def test_mask_layout_and_packing():
    e = _tiny_engine()
    assert e.num_dropout() == 1
    layout, words = e.mask_layout(3, 1)
    assert layout == [(0, 3 * 16 * 16 * 32)] and words == 3 * 16 * 16 * 32 // 32
    m = np.zeros((3, 16, 16, 32), dtype=bool)
    m.reshape(-1)[[0, 33, 24575]] = True
    buf = e.pack_masks([m], 3, 1)
    assert buf[0] == 1 and buf[1] == 2 and buf[-1] == 1 << 31 and buf.sum() == 1 + 2 + (1 << 31)


def test_adversarial_weights_are_what_they_say():
    from oracle import cpu_ref
    from byolo import synth
    shapes = cpu_ref.variable_shapes(VARIANT, 2)
    a = synth.adversarial_params(shapes, VARIANT, 2, "scales")
    b = synth.adversarial_params(shapes, VARIANT, 2, "scales")
    assert all(np.array_equal(a[k], b[k]) for k in a)                        # seeded
    k = a["det_net_2/conv_3/conv2d/kernel"]
    col = np.abs(k).reshape(-1, k.shape[-1]).max(0)
    assert col.max() / col.min() > 2.0 ** 15                                 # 2^+-10 per channel inside ONE layer
    g = a["darknet53/conv_20/batch_normalization/gamma"]
    assert g.max() > 20 and g.min() < 0.12
    assert np.all(a["det_net_1/conv_5/batch_normalization/gamma"] == 1)      # the layer in front of a detection head
    t = synth.adversarial_params(shapes, VARIANT, 2, "tails")
    k = t["det_net_1/conv_1/conv2d/kernel"]
    assert np.abs(k).max() / k.std() > 15                                    # Student-t outliers
    assert sum(bool(np.all(v == 250)) for n, v in t.items() if n.endswith("gamma")) >= 10


# ------------------------------------------------------------------------------------------------------------------
# MI355X
# ------------------------------------------------------------------------------------------------------------------
_CASES = {}


def _case(kind, size):
    """Weights (BN statistics calibrated on the device in the fp32 mode, ONE set for both precisions), images and the
    float64 / float32 oracle rows of one adversarial configuration."""
    key = (kind, size)
    if key in _CASES:
        return _CASES[key]
    import torch
    from oracle import cpu_ref
    from byolo import synth
    B, T = 2, 3
    yolo, m = build_model(VARIANT, size, size, T=T)
    eng = m.engine
    eng.set_precision("f32")
    eng.set_params(synth.adversarial_params(eng.param_shapes(), VARIANT, 2, kind, seed=7))
    eng.finalize()
    eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(2, size, size, seed=999)).cuda())
    params = eng.get_params()
    eng.close()
    imgs = synth.synthetic_images(B, size, size, seed=1234)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        f64 = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, torch.float64), imgs, VARIANT, T=T, seed=5, dtype=torch.float64,
                                   taps=(10, 36, 61, 74))
        ref32, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, VARIANT, T=T, seed=5)
    ref64 = f64[0].numpy()
    amax = {i: float(t.abs().max()) for i, t in f64[1]["layers"].items()}
    _CASES[key] = dict(params=params, imgs=imgs, ref64=ref64, ref32=ref32.numpy(), amax=amax, B=B, T=T)
    return _CASES[key]


def _device_rows(case, size, precision, monkeypatch, per_layer=False):
    import torch
    monkeypatch.setenv("BYOLO_PRECISION", precision)
    monkeypatch.setenv("BYOLO_WSHIFT_PER_LAYER", "1" if per_layer else "0")
    yolo, m = build_model(VARIANT, size, size, T=case["T"], params=case["params"])
    m.finalize()
    assert m.engine.precision == precision
    out = m.engine.forward(torch.from_numpy(case["imgs"]).cuda(), T=case["T"], seed=5, want_boxes=True)     # raises on BYOLO_ERR_RANGE
    torch.cuda.synchronize()
    assert m.engine.status() == (0, -1) or precision == "f32"
    return m, out


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["split", "f32"])
@pytest.mark.parametrize("size", [416, 608])
@pytest.mark.parametrize("kind", ["scales", "tails"])
def test_adversarial_weights_parity(kind, size, precision, monkeypatch):
    from test_gpu_parity import _check_nms_against_oracle
    case = _case(kind, size)
    m, out = _device_rows(case, size, precision, monkeypatch)
    boxes = out["boxes"].cpu().numpy()
    floor = rows_report(case["ref32"], case["ref64"], VARIANT)
    what = "%s weights %dx%d B=2 T=3, %s" % (kind, size, size, precision)
    print("%s: largest |activation| of the float64 run at layers 10 / 36 / 61 / 74: %s" % (what, ", ".join("%.3g" % case["amax"][i] for i in (10, 36, 61, 74))))
    print("%s: float32 oracle vs float64: %s" % (what, format_report(floor)))
    rep = assert_rows_close(boxes, case["ref64"], VARIANT, what + " vs the float64 oracle", allowed=allowance(floor))         # E(g) <= max(1, F(g))
    print("%s: device vs float64: %s" % (what, format_report(rep)))
    vs32 = assert_rows_close(boxes, case["ref32"], VARIANT, what + " vs the float32 oracle", allowed=allowance(floor, "float32"))     # D(g) <= max(1, F(g)) + F(g)
    print("%s: device vs float32 oracle: %s" % (what, format_report(vs32)))
    _check_nms_against_oracle(boxes, out, VARIANT)             # kept indices / gathered rows bit-exact on the device's rows


@pytest.mark.gpu
def test_per_channel_weight_scales_beat_one_scale_per_layer(monkeypatch):
    """'scales' weights: inside one layer the filters differ by up to 2^20.  With ONE power of two per layer the small
    filters' hi/lo pairs lose up to 20 of their 22 bits; with one per output channel (folded into the BN scale, exactly)
    every filter keeps them.  Measured against the float64 oracle, per column group."""
    case = _case("scales", 416)
    _, a = _device_rows(case, 416, "split", monkeypatch)
    rep_c = rows_report(a["boxes"].cpu().numpy(), case["ref64"], VARIANT)
    try:
        _, b = _device_rows(case, 416, "split", monkeypatch, per_layer=True)
        rep_l = rows_report(b["boxes"].cpu().numpy(), case["ref64"], VARIANT)
    except Exception as e:                                           # a range error is a legitimate outcome of the coarse scaling
        print("per-layer scales:", e)
        return
    print("per-channel scales:", format_report(rep_c))
    print("per-layer scales:  ", format_report(rep_l))
    worst_c = max(v["worst_in_bounds"] for k, v in rep_c.items() if k != "ids")
    worst_l = max(v["worst_in_bounds"] for k, v in rep_l.items() if k != "ids")
    assert worst_l > 3.0 * worst_c, "one scale per layer was expected to lose the small filters (%.2f vs %.2f bounds)" % (worst_l, worst_c)


def _overflowing_params(gamma=3e4):
    p = golden_params(VARIANT)
    p = {k: v.copy() for k, v in p.items()}
    p["darknet53/conv_10/batch_normalization/gamma"][:] = gamma      # post-BN activations ~ N(0, gamma^2): far beyond 16376
    return p


@pytest.mark.gpu
def test_an_activation_beyond_the_split_range_is_an_error_not_inf_rows(monkeypatch):
    import torch
    from byolo import ByoloError, _lib
    monkeypatch.setenv("BYOLO_PRECISION", "split")
    params = _overflowing_params()
    yolo, m = build_model(VARIANT, 64, 96, T=3, params=params)
    m.finalize()
    eng = m.engine
    x = torch.from_numpy(golden_images(1)).cuda()
    with pytest.raises(ByoloError) as ei:
        eng.forward(x, T=3, seed=42, want_boxes=True)
    assert ei.value.code == _lib.ERR_RANGE and "darknet53/conv_10" in str(ei.value) and "16376" in str(ei.value)
    assert eng.status() == (0, -1)                                   # the failing call cleared the words for the next one
    # deferred form: nothing waits inside forward, the caller asks where it synchronises anyway
    eng.set_async(True)
    out = eng.forward(x, T=3, seed=42, want_boxes=True)
    flags, layer = eng.status()
    names = [n for n in eng.param_shapes() if n.endswith("/conv2d/kernel")]
    assert flags & 1 and layer >= 0
    with pytest.raises(ByoloError) as ei:
        eng.check_status()
    assert "darknet53/conv_10" in str(ei.value), (str(ei.value), names[:12])
    raws = [dl.raw_output.cpu().numpy() for dl in m.det_layers]
    assert not all(np.isfinite(r).all() for r in raws)               # what the caller would have got without the check
    eng.clear_status()
    assert eng.status() == (0, -1)
    eng.set_async(False)
    # the same weights in the fp32 mode: plain numbers, as in the reference
    from oracle import cpu_ref
    eng.set_precision("f32")
    eng.finalize()
    ok = eng.forward(x, T=3, seed=42, want_boxes=True)["boxes"].cpu().numpy()
    with torch.no_grad():
        f = cpu_ref.forward(cpu_ref.to_torch_params(params), golden_images(1), VARIANT, T=3, seed=42)
    for k, dl in enumerate(m.det_layers):                            # raw detection outputs of ~1e5: finite, and the oracle's
        r = dl.raw_output.cpu().numpy()
        assert np.isfinite(r).all() and np.abs(r).max() > 1e3
        want = f["raw"][k].numpy()                                       # sums of ~1e6-sized terms: float32 agreement relative to the tensor's scale
        assert np.abs(r - want).max() <= 1e-4 * np.abs(want).max()
    # Model.run: the reference-shaped entry point re-runs the batch in the fp32 mode by itself
    yolo2, m2 = build_model(VARIANT, 64, 96, T=3, params=params)
    m2.finalize()
    assert m2.engine.precision == "split"
    res = m2.run(x, seed=42)
    got = res["boxes"].cpu().numpy()
    # ... THAT batch only (round 5): on the fp32 twin handle -- both weight packs stay resident --, the model stays in split-f16
    assert res["precision"] == "f32" and m2.engine.precision == "split" and m2.precision_switches == 2
    assert np.array_equal(got.view(np.uint32), ok.view(np.uint32))
    assert m2.engine.status() == (0, -1)                               # the words were cleared for the next batch
    for k, dl in enumerate(m2.det_layers):                             # the accessors read the handle that ran the batch
        assert np.isfinite(dl.raw_output.cpu().numpy()).all()
    res = m2.run(x, seed=43)
    assert res["precision"] == "f32" and m2.engine.precision == "split" and m2.precision_switches == 4


@pytest.mark.gpu
def test_a_non_finite_detection_output_is_an_error_in_split_mode(monkeypatch):
    import torch
    from byolo import ByoloError, _lib
    monkeypatch.setenv("BYOLO_PRECISION", "split")
    p = {k: v.copy() for k, v in golden_params(VARIANT).items()}
    p["det_net_2/detection/conv2d/kernel"][:] = 1e36                 # conv ~ 1e37 .. 1e38, + 3e38: float32 overflows (the reference's too)
    p["det_net_2/detection/conv2d/bias"][:] = 3e38
    yolo, m = build_model(VARIANT, 64, 96, T=3, params=p)
    m.finalize()
    with pytest.raises(ByoloError) as ei:
        m.engine.forward(torch.from_numpy(golden_images(1)).cuda(), T=3, seed=1, want_boxes=True)
    assert ei.value.code == _lib.ERR_RANGE and "detection output" in str(ei.value)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["split", "split-winograd", "f32"])
def test_injected_dropout_masks(precision, monkeypatch):
    """tf.layers.dropout draws unseeded noise (layers.py:521-524): ANY Bernoulli(0.9) array is a run of the reference.
    Masks drawn by numpy go to the oracle (cpu_ref.forward(masks=...), pinned to the reference's call order by
    tests/test_oracle_vs_reference.py) and, packed as bits, to byolo_forward: the rows must agree at the bound -- with
    no restatement of csrc/byolo_rng.h in between.  And fed its OWN stream as bits, the device reproduces its seeded run
    bit for bit (injection and hash index the same elements)."""
    import torch
    from oracle import cpu_ref, rng
    if precision == "split-winograd":      # the six big head convolutions through csrc/wino_split.hip (its epilogue reads the bits too)
        precision = "split"
        monkeypatch.setenv("BYOLO_WINO_SPLIT", "2")
    monkeypatch.setenv("BYOLO_PRECISION", precision)
    B, T = 2, 3
    params = golden_params(VARIANT)
    yolo, m = build_model(VARIANT, 64, 96, T=T, params=params)
    m.finalize()
    eng = m.engine
    layout, words = eng.mask_layout(B, T)
    assert len(layout) == 15
    imgs = golden_images(B)
    with torch.no_grad():
        probe = cpu_ref.forward(cpu_ref.to_torch_params(params), imgs, VARIANT, T=T, seed=0, taps="all")
    shapes = [tuple(probe["layers"][i].shape) for i, l in enumerate(probe["topo"]) if l["op"] == "conv" and l["norm"] == "dropout_bn"]
    assert [int(np.prod(s)) for s in shapes] == [n for _, n in layout]
    g = np.random.default_rng(20260929)
    masks = [g.random(s) < 0.9 for s in shapes]
    x = torch.from_numpy(imgs).cuda()
    bits = torch.from_numpy(eng.pack_masks(masks, B, T).view(np.int32)).cuda()
    eng.set_profiling(2)
    got = eng.forward(x, T=T, seed=12345, want_boxes=True, mask_bits=bits)["boxes"].cpu().numpy()
    wino = [s for s in eng.step_profile() if s["variant"] == 140]
    eng.set_profiling(0)
    assert bool(wino) == (os.environ.get("BYOLO_WINO_SPLIT") == "2"), "Winograd-in-split launches: %d" % len(wino)
    with torch.no_grad():
        ref64, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, torch.float64), imgs, VARIANT, T=T, seed=777, dtype=torch.float64, masks=masks)
        ref32, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, VARIANT, T=T, seed=777, masks=masks)
    floor = rows_report(ref32.numpy(), ref64.numpy(), VARIANT)
    rep = assert_rows_close(got, ref64.numpy(), VARIANT, "injected masks (%s) vs the float64 oracle" % precision, allowed=allowance(floor))
    assert_rows_close(got, ref32.numpy(), VARIANT, "injected masks (%s) vs the float32 oracle" % precision, allowed=allowance(floor, "float32"))
    print("injected masks (%s): %s | float32 oracle vs float64: %s" % (precision, format_report(rep), format_report(floor)))
    # the library's own stream, handed back as bits
    own = [rng.keep_mask(4242, k, s, 0.1) for k, s in enumerate(shapes)]
    a = eng.forward(x, T=T, seed=4242, want_boxes=True)["boxes"].cpu().numpy()
    b = eng.forward(x, T=T, seed=1, want_boxes=True, mask_bits=torch.from_numpy(eng.pack_masks(own, B, T).view(np.int32)).cuda())["boxes"].cpu().numpy()
    # injected bits index THIS call's tensors: the position of the call in a larger logical batch (first_image) does not enter
    b3 = eng.forward(x, T=T, seed=1, want_boxes=True, first_image=3, mask_bits=torch.from_numpy(eng.pack_masks(own, B, T).view(np.int32)).cuda())["boxes"].cpu().numpy()
    assert np.array_equal(b.view(np.uint32), b3.view(np.uint32)), "injected masks depend on first_image"
    if precision == "split":
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "injected bits and the counter hash disagree on an element"
    else:        # fp32 mode: the injected call runs the convolutions on other kernels (no Winograd): same masks, float32 rounding apart
        assert np.nanmax(np.abs(a - b) / np.maximum(1.0, np.abs(a))) < 1e-4


@pytest.mark.gpu
def test_fp32_fallback_graph_runs(monkeypatch):
    """The 30-channel graph of the CPU test above, on the device: finalize picks the fp32 mode, the forward runs."""
    import torch
    from byolo import synth
    monkeypatch.setenv("BYOLO_QUIET", "1")
    monkeypatch.setenv("BYOLO_PRECISION", "split")
    e = _tiny_engine(filters=30)
    e.set_params(synth.base_params(e.param_shapes(), "yolov3", 2, seed=3))
    e.finalize()
    assert e.precision == "f32" and e.precision_note
    out = e.forward(torch.from_numpy(synth.synthetic_images(2, 32, 32, seed=1)).cuda(), want_boxes=True)
    torch.cuda.synchronize()
    assert np.isfinite(out["boxes"].cpu().numpy()).all()
