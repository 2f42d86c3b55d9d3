import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "bayesian-yolov3_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---------------------------------------------------------------------------------------------
# shared helpers
# ---------------------------------------------------------------------------------------------
RTOL = ATOL = 1e-4      # BASELINE.json north_star: coords / scores / sigma within 1e-4 fp32


def golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".json"):
        return json.load(open(path))
    return np.load(path)


@pytest.fixture(scope="session")
def fwd_meta():
    return golden("fwd_meta.json")


def golden_params(variant, cls_cnt=2):
    """Seeded base weights + the calibrated BN statistics the fixtures were generated with."""
    from oracle import cpu_ref
    from byolo import synth
    shapes = cpu_ref.variable_shapes(variant, cls_cnt)
    p = synth.base_params(shapes, variant, cls_cnt, seed=7)
    stats = golden("bn_stats.npz")
    for k in stats.files:
        p[k] = stats[k].astype(np.float32)
    return p


def golden_images(B):
    from byolo import synth
    return synth.synthetic_images(B, 64, 96, seed=1234)


def make_config(variant, H, W, T=3, cls_cnt=2, **kw):
    from lib_yolo import yolov3
    c = {"full_img_size": [H, W, 3], "crop": False, "cls_cnt": cls_cnt, "priors": yolov3.ECP_9_PRIORS,
         "aleatoric_loss": False, "inference_mode": True, "T": T, "implicit_background_class": True}
    c.update(kw)
    return c


def build_model(variant, H, W, T=3, params=None, engine_options=None, B=None, **kw):
    """Build the product model through the reference-shaped API."""
    from lib_yolo import yolov3, model
    cfg = make_config(variant, H, W, T=T, engine_options=engine_options or {}, **kw)
    yolo = getattr(yolov3, variant)(cfg)
    m = yolo.init_model(inputs=model.Placeholder((B, H, W, 3)), training=False).get_model()
    if params is not None:
        m.engine.set_params(params)
    return yolo, m


def _literal_tol(ref, atol, rtol):
    """north_star: "within 1e-4 fp32" -- absolute 1e-4 for |v| <= 1, relative 1e-4 beyond (an fp32 value of
    magnitude 10 has an ulp of 1e-6 and a 75-layer fp32 network a relative error of ~1e-5: no fp32 evaluation can
    hold an ABSOLUTE 1e-4 on it).  One bound, not the sum of the two."""
    return np.maximum(atol, rtol * np.abs(ref))


def assert_close(a, b, what, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "%s: NaN pattern differs" % what
    err = np.abs(a - b)
    tol = _literal_tol(b, atol, rtol)
    bad = (err > tol) & ~nan_a
    if bad.any():
        i = np.unravel_index(np.nanargmax(np.where(nan_a, 0, err - tol)), a.shape)
        raise AssertionError("%s: %d / %d elements out of tolerance; worst at %s: %r vs %r (err %.3e)"
                             % (what, bad.sum(), a.size, i, a[i], b[i], err[i]))
    return float(np.nanmax(err)) if err.size else 0.0


def column_groups(variant, C=2):
    """Columns of a pre-NMS row by meaning (lib_yolo/layers.py:250-258, :330-346, :480-499).
    `(exp)`: exp(logvar) of network outputs (layers.py:309-313, :465-468) -- unbounded, the only columns beyond 1."""
    if variant == "yolov3":
        return {"coords": list(range(0, 4)), "scores": list(range(4, 5 + C))}
    if variant == "yolov3_aleatoric":
        return {"coords": list(range(0, 4)), "sigma_ale(exp)": [4, 5, 6, 7, 8],
                "scores": [9] + list(range(11, 11 + C)), "entropy": [10, 11 + C], "ids": [12 + C, 13 + C]}
    return {"coords": list(range(0, 4)), "sigma_epi": [4, 5, 6, 7, 12], "sigma_ale(exp)": [8, 9, 10, 11, 13],
            "scores": [14] + list(range(17, 17 + C)), "mutual_info/entropy": [15, 16, 17 + C, 18 + C],
            "ids": [19 + C, 20 + C]}


def rows_report(got, ref, variant, C=2):
    """Per column group: max |err|, max |ref|, max relative err over |ref| > 1, and the worst error in units of the
    north_star's bound taken literally, 1e-4 * max(1, |ref|)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    rep = {}
    for name, cols in column_groups(variant, C).items():
        g, r = got[..., cols], ref[..., cols]
        ok = np.isfinite(r) & np.isfinite(g)
        r0 = np.where(ok, r, 0)
        err = np.where(ok, np.abs(g - r), 0.0)
        big = ok & (np.abs(r) > 1)
        units = err / _literal_tol(r0, ATOL, RTOL)
        rep[name] = dict(max_abs_err=float(err.max()), max_ref=float(np.abs(r0).max()),
                         max_rel_err_over_1=float((err[big] / np.abs(r[big])).max()) if big.any() else 0.0,
                         worst_in_bounds=float(units.max()), ref_at_worst=float(r0.reshape(-1)[int(units.argmax())]),
                         nonfinite=int((~ok).sum()))
    return rep


def assert_rows_close_vs_oracle(got, params, imgs, variant, what, T=1, seed=0, cls_cnt=2, **kw):
    """Pre-NMS rows against the oracle run in FLOAT64 (the exact value of the reference's graph), at the literal bound --
    or, on a column group where the oracle's own float32 run does not reach it (variances over two or three MC samples,
    exp(logvar) of a single pass), no further from the float64 result than that float32 run (x 1.1).  Two float32-grade
    evaluations may differ from EACH OTHER by about the bound there, so the float32 run is the
    yardstick, not the reference, on such groups.  Also checks the device against the float32 run loosely (2 bounds: a
    wrong result is off by orders of magnitude).  Returns the report against float64."""
    import torch
    from oracle import cpu_ref
    with torch.no_grad():
        ref64, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, torch.float64), imgs, variant, T=T, seed=seed, cls_cnt=cls_cnt,
                                        dtype=torch.float64, **kw)
        ref32, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=seed, cls_cnt=cls_cnt, **kw)
    floor = rows_report(ref32.numpy(), ref64.numpy(), variant, cls_cnt)
    rep = assert_rows_close(got, ref64.numpy(), variant, what + " vs the float64 oracle", C=cls_cnt, floor=floor)
    loose = rows_report(got, ref32.numpy(), variant, cls_cnt)
    assert all(v["worst_in_bounds"] <= 2.0 for v in loose.values()), "%s vs the float32 oracle: %s" % (what, format_report(loose))
    print("%s: vs float64 %s | float32 oracle vs float64 %s" % (what, format_report(rep), format_report(floor)))
    return rep


def format_report(rep):
    return "; ".join("%s: |err| %.2e (|ref| <= %.3g, rel>1 %.1e, %.2f of bound at ref %.3g)"
                     % (k, v["max_abs_err"], v["max_ref"], v["max_rel_err_over_1"], v["worst_in_bounds"], v["ref_at_worst"])
                     for k, v in rep.items())


def assert_rows_close(got, ref, variant, what, C=2, floor=None):
    """Pre-NMS rows against the oracle, per column group, at the north_star's literal bound: ids exact, everything
    else |err| <= 1e-4 * max(1, |ref|); NaN / inf patterns equal (entropies are NaN exactly at saturated
    probabilities, layers.py:349-358).  `floor`: a rows_report of the float32 CPU restatement against the SAME
    (float64) reference -- where float32 arithmetic itself does not reach the bound on a group, the device must be
    no further from the reference than that float32 evaluation (x 1.1).  Returns the per-group report."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "%s: NaN pattern differs" % what
    assert np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)]), "%s: inf pattern" % what
    rep = rows_report(got, ref, variant, C)
    if "ids" in rep:
        assert rep["ids"]["max_abs_err"] == 0.0, "%s: layer / prior ids differ" % what
    allowed = {k: max(1.0, 1.1 * floor[k]["worst_in_bounds"]) if floor else 1.0 for k in rep}
    bad = {k: v for k, v in rep.items() if v["worst_in_bounds"] > allowed[k]}
    assert not bad, "%s: beyond 1e-4 * max(1, |ref|)%s: %s" % (what, " and beyond the float32 floor" if floor else "", format_report(bad))
    return rep
