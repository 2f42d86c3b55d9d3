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
from oracle.report import RTOL, ATOL, _literal_tol, column_groups, rows_report, format_report, allowance, check      # noqa: E402,F401  (one definition, shared with bench.py's parity_note)


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


def assert_rows_close_vs_oracle(got, params, imgs, variant, what, T=1, seed=0, cls_cnt=2, **kw):
    """Pre-NMS rows under THE PARITY CONTRACT of oracle/report.py: against the oracle run in FLOAT64 (the exact value of the
    reference's graph) at max(1, F(g)); against the oracle run in FLOAT32 at max(1, F(g)) + F(g); F(g) = that float32 run's own
    distance from the float64 one, measured here on the same input.  All three distances are recorded per group
    (profiles/*_parity_table.json).  Returns the report against float64."""
    import torch
    from oracle import cpu_ref
    from oracle.report import allowance
    with torch.no_grad():
        ref64, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, torch.float64), imgs, variant, T=T, seed=seed, cls_cnt=cls_cnt,
                                        dtype=torch.float64, **kw)
        ref32, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params), imgs, variant, T=T, seed=seed, cls_cnt=cls_cnt, **kw)
    floor = rows_report(ref32.numpy(), ref64.numpy(), variant, cls_cnt)
    record_parity(what + ": float32 oracle vs float64 oracle (the floor)", floor)
    rep = assert_rows_close(got, ref64.numpy(), variant, what + " vs the float64 oracle", C=cls_cnt, allowed=allowance(floor))
    loose = assert_rows_close(got, ref32.numpy(), variant, what + " vs the float32 oracle", C=cls_cnt, allowed=allowance(floor, "float32"))
    print("%s: device vs float64 %s | device vs float32 %s | float32 oracle vs float64 %s"
          % (what, format_report(rep), format_report(loose), format_report(floor)))
    return rep


def record_parity(what, rep, kind="rows"):
    """Every per-group distance a GPU test computes goes into ONE json file instead of `pytest -q`'s swallowed stdout:
    gpurun_out/parity_table.json (BYOLO_PARITY_TABLE overrides; gpurun merges that directory back, the builder copies the table to
    profiles/rN_parity_table.json).  {comparison: {column group: {worst_in_bounds, max_abs_err, max_ref}}}, in units of the bound
    1e-4 * max(1, |ref|); the precision in effect (BYOLO_PRECISION) is part of the key."""
    import json
    path = os.environ.get("BYOLO_PARITY_TABLE") or os.path.join(REPO, "gpurun_out", "parity_table.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        table = json.load(open(path)) if os.path.exists(path) else {}
        key = "%s [%s]" % (what, os.environ.get("BYOLO_PRECISION", "default"))
        table[key] = {k: {"worst_in_bounds": round(v["worst_in_bounds"], 4), "max_abs_err": float("%.3g" % v["max_abs_err"]),
                          "max_ref": float("%.4g" % v["max_ref"])} for k, v in rep.items()}
        with open(path + ".tmp", "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
        os.replace(path + ".tmp", path)
    except OSError:
        pass                                    # a read-only tree must not fail a parity test


def assert_rows_close(got, ref, variant, what, C=2, allowed=None):
    """Pre-NMS rows against the oracle, per column group, at the north_star's literal bound: ids exact, everything
    else |err| <= 1e-4 * max(1, |ref|); NaN / inf patterns equal (entropies are NaN exactly at saturated
    probabilities, layers.py:349-358).  `allowed`: oracle.report.allowance(floor[, "float32"]) -- derived from the float32 oracle's
    own distance from the float64 run MEASURED IN THE SAME TEST (THE PARITY CONTRACT, oracle/report.py); there is no constant
    above 1 anywhere.  Returns the per-group report."""
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "%s: NaN pattern differs" % what
    assert np.array_equal(np.isinf(got), np.isinf(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)]), "%s: inf pattern" % what
    rep = rows_report(got, ref, variant, C)
    record_parity(what, rep)
    if "ids" in rep:
        assert rep["ids"]["max_abs_err"] == 0.0, "%s: layer / prior ids differ" % what
    allowed = allowed or {}
    bad = {k: v for k, v in rep.items() if v["worst_in_bounds"] > allowed.get(k, 1.0)}
    assert not bad, "%s: beyond 1e-4 * max(1, |ref|)%s: %s" % (what, (" and beyond the allowance from the float32 oracle's own measured distance from float64 %s"
                                                                     % {k: round(allowed[k], 3) for k in bad if k in allowed}) if allowed else "", format_report(bad))
    return rep
