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


def assert_close(a, b, what, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "%s: NaN pattern differs" % what
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = (err > tol) & ~nan_a
    if bad.any():
        i = np.unravel_index(np.nanargmax(np.where(nan_a, 0, err - tol)), a.shape)
        raise AssertionError("%s: %d / %d elements out of tolerance; worst at %s: %r vs %r (err %.3e)"
                             % (what, bad.sum(), a.size, i, a[i], b[i], err[i]))
    return float(np.nanmax(err)) if err.size else 0.0
