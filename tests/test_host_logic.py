"""Host logic without a GPU: the reference-shaped Python API (lib_yolo mirror), graph construction
and lowering errors, variable / layer naming, workspace planning, cost model, Darknet loader."""
import copy

import numpy as np
import pytest

from conftest import make_config, build_model, golden

VARIANTS = ("yolov3", "yolov3_aleatoric", "bayesian_yolov3_aleatoric")


@pytest.mark.parametrize("variant", VARIANTS)
def test_names_match_reference(variant, fwd_meta):
    """Variable names (creation order) and layer tensor names as the reference produced them."""
    yolo, m = build_model(variant, 64, 96)
    meta = fwd_meta[variant]
    assert list(m.engine.param_shapes()) == meta["var_names"]
    assert len(m.layers) == meta["n_layers"] == m.engine.num_layers()
    mine = [l.name for l in m.layers]
    for a, b in zip(mine, meta["layer_names"]):
        # op-type suffixes of view layers are cosmetic; conv / residual / detection names are load-bearing
        if "LeakyRelu" in b or "detection" in b or "residual" in b:
            assert a == b
    assert (m.obj_idx, m.cls_start_idx) == {"yolov3": (4, 5), "yolov3_aleatoric": (9, 11)}.get(variant, (14, 17))
    assert [(d.h, d.w, d.downsample) for d in m.det_layers] == [(2, 3, 32), (4, 6, 16), (8, 12, 8)]
    assert m.engine.num_boxes() == (378, {"yolov3": 7, "yolov3_aleatoric": 16}.get(variant, 23))
    assert m.matches_blueprint(yolo.blueprint)
    assert m.dn_out.index == 74 and m.det_net_1_out.name == "det_net_1/detection/conv2d/BiasAdd:0"


def test_flops_match_baseline_table():
    """BASELINE.md section 2 (GFLOP per image, conv FLOPs as written)."""
    for variant, hw, T, want in (("yolov3", 416, 1, 65.30), ("yolov3_aleatoric", 416, 1, 65.35),
                                 ("bayesian_yolov3_aleatoric", 416, 10, 212.20), ("bayesian_yolov3_aleatoric", 608, 30, 1150.34),
                                 ("bayesian_yolov3_aleatoric", 1024, 50, 5240.29)):
        _, m = build_model(variant, hw, hw, T=T)
        assert abs(m.engine.flops(1, T) / 1e9 - want) < 0.01, (variant, hw, T)
    _, m = build_model("bayesian_yolov3_aleatoric", 608, 608, T=30)
    assert m.engine.num_boxes() == (22743, 23)


def test_config_contract():
    from lib_yolo import yolov3, model
    # required keys (yolov3.py:315, :456, :461, :467)
    cfg = make_config("x", 64, 64)
    for cls, key in ((yolov3.yolov3_aleatoric, "aleatoric_loss"), (yolov3.bayesian_yolov3_aleatoric, "aleatoric_loss"),
                     (yolov3.bayesian_yolov3_aleatoric, "inference_mode"), (yolov3.bayesian_yolov3_aleatoric, "T")):
        c = dict(cfg); del c[key]
        with pytest.raises(KeyError):
            cls(c)
    with pytest.raises(AssertionError):                         # % 32 (yolov3.py:207)
        yolov3.yolov3(make_config("x", 100, 64))
    y = yolov3.yolov3(cfg)
    with pytest.raises(AssertionError, match="Call init_model first"):
        y.get_model()
    y.init_model(model.Placeholder((1, 64, 64, 3)), training=False)
    with pytest.raises(Exception, match="only be initialized once"):
        y.init_model(model.Placeholder((1, 64, 64, 3)), training=False)
    with pytest.raises(NotImplementedError):
        yolov3.yolov3(cfg).init_model(model.Placeholder((1, 64, 64, 3)), training=True)
    # Bayesian model outside inference mode: plain dropout network with the aleatoric decode, no stacking
    _, m = build_model("bayesian_yolov3_aleatoric", 64, 64, inference_mode=False)
    assert len(m.layers) == 104 and m.T == 1 and all(d.kind == 1 for d in m.det_layers)


def test_crop_rescales_and_mutates_priors():
    """model.py:10-15 incl. the in-place mutation of the shared table (App. D.13)."""
    from lib_yolo import yolov3, model
    table = copy.deepcopy(yolov3.ECP_9_PRIORS)
    cfg = make_config("x", 128, 256, priors=table, crop=True, crop_img_size=[64, 64, 3])
    h0 = table[32][0].h
    img_size, pri = model.img_size_and_priors_if_crop(cfg)
    assert img_size == [64, 64, 3] and pri is table
    assert table[32][0].h == h0 * 2.0 and abs(table[8][2].w - yolov3.ECP_9_PRIORS[8][2].w * 4.0) < 1e-12


def test_graph_errors():
    from byolo import Engine, ByoloError
    with pytest.raises(ByoloError, match="cls_cnt outside"):
        Engine((64, 64, 3), 129)
    e = Engine((64, 64, 3), 2)
    with pytest.raises(ByoloError, match="invalid kernel size"):
        e.add_conv("a", 8, 5, 1, 1)
    with pytest.raises(ByoloError, match="invalid strides"):
        e.add_conv("a", 8, 3, 3, 1)
    with pytest.raises(ByoloError, match="too many routes"):
        e.add_route([0, 0, 0])
    with pytest.raises(ByoloError, match="too few routes"):
        e.add_route([])
    i0 = e.add_conv("a", 32, 3, 1, 1)
    with pytest.raises(ByoloError, match="duplicate scope"):
        e.add_conv("a", 32, 3, 1, 1)
    i1 = e.add_conv("b", 64, 3, 2, 1)
    with pytest.raises(ByoloError, match="shape mismatch"):
        e.add_residual(0)
    with pytest.raises(ByoloError, match="bad route"):
        e.add_route([17])
    assert (i0, i1) == (0, 1)
    with pytest.raises(ByoloError, match="no detection layer"):
        e.workspace_bytes(1, 1)
    e.add_detection("d/detection", 1, [(0.1, 0.1)] * 3)
    with pytest.raises(ByoloError, match="mixed detection kinds"):
        e.add_detection("d2/detection", 0, [(0.1, 0.1)] * 3)
    assert e.workspace_bytes(2, 1) > 0
    with pytest.raises(ByoloError, match="frozen"):
        e.add_upsample()
    with pytest.raises(ByoloError, match="unknown variable"):
        e.set_param("nope", np.zeros(3))
    with pytest.raises(ByoloError, match="expects"):
        e.set_param("a/conv2d/kernel", np.zeros(3))
    # a residual whose producer is shared cannot ride in that convolution's epilogue: it becomes a step of its own
    e2 = Engine((64, 64, 3), 2)
    e2.add_conv("a", 32, 3, 1, 1); e2.add_conv("b", 32, 1, 1, 1); e2.add_residual(0); e2.add_route([1])
    e2.add_detection("d/detection", 0, [(0.1, 0.1)] * 3)
    assert e2.workspace_bytes(1, 1) > 0


def test_absurd_sizes_are_refused_not_allocated():
    """A builder call with a size no network has is a status, never a 200 GB std::vector (filters = INT_MAX used to be one: on a host
    with enough memory to start filling it, the random builder sweep -- tools/fuzz_builder.py -- took the machine down) and never a C++
    exception through the C-ABI.  Child process under an address-space limit: the refusal must not depend on the allocation failing."""
    import os
    import resource
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
from byolo import Engine, ByoloError
e = Engine((96, 64, 3), 2)
for args, what in ((("a", 2**31 - 1, 3, 1, 1), "filters"), (("a", 1 << 20, 1, 1, 1), "filters"), (("a", 65536, 3, 1, 1), None)):
    try:
        e.add_conv(*args)
        assert what is None, args
    except ByoloError as ex:
        assert what and what in str(ex), (args, str(ex))
try:
    e.add_conv("b", 65536, 3, 1, 1)           # 9 * 65536 * 65536 weights
    raise SystemExit("a 38 G-weight kernel was accepted")
except ByoloError as ex:
    assert "weights" in str(ex), str(ex)
e.add_conv("c", 32, 1, 1, 1)
e.add_detection("d/detection", 0, [(0.1, 0.1)] * 3)
for b, t in ((2**31 - 1, 2**31 - 1), (1 << 20, 30), (8, 1 << 20)):
    try:
        e.workspace_bytes(b, t)
    except ByoloError:
        pass
print("OK")
""" % os.path.join(REPO, "bayesian-yolov3_amd")

    def limit():
        resource.setrlimit(resource.RLIMIT_AS, (16 << 30, 16 << 30))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, preexec_fn=limit)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.returncode, p.stdout[-500:], p.stderr[-1500:])


def test_random_builder_call_sequences_return_statuses():
    """tools/fuzz_builder.py (random call sequences against the graph-builder half of the C-ABI; no device needed; every child under an
    address-space limit), a short sweep: no crash, no inconsistent size."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "fuzz_builder.py"), "--runs", "10", "--seed", "150"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "0 crashed or inconsistent" in p.stdout, (p.stdout[-1500:], p.stderr[-500:])


def test_views_get_tensors_where_a_loader_cannot_express_them():
    """Route / upsample / stack layers are views inside the next convolution's loader; where that is not enough -- a
    view as residual shortcut (once read as if it were a tensor: a device fault), a concat of a concat, an upsample of
    an upsample -- the lowering copies the inner view into a tensor of its own (numerics: tools/fuzz_graph.py and
    tests/test_gpu_layers.py on the GPU)."""
    from byolo import Engine
    pri = [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)]
    eng = Engine((64, 64, 3), 2)
    eng.add_conv("a", 32, 3, 1, 1)
    s = eng.add_stack(0)
    eng.add_conv("b", 32, 3, 1, 1)
    eng.add_residual(s)                                   # shortcut = the stacked view of a
    eng.add_detection("d/detection", 2, pri)
    assert eng.workspace_bytes(1, 2) > 0
    eng2 = Engine((64, 64, 3), 2)
    a = eng2.add_conv("a", 32, 3, 2, 1)
    b = eng2.add_conv("b", 32, 1, 1, 1)
    c = eng2.add_route([a, b])
    d = eng2.add_conv("c", 32, 1, 1, 1)
    eng2.add_route([c, d])                                # nested concat: 64 + 32 channels
    eng2.add_upsample(); eng2.add_upsample()              # double upsample
    eng2.add_conv("e", 32, 3, 1, 1)
    eng2.add_detection("d/detection", 0, pri)
    assert eng2.workspace_bytes(2, 1) > 0


def test_forward_requires_finalize_and_device_tensors():
    import torch
    from byolo import ByoloError
    _, m = build_model("yolov3", 64, 64)
    with pytest.raises(TypeError):
        m.engine.forward(torch.zeros((1, 64, 64, 3)))            # CPU tensor: no CPU path in the product


def test_workspace_plan():
    v = "bayesian_yolov3_aleatoric"
    _, keep = build_model(v, 608, 608, T=30, engine_options={"keep_all_outputs": True})
    _, reuse = build_model(v, 608, 608, T=30)
    a, b = keep.engine.workspace_bytes(8, 30), reuse.engine.workspace_bytes(8, 30)
    assert b < a / 2                                              # liveness-based reuse pays
    assert reuse.engine.workspace_bytes(4, 30) < b < reuse.engine.workspace_bytes(8, 50)
    # the three 76x76 3x3 outputs (1.42 GB each) never exist (back-to-back fusion); a fused pair needs its 128-channel input
    # and its follower's 128-channel output at once
    live = 2 * 240 * 76 * 76 * 128 * 4
    assert live <= b < 6e9
    # 32-bit source offsets: one launch sequence takes as many images as keep every activation below 3 GiB; byolo_forward cuts a
    # larger batch into such pieces itself, in one workspace sized for the largest piece
    per_image = 30 * 76 * 76 * 256 * 4
    cap = reuse.engine.max_images(30)
    assert cap == 0xBFF00000 // per_image == 18
    at_cap = reuse.engine.workspace_bytes(cap, 30)
    assert at_cap > reuse.engine.workspace_bytes(cap - 1, 30)
    assert reuse.engine.workspace_bytes(cap + 1, 30) == at_cap == reuse.engine.workspace_bytes(64, 30) == reuse.engine.workspace_bytes(3 * cap, 30)
    assert reuse.engine.max_images(1) == 0xBFF00000 // (608 * 608 * 32 * 4)       # the stem's output is the largest tensor


def test_darknet_weight_loader(tmp_path):
    """lib_yolo/darknet.py:42-122: header, [beta, gamma, mean, var] then kernel as [cout,cin,kh,kw]."""
    yolo, m = build_model("yolov3", 64, 64)
    shapes = m.engine.param_shapes()
    names = [n for n in shapes if n.startswith("darknet53/")]
    rng = np.random.default_rng(0)
    want = {}
    blob = [np.array([0, 2, 0, 0, 0], dtype=np.int32).tobytes()]
    scopes = []
    for n in names:
        s = n.rsplit("/", 2)[0]
        if s not in scopes:
            scopes.append(s)
    for s in scopes:
        for v in ("beta", "gamma", "moving_mean", "moving_variance"):
            k = "%s/batch_normalization/%s" % (s, v)
            want[k] = rng.standard_normal(shapes[k]).astype(np.float32)
            blob.append(want[k].tobytes())
        k = s + "/conv2d/kernel"
        w = rng.standard_normal(shapes[k]).astype(np.float32)            # HWIO
        want[k] = w
        blob.append(np.ascontiguousarray(w.transpose(3, 2, 0, 1)).tobytes())   # darknet: [n, c, h, w]
    path = tmp_path / "darknet53.conv.74"
    path.write_bytes(b"".join(blob))
    assigned = yolo.load_darknet53_weights(str(path))
    assert len(assigned) == 52 * 5
    for k, w in want.items():
        assert np.array_equal(m.engine.get_param(k, shapes[k]), w), k
    # a truncated file trips the reference's `assert ptr == len(weights)` ... from the other side
    (tmp_path / "short").write_bytes(b"".join(blob)[:-400])
    with pytest.raises((AssertionError, ValueError)):
        yolo2, _ = build_model("yolov3", 64, 64)
        yolo2.load_darknet53_weights(str(tmp_path / "short"))


@pytest.mark.parametrize("case", ["yolov3/backbone", "yolov3/all", "bayesian_yolov3_aleatoric/all"])
def test_darknet_weight_loader_equals_the_references_own_loader(case, tmp_path):
    """SURVEY 8(f) item 2, pinned to REFERENCE OUTPUT: tests/golden/darknet_weights.json holds, per variable,
    the SHA-256 of what the reference's own importer (lib_yolo/darknet.py:42-122, run unmodified under the shim by
    oracle/make_golden_weights.py) assigned from a file of structureless seeded float32 words.  The same bytes are regenerated
    here -- five int32 header words + numpy's default_rng(seed).standard_normal stream, no reading of the format involved --,
    the product's loader runs, and every variable must hold the same bits.  `backbone`: yolo.load_darknet53_weights (the call the
    training scripts make, lib_yolo/yolov3.py:220-222); `all`: load_darknet_weights(model.layers, ...), head scopes included; the
    detection convolutions are skipped by the reference (no 'LeakyRelu' in their layer name, darknet.py:56) and stay untouched."""
    import hashlib
    from lib_yolo import darknet
    g = golden("darknet_weights.json")
    c = g["cases"][case]
    variant, which = case.split("/")
    yolo, m = build_model(variant, 64, 64, T=2)
    shapes = m.engine.param_shapes()
    before = {n: m.engine.get_param(n, shapes[n]).copy() for n in c["untouched"]}
    path = tmp_path / "synthetic.weights"
    with open(path, "wb") as f:
        f.write(np.asarray(g["header"], dtype=np.int32).tobytes())
        rng = np.random.default_rng(g["seed"])
        left = c["floats"]
        while left > 0:
            n = min(left, 1 << 22)
            f.write(rng.standard_normal(n, dtype=np.float32).tobytes())
            left -= n
    assigned = yolo.load_darknet53_weights(str(path)) if which == "backbone" else darknet.load_darknet_weights(m.layers, str(path))
    assert len(assigned) == c["assign_ops"] and sorted(assigned) == sorted(c["variables"])
    for name, d in c["variables"].items():
        got = np.ascontiguousarray(m.engine.get_param(name, shapes[name]), dtype=np.float32)
        assert list(got.shape) == d["shape"], name
        assert hashlib.sha256(got.tobytes()).hexdigest() == d["sha256"], "%s: not what the reference's loader assigns (first values %s vs %s)" % (
            name, got.reshape(-1)[:3], d["first"])
    for n in c["untouched"]:
        assert np.array_equal(m.engine.get_param(n, shapes[n]), before[n]), "%s must stay untouched (the reference's loader skips it)" % n


def test_winograd_f2x2_3x3_identity():
    """The algebra behind csrc/winograd.hip / wino_fused.hip, with the matrices as written there:
    for one 4x4 input patch d and one 3x3 filter g, A^T [(G g G^T) * (B^T d B)] A equals the 2x2 outputs of the
    stride-1 correlation, and the fused kernel's fold coefficients A^T[dy][i] * A^T[dx][j] reproduce A^T M A."""
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        M = (G @ g @ G.T) * (Bt @ d @ Bt.T)
        y = At @ M @ At.T
        ref = np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(2)] for i in range(2)])
        assert np.allclose(y, ref, atol=1e-12)
        folded = np.zeros((2, 2))
        for xi in range(16):
            i, j = xi >> 2, xi & 3
            for dy in range(2):
                for dx in range(2):
                    folded[dy, dx] += At[dy, i] * At[dx, j] * M[i, j]
        assert np.allclose(folded, y, atol=1e-12)
    assert int((np.abs(np.kron(At, At)) > 0).sum()) == 36      # the 36 non-zero (xi, output) pairs of the fold
