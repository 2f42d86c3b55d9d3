"""Layer-level numerics of the HIP convolution family through the C-ABI graph builder, against a PLAIN PyTorch fp32
reference of the same ops written out here (F.conv2d, explicit pads, the dropout mask of oracle/rng.py).

The graphs are not YOLOv3: they are chosen to hit the launch shapes the network itself never produces -- output
channels that do not fill a column tile (96, 40), an input-channel count that is not a multiple of 32 (generic
direct kernel), row counts that do not fill a row tile, the Darknet stride-2 pad, a fused residual, the two-source
(upsampled + plain) concat loader, dropout in a non-stacked graph, a detection head with bias.
Tolerance: 1e-4 abs / rel like everywhere else (fp32, different summation order)."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

BN, DROP = 1, 2


@pytest.fixture(autouse=True, params=["split", "f32"])
def precision(request, monkeypatch):
    """Both arithmetic modes of the convolution stack (include/byolo.h: BYOLO_PREC_SPLIT_F16, the default, and
    BYOLO_PREC_F32); tests of fp32-only machinery skip the other one."""
    monkeypatch.setenv("BYOLO_PRECISION", request.param)
    return request.param


def _ref_conv(x, p, scope, k, stride, flags, drop=None):
    """x NHWC torch fp32 -> conv (HWIO kernel) -> [dropout] -> [BN, leaky 0.1];  lib_yolo/layers.py:533-575"""
    import torch
    import torch.nn.functional as F
    w = torch.from_numpy(p[scope + "/conv2d/kernel"]).permute(3, 2, 0, 1)
    xin = x.permute(0, 3, 1, 2)
    if k == 3 and stride == 2:
        xin = F.pad(xin, (1, 0, 1, 0))                     # Darknet downsample: one row / column on top / left
        y = F.conv2d(xin, w, stride=2)
    else:
        y = F.conv2d(xin, w, stride=stride, padding=(k - 1) // 2)
    y = y.permute(0, 2, 3, 1).contiguous()
    if drop is not None:
        seed, ordinal, prob = drop
        from oracle import rng
        m = torch.from_numpy(rng.keep_mask(seed, ordinal, tuple(y.shape), drop_prob=prob))
        y = y / torch.tensor(1.0 - np.float32(prob)) * m
    if flags & BN:
        g, b = (torch.from_numpy(p[scope + "/batch_normalization/" + n]) for n in ("gamma", "beta"))
        mu, var = (torch.from_numpy(p[scope + "/batch_normalization/" + n]) for n in ("moving_mean", "moving_variance"))
        y = (y - mu) * (g * torch.rsqrt(var + 1e-5)) + b
        y = torch.maximum(y, 0.1 * y)
    return y


def _random_params(eng, seed):
    g = np.random.default_rng(seed)
    p = {}
    for name, shape in eng.param_shapes().items():
        if name.endswith("kernel"):
            fan_in = int(np.prod(shape[:3]))
            p[name] = (g.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif name.endswith("moving_variance"):
            p[name] = (g.random(shape) + 0.5).astype(np.float32)
        elif name.endswith("gamma"):
            p[name] = (g.random(shape) + 0.5).astype(np.float32)
        else:
            p[name] = (g.standard_normal(shape) * 0.1).astype(np.float32)
    return p


@pytest.mark.parametrize("wino", ["0", "2", "2f"])       # direct / Winograd F(2x2,3x3) on every eligible 3x3 (layer d) / fused kernel
@pytest.mark.parametrize("ksplit", ["-1", "3"])          # planner's choice / K slices forced on every launch
@pytest.mark.parametrize("H,W,B", [(64, 64, 1), (32, 96, 3)])
def test_custom_graph_layer_by_layer(H, W, B, ksplit, wino, monkeypatch, precision):
    import torch
    from byolo import Engine
    if precision != "f32" and wino != "0":
        pytest.skip("Winograd exists in the fp32 mode only")
    monkeypatch.setenv("BYOLO_KSPLIT", ksplit)
    monkeypatch.setenv("BYOLO_WINOGRAD", wino[0])
    monkeypatch.setenv("BYOLO_WINO_FUSED", "2" if wino.endswith("f") else "0")
    eng = Engine((H, W, 3), 2, drop_prob=0.25, keep_all_outputs=True)
    L = {}
    L["a"] = eng.add_conv("a", 32, 3, 1, BN)               # 3 -> 32: stem kernel
    L["b"] = eng.add_conv("b", 64, 3, 2, BN)               # Darknet stride-2 pad
    L["c"] = eng.add_conv("c", 32, 1, 1, BN | DROP)        # 1x1 + dropout (ordinal 0)
    L["d"] = eng.add_conv("d", 64, 3, 1, BN)
    L["res"] = eng.add_residual(L["b"])                    # fused into d's epilogue
    L["e"] = eng.add_conv("e", 96, 3, 2, BN | DROP)        # 96 of a 128-wide column tile, dropout ordinal 1
    L["up"] = eng.add_upsample()
    L["cat"] = eng.add_route([L["up"], L["res"]])          # 96 (x2 upsampled) + 64 channels
    L["f"] = eng.add_conv("f", 40, 1, 1, BN)               # two-source loader; 40 output channels
    L["g"] = eng.add_conv("g", 64, 3, 1, BN)               # Cin = 40: generic direct kernel
    L["det"] = eng.add_detection("h/detection", 0, [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])
    p = _random_params(eng, 3)
    eng.set_params(p)
    eng.finalize()
    g = np.random.default_rng(8)
    img = g.random((B, H, W, 3)).astype(np.float32)
    seed = 77
    eng.forward(torch.from_numpy(img).cuda(), T=1, seed=seed, want_boxes=True, want_nms=False)
    torch.cuda.synchronize()

    x = torch.from_numpy(img)
    a = _ref_conv(x, p, "a", 3, 1, BN)
    b = _ref_conv(a, p, "b", 3, 2, BN)
    c = _ref_conv(b, p, "c", 1, 1, BN, drop=(seed, 0, 0.25))
    d = _ref_conv(c, p, "d", 3, 1, BN) + b
    e = _ref_conv(d, p, "e", 3, 2, BN, drop=(seed, 1, 0.25))
    up = e.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    cat = torch.cat([up, d], dim=3)
    f = _ref_conv(cat, p, "f", 1, 1, BN)
    gg = _ref_conv(f, p, "g", 3, 1, BN)
    wdet = torch.from_numpy(p["h/detection/conv2d/kernel"]).permute(3, 2, 0, 1)
    det = torch.nn.functional.conv2d(gg.permute(0, 3, 1, 2), wdet).permute(0, 2, 3, 1) + torch.from_numpy(p["h/detection/conv2d/bias"])

    for name, ref in (("a", a), ("b", b), ("c", c), ("e", e), ("f", f), ("g", gg), ("det", det)):
        got = eng.layer_output(L[name]).cpu().numpy()
        assert_close(got, ref.numpy(), "layer %s (%dx%d, B=%d)" % (name, H, W, B))
    # the residual add is fused into d's epilogue: one of the two layers owns the (summed) tensor, the other none
    from byolo import ByoloError
    try:
        got = eng.layer_output(L["res"])
    except ByoloError:
        got = eng.layer_output(L["d"])
    assert_close(got.cpu().numpy(), d.numpy(), "fused residual")


def test_general_direct_convolution_and_view_shortcuts():
    """Found by tools/fuzz_graph.py.  (1) Convolutions the implicit-GEMM loader does not take -- input channels not a
    multiple of 32 -- run on the general direct kernel with everything the loader folds in: a two-source concat, a x2
    upsampled source, an output-channel count that is not a multiple of 8, weights too large for LDS, a detection head.
    (2) A residual whose shortcut is seen through an identity route is resolved to the tensor behind it."""
    import torch
    from byolo import Engine
    H, W, B = 64, 96, 2
    eng = Engine((H, W, 3), 2, drop_prob=0.25, keep_all_outputs=True)
    L = {}
    L["a"] = eng.add_conv("a", 24, 3, 1, BN)               # 3 -> 24 (not the 32-channel stem kernel)
    L["b"] = eng.add_conv("b", 40, 3, 2, BN | DROP)        # Cin = 24, stride 2
    L["id"] = eng.add_route([L["b"]])                      # identity view of b
    L["c"] = eng.add_conv("c", 40, 3, 1, BN)               # Cin = 40
    L["res"] = eng.add_residual(L["id"])                   # shortcut through the view
    L["d"] = eng.add_conv("d", 256, 3, 2, BN)              # Cin = 40, 9 * 40 * 256 weights = 368 KB: not in LDS
    L["up"] = eng.add_upsample()
    L["cat"] = eng.add_route([L["up"], L["res"]])          # 256 (x2 upsampled) + 40 channels: 296, not a multiple of 32
    L["e"] = eng.add_conv("e", 20, 1, 1, BN)               # two-source direct convolution, 20 output channels
    L["det"] = eng.add_detection("h/detection", 0, [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])    # Cin = 20, 21 channels, bias
    p = _random_params(eng, 5)
    eng.set_params(p)
    eng.finalize()
    img = np.random.default_rng(9).random((B, H, W, 3)).astype(np.float32)
    seed = 5
    eng.forward(torch.from_numpy(img).cuda(), T=1, seed=seed, want_boxes=True, want_nms=False)
    torch.cuda.synchronize()
    x = torch.from_numpy(img)
    a = _ref_conv(x, p, "a", 3, 1, BN)
    b = _ref_conv(a, p, "b", 3, 2, BN, drop=(seed, 0, 0.25))
    c = _ref_conv(b, p, "c", 3, 1, BN) + b
    d = _ref_conv(c, p, "d", 3, 2, BN)
    cat = torch.cat([d.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2), c], dim=3)
    e = _ref_conv(cat, p, "e", 1, 1, BN)
    wdet = torch.from_numpy(p["h/detection/conv2d/kernel"]).permute(3, 2, 0, 1)
    det = torch.nn.functional.conv2d(e.permute(0, 3, 1, 2), wdet).permute(0, 2, 3, 1) + torch.from_numpy(p["h/detection/conv2d/bias"])
    for name, ref in (("a", a), ("b", b), ("res", c), ("d", d), ("e", e), ("det", det)):
        assert_close(eng.layer_output(L[name]).cpu().numpy(), ref.numpy(), "layer %s" % name)



def test_deduplicated_concat_convolution_with_residual():
    """Found by tools/fuzz_graph.py: a convolution over [stacked tensor, T-fold tile of an unstacked one] is split into a
    once-per-image partial sum and a main launch that picks it up as an addend; when a residual add is fused into the
    same launch the epilogue has BOTH an addend and a residual (it used to add the addend twice)."""
    import torch
    from byolo import Engine
    H, W, B, T = 64, 96, 2, 3
    eng = Engine((H, W, 3), 2, keep_all_outputs=True)
    L = {}
    L["a"] = eng.add_conv("a", 32, 3, 1, BN)
    L["s"] = eng.add_stack(L["a"])
    L["b"] = eng.add_conv("b", 64, 1, 1, BN)
    L["cat"] = eng.add_route([L["b"], L["s"]])               # 64 stacked + 32 tiled channels
    L["c"] = eng.add_conv("c", 64, 3, 1, BN)
    L["res"] = eng.add_residual(L["b"])
    L["det"] = eng.add_detection("h/detection", 2, [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])
    p = _random_params(eng, 13)
    eng.set_params(p)
    eng.finalize()
    img = np.random.default_rng(2).random((B, H, W, 3)).astype(np.float32)
    eng.forward(torch.from_numpy(img).cuda(), T=T, seed=1, want_boxes=True, want_nms=False)
    torch.cuda.synchronize()
    a = _ref_conv(torch.from_numpy(img), p, "a", 3, 1, BN)
    s = a.repeat_interleave(T, dim=0)
    b = _ref_conv(s, p, "b", 1, 1, BN)
    c = _ref_conv(torch.cat([b, s], dim=3), p, "c", 3, 1, BN) + b
    assert_close(eng.layer_output(L["res"]).cpu().numpy(), c.numpy(), "concat convolution + residual")


def test_slab_handoff_stress_on_three_streams():
    """tools/stress_handoff.py: split-K slices, stream-K segments and the planner's own plan on three concurrent HIP
    streams, 150 iterations here (1000 in the tool's default run, logged under profiles/): every result bit-identical to
    the handle's first run -- the ordered reduce of the last arriver never sees a stale slab or a left-over ticket."""
    import importlib.util
    import os
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("stress_handoff", os.path.join(REPO, "tools", "stress_handoff.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(150, verbose=True)


def test_launch_graphs_of_four_handles_survive_three_hundred_replays():
    """tools/graph_stress.py: four handles (K slices forced, stream-K forced, two planner plans) replay their captured forwards 300
    times.  With the ticket words zeroed by a hipMemsetAsync NODE the split-K handles' rows turned to inf at replay 206 (~8 192 graph
    operations of a process with >= 3 executable graphs; 274 with three) -- the forward zeroes them with a kernel of its own now."""
    import importlib.util
    import os
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("graph_stress", os.path.join(REPO, "tools", "graph_stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run([0, 1, 2, 2], 300)
    assert mod.LAST_BAD == 0
