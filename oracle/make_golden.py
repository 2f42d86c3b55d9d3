"""Oracle (test infrastructure): generate the golden fixtures under tests/golden/ by IMPORTING THE
REFERENCE (/root/reference) in this container.

    python -m oracle.make_golden            # from the repo root; needs /root/reference

Two kinds of fixtures:
  (A) produced by the reference's own pure-numpy/Python code running as written (only a bare stub
      `tensorflow` module is needed to import it): prior tables, `predictions_to_boxes_numpy_
      reference_implementation` (lib_yolo/utils.py:72-123), the three `bbox_to_ecp_format`s,
      `detect.filter_boxes` / `preproces_boxes`.  These are genuine reference outputs.
  (B) produced by the reference's graph-construction code (`lib_yolo/{yolov3,model,layers}.py`,
      `inference_*.concat_bbox` / `nms`) executed UNMODIFIED under `oracle/tf1_shim.py`, where the
      TensorFlow primitives are restated from documented TF semantics ("parity unpinned" at that
      boundary -- see oracle/__init__.py): full forwards at 64x96 for all three variants, the
      batch-1 loop that defines the batched-epistemic generalisation, NMS results.
Fixtures are data only (inputs, seeds, expected outputs).  Weights are regenerated from the seed
(byolo/synth.py); the calibrated BN statistics they were generated with are stored (bn_stats.npz).
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))

from oracle import tf1_shim as shim          # noqa: E402
from oracle import cpu_ref, nms_ref          # noqa: E402

H, W = 64, 96
SEED_W, SEED_DROP, T_EPI = 7, 42, 3


def _synth():
    import importlib.util
    spec = importlib.util.spec_from_file_location("byolo_synth", os.path.join(REPO, "bayesian-yolov3_amd", "byolo", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _synth()


def _purge_reference_modules():
    for k in list(sys.modules):
        if k == "lib_yolo" or k.startswith("lib_yolo.") or k in ("inference_epistemic", "inference_aleatoric",
                                                                    "inference_standard_yolov3", "detect",
                                                                    "vis_uncertainty"):
            del sys.modules[k]


_SAVED_PATH = None        # sys.path before the first import_reference()


def restore_environment():
    """Undo import_reference() / shim.install(): the caller's sys.path is back, the reference's `lib_yolo`,
    `inference_*`, `detect`, `vis_uncertainty` and the stand-in `tensorflow` / `cv2` leave sys.modules, so a later
    `import lib_yolo` finds the build's own package again (tests/test_oracle_vs_reference.py tears down with this)."""
    global _SAVED_PATH
    if _SAVED_PATH is not None:
        sys.path[:] = _SAVED_PATH
        _SAVED_PATH = None
    _purge_reference_modules()
    if getattr(sys.modules.get("tensorflow"), "__version__", "") == "1.12-shim":
        del sys.modules["tensorflow"]
    cv2 = sys.modules.get("cv2")
    if cv2 is not None and not hasattr(cv2, "__file__"):          # the bare stub of shim.install()
        del sys.modules["cv2"]


def import_reference():
    """Import the reference modules under the shim (module level only defines functions/tables).
    Changes sys.path / sys.modules for the process; restore_environment() undoes it."""
    global _SAVED_PATH
    os.environ.setdefault("MPLBACKEND", "Agg")
    if _SAVED_PATH is None:
        _SAVED_PATH = list(sys.path)
    _purge_reference_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # make sure `lib_yolo` resolves to the REFERENCE package here, not the build's mirror
    sys.path[:] = [p for p in sys.path if not p.rstrip("/").endswith("bayesian-yolov3_amd")]
    import lib_yolo.yolov3 as ryolo
    import lib_yolo.utils as rutils
    import inference_epistemic as repi
    import inference_aleatoric as rale
    import inference_standard_yolov3 as rstd
    import detect as rdetect
    assert ryolo.__file__.startswith(REF)
    return ryolo, rutils, repi, rale, rstd, rdetect


def ref_config(ryolo, variant, T=T_EPI, hw=None):
    return {"full_img_size": [hw[0] if hw else H, hw[1] if hw else W, 3], "crop": False, "cls_cnt": 2, "priors": ryolo.ECP_9_PRIORS,
            "aleatoric_loss": False, "inference_mode": True, "T": T, "implicit_background_class": True}


def calibrated_params():
    """Seeded base weights + BN statistics calibrated with the CPU restatement on one image."""
    shapes = cpu_ref.variable_shapes("yolov3_aleatoric", 2)
    base = synth.base_params(shapes, "yolov3_aleatoric", 2, seed=SEED_W)
    p = cpu_ref.to_torch_params(base)
    calib = synth.synthetic_images(16, H, W, seed=999)   # 16 x (2x3) = 96 samples/channel at stride 32
    cpu_ref.forward(p, calib, "yolov3_aleatoric", calibrate=True)
    stats = {k: v.numpy() for k, v in p.items() if k.endswith("moving_mean") or k.endswith("moving_variance")}
    return stats


def params_for(variant, stats):
    shapes = cpu_ref.variable_shapes(variant, 2)
    p = synth.base_params(shapes, variant, 2, seed=SEED_W)
    for k, v in stats.items():
        assert p[k].shape == v.shape
        p[k] = v.astype(np.float32)
    return p


def run_reference(variant, params, imgs, dtype, seed=SEED_DROP, sample_offset=0, T=T_EPI, masks=None, taps=(0, 1, 4, 36, 61, 74)):
    """Execute the reference's model class + concat_bbox + nms under the shim.  `masks`: keep-masks handed to the
    reference's tf.layers.dropout calls in call order (instead of the build-defined stream)."""
    shim.install(dtype=dtype, param_provider=lambda name, shape: params[name], seed=seed,
                 sample_offset=sample_offset, masks=masks)
    ryolo, rutils, repi, rale, rstd, rdetect = import_reference()
    cls = getattr(ryolo, variant)
    yolo = cls(ref_config(ryolo, variant, T, hw=imgs.shape[1:3]))
    x = shim.input_tensor(imgs)
    model = yolo.init_model(inputs=x, training=False).get_model()
    mod = {"yolov3": rstd, "yolov3_aleatoric": rale, "bayesian_yolov3_aleatoric": repi}[variant]
    bbox = mod.concat_bbox([dl.bbox for dl in model.det_layers])
    if variant == "bayesian_yolov3_aleatoric":
        kept = [mod.nms(bbox, model).numpy()]
    else:
        # the reference's batched nms concatenates per-image results on axis 0, which needs equal
        # kept counts (inference_aleatoric.py:137-143) -- false at this tiny
        # size, so call it once per image with a batch of one.
        kept = [mod.nms(bbox[i:i + 1], model).numpy()[0] for i in range(bbox.shape.as_list()[0])]
    out = {
        "n_layers": len(model.layers), "n_vars": len(shim.STATE.variables),
        "var_names": [v[:-2] for v in shim.STATE.variables],
        "layer_names": [l.name for l in model.layers],
        "obj_idx": model.obj_idx, "cls_start_idx": model.cls_start_idx,
        "layers": {i: model.layers[i].numpy() for i in taps},
        "raw": [dl.raw_output.numpy() for dl in model.det_layers],
        "bbox": bbox.numpy(), "nms_rows": kept,
        "dropout_calls": list(shim.STATE.dropout_calls),
    }
    return out


def tap_subsample(i, a):
    """Keep fixtures small: early (large) layers are stored on a pixel-strided grid."""
    if i == 0:
        return a[:, ::8, ::8, :]
    if i in (1, 4):
        return a[:, ::4, ::4, :]
    if i == 36:
        return a[:, ::2, ::2, :]
    return a


def gen_A(ryolo, rutils, repi, rale, rstd, rdetect):
    # 1. prior tables
    tables = {}
    for n in ("CITY_PERSONS_9_PRIORS", "ECP_9_PRIORS", "ECP_NIGHT_9_PRIORS", "ECP_DAY_NIGHT_9_PRIORS", "ECP_BIC_9_PRIORS"):
        t = getattr(ryolo, n)
        tables[n] = {str(s): [[p.h, p.w] for p in t[s]] for s in (32, 16, 8)}
    json.dump(tables, open(os.path.join(OUT, "priors.json"), "w"), indent=1)

    # 2. the reference's numpy decode
    g = np.random.default_rng(11)
    dec = {}
    for (lh, lw) in ((2, 3), (4, 6)):
        pred = (g.standard_normal((2, lh, lw, 42)) * 1.5).astype(np.float32)
        dec["pred_%dx%d" % (lh, lw)] = pred
        for fmt in ("xywh", "corners"):
            dec["out_%dx%d_%s" % (lh, lw, fmt)] = rutils.predictions_to_boxes_numpy_reference_implementation(
                pred, 2, ryolo.ECP_9_PRIORS[16], box_format=fmt)
    np.savez_compressed(os.path.join(OUT, "numpy_decode.npz"), **dec)

    # 3. ECP dicts, 4. detect.py post-filter
    class M:
        pass
    ecp = {}
    post = {}
    for variant, mod, D, obj, cs in (("yolov3", rstd, 7, 4, 5), ("yolov3_aleatoric", rale, 16, 9, 11),
                                     ("bayesian_yolov3_aleatoric", repi, 23, 14, 17)):
        m = M(); m.obj_idx, m.cls_start_idx, m.cls_cnt = obj, cs, 2
        rows = g.random((6, D)).astype(np.float32)
        cases = []
        for ibc in (True, False):
            cfgd = {"implicit_background_class": ibc}
            dicts = [mod.bbox_to_ecp_format(r, [1024, 1920, 3], m, cfgd) for r in rows]
            dicts = json.loads(json.dumps(dicts, default=lambda x: x.tolist()))
            cases.append({"implicit_background_class": ibc, "dicts": dicts})
        ecp[variant] = {"rows": rows.tolist(), "img_size": [1024, 1920, 3], "cases": cases}
        # cls scores are kept < obj column positions valid for `cls_idx + cls_start_idx` with +1
        filt = rdetect.filter_boxes(rows, obj, 0.5)
        # detect.py:43-51 reads the class score at cls_idx + cls_start_idx AFTER the +1 of the implicit
        # background class: for the 7-column standard rows that is out of bounds when class 1 wins
        # (IndexError in the reference) -- record which rows raise and keep the others.
        raises = []
        ok_rows = []
        for r in filt:
            try:
                rdetect.preproces_boxes([1024, 1920, 3], [r], obj, cs, 2, {"implicit_background_class": True})
                ok_rows.append(r); raises.append(False)
            except IndexError:
                raises.append(True)
        pp = rdetect.preproces_boxes([1024, 1920, 3], ok_rows, obj, cs, 2, {"implicit_background_class": True},
                                     cls_mapping={1: "ped", 2: "rider"})
        pp2 = rdetect.preproces_boxes([1024, 1920, 3], filt, obj, cs, 2, {"implicit_background_class": False})
        conv = lambda L: json.loads(json.dumps(L, default=lambda x: x.item() if hasattr(x, "item") else x))
        post[variant] = {"rows": rows.tolist(), "thresh": 0.5, "n_filtered": len(filt), "ibc_raises": raises,
                         "pre_ibc": conv(pp),
                         "pre_noibc": conv(pp2)}
    json.dump(ecp, open(os.path.join(OUT, "ecp_dicts.json"), "w"), indent=0)
    json.dump(post, open(os.path.join(OUT, "detect_post.json"), "w"), indent=0)


def gen_B(stats):
    np.savez_compressed(os.path.join(OUT, "bn_stats.npz"), **stats)
    imgs2 = synth.synthetic_images(2, H, W, seed=1234)
    meta = {"H": H, "W": W, "seed_w": SEED_W, "seed_drop": SEED_DROP, "T": T_EPI, "img_seed": 1234}
    for variant in cpu_ref.VARIANTS:
        params = params_for(variant, stats)
        bayes = variant == "bayesian_yolov3_aleatoric"
        imgs = imgs2[:1] if bayes else imgs2
        r32 = run_reference(variant, params, imgs, torch.float32)
        r64 = run_reference(variant, params, imgs, torch.float64)
        D, obj_idx, cs = cpu_ref.row_layout(variant, 2)
        assert r32["obj_idx"] == obj_idx and r32["cls_start_idx"] == cs
        save = {"bbox": r32["bbox"].astype(np.float32), "bbox_f64": r64["bbox"].astype(np.float64)}
        for k, rows in enumerate(r32["nms_rows"]):
            save["nms_rows_%d" % k] = rows.astype(np.float32)
        for i, a in r32["layers"].items():
            save["layer_%d" % i] = tap_subsample(i, a).astype(np.float32)
        for k, a in enumerate(r32["raw"]):
            save["raw_%d" % k] = a.astype(np.float32)
            save["raw64_%d" % k] = r64["raw"][k].astype(np.float64)
        np.savez_compressed(os.path.join(OUT, "fwd_%s.npz" % variant), **save)
        meta[variant] = {"n_layers": r32["n_layers"], "n_vars": r32["n_vars"], "var_names": r32["var_names"],
                         "layer_names": r32["layer_names"], "dropout_calls": [[o, list(s)] for o, s in r32["dropout_calls"]],
                         "B": int(imgs.shape[0])}
        print(variant, "layers", r32["n_layers"], "vars", r32["n_vars"], "bbox", r32["bbox"].shape,
              "nms", [r.shape for r in r32["nms_rows"]], "max|f32-f64|", float(np.nanmax(np.abs(r32["bbox"] - r64["bbox"]))))
    # 7. batched-epistemic generalisation == loop of batch-1 reference runs (sample s = img*T + t)
    variant = "bayesian_yolov3_aleatoric"
    params = params_for(variant, stats)
    loop = [run_reference(variant, params, imgs2[i:i + 1], torch.float32, sample_offset=i * T_EPI) for i in range(2)]
    np.savez_compressed(os.path.join(OUT, "fwd_bayesian_b2_loop.npz"),
                        bbox=np.stack([r["bbox"] for r in loop]).astype(np.float32),
                        nms_rows_0=loop[0]["nms_rows"][0].astype(np.float32), nms_rows_1=loop[1]["nms_rows"][0].astype(np.float32))
    # standard_test_dropout quirk: dropout result discarded => deterministic
    json.dump(meta, open(os.path.join(OUT, "fwd_meta.json"), "w"), indent=0)


def gen_tail():
    """Hand-made NMS cases; expected kept indices from the pure-python scalar restatement."""
    g = np.random.default_rng(5)
    cases = {}

    def add(name, boxes, scores, max_out=1000, cand=None):
        boxes = np.asarray(boxes, np.float32); scores = np.asarray(scores, np.float32)
        keep = nms_ref.nms_tf_py(boxes, scores, max_out, 0.5, candidates=cand)
        cases[name + "_boxes"] = boxes; cases[name + "_scores"] = scores
        cases[name + "_keep"] = keep; cases[name + "_max_out"] = np.int32(max_out)

    # random overlapping boxes
    c = g.random((300, 2)).astype(np.float32); s = (g.random((300, 2)) * 0.2 + 0.02).astype(np.float32)
    add("random", np.concatenate([c - s, c + s], 1), g.random(300))
    # exact score ties -> lower index first
    b = np.concatenate([c[:64] - s[:64], c[:64] + s[:64]], 1)
    add("ties", b, np.repeat(np.float32([0.9, 0.5, 0.5, 0.1]), 16))
    # zero-area, flipped corners, IoU exactly 0.5 (not suppressed: strict >), NaN / -inf scores
    edge = np.float32([[0, 0, 1, 1], [0, 0, 1, 0.5], [0.2, 0.2, 0.2, 0.8], [1, 1, 0, 0], [0, 0, 0.5, 0.5],
                       [0.1, 0.1, 0.9, 0.9], [0.5, 0.5, 0.6, 0.6], [0.5, 0.5, 0.6, 0.6]])
    add("edge", edge, np.float32([0.9, 0.8, 0.7, 0.6, 0.5, np.nan, -np.inf, 0.3]))
    # N > max_out
    c = g.random((500, 2)).astype(np.float32); s = np.full((500, 2), 0.005, np.float32)
    add("cap", np.concatenate([c - s, c + s], 1), g.random(500), max_out=50)
    np.savez_compressed(os.path.join(OUT, "tail_cases.npz"), **cases)


def gen_vis():
    """vis_uncertainty.py colorize / color_map (reference :15-47) on seeded maps, executed under the shim."""
    import matplotlib
    import matplotlib.cm
    if not hasattr(matplotlib.cm, "get_cmap"):          # removed in matplotlib 3.9; the reference calls it (:27)
        matplotlib.cm.get_cmap = lambda name: matplotlib.colormaps[name]
    shim.install(dtype=torch.float32)
    import_reference()
    import vis_uncertainty as rvis
    g = np.random.default_rng(21)
    out = {}
    for name, (h, w, stride) in {"s32": (2, 3, 32), "s8": (8, 12, 8)}.items():
        img = g.random((1, h * stride, w * stride, 3)).astype(np.float32)
        unc = (g.random((h, w, 1)) ** 3).astype(np.float32)
        unc[0, 0, 0] = 50.0                                  # outlier above the 99th percentile -> clipped
        res = rvis.color_map(shim.input_tensor(img), shim.input_tensor(unc), stride, 0, None).numpy()
        out[name + "_img"], out[name + "_unc"], out[name + "_map"] = img, unc, res
        out[name + "_colorized"] = rvis.colorize(shim.input_tensor(unc), 0, None).numpy()
    np.savez_compressed(os.path.join(OUT, "vis_maps.npz"), **out)


def gen_default_frame():
    """The reference at ITS OWN default workload (inference_epistemic.py:212-240: the full 1024 x 1920 ECP frame, T = 50, one
    image, class-agnostic NMS :99-102): the reference's model class, concat_bbox and nms executed under the shim on one synthetic
    frame with the golden weights.  Stored: every 16th pre-NMS row (7560 x 23), the kept rows, and the kept rows' positions in
    the box list -- 0.8 MB instead of 11 MB.  Takes a few minutes and ~20 GB on the build container's 8 cores."""
    import time
    variant = "bayesian_yolov3_aleatoric"
    HD, WD, TD = 1024, 1920, 50
    t0 = time.time()
    # BN statistics calibrated AT THIS SIZE with the CPU restatement on one frame (as bn_stats.npz was at 64x96: with the 64x96
    # statistics the 128x240-cell heads put out logits of +-20, exp() of which no two float32 evaluations agree on to 1e-4) --
    # stored in the fixture, so that the test rebuilds exactly these weights
    shapes = cpu_ref.variable_shapes("yolov3_aleatoric", 2)
    p = cpu_ref.to_torch_params(synth.base_params(shapes, "yolov3_aleatoric", 2, seed=SEED_W))
    with torch.no_grad():
        cpu_ref.forward(p, synth.synthetic_images(1, HD, WD, seed=999), "yolov3_aleatoric", calibrate=True)
    stats = {k: v.numpy().astype(np.float32) for k, v in p.items() if k.endswith("moving_mean") or k.endswith("moving_variance")}
    del p
    print("calibrated at %dx%d in %.0f s" % (HD, WD, time.time() - t0), flush=True)
    params = params_for(variant, stats)
    img = synth.synthetic_images(1, HD, WD, seed=1234)
    with torch.no_grad():
        r = run_reference(variant, params, img, torch.float32, T=TD, taps=())
    bbox, kept_rows = r["bbox"].astype(np.float32), r["nms_rows"][0].astype(np.float32)
    assert bbox.shape == (3 * (32 * 60 + 64 * 120 + 128 * 240), 23), bbox.shape
    # positions of the kept rows in the box list (tf.gather of the indices: every kept row is a row of bbox, bit for bit)
    key = {row.tobytes(): i for i, row in enumerate(bbox)}
    kept_idx = np.array([key[row.tobytes()] for row in kept_rows], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "fwd_default_frame.npz"), rows_every_16th=bbox[::16], kept_rows=kept_rows, kept_idx=kept_idx,
                        meta=np.array([HD, WD, TD, SEED_W, SEED_DROP, 1234, 16], dtype=np.int32),
                        **{"bn/" + k: v for k, v in stats.items()})
    print("largest |row value| %.3g, largest sigma_ale %.3g" % (float(np.nanmax(np.abs(bbox[:, :4]))), float(np.nanmax(bbox[:, 8:12]))))
    print("default frame: bbox", bbox.shape, "kept", kept_rows.shape, "%.0f s" % (time.time() - t0),
          os.path.getsize(os.path.join(OUT, "fwd_default_frame.npz")), "bytes")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vis":      # regenerate only this fixture
        gen_vis()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "default_frame":
        gen_default_frame()
        return
    shim.install(dtype=torch.float32)
    mods = import_reference()
    gen_A(*mods)
    gen_vis()
    gen_tail()
    stats = calibrated_params()
    gen_B(stats)
    print("fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-34s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
