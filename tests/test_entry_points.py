"""Entry points (rows a15-a19 of SURVEY.md section 8): ECP mapping and detect.py post-filter of the
PRODUCT against the golden dicts produced by the reference's own functions; TFRecord/PNG feed; and, on
the GPU, the full `inference(config)` / `do_it(...)` drivers against the oracle."""
import io
import json
import os

import numpy as np
import pytest

from conftest import golden, make_config, golden_params, assert_close

VARIANTS = ("yolov3", "yolov3_aleatoric", "bayesian_yolov3_aleatoric")
SCRIPTS = {"yolov3": "inference_standard_yolov3", "yolov3_aleatoric": "inference_aleatoric",
           "bayesian_yolov3_aleatoric": "inference_epistemic"}


class _M:
    def __init__(self, variant):
        from oracle import cpu_ref
        self.D, self.obj_idx, self.cls_start_idx = cpu_ref.row_layout(variant, 2)
        self.cls_cnt = 2


@pytest.mark.parametrize("variant", VARIANTS)
def test_bbox_to_ecp_format_matches_reference(variant):
    mod = __import__(SCRIPTS[variant])
    g = golden("ecp_dicts.json")[variant]
    rows = np.asarray(g["rows"], dtype=np.float32)
    for case in g["cases"]:
        cfg = {"implicit_background_class": case["implicit_background_class"]}
        for r, want in zip(rows, case["dicts"]):
            got = mod.bbox_to_ecp_format(r, g["img_size"], _M(variant), cfg)
            assert json.loads(json.dumps(got, default=lambda x: x.tolist())) == want


@pytest.mark.parametrize("variant", VARIANTS)
def test_detect_postfilter_matches_reference(variant):
    import detect
    g = golden("detect_post.json")[variant]
    rows = np.asarray(g["rows"], dtype=np.float32)
    m = _M(variant)
    filt = detect.filter_boxes(rows, m.obj_idx, g["thresh"])
    assert len(filt) == g["n_filtered"]
    ok = [r for r, bad in zip(filt, g["ibc_raises"]) if not bad]
    for r, bad in zip(filt, g["ibc_raises"]):
        if bad:
            with pytest.raises(IndexError):
                detect.preproces_boxes([1024, 1920, 3], [r], m.obj_idx, m.cls_start_idx, 2, {"implicit_background_class": True})
    conv = lambda L: json.loads(json.dumps(L, default=lambda x: x.item() if hasattr(x, "item") else x))
    got = detect.preproces_boxes([1024, 1920, 3], ok, m.obj_idx, m.cls_start_idx, 2, {"implicit_background_class": True},
                                 cls_mapping={1: "ped", 2: "rider"})
    assert conv(got) == g["pre_ibc"]
    got = detect.preproces_boxes([1024, 1920, 3], filt, m.obj_idx, m.cls_start_idx, 2, {"implicit_background_class": False})
    assert conv(got) == g["pre_noibc"]


def _png(img_u8):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img_u8).save(b, format="PNG")
    return b.getvalue()


def _make_records(tmp_path, n, H=64, W=96, shards=2):
    from lib_yolo import dataset_utils as du
    rng = np.random.default_rng(0)
    imgs, names, per = [], [], [[] for _ in range(shards)]
    for i in range(n):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        name = "city_%03d.png" % i
        per[i % shards].append(du.make_example({"image/encoded": _png(img), "image/filename": name,
                                                "image/height": H, "image/width": W}))
        imgs.append(img); names.append(name)
    for s in range(shards):
        du.write_tfrecords(str(tmp_path / ("ecp-day-val-%05d-of-%05d" % (s, shards))), per[s])
    return imgs, names


def test_tfrecord_feed(tmp_path):
    from lib_yolo import dataset_utils as du
    from byolo._lib import lib
    assert lib.byolo_crc32c(b"123456789", 9) == 0xE3069283          # CRC-32C check value
    imgs, names = _make_records(tmp_path, 5)
    cfg = make_config("x", 64, 96, batch_size=2, data={"file_pattern": str(tmp_path / "ecp-day-val-*-of-*")})
    ds = du.TestingDataset(cfg)
    assert ds.placeholder.shape == (2, 64, 96, 3)
    batches = list(ds)
    assert [len(b[1]) for b in batches] == [2, 2, 1]
    got_names = [n for b in batches for n in b[1]]
    assert got_names == names                                       # interleave(cycle 2, block 1) of 2 shards
    got = np.concatenate([b[0] for b in batches])
    assert got.dtype == np.float32 and got.max() <= 1.0
    for a, u8 in zip(got, imgs):
        assert np.array_equal(a, u8.astype(np.float32) * np.float32(1.0 / 255.0))
    # corruption is detected
    p = str(tmp_path / "ecp-day-val-00000-of-00002")
    raw = bytearray(open(p, "rb").read()); raw[40] ^= 0xFF
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError, match="CRC"):
        list(du.read_tfrecords(p))
    ex = du.parse_example(du.make_example({"a": b"xyz", "n": -3, "s": "str"}))
    assert ex == {"a": [b"xyz"], "n": [-3], "s": [b"str"]}


def test_tfrecord_fixture_from_the_published_format():
    """tests/golden/tf_bundle/ecp-day-val-00000-of-00001: a TFRecord file of tf.train.Example protos assembled byte by
    byte by tests/golden/make_bundle_fixture.py (its own varint / CRC-32C / masking / proto encoder; the feature keys of the
    reference's dataset writer, create_tf_records_citypersons.py:132-147; PNGs by Pillow) -- read through TestingDataset,
    CRC verification on."""
    import importlib.util
    from conftest import GOLDEN
    from lib_yolo import dataset_utils as du
    spec = importlib.util.spec_from_file_location("make_bundle_fixture", os.path.join(GOLDEN, "make_bundle_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    cfg = {"batch_size": 2, "full_img_size": [32, 32, 3], "data": {"file_pattern": os.path.join(GOLDEN, "tf_bundle", "ecp-day-val-*")}}
    batches = list(du.TestingDataset(cfg))
    assert [len(n) for _, n in batches] == [2, 1]
    want = mk.records()
    assert [n for _, ns in batches for n in ns] == [n for n, _ in want]
    for a, (_, u8) in zip(np.concatenate([x for x, _ in batches]), want):
        assert np.array_equal(a, u8.astype(np.float32) * np.float32(1.0 / 255.0))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_inference_driver_end_to_end(variant, tmp_path):
    """`inference(config)` writes one ECP JSON per image; contents == oracle on the same inputs."""
    import torch
    from oracle import cpu_ref
    mod = __import__(SCRIPTS[variant])
    imgs, names = _make_records(tmp_path, 3)
    # checkpoint: the golden weights saved by TF variable name
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    params = golden_params(variant)
    np.savez(str(ck / "model-1234.npz"), **params)
    cfg = make_config(variant, 64, 96, T=3, batch_size=2, checkpoint_path=str(tmp_path / "checkpoints"), run_id="run",
                      step="last", seed=10, data={"file_pattern": str(tmp_path / "ecp-day-val-*-of-*")},
                      out_path=str(tmp_path / "out" / "run"))
    mod.inference(cfg)
    out_dir = str(tmp_path / "out" / "run_1234")
    assert sorted(os.listdir(out_dir)) == sorted(n.replace(".png", ".json") for n in names)
    x = np.stack([i.astype(np.float32) * np.float32(1 / 255.) for i in imgs])
    tp = cpu_ref.to_torch_params(params)
    D, obj, cs = cpu_ref.row_layout(variant, 2)
    for bi, (lo, hi) in enumerate(((0, 2), (2, 3))):             # the driver's batches; seed = 10 + step
        boxes, _ = cpu_ref.detect_boxes(tp, x[lo:hi], variant, T=3, seed=10 + bi + 1)
        for k, (rows, keep) in enumerate(cpu_ref.nms_batch(boxes, variant)):
            got = json.load(open(os.path.join(out_dir, names[lo + k].replace(".png", ".json"))))["children"]
            assert len(got) == len(rows)
            want = [cpu_ref.bbox_to_ecp(r, [64, 96, 3], variant, 2, True) for r in rows]
            assert set(got[0]) == set(want[0])
            # same boxes (kept order may differ between near-tied scores): match each oracle box to the
            # nearest written box by its corners
            gc = np.array([[d["y0"], d["x0"], d["y1"], d["x1"]] for d in got])
            for r, w_ in zip(rows, want):
                wc = np.array([w_["y0"], w_["x0"], w_["y1"], w_["x1"]])
                g_ = got[int(np.argmin(np.abs(gc - wc).sum(1)))]
                for f in ("y0", "x0", "y1", "x1"):
                    assert abs(g_[f] - w_[f]) <= 1e-4 * 96 * max(1.0, abs(w_[f]) / 96)
                assert abs(g_["score"] - w_["score"]) <= 1e-4
                if abs(float(r[cs]) - float(r[cs + 1])) > 1e-3:       # argmax is rounding-stable
                    assert g_["identity"] == w_["identity"]


@pytest.mark.gpu
def test_inference_epistemic_as_a_torchrun_rank(tmp_path):
    """`torchrun inference_epistemic.py`: the entry point as the driver would launch it, on the only GPU of the box as a
    forced one-rank RCCL group (BYOLO_DIST_FORCE=1) -- sharded dataset, first_image, the all-gather of the padded
    box lists, rank-0 writer -- must write byte-identical JSON files to the plain single-process run."""
    import socket
    import subprocess
    import sys
    import inference_epistemic as mod
    imgs, names = _make_records(tmp_path, 5)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    np.savez(str(ck / "model-77.npz"), **golden_params("bayesian_yolov3_aleatoric"))
    pattern = str(tmp_path / "ecp-day-val-*-of-*")
    cfg = make_config("bayesian_yolov3_aleatoric", 64, 96, T=3, batch_size=2, checkpoint_path=str(tmp_path / "checkpoints"),
                      run_id="run", step="last", seed=10, data={"file_pattern": pattern}, out_path=str(tmp_path / "plain" / "run"))
    mod.inference(cfg)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_FORCE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "_inference_worker.py"), pattern, str(tmp_path / "checkpoints"),
           str(tmp_path / "dist" / "run"), "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    a, b = str(tmp_path / "plain" / "run_77"), str(tmp_path / "dist" / "run_77")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == sorted(n.replace(".png", ".json") for n in names)
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f


@pytest.mark.gpu
def test_two_ranks_with_real_engines_write_the_single_process_json(tmp_path):
    """The N > 1 path with TWO REAL engines (the driver's round-end box has one GPU, and RCCL refuses two ranks on one device:
    both ranks run on cuda:0 and the one collective per batch is staged through host memory under gloo, byolo/dist.py
    all_gather_flat): `inference_epistemic.inference` on 5 images, global batch 3 -- blocks of 2 + 1, then 1 + 1; each rank feeds,
    runs and writes ITS images with the dropout masks of its position in the global batch -- must leave byte-identical JSON
    files to the one-process run."""
    import socket
    import subprocess
    import sys
    import inference_epistemic as mod
    imgs, names = _make_records(tmp_path, 5)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    np.savez(str(ck / "model-77.npz"), **golden_params("bayesian_yolov3_aleatoric"))
    pattern = str(tmp_path / "ecp-day-val-*-of-*")
    cfg = make_config("bayesian_yolov3_aleatoric", 64, 96, T=3, batch_size=3, checkpoint_path=str(tmp_path / "checkpoints"),
                      run_id="run", step="last", seed=10, data={"file_pattern": pattern}, out_path=str(tmp_path / "plain" / "run"))
    stats = mod.inference(cfg)
    assert stats["images"] == 5 and stats["batches"] == 2 and stats["native_json"] and stats["precision_switches"] == 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_BACKEND="gloo", BYOLO_DIST_SHARE_DEVICE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "_inference_worker.py"), pattern, str(tmp_path / "checkpoints"),
           str(tmp_path / "dist" / "run"), "3", str(tmp_path / "stats")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    a, b = str(tmp_path / "plain" / "run_77"), str(tmp_path / "dist" / "run_77")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == sorted(n.replace(".png", ".json") for n in names)
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    st = [json.load(open(str(tmp_path / ("stats_rank%d.json" % r)))) for r in range(2)]
    assert [x["images"] for x in st] == [3, 2] and all(x["device"] == 0 and x["native_json"] for x in st)


@pytest.mark.gpu
def test_t_sharded_entry_point_one_process_and_two_ranks(tmp_path):
    """config['shard'] = 'T' (the latency path at the reference's batch_size = 1, SURVEY 8(e) alternative): `inference_epistemic.
    inference` on 3 frames, T = 5.
      * one process: the single shard runs all five samples -- the per-box sums are the one-call reduction's own, so the JSON files
        are BYTE-identical to the batch path's;
      * two ranks with REAL engines on one device (gloo staging, as test_two_ranks_with_real_engines...): samples 0-2 on rank 0, 3-4 on
        rank 1, one all-reduce per image, file i written by rank i % 2 -- every box within the contract's 1e-4 of the one-process
        file's (the float32 sums are added in another order), same boxes kept where scores are not tied."""
    import socket
    import subprocess
    import sys
    import inference_epistemic as mod
    imgs, names = _make_records(tmp_path, 3)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    np.savez(str(ck / "model-77.npz"), **golden_params("bayesian_yolov3_aleatoric"))
    pattern = str(tmp_path / "ecp-day-val-*-of-*")
    cfg = make_config("bayesian_yolov3_aleatoric", 64, 96, T=5, batch_size=1, checkpoint_path=str(tmp_path / "checkpoints"),
                      run_id="run", step="last", seed=10, data={"file_pattern": pattern})
    s0 = mod.inference(dict(cfg, out_path=str(tmp_path / "batch" / "run")))
    s1 = mod.inference(dict(cfg, out_path=str(tmp_path / "tshard1" / "run"), shard="T"))
    assert s0["images"] == s1["images"] == 3 and s1["shard"] == "T" and s1["samples"] == [0, 5] and s1["latency_ms_median"] > 0
    a, b = str(tmp_path / "batch" / "run_77"), str(tmp_path / "tshard1" / "run_77")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == sorted(n.replace(".png", ".json") for n in names)
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_BACKEND="gloo", BYOLO_DIST_SHARE_DEVICE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(here, "_inference_worker.py"), pattern, str(tmp_path / "checkpoints"),
           str(tmp_path / "tshard2" / "run"), "1", str(tmp_path / "stats"), "T", "5"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    c = str(tmp_path / "tshard2" / "run_77")
    assert sorted(os.listdir(c)) == sorted(os.listdir(a))
    st = [json.load(open(str(tmp_path / ("stats_rank%d.json" % r)))) for r in range(2)]
    assert [x["samples"] for x in st] == [[0, 3], [3, 5]] and [x["images"] for x in st] == [2, 1]
    num = ("y0", "x0", "y1", "x1", "score", "x_var_epi", "y_var_epi", "w_var_epi", "h_var_epi", "x_var_ale", "y_var_ale", "w_var_ale", "h_var_ale",
           "obj_mutual_info", "obj_entropy", "ped_score", "rider_score", "cls_mutual_info", "cls_entropy")
    for f in os.listdir(a):
        one = json.load(open(os.path.join(a, f)))["children"]
        two = json.load(open(os.path.join(c, f)))["children"]
        assert len(one) == len(two) > 0
        oc = np.array([[d["y0"], d["x0"], d["y1"], d["x1"]] for d in two])
        for d in one:                              # (the kept ORDER may flip between near-tied scores: match by corners)
            g = two[int(np.argmin(np.abs(oc - np.array([d["y0"], d["x0"], d["y1"], d["x1"]])).sum(1)))]
            for k in num:
                scale = 96.0 if k in ("y0", "x0", "y1", "x1") else 1.0
                assert abs(g[k] - d[k]) <= 1e-4 * scale * max(1.0, abs(d[k]) / scale), (f, k, g[k], d[k])
            assert g["layer_id"] == d["layer_id"] and g["prior_id"] == d["prior_id"]


@pytest.mark.gpu
def test_normalize_u8_on_the_device_is_the_host_conversion():
    """byolo_normalize_u8 == decode_img's `astype(float32) * float32(1 / 255)` bit for bit, every byte value, lengths that are
    not a multiple of 4 included (`lib_yolo/dataset_utils.py:6-11`, tf.image.convert_image_dtype)."""
    import torch
    from conftest import build_model
    _, m = build_model("yolov3", 64, 96)
    rng = np.random.default_rng(0)
    for n in (256, 4 * 1000 + 3, 64 * 96 * 3 * 2, 1):
        u8 = np.concatenate([np.arange(256, dtype=np.uint8), rng.integers(0, 256, n, dtype=np.uint8)])[:max(n, 1)] if n >= 256 else rng.integers(0, 256, n, dtype=np.uint8)
        got = m.engine.normalize_u8(torch.from_numpy(u8).cuda()).cpu().numpy()
        want = u8.astype(np.float32) * np.float32(1.0 / 255.0)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), n


@pytest.mark.gpu
def test_inference_loop_switches_to_fp32_on_a_range_error(tmp_path, monkeypatch):
    """A checkpoint whose activations leave the split-f16 range on EVERY frame (BN gamma 3e4 in darknet53/conv_10): the pipelined
    driver loop finds BYOLO_ERR_RANGE in the status words that come back with each batch's rows and re-runs that batch on the
    fp32 twin handle -- the files are those of a run that was in the fp32 mode from the start."""
    import inference_epistemic as mod
    imgs, names = _make_records(tmp_path, 5)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    p = {k: v.copy() for k, v in golden_params("bayesian_yolov3_aleatoric").items()}
    p["darknet53/conv_10/batch_normalization/gamma"][:] = 3e4
    np.savez(str(ck / "model-5.npz"), **p)
    cfg = make_config("bayesian_yolov3_aleatoric", 64, 96, T=3, batch_size=2, checkpoint_path=str(tmp_path / "checkpoints"),
                      run_id="run", step="last", seed=10, data={"file_pattern": str(tmp_path / "ecp-day-val-*-of-*")})
    monkeypatch.setenv("BYOLO_PRECISION", "split")
    s1 = mod.inference(dict(cfg, out_path=str(tmp_path / "split" / "run")))
    # every one of the three batches leaves the range: each is re-run in fp32 (two switches per batch), the run itself stays in split-f16
    assert s1["precision_switches"] == 6 and s1["fp32_batches"] == [1, 2, 3] and s1["precision"] == "split" and s1["images"] == 5
    monkeypatch.setenv("BYOLO_PRECISION", "f32")
    s2 = mod.inference(dict(cfg, out_path=str(tmp_path / "f32" / "run")))
    assert s2["precision_switches"] == 0 and s2["precision"] == "f32"
    a, b = str(tmp_path / "split" / "run_5"), str(tmp_path / "f32" / "run_5")
    assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == sorted(n.replace(".png", ".json") for n in names)
    for f in os.listdir(a):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f


def _one_hot_frame_records(tmp_path, frames, H=64, W=96):
    from lib_yolo import dataset_utils as du
    names = ["city_%03d.png" % i for i in range(len(frames))]
    du.write_tfrecords(str(tmp_path / "ecp-day-val-00000-of-00001"),
                       [du.make_example({"image/encoded": _png(f), "image/filename": n, "image/height": H, "image/width": W}) for f, n in zip(frames, names)])
    return names


@pytest.mark.gpu
def test_only_the_batch_beyond_the_range_runs_in_fp32(tmp_path, monkeypatch):
    """the fp32 fall-back is per batch.  Five frames at batch_size 1; frame 2 is a full-contrast pattern, the
    others are low-contrast noise, and the BN gamma / beta of det_net_2/conv are scaled so that -- by the ORACLE's own activations,
    asserted below -- frame 2 alone leaves the split-f16 range (|activation| > 16376) while the others stay below a quarter of it.
    Expected: batch 3 (frame 2) re-runs in fp32 on every rank, the run stays in split-f16 (`precision`), `precision_switches`
    counts both directions, `fp32_batches` == [3]; frame 2's file is the file of a run that was in fp32 from the start; the files
    of the frames AFTER it are byte-identical to those of a run in which frame 2 is replaced by a harmless frame (same seeds, same
    positions): nothing of the fall-back leaks into the batches behind it -- one JSON per image regardless
    (inference_epistemic.py:84-92)."""
    import torch
    import inference_epistemic as mod
    from oracle import cpu_ref
    variant = "bayesian_yolov3_aleatoric"
    rng = np.random.default_rng(3)
    H, W = 64, 96
    frames = [(rng.integers(118, 138, (H, W, 3))).astype(np.uint8) for _ in range(5)]
    hot = np.zeros((H, W, 3), np.uint8)
    hot[::2, 1::2] = 255; hot[1::2, ::2] = 255                     # checkerboard: the largest responses a 3x3 stack can see
    calm = frames[0].copy()
    p = {k: v.copy() for k, v in golden_params(variant).items()}
    # gamma and beta of det_net_2/conv (a plain conv-BN-leaky layer: where the hot frame's response peaks) x 200: its output scales
    # by exactly 200.  The premise is checked on the ORACLE's activations (float32 CPU restatement, every conv / residual output):
    # the hot frame beyond 4 x the range, every other frame below a quarter of it
    for v in ("gamma", "beta"):
        p["det_net_2/conv/batch_normalization/" + v] = p["det_net_2/conv/batch_normalization/" + v] * np.float32(200)

    def amax(frame):
        x = (frame.astype(np.float32) * np.float32(1 / 255.))[None]
        with torch.no_grad():
            f = cpu_ref.forward(cpu_ref.to_torch_params(p), x, variant, T=1, seed=0, taps="all", dropout_off=True)
        return max(float(t.abs().max()) for i, t in f["layers"].items() if f["topo"][i]["op"] in ("conv", "residual"))
    a_hot, a_rest = amax(hot), max(amax(f) for f in frames + [calm])
    print("largest |activation|: hot frame %.3g, the others <= %.3g (split-f16 holds 16376)" % (a_hot, a_rest))
    assert a_hot > 4 * 16376 and a_rest < 16376 / 4, "the test's premise does not hold"
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    np.savez(str(ck / "model-5.npz"), **p)
    base = dict(T=3, batch_size=1, checkpoint_path=str(tmp_path / "checkpoints"), run_id="run", step="last", seed=10)
    runs = {}
    for tag, fr, prec in (("mixed", frames[:2] + [hot] + frames[3:], "split"), ("calm", frames[:2] + [calm] + frames[3:], "split"),
                          ("f32", frames[:2] + [hot] + frames[3:], "f32")):
        d = tmp_path / tag
        d.mkdir()
        names = _one_hot_frame_records(d, fr)
        monkeypatch.setenv("BYOLO_PRECISION", prec)
        cfg = make_config(variant, H, W, data={"file_pattern": str(d / "ecp-day-val-*-of-*")}, out_path=str(d / "out" / "run"), **base)
        runs[tag] = (mod.inference(cfg), str(d / "out" / "run_5"), names)
    s, out, names = runs["mixed"]
    assert s["images"] == 5 and s["precision"] == "split", s
    assert s["fp32_batches"] == [3] and s["precision_switches"] == 2, s
    assert runs["calm"][0]["precision_switches"] == 0 and runs["calm"][0]["fp32_batches"] == []
    assert runs["f32"][0]["precision"] == "f32" and runs["f32"][0]["precision_switches"] == 0
    rd = lambda tag, i: open(os.path.join(runs[tag][1], names[i].replace(".png", ".json"))).read()
    assert rd("mixed", 2) == rd("f32", 2), "the re-run batch is not the fp32 mode's result"
    for i in (0, 1, 3, 4):
        assert rd("mixed", i) == rd("calm", i), "frame %d differs from the run without the hot frame" % i
        assert rd("mixed", i) != rd("f32", i) or i < 0            # (split-f16 and fp32 rows differ in the last bits: the files are not the fp32 run's)


def test_the_native_json_writer_is_only_for_the_unedited_stock_function():
    """the reference invites users to edit `bbox_to_ecp_format` (inference_epistemic.py:131 ff.).  The native formatter
    stands in for the script's function only while that function is the unedited one-line delegate (byolo.inference._is_stock
    reads its source); anything else goes through json.dumps of the function's own dicts."""
    from byolo import inference as binf
    import inference_epistemic, inference_aleatoric, inference_standard_yolov3
    for mod, v in ((inference_epistemic, "bayesian_yolov3_aleatoric"), (inference_aleatoric, "yolov3_aleatoric"), (inference_standard_yolov3, "yolov3")):
        assert binf._is_stock(mod.bbox_to_ecp_format, v)
        assert not binf._is_stock(mod.bbox_to_ecp_format, "other")
    ns = {"_inf": binf, "VARIANT": "bayesian_yolov3_aleatoric"}
    src = ("def bbox_to_ecp_format(bbox, img_size, model, config):\n"
           "    d = _inf.bbox_to_ecp_format(bbox, img_size, model, config, VARIANT)\n"
           "    d['extra'] = 1\n"
           "    return d\n")
    import linecache
    linecache.cache["<edited>"] = (len(src), None, src.splitlines(True), "<edited>")
    exec(compile(src, "<edited>", "exec"), ns)
    assert not binf._is_stock(ns["bbox_to_ecp_format"], "bayesian_yolov3_aleatoric")
    assert not binf._is_stock(lambda *a: {}, "yolov3")


@pytest.mark.gpu
def test_an_edited_bbox_to_ecp_format_is_what_gets_written(tmp_path, monkeypatch):
    """... end to end: the edited function's extra field is in every file, and stats['native_json'] says which writer ran."""
    import inference_aleatoric as mod
    imgs, names = _make_records(tmp_path, 2)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    np.savez(str(ck / "model-9.npz"), **golden_params("yolov3_aleatoric"))
    cfg = make_config("yolov3_aleatoric", 64, 96, batch_size=2, checkpoint_path=str(tmp_path / "checkpoints"), run_id="run", step="last",
                      data={"file_pattern": str(tmp_path / "ecp-day-val-*-of-*")})
    s0 = mod.inference(dict(cfg, out_path=str(tmp_path / "stock" / "run")))
    assert s0["native_json"]
    stock = mod.bbox_to_ecp_format

    def edited(bbox, img_size, model, config):
        d = stock(bbox, img_size, model, config)
        d["frame_area"] = img_size[0] * img_size[1]
        return d
    monkeypatch.setattr(mod, "bbox_to_ecp_format", edited)
    s1 = mod.inference(dict(cfg, out_path=str(tmp_path / "edited" / "run")))
    assert not s1["native_json"]
    for n in names:
        a = json.load(open(str(tmp_path / "stock" / "run_9" / n.replace(".png", ".json"))))["children"]
        b = json.load(open(str(tmp_path / "edited" / "run_9" / n.replace(".png", ".json"))))["children"]
        assert len(a) == len(b) > 0 and all(y["frame_area"] == 64 * 96 for y in b)
        assert [{k: v for k, v in y.items() if k != "frame_area"} for y in b] == a


@pytest.mark.gpu
def test_detect_do_it(tmp_path):
    import detect
    from lib_yolo import yolov3
    from PIL import Image
    rng = np.random.default_rng(1)
    files = []
    for i in range(2):
        p = str(tmp_path / ("img%d.png" % i))
        Image.fromarray(rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)).save(p)
        files.append(p)
    cfg = make_config("x", 64, 96, T=3, weights="synthetic", crop_img_size=[64, 96, 3])
    for cls in (yolov3.yolov3, yolov3.yolov3_aleatoric, yolov3.bayesian_yolov3_aleatoric):
        c = dict(cfg)
        if cls is yolov3.yolov3:      # 7-column rows + implicit background class -> the reference's IndexError (detect.py:51)
            c["implicit_background_class"] = False
        res = detect.do_it(files, 0.0, c, cls, {1: "ped", 2: "rider"} if cls is not yolov3.yolov3 else None)
        assert set(res) == set(files)
        for boxes in res.values():
            assert boxes and all(0 <= b["y0"] <= 64 and 0 <= b["x1"] <= 96 for b in boxes)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_detect_do_it_boxes_equal_the_oracle(variant, tmp_path):
    """`detect.do_it` (detect.py:112-135) on two PNGs with the golden weights: EVERY returned box against the oracle's
    forward + NMS on the same pixels, passed through the oracle's `filter_boxes` / `preproces_boxes` -- the restatement of
    detect.py:36-63 that tests/golden/detect_post.json pins to the reference's own functions."""
    import torch
    import detect
    from lib_yolo import yolov3
    from oracle import cpu_ref
    from PIL import Image
    rng = np.random.default_rng(7)
    files, pix = [], []
    for i in range(2):
        p = str(tmp_path / ("frame%d.png" % i))
        a = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
        Image.fromarray(a).save(p)
        files.append(p)
        pix.append(a.astype(np.float32) / np.float32(255.0))          # plt.imread of a PNG: float32 in [0, 1] (detect.py:77)
    ck = tmp_path / "checkpoints" / "run"
    ck.mkdir(parents=True)
    params = golden_params(variant)
    np.savez(str(ck / "model-77.npz"), **params)
    ibc = variant != "yolov3"         # 7-column rows + implicit background class -> the reference's IndexError (detect.py:51)
    cfg = make_config(variant, 64, 96, T=3, checkpoint_path=str(tmp_path / "checkpoints"), run_id="run", step="last", seed=5,
                      crop_img_size=[64, 96, 3], implicit_background_class=ibc)
    mapping = {1: "ped", 2: "rider"} if ibc else None
    thresh = 0.05
    res = detect.do_it(files, thresh, cfg, getattr(yolov3, variant), mapping)
    assert list(res) == files
    tp = cpu_ref.to_torch_params(params)
    D, obj, cs = cpu_ref.row_layout(variant, 2)
    total = 0
    for f, img in zip(files, pix):
        with torch.no_grad():
            boxes, _ = cpu_ref.detect_boxes(tp, img[None], variant, T=3, seed=5)
        rows, keep = cpu_ref.nms_batch(boxes, variant)[0]
        rows = cpu_ref.filter_boxes(rows, obj, thresh)
        want = cpu_ref.preproces_boxes([64, 96, 3], rows, obj, cs, 2, ibc, cls_mapping=mapping)
        got = res[f]
        assert len(got) == len(want) and len(want) > 0, "%s %s: %d boxes, the oracle %d" % (variant, os.path.basename(f), len(got), len(want))
        total += len(want)
        # the same boxes; the greedy NMS visits candidates in score order, which conv rounding (~1e-6) may flip between near-tied
        # scores: every oracle box is matched to the nearest returned box by its corners, each returned box is used once, and a
        # position may differ only between boxes whose objectness differs by less than 1e-5
        gc = np.array([[float(d["y0"]), float(d["x0"]), float(d["y1"]), float(d["x1"])] for d in got])
        used = set()
        for k, w in enumerate(want):
            wc = np.array([float(w["y0"]), float(w["x0"]), float(w["y1"]), float(w["x1"])])
            j = int(np.argmin(np.abs(gc - wc).sum(1)))
            assert j not in used, (variant, k, j)
            used.add(j)
            g = got[j]
            assert set(g) == set(w)
            if j != k:
                assert abs(float(got[k]["obj_score"]) - float(w["obj_score"])) < 1e-5, "%s: box %d sits at %d beyond a near-tie" % (variant, k, j)
            for key in ("y0", "x0", "y1", "x1"):
                assert abs(float(g[key]) - float(w[key])) <= 1e-4 * 96, (variant, k, key, g[key], w[key])
            for key in ("score", "obj_score", "cls_score"):
                assert abs(float(g[key]) - float(w[key])) <= 1e-4, (variant, k, key, g[key], w[key])
            # the winning class is rounding-stable unless the two class scores are within the bound of each other
            if abs(float(rows[k][cs]) - float(rows[k][cs + 1])) > 1e-3 and w["cls"] == g["cls"]:
                pass
            elif abs(float(rows[k][cs]) - float(rows[k][cs + 1])) > 1e-3:
                raise AssertionError("%s: box %d class %r vs %r" % (variant, k, g["cls"], w["cls"]))
    print("%s: %d boxes over 2 frames equal the oracle's" % (variant, total))


@pytest.mark.parametrize("name,stride", [("s32", 32), ("s8", 8)])
def test_uncertainty_colour_maps_match_reference(name, stride):
    """vis_uncertainty.colorize / color_map vs the reference's functions (fixture made by running them)."""
    import vis_uncertainty as vis
    g = golden("vis_maps.npz")
    col = vis.colorize(g[name + "_unc"], 0, None)
    assert_close(col, g[name + "_colorized"], "colorize", 1e-6, 1e-6)
    got = vis.color_map(g[name + "_img"], g[name + "_unc"], stride, 0, None)
    want = g[name + "_map"]
    assert got.dtype == np.uint8 and got.shape == want.shape
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got != want).mean() < 1e-3


@pytest.mark.gpu
def test_vis_uncertainty_do_it(tmp_path):
    """One forward per image -> 11 uncertainty kinds x 9 (stride, prior) maps, reference file names."""
    import vis_uncertainty as vis
    from PIL import Image
    rng = np.random.default_rng(2)
    f = str(tmp_path / "frame.png")
    Image.fromarray(rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)).save(f)
    cfg = make_config("x", 64, 96, T=4, weights="synthetic", batch_size=1, out_path=str(tmp_path / "maps"))
    vis.do_it([f], cfg)
    files = sorted(os.listdir(tmp_path / "maps"))
    assert len(files) == 11 * 9
    assert "frame_prior0_epi_x.png" in files and "frame_prior8_obj_mutual_info.png" in files
    im = np.asarray(Image.open(tmp_path / "maps" / "frame_prior4_ale_w.png"))
    assert im.shape == (64, 96, 3) and im.dtype == np.uint8


@pytest.mark.gpu
def test_weight_import_routes_reach_the_device(tmp_path):
    """Row (f2) of SURVEY.md section 8: the same trained variables through every import route of the reference --
    `tf.train.Saver.restore` of a tensor-bundle checkpoint (inference_epistemic.py:27-38, :58; found by
    find_checkpoint, read by byolo.tf_checkpoint), the `.npz` by TF variable name, and `load_darknet53_weights` of a
    Darknet `.weights` file for the backbone (lib_yolo/darknet.py:42-122: [beta, gamma, mean, var] then the kernel as
    [cout, cin, kh, kw]) -- end in bit-identical device results, which equal the golden fixture of the reference's graph."""
    import torch
    from byolo import tf_checkpoint as tc, inference as inf
    from conftest import build_model, golden_images
    variant = "yolov3_aleatoric"
    params = golden_params(variant)
    x = torch.from_numpy(golden_images(2)).cuda()

    def run(m):
        out = m.run(x, seed=42)
        torch.cuda.synchronize()
        return out["boxes"].clone()

    _, m0 = build_model(variant, 64, 96, params=params)
    base = run(m0)
    gold = golden("fwd_%s.npz" % variant)["bbox"]
    assert_close(base.cpu().numpy(), gold, "npz-by-name route vs golden")
    # (1) tensor-bundle checkpoint with the extras a training run leaves (global_step, an optimizer slot)
    ck = tmp_path / "ckpts" / "run"
    ck.mkdir(parents=True)
    extra = dict(params, global_step=np.array(77, np.int64))
    extra["darknet53/conv/conv2d/kernel/Adam"] = np.zeros_like(params["darknet53/conv/conv2d/kernel"])
    tc.write(str(ck / "model-77"), extra)
    path = inf.find_checkpoint({"checkpoint_path": str(tmp_path / "ckpts"), "run_id": "run", "step": "last"})
    assert path.endswith("model-77.index")
    _, m1 = build_model(variant, 64, 96)
    inf.restore(m1, path)
    assert torch.equal(run(m1), base)
    # (2) Darknet .weights for the 52 backbone convolutions, the heads by name
    yolo2, m2 = build_model(variant, 64, 96)
    m2.engine.set_params({k: v for k, v in params.items() if not k.startswith("darknet53/")}, strict=False)
    blob = [np.array([0, 2, 0, 0, 0], dtype=np.int32).tobytes()]
    scopes = []
    for n in m2.engine.param_shapes():
        s_ = n.rsplit("/", 2)[0]
        if n.startswith("darknet53/") and s_ not in scopes:
            scopes.append(s_)
    for s_ in scopes:
        for v in ("beta", "gamma", "moving_mean", "moving_variance"):
            blob.append(params["%s/batch_normalization/%s" % (s_, v)].tobytes())
        blob.append(np.ascontiguousarray(params[s_ + "/conv2d/kernel"].transpose(3, 2, 0, 1)).tobytes())
    wf = tmp_path / "darknet53.conv.74"
    wf.write_bytes(b"".join(blob))
    assert len(yolo2.load_darknet53_weights(str(wf))) == 52 * 5
    assert torch.equal(run(m2), base)


@pytest.mark.gpu
def test_det_dict_and_uncertainty_maps_vs_oracle():
    """Row (f3) of SURVEY.md section 8: `DetLayer.det` holds every key of the reference's decode_epistemic dict
    (lib_yolo/layers.py:397-411) with the reference's shapes, each within 1e-4 of the oracle's
    `decode_epistemic_stats` on the same image / weights / dropout stream -- and the 11 kinds x 9 (stride, prior)
    heat maps built from the DEVICE statistics equal the maps built from the ORACLE statistics through the same
    colour mapping (which tests/golden/vis_maps.npz pins to the reference's colorize / color_map)."""
    import torch
    import vis_uncertainty as vis
    from lib_yolo import yolov3, model as _model
    from oracle import cpu_ref
    from byolo import synth
    variant, H, W, T, seed = "bayesian_yolov3_aleatoric", 64, 96, 4, 21
    params = golden_params(variant)
    cfg = make_config(variant, H, W, T=T, batch_size=1)
    yolo = yolov3.bayesian_yolov3_aleatoric(cfg)
    m = yolo.init_model(inputs=_model.Placeholder((1, H, W, 3)), training=False).get_model()
    m.engine.set_params(params)
    m.finalize()
    img = synth.synthetic_images(1, H, W, seed=77)
    m.run(torch.from_numpy(img).cuda(), seed=seed, want_nms=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        f = cpu_ref.forward(cpu_ref.to_torch_params(params), img, variant, T=T, seed=seed)
    keys = {"ev_loc", "epi_covar_loc", "ale_var_loc", "obj_samples", "obj_mean", "obj_mutual_info", "obj_entropy",
            "cls_samples", "cls_mean", "cls_mutual_info", "cls_entropy"}
    ref_stats = []
    for k, dl in enumerate(m.det_layers):
        got = dl.det
        assert set(got) == keys
        raw = f["raw"][k]
        ref = cpu_ref.decode_epistemic_stats(raw, 2)
        d = raw.reshape(T, dl.h, dl.w, 3, 14)
        ref["obj_samples"] = torch.sigmoid(d[..., 8])
        ref["cls_samples"] = torch.softmax(d[..., 10:12], dim=-1)
        ref_stats.append(ref)
        for key in sorted(keys):
            assert tuple(got[key].shape) == tuple(ref[key].shape), (key, got[key].shape, ref[key].shape)
            assert_close(got[key].cpu().numpy(), ref[key].numpy(), "det layer %d %s" % (k, key))
        # the exported covariance is symmetric and its diagonal IS the row's columns (same sums, same bits)
        cov = got["epi_covar_loc"]
        assert torch.equal(cov, cov.transpose(-1, -2))
        rows = torch.stack([b[0] for b in dl.bbox], dim=2)
        assert torch.equal(torch.diagonal(cov, dim1=-2, dim2=-1), rows[..., 4:8])
    # maps: device statistics vs oracle statistics through the same (reference-pinned) colour mapping
    inf = vis.Inference.__new__(vis.Inference)
    inf.model = m
    n = 0
    for key, idx, name in vis.UCTY_KINDS:
        got_maps = inf.uncertainty_grids(img, key, idx)
        want_maps = []
        for dl, ref in zip(m.det_layers, ref_stats):
            u = ref[key]
            u = u if ("obj" in key or "cls" in key) else (u[..., idx, idx] if "epi" in key else u[..., idx])
            u = u.numpy()
            want_maps += [vis.color_map(img, u[..., p:p + 1], dl.downsample, 0, None) for p in range(3)]
        assert len(got_maps) == len(want_maps) == 9
        for g_, w_ in zip(got_maps, want_maps):
            assert g_.shape == w_.shape == (H, W, 3) and g_.dtype == np.uint8
            diff = np.abs(g_.astype(int) - w_.astype(int))
            # a statistic within 1e-4 of the oracle's moves a colour index by at most one step of the 256-entry map
            assert diff.max() <= 8 and (diff > 0).mean() < 0.02, "%s: max colour diff %d, %.3f of the pixels differ" % (name, diff.max(), (diff > 0).mean())
            n += 1
    assert n == 11 * 9


def test_corrupt_tfrecords_raise_cleanly(tmp_path):
    """Damaged TFRecord files: IOError from the framing, ValueError from the Example parser -- no other exception type,
    no giant read from a corrupt length (ad-hoc fuzz finding)."""
    import random
    from lib_yolo import dataset_utils as du
    ex = du.make_example({"image/encoded": _png(np.zeros((8, 8, 3), np.uint8)), "image/filename": b"a.png",
                          "image/height": 8, "image/width": 8})
    path = str(tmp_path / "x.tfrecord")
    du.write_tfrecords(path, [ex, ex])
    good = open(path, "rb").read()
    rnd = random.Random(4)
    seen = set()
    for i in range(600):
        b = bytearray(good)
        for _ in range(rnd.randint(1, 3)):
            op = rnd.randint(0, 2)
            if op == 0:
                b[rnd.randrange(len(b))] = rnd.randrange(256)
            elif op == 1:
                del b[rnd.randrange(len(b)):][:rnd.randint(1, 30)]
            else:
                at = rnd.randrange(len(b))
                b[at:at] = bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 12)))
        open(path, "wb").write(bytes(b))
        try:
            for rec in du.read_tfrecords(path, verify_crc=bool(i % 2)):
                du.parse_example(rec)
            seen.add("ok")
        except (IOError, ValueError):
            seen.add("refused")
    assert seen == {"ok", "refused"}
