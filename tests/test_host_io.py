"""Host side of the entry points (csrc/host_io.cpp through the C-ABI; no GPU involved): the native ECP-JSON text must be
byte-identical to what the reference's writer produces -- `json.dump({'children': [bbox_to_ecp_format(b) ...]}, f)`,
inference_epistemic.py:84-92 -- and the native PNG pool must decode what `tf.image.decode_png` / Pillow decode."""
import io
import json
import os

import numpy as np
import pytest

LAYOUT = {"yolov3": (7, 4, 5), "yolov3_aleatoric": (15, 9, 11), "bayesian_yolov3_aleatoric": (23, 14, 17)}     # D, obj, cls_start (C = 2)


class _M:
    pass


def _rows(variant, n, seed):
    D, obj, cs = LAYOUT[variant]
    rng = np.random.default_rng(seed)
    rows = rng.random((n, D), dtype=np.float32)
    h = n // 2                                              # every decade a float32 has, and raw bit patterns (subnormals, NaNs, infs)
    rows[:h] = (rng.standard_normal((h, D)) * np.exp(rng.uniform(-80, 80, (h, D)))).astype(np.float32)
    rows[h:h + n // 4] = rng.integers(0, 2 ** 32, (n // 4, D), dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, 1e16, 9999999827968.0, 1e-4, 9.999e-5, 123456789.0, 1e22, 1e-45, 3.4028235e38,
               0.1, 0.5, 100.0, 16777216.0, 1e15, 1.5e16, 1e-5, 0.001]
    for i, v in enumerate(special):
        rows[h + n // 4 + i] = v
    rows[-1, cs] = rows[-1, cs + 1] = 0.5                   # argmax tie -> first
    rows[-2, cs] = np.nan                                   # np.argmax: NaN is the maximum
    rows[-3, cs + 1] = np.nan
    return rows


@pytest.mark.parametrize("variant", sorted(LAYOUT))
@pytest.mark.parametrize("bg", [True, False])
def test_native_ecp_json_is_what_json_dump_writes(variant, bg):
    from byolo import hostio, inference as inf
    D, obj, cs = LAYOUT[variant]
    m = _M()
    m.cls_cnt, m.obj_idx, m.cls_start_idx = 2, obj, cs
    cfg = {"implicit_background_class": bg}
    rows = _rows(variant, 3000, seed=len(variant) + bg)
    fmt = hostio.EcpJsonFormatter(variant, [1024, 1920, 3], 2, obj, cs, bg, inf.LABEL_TO_CLS_NAME)
    got = fmt.format(rows)
    buf = io.StringIO()
    with np.errstate(all="ignore"):
        json.dump({"children": [inf.bbox_to_ecp_format(b, [1024, 1920, 3], m, cfg, variant) for b in rows]}, buf, default=lambda x: x.tolist())
    want = buf.getvalue().encode()
    if got != want:
        i = next(k for k in range(min(len(got), len(want))) if got[k] != want[k])
        raise AssertionError("first difference at byte %d: %r vs %r" % (i, got[max(0, i - 60):i + 30], want[max(0, i - 60):i + 30]))
    assert json.loads(got)["children"][0].keys() == json.loads(want)["children"][0].keys()
    assert fmt.format(rows[:0]) == b'{"children": []}' == json.dumps({"children": []}).encode()
    # an identity outside the label table is written as the integer (LABEL_TO_CLS_NAME.get(cls, cls))
    fmt1 = hostio.EcpJsonFormatter(variant, [64, 96, 3], 2, obj, cs, bg, {1: "pedestrian"})
    old = dict(inf.LABEL_TO_CLS_NAME)
    try:
        inf.LABEL_TO_CLS_NAME.clear(); inf.LABEL_TO_CLS_NAME[1] = "pedestrian"
        with np.errstate(all="ignore"):
            want1 = json.dumps({"children": [inf.bbox_to_ecp_format(b, [64, 96, 3], m, cfg, variant) for b in rows[:200]]}, default=lambda x: x.tolist())
    finally:
        inf.LABEL_TO_CLS_NAME.clear(); inf.LABEL_TO_CLS_NAME.update(old)
    assert fmt1.format(rows[:200]) == want1.encode()
    with pytest.raises(ValueError):
        hostio.EcpJsonFormatter(variant, [64, 96, 3], 2, obj, cs, bg, {1: 'ped"estrian'})
    with pytest.raises(ValueError):
        fmt.format(rows[:, :D - 2])


def _png(img, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img[:, :, 0] if img.shape[2] == 1 else img).save(b, format="PNG", **kw)
    return b.getvalue()


def test_native_png_decode_equals_pillow():
    from byolo import hostio, _lib
    rng = np.random.default_rng(3)
    for shape in ((64, 96, 3), (33, 17, 1), (50, 31, 4), (7, 5, 2), (300, 301, 3)):
        y, x = np.mgrid[0:shape[0], 0:shape[1]]
        smooth = np.clip((127 + 80 * np.sin(x / 7.) * np.cos(y / 5.))[:, :, None] + rng.normal(0, 6, shape), 0, 255).astype(np.uint8)
        imgs = [rng.integers(0, 256, shape, dtype=np.uint8), smooth, np.zeros(shape, np.uint8), smooth[::-1].copy()]
        enc = [_png(im, compress_level=lvl) for im, lvl in zip(imgs, (1, 6, 9, 0))]       # the encoder picks all five filter types on these
        out, st, found = hostio.decode_png_batch(enc, shape, threads=3)
        assert st.tolist() == [0, 0, 0, 0] and all(tuple(f) == shape for f in found)
        for o, i in zip(out, imgs):
            assert np.array_equal(o, i)
    # what it refuses, and how
    good = enc[1]
    shape = (300, 301, 3)
    cases = {"shape": good, "truncated": good[:len(good) // 2], "not a png": b"nope" * 20, "crc": good[:100] + bytes([good[100] ^ 1]) + good[101:]}
    out, st, found = hostio.decode_png_batch([cases["shape"]], (300, 300, 3))
    assert st[0] == _lib.PNG_SHAPE and tuple(found[0]) == shape
    out, st, _ = hostio.decode_png_batch([cases["truncated"], cases["not a png"], cases["crc"]], shape)
    assert st.tolist() == [_lib.PNG_CORRUPT] * 3
    from PIL import Image
    b = io.BytesIO(); Image.fromarray(imgs[1]).convert("P").save(b, format="PNG")
    b16 = io.BytesIO(); Image.fromarray((imgs[1][:, :, 0].astype(np.uint16) * 257)).save(b16, format="PNG")
    out, st, _ = hostio.decode_png_batch([b.getvalue(), b16.getvalue()], (300, 301, 1))
    assert st.tolist() == [_lib.PNG_UNSUPPORTED] * 2


def test_device_normalisation_constant_is_the_host_one():
    """byolo_normalize_u8 multiplies by the float32 constant 1.0f / 255.0f; decode_img by np.float32(1.0 / 255.0): the same number,
    so float(u8) * k is the same float32 for every byte (the kernel itself is compared on the device, tests/test_entry_points.py)."""
    assert np.float32(1.0) / np.float32(255.0) == np.float32(1.0 / 255.0)


def test_feed_falls_back_to_the_general_decoder(tmp_path):
    """A record whose PNG the native pool does not read (palette) goes through Pillow, like every record did before; a frame of the
    wrong size is the reference's set_shape failure."""
    from lib_yolo import dataset_utils as du
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    b = io.BytesIO(); Image.fromarray(grey).convert("P").save(b, format="PNG")
    ex = [du.make_example({"image/encoded": _png(img), "image/filename": "a.png"}),
          du.make_example({"image/encoded": b.getvalue(), "image/filename": "b.png"})]
    du.write_tfrecords(str(tmp_path / "x-00000"), ex)
    cfg = {"batch_size": 2, "full_img_size": [32, 32, 3], "cpu_thread_cnt": 2, "data": {"file_pattern": str(tmp_path / "x-*")}}
    with pytest.raises(ValueError, match="shape"):          # Pillow yields [32,32,1] palette indices for record b: not [32,32,3]
        list(du.TestingDataset(cfg))
    cfg1 = dict(cfg, full_img_size=[32, 32, 1])
    du.write_tfrecords(str(tmp_path / "y-00000"), ex[1:])
    cfg1["data"] = {"file_pattern": str(tmp_path / "y-*")}
    (x, names), = list(du.TestingDataset(cfg1))
    assert names == ["b.png"] and np.array_equal(x[0, :, :, 0], np.asarray(Image.open(io.BytesIO(b.getvalue()))).astype(np.float32) * np.float32(1 / 255.))
    cfg2 = dict(cfg, full_img_size=[32, 40, 3])
    du.write_tfrecords(str(tmp_path / "z-00000"), ex[:1])
    cfg2["data"] = {"file_pattern": str(tmp_path / "z-*")}
    with pytest.raises(ValueError, match=r"\(32, 32, 3\) != config full_img_size \(32, 40, 3\)"):
        list(du.TestingDataset(cfg2))
    # a record without an image, and a name longer than the native buffer: the general path reports / reads them
    du.write_tfrecords(str(tmp_path / "w-00000"), [du.make_example({"image/filename": "c.png"})])
    with pytest.raises(ValueError, match="without image/encoded"):
        list(du.TestingDataset(dict(cfg, data={"file_pattern": str(tmp_path / "w-*")})))
    long = "d" * 1500 + ".png"
    du.write_tfrecords(str(tmp_path / "v-00000"), [du.make_example({"image/encoded": _png(img), "image/filename": long})])
    (x, names), = list(du.TestingDataset(dict(cfg, data={"file_pattern": str(tmp_path / "v-*")})))
    assert names == [long] and np.array_equal(x[0], img.astype(np.float32) * np.float32(1 / 255.))
