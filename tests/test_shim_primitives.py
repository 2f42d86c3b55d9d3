"""The oracle's TensorFlow stand-in (`oracle/tf1_shim.py`) against the PUBLISHED definitions of the TF 1.x
primitives the reference calls, written out literally here (nested loops, float64).

The reference's arithmetic lives in TensorFlow 1.x (not installable here), so parity is pinned at this boundary
by restating the documented algorithms twice, independently: once in the shim (vectorised torch) and once
below (index-by-index numpy).  A disagreement is a shim bug; an agreement does not prove TensorFlow behaves
so -- that is what "parity unpinned at the TF boundary" in DESIGN.md means.

  tf.nn.conv2d (r1.12 docs): output[b,i,j,k] = sum_{di,dj,q} input[b, s*i+di, s*j+dj, q] * filter[di,dj,q,k];
      padding 'SAME', stride 1, odd k: (k-1)/2 zeros on every side; 'VALID': none.
  tf.layers.batch_normalization (inference): (x - moving_mean) * gamma / sqrt(moving_variance + eps) + beta
  tf.nn.leaky_relu: max(x, alpha*x);   tf.image.resize_nearest_neighbor (align_corners=False): src = floor(dst*in/out)
  tf.layers.dropout (TF <= 1.12): x / keep_prob * mask;   tf.pad CONSTANT;   tf.nn.softmax;   tf.linalg.det
  tf.image.non_max_suppression (core/kernels/non_max_suppression_op.cc): candidates by descending score, a
      candidate is kept unless IoU with an already kept box > threshold, IoU = 0 for non-positive areas, corners
      may come in any order.
"""
import numpy as np
import pytest
import torch

from oracle import tf1_shim as shim


def _tf(params=None, dtype=torch.float64, **kw):
    return shim.install(dtype=dtype, param_provider=(lambda name, shape: params[name]) if params else None, **kw)


def _conv_literal(x, w, stride, pad):
    B, H, W, C = x.shape
    k, _, _, O = w.shape
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, C))
    xp[:, pad:pad + H, pad:pad + W] = x
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.zeros((B, oh, ow, O))
    for b in range(B):
        for i in range(oh):
            for j in range(ow):
                for di in range(k):
                    for dj in range(k):
                        y[b, i, j] += xp[b, stride * i + di, stride * j + dj] @ w[di, dj]
    return y


@pytest.mark.parametrize("k,stride,padding", [(3, 1, "SAME"), (1, 1, "SAME"), (3, 2, "VALID"), (1, 1, "VALID")])
def test_conv2d(k, stride, padding):
    g = np.random.default_rng(k * 10 + stride)
    x = g.standard_normal((2, 7, 9, 5))
    w = g.standard_normal((k, k, 5, 4))
    b = g.standard_normal(4)
    tf = _tf({"s/conv2d/kernel": w, "s/conv2d/bias": b})
    with tf.variable_scope("s"):
        y = tf.layers.conv2d(shim.input_tensor(x), 4, k, strides=stride, padding=padding, use_bias=True).numpy()
    ref = _conv_literal(x, w, stride, (k - 1) // 2 if padding == "SAME" else 0) + b
    assert y.shape == ref.shape and np.abs(y - ref).max() < 1e-12


def test_darknet_downsample_padding():
    """lib_yolo/layers.py:533-540 pads one row/column at the TOP/LEFT only, then convolves VALID with stride 2."""
    g = np.random.default_rng(3)
    x = g.standard_normal((1, 8, 6, 3))
    w = g.standard_normal((3, 3, 3, 2))
    tf = _tf({"d/conv2d/kernel": w})
    with tf.variable_scope("d"):
        xp = tf.pad(shim.input_tensor(x), [[0, 0], [1, 0], [1, 0], [0, 0]])
        y = tf.layers.conv2d(xp, 2, 3, strides=2, padding="VALID", use_bias=False).numpy()
    xp_ref = np.zeros((1, 9, 7, 3)); xp_ref[:, 1:, 1:] = x
    assert np.array_equal(xp.numpy(), xp_ref)
    ref = _conv_literal(xp_ref, w, 2, 0)
    assert y.shape == (1, 4, 3, 2) and np.abs(y - ref).max() < 1e-12


def test_batch_norm_leaky_softmax_resize():
    g = np.random.default_rng(4)
    x = g.standard_normal((2, 3, 4, 6))
    p = {"n/batch_normalization/gamma": g.standard_normal(6), "n/batch_normalization/beta": g.standard_normal(6),
         "n/batch_normalization/moving_mean": g.standard_normal(6),
         "n/batch_normalization/moving_variance": g.random(6) + 0.1}
    tf = _tf(p)
    with tf.variable_scope("n"):
        y = tf.layers.batch_normalization(shim.input_tensor(x), training=False, epsilon=1e-5).numpy()
    ref = np.empty_like(x)
    for c in range(6):
        ref[..., c] = ((x[..., c] - p["n/batch_normalization/moving_mean"][c]) * p["n/batch_normalization/gamma"][c]
                       / np.sqrt(p["n/batch_normalization/moving_variance"][c] + 1e-5) + p["n/batch_normalization/beta"][c])
    assert np.abs(y - ref).max() < 1e-12
    lr = tf.nn.leaky_relu(shim.input_tensor(x), alpha=0.1).numpy()
    assert np.array_equal(lr, np.where(x > 0, x, 0.1 * x))
    sm = tf.nn.softmax(shim.input_tensor(x)).numpy()
    e = np.exp(x - x.max(-1, keepdims=True))
    assert np.abs(sm - e / e.sum(-1, keepdims=True)).max() < 1e-14
    up = tf.image.resize_nearest_neighbor(shim.input_tensor(x), (6, 8)).numpy()
    for i in range(6):
        for j in range(8):
            assert np.array_equal(up[:, i, j], x[:, i // 2, j // 2])


def test_dropout_forms_and_det():
    g = np.random.default_rng(5)
    x = g.standard_normal((2, 3, 3, 8))
    from oracle import rng
    for form, f in (("div", lambda v, m: v / 0.75 * m), ("mul", lambda v, m: v * (1.0 / 0.75) * m)):
        tf = _tf(seed=9, drop_form=form)
        y = tf.layers.dropout(shim.input_tensor(x), rate=0.25, training=True).numpy()
        m = rng.keep_mask(9, 0, x.shape, drop_prob=0.25)
        assert np.array_equal(y, f(x, m.astype(np.float64)))
        assert np.array_equal(tf.layers.dropout(shim.input_tensor(x), rate=0.25, training=False).numpy(), x)
    a = g.standard_normal((5, 4, 4))
    tf = _tf()
    assert np.allclose(tf.linalg.det(shim.input_tensor(a)).numpy(), np.linalg.det(a), rtol=1e-12, atol=1e-12)


def _nms_literal(boxes, scores, max_out, thr):
    def area(b):
        return (max(b[0], b[2]) - min(b[0], b[2])) * (max(b[1], b[3]) - min(b[1], b[3]))

    def iou(a, b):
        aa, ab = np.float32(area(a)), np.float32(area(b))
        if aa <= 0 or ab <= 0:
            return np.float32(0)
        y0, x0 = max(min(a[0], a[2]), min(b[0], b[2])), max(min(a[1], a[3]), min(b[1], b[3]))
        y1, x1 = min(max(a[0], a[2]), max(b[0], b[2])), min(max(a[1], a[3]), max(b[1], b[3]))
        inter = np.float32(max(np.float32(y1 - y0), np.float32(0))) * np.float32(max(np.float32(x1 - x0), np.float32(0)))
        return inter / (aa + ab - inter)
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    keep = []
    for i in order:
        if len(keep) >= max_out:
            break
        if all(not (iou(boxes[i], boxes[j]) > np.float32(thr)) for j in keep):
            keep.append(i)
    return keep


def test_non_max_suppression():
    g = np.random.default_rng(6)
    c = g.random((300, 2)).astype(np.float32)
    s = (g.random((300, 2)) * 0.2 + 0.02).astype(np.float32)
    boxes = np.concatenate([c - s, c + s], 1).astype(np.float32)
    boxes[::7] = boxes[::7][:, [2, 3, 0, 1]]            # flipped corners
    boxes[5] = [0.3, 0.3, 0.3, 0.9]                      # zero area
    scores = g.random(300).astype(np.float32)
    scores[10:14] = scores[10]                           # ties: lower index first
    tf = _tf()
    for max_out, thr in ((1000, 0.5), (20, 0.3)):
        got = tf.image.non_max_suppression(shim.input_tensor(boxes), shim.input_tensor(scores), max_out, thr).numpy()
        assert got.tolist() == _nms_literal(boxes, scores, max_out, thr)
