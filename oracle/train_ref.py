"""Oracle (TEST INFRASTRUCTURE, not product): CPU restatement of the reference's ground-truth encoding and training loss
(SURVEY.md section 8 row f4).

    encode_boxes   lib_yolo/tfdata.py:77-171 (+ create_prior_data :16-75, calc_iou :174-189, lib_yolo/data.py:125-166)
    loss           lib_yolo/layers.py:126-188 (`loss_tf`) on the dict of `split_detection` / `split_detection_aleatoric`
                   (lib_yolo/layers.py:11-84)
    loss_grad      d(loc + obj + cls) / d(raw detection output): what `optimizer.minimize` (lib_yolo/train.py:88) would
                   back-propagate into the network -- analytic, checked against finite differences in tests/test_loss.py
    l2_regularization   `tf.contrib.layers.l2_regularizer(l2_scale)` on every kernel and the detection biases
                   (lib_yolo/model.py:27, lib_yolo/layers.py:553-554, :604, :612): scale * sum(w ** 2) / 2 per tensor

numpy, op for op in the reference's order and in the working precision asked for (float32 = what the TF graph computes in;
float64 for the parity bounds).  Pinned by tests/golden/loss_gt.npz, which oracle/make_golden_loss.py generates by running the
reference's OWN `tfdata.encode_boxes` / `layers.loss_tf` under oracle/tf1_shim.py, and live by tests/test_oracle_vs_reference.py.

PARITY STATUS: "parity unpinned" at the TensorFlow-primitive boundary (tf.where, tf.nn.sigmoid_cross_entropy_with_logits,
tf.nn.sparse_softmax_cross_entropy_with_logits, tf.clip_by_value are restated from their documented formulas), see
oracle/__init__.py.
"""
import numpy as np


def prior_data(layers, dtype=np.float32):
    """`tfdata.create_prior_data` over `data.create_prior_data` (lib_yolo/data.py:125-166): per prior box, flattened
    [row, col, box] per layer and concatenated over the layers.  `layers`: list of (h, w, [(prior_h, prior_w), ...]).
    The reference fills float32 arrays from Python doubles and hands them to the graph as float32 constants."""
    out = {k: [] for k in ("bboxes", "bbox_areas", "cx", "cy", "pw", "ph", "lw", "lh")}
    for (h, w, priors) in layers:
        n = len(priors)
        bb = np.zeros((h, w, n, 4), np.float32)
        ar = np.zeros((h, w, n), np.float32)
        cx = np.zeros((h, w, n), np.float32)
        cy = np.zeros((h, w, n), np.float32)
        pw = np.zeros((h, w, n), np.float32)
        ph = np.zeros((h, w, n), np.float32)
        for row in range(h):
            for col in range(w):
                for b, (p_h, p_w) in enumerate(priors):
                    y_center = (row + 0.5) / h
                    x_center = (col + 0.5) / w
                    h2, w2 = p_h / 2., p_w / 2.
                    bb[row, col, b] = [y_center - h2, x_center - w2, y_center + h2, x_center + w2]
                    ar[row, col, b] = p_h * p_w
                    cx[row, col, b] = col / float(w)
                    cy[row, col, b] = row / float(h)
                    pw[row, col, b] = p_w
                    ph[row, col, b] = p_h
        out["bboxes"].append(bb.reshape(-1, 4)); out["bbox_areas"].append(ar.reshape(-1))
        out["cx"].append(cx.reshape(-1)); out["cy"].append(cy.reshape(-1))
        out["pw"].append(pw.reshape(-1)); out["ph"].append(ph.reshape(-1))
        out["lw"].append(np.full(h * w * n, w, np.float32)); out["lh"].append(np.full(h * w * n, h, np.float32))
    return {k: np.concatenate(v, axis=0).astype(dtype) for k, v in out.items()}


def calc_iou(ref_bbox, pd):
    """lib_yolo/tfdata.py:174-189."""
    bb = pd["bboxes"]
    int_ymin = np.maximum(bb[..., 0], ref_bbox[0])
    int_xmin = np.maximum(bb[..., 1], ref_bbox[1])
    int_ymax = np.minimum(bb[..., 2], ref_bbox[2])
    int_xmax = np.minimum(bb[..., 3], ref_bbox[3])
    zero = bb.dtype.type(0)
    h = np.maximum(int_ymax - int_ymin, zero)
    w = np.maximum(int_xmax - int_xmin, zero)
    inter = h * w
    union = pd["bbox_areas"] - inter + ((ref_bbox[2] - ref_bbox[0]) * (ref_bbox[3] - ref_bbox[1]))
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / union


def _logit(x):
    """lib_yolo/tfdata.py:7-11."""
    one = x.dtype.type(1)
    return -np.log((one / x) - one)


def encode_boxes(bboxes, labels, layers, ign_thresh, dtype=np.float32):
    """lib_yolo/tfdata.py:77-171 for ONE image.  bboxes [n, 4] (ymin, xmin, ymax, xmax; image fractions), labels [n] int.
    Returns one dict per detection layer: loc [h, w, 3, 4], cls [h, w, 3] int32, obj, ign [h, w, 3]."""
    pd = prior_data(layers, dtype)
    bboxes = np.asarray(bboxes, dtype).reshape(-1, 4)
    labels = np.asarray(labels, np.int32).reshape(-1)
    total = pd["cx"].shape[0]
    loc_x = np.zeros(total, dtype); loc_y = np.zeros(total, dtype)
    loc_w = np.zeros(total, dtype); loc_h = np.zeros(total, dtype)
    obj = np.zeros(total, dtype); cls = np.zeros(total, np.int32); ign = np.ones(total, dtype)
    two = dtype(2)
    w = bboxes[..., 3] - bboxes[..., 1]
    h = bboxes[..., 2] - bboxes[..., 0]
    x = (bboxes[..., 3] + bboxes[..., 1]) / two
    y = (bboxes[..., 2] + bboxes[..., 0]) / two
    eps = dtype(1e-7)
    hi = dtype(1 - 1e-7)
    for i in range(labels.shape[0]):
        dx = pd["lw"] * (x[i] - pd["cx"])
        dy = pd["lh"] * (y[i] - pd["cy"])
        obj_mask = (dx >= 0) & (dx <= 1) & (dy >= 0) & (dy <= 1)
        iou = calc_iou(bboxes[i], pd)
        best = iou >= np.max(iou)
        obj_mask = best & obj_mask
        ign_mask = iou >= dtype(ign_thresh)
        with np.errstate(divide="ignore", invalid="ignore"):
            loc_x = np.where(obj_mask, _logit(np.clip(dx, eps, hi)), loc_x)
            loc_y = np.where(obj_mask, _logit(np.clip(dy, eps, hi)), loc_y)
            loc_w = np.where(obj_mask, np.log(np.maximum(w[i] / pd["pw"], eps)), loc_w)
            loc_h = np.where(obj_mask, np.log(np.maximum(h[i] / pd["ph"], eps)), loc_h)
        cls = np.where(obj_mask, labels[i], cls).astype(np.int32)
        obj = np.where(obj_mask, dtype(1), obj)
        ign = np.where(ign_mask, dtype(0), ign)
    loc = np.stack([loc_x, loc_y, loc_w, loc_h], axis=1)
    ign = np.maximum(ign, obj)
    out, off = [], 0
    for (lh, lw, priors) in layers:
        n = lh * lw * len(priors)
        shape = (lh, lw, len(priors))
        out.append({"loc": loc[off:off + n].reshape(shape + (4,)), "cls": cls[off:off + n].reshape(shape),
                    "obj": obj[off:off + n].reshape(shape), "ign": ign[off:off + n].reshape(shape)})
        off += n
    return out


def split_detection(raw, cls_cnt, aleatoric):
    """lib_yolo/layers.py:11-38 / :41-84: raw [b, h, w, 3 * blk] -> dict of [b, h, w, 3, ...] arrays (prior-major blocks)."""
    b, h, w, F = raw.shape
    blk = (10 + 2 * cls_cnt) if aleatoric else (5 + cls_cnt)
    assert F == 3 * blk
    r = raw.reshape(b, h, w, 3, blk)
    if not aleatoric:
        return {"loc": r[..., 0:4], "obj": r[..., 4], "cls": r[..., 5:5 + cls_cnt]}
    return {"loc": r[..., 0:4], "log_loc_var": r[..., 4:8], "obj": r[..., 8], "log_obj_stddev": r[..., 9],
            "cls": r[..., 10:10 + cls_cnt], "log_cls_stddev": r[..., 10 + cls_cnt:10 + 2 * cls_cnt]}


def loss(raw, gt, cls_cnt, aleatoric, aleatoric_loss, dtype=np.float64, want_grad=False):
    """lib_yolo/layers.py:126-188 for one detection layer.  raw [b, h, w, F]; gt: loc [b, h, w, 3, 4], obj, ign [b, h, w, 3],
    cls [b, h, w, 3] int.  Returns {'loc', 'obj', 'cls'} (and 'grad' [b, h, w, F] = d(loc + obj + cls) / d raw)."""
    raw = np.asarray(raw, dtype)
    det = split_detection(raw, cls_cnt, aleatoric)
    g_loc = np.asarray(gt["loc"], dtype); g_obj = np.asarray(gt["obj"], dtype); g_ign = np.asarray(gt["ign"], dtype)
    g_cls = np.asarray(gt["cls"]).astype(np.int64)
    bs = dtype(raw.shape[0])
    diff = g_loc - det["loc"]
    loc_loss = diff ** 2
    if aleatoric_loss:
        assert aleatoric
        lv = np.clip(det["log_loc_var"], dtype(-40), dtype(40))
        loc_loss = loc_loss * np.exp(-lv)
        loc_loss = loc_loss + lv
    loc_loss = loc_loss * g_obj[..., None]
    loc_all = np.sum(loc_loss) / (dtype(2) * bs)
    # tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x * z + log(1 + exp(-|x|))
    xo = det["obj"]
    obj_loss = np.maximum(xo, 0) - xo * g_obj + np.log1p(np.exp(-np.abs(xo)))
    obj_all = np.sum(obj_loss * g_ign) / bs
    # tf.nn.sparse_softmax_cross_entropy_with_logits: -log_softmax(x)[label]
    xc = det["cls"]
    m = np.max(xc, axis=-1, keepdims=True)
    lse = np.log(np.sum(np.exp(xc - m), axis=-1, keepdims=True)) + m
    picked = np.take_along_axis(xc, g_cls[..., None], axis=-1)
    cls_loss = (lse - picked)[..., 0]
    cls_all = np.sum(cls_loss * g_obj) / bs
    out = {"loc": loc_all, "obj": obj_all, "cls": cls_all}
    if want_grad:
        b, h, w, F = raw.shape
        blk = F // 3
        g = np.zeros((b, h, w, 3, blk), dtype)
        if aleatoric_loss:
            raw_lv = det["log_loc_var"]
            inside = (raw_lv >= -40) & (raw_lv <= 40)         # tf.clip_by_value passes the gradient inside [min, max]
            e = np.exp(-lv)
            g[..., 0:4] = -diff * e * g_obj[..., None] / bs
            g[..., 4:8] = np.where(inside, (1 - diff ** 2 * e), 0) * g_obj[..., None] / (2 * bs)
        else:
            g[..., 0:4] = -diff * g_obj[..., None] / bs
        o = 8 if aleatoric else 4
        sig = 1 / (1 + np.exp(-xo))
        g[..., o] = (sig - g_obj) * g_ign / bs
        c = 10 if aleatoric else 5
        sm = np.exp(xc - lse)
        onehot = np.zeros_like(sm)
        np.put_along_axis(onehot, g_cls[..., None], 1, axis=-1)
        g[..., c:c + cls_cnt] = (sm - onehot) * g_obj[..., None] / bs
        out["grad"] = g.reshape(b, h, w, F)
    return out


def l2_regularization(params, scale=0.0005, dtype=np.float64):
    """`tf.contrib.layers.l2_regularizer(scale)` = scale * tf.nn.l2_loss(w) = scale * sum(w ** 2) / 2, on every conv kernel and
    on the detection layers' biases (lib_yolo/layers.py:553-554 `use_bias=False` elsewhere, :604, :612), summed
    (`tf.losses.get_regularization_loss`, lib_yolo/model.py:200)."""
    total = dtype(0)
    for name in sorted(params):
        if name.endswith("/kernel") or name.endswith("/bias"):
            w = np.asarray(params[name], dtype)
            total = total + dtype(scale) * (np.sum(w * w) / dtype(2))
    return total
