"""Oracle (test infrastructure): restatement of ``tf.image.non_max_suppression`` as the reference
calls it -- ``tf.image.non_max_suppression(boxes[:, :4], boxes[:, obj_idx], 1000)``
(`inference_epistemic.py:101`, `inference_aleatoric.py:107`, `inference_standard_yolov3.py:107`):
class-agnostic greedy NMS, IoU threshold 0.5 (suppress iff IoU > thr, strict), no score
threshold, at most ``max_output_size`` outputs.

TensorFlow (third-party, unpinned TF 1.x; algorithm published in
tensorflow/core/kernels/non_max_suppression_op.cc) is restated here:
  * candidates = boxes whose score > lowest-finite-float (so NaN and -inf scores never enter);
  * visited in descending score; ties: **lower index first** (TF1 left this unspecified
    -- std::priority_queue; TF2 made it lower-index-first; the build fixes that);
  * IoU from per-box min/max of the two y's and the two x's (flipped corners tolerated),
    ``area <= 0`` on either box => IoU 0, ``inter / (a_i + a_j - inter)`` in float32,
    std::min/std::max semantics ``(b < a) ? b : a`` / ``(a < b) ? b : a`` (matters for NaN).
All arithmetic is float32 with one rounding per operation (no FMA contraction), which the HIP
kernel reproduces bit-exactly ("kept indices bit-exact", BASELINE.json north_star).

Also: the 2-class variant the reference keeps as commented-out code
(`inference_epistemic.py:104-126`): ped iff cls0 > cls1, rider iff cls1 > cls0 (strict; ties
dropped), NMS(1000) on each subset, concat ped then rider.
"""
import ctypes
import os

import numpy as np

_F = np.float32


def _smin(a, b):      # std::min<float>(a, b)
    return b if b < a else a


def _smax(a, b):      # std::max<float>(a, b)
    return b if a < b else a


def iou_scalar(bi, bj):
    """float32 IoU of two [y0,x0,y1,x1] boxes, operation-for-operation as TF's IOU()."""
    ymin_i, xmin_i = _smin(bi[0], bi[2]), _smin(bi[1], bi[3])
    ymax_i, xmax_i = _smax(bi[0], bi[2]), _smax(bi[1], bi[3])
    ymin_j, xmin_j = _smin(bj[0], bj[2]), _smin(bj[1], bj[3])
    ymax_j, xmax_j = _smax(bj[0], bj[2]), _smax(bj[1], bj[3])
    with np.errstate(all="ignore"):
        area_i = _F(_F(ymax_i - ymin_i) * _F(xmax_i - xmin_i))
        area_j = _F(_F(ymax_j - ymin_j) * _F(xmax_j - xmin_j))
        if area_i <= 0 or area_j <= 0:
            return _F(0.0)
        iy0, ix0 = _smax(ymin_i, ymin_j), _smax(xmin_i, xmin_j)
        iy1, ix1 = _smin(ymax_i, ymax_j), _smin(xmax_i, xmax_j)
        inter = _F(_smax(_F(iy1 - iy0), _F(0.0)) * _smax(_F(ix1 - ix0), _F(0.0)))
        return _F(inter / _F(_F(area_i + area_j) - inter))


def sort_order(scores):
    """Candidate visiting order: descending score, lower index first among equals;
    scores that are NaN or <= -FLT_MAX are excluded (TF: ``score > lowest()``)."""
    s = np.asarray(scores, dtype=np.float32)
    idx = np.nonzero(s > np.finfo(np.float32).min)[0]
    order = idx[np.argsort(-s[idx].astype(np.float64), kind="stable")]
    return order.astype(np.int64)


def nms_tf_py(boxes, scores, max_out, iou_thr=0.5, candidates=None):
    """Pure-python/numpy-scalar reference (slow; small cases)."""
    boxes = np.asarray(boxes, dtype=np.float32)
    order = sort_order(scores)
    if candidates is not None:
        cand = np.asarray(candidates, dtype=bool)
        order = order[cand[order]]
    thr = _F(iou_thr)
    sel = []
    for c in order:
        if len(sel) >= max_out:
            break
        ok = True
        for j in reversed(sel):
            if iou_scalar(boxes[c], boxes[j]) > thr:
                ok = False
                break
        if ok:
            sel.append(int(c))
    return np.asarray(sel, dtype=np.int32)


# ------------------------------------------------------------------------------------------
# C restatement (oracle/nms_ref.c) -- same algorithm, used for full-size cases / the CPU baseline
# ------------------------------------------------------------------------------------------
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "liboracle_nms.so")
        if not os.path.exists(path):
            from . import build as _b
            _b.build_c()
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def nms_tf(boxes, scores, max_out, iou_thr=0.5, candidates=None):
    """boxes [N,>=4] float32 (row stride = boxes.shape[1]), scores [N] -> kept indices int32."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    n = boxes.shape[0]
    out = np.empty(max(max_out, 1), dtype=np.int32)
    cand = None
    if candidates is not None:
        cand = np.ascontiguousarray(candidates, dtype=np.uint8)
    cnt = _lib().oracle_nms(boxes.ctypes.data, boxes.shape[1], scores.ctypes.data,
                            cand.ctypes.data if cand is not None else None,
                            n, int(max_out), float(iou_thr), out.ctypes.data)
    return out[:cnt].copy()


def nms_agnostic(rows, obj_idx, max_out=1000, iou_thr=0.5):
    """`inference_epistemic.py:99-102`: returns (kept_rows, kept_idx)."""
    rows = np.asarray(rows, dtype=np.float32)
    keep = nms_tf(rows[:, :4], rows[:, obj_idx], max_out, iou_thr)
    return rows[keep], keep


def nms_two_class(rows, obj_idx, cls_start_idx, max_out=1000, iou_thr=0.5):
    """`inference_epistemic.py:104-126` (commented-out paper variant; `b` read as `boxes`).
    Returns (rows_ped ++ rows_rider, global kept indices, count_ped)."""
    rows = np.asarray(rows, dtype=np.float32)
    c0, c1 = rows[:, cls_start_idx], rows[:, cls_start_idx + 1]
    keep_p = nms_tf(rows[:, :4], rows[:, obj_idx], max_out, iou_thr, candidates=(c0 > c1))
    keep_r = nms_tf(rows[:, :4], rows[:, obj_idx], max_out, iou_thr, candidates=(c1 > c0))
    keep = np.concatenate([keep_p, keep_r])
    return rows[keep], keep, len(keep_p)
