"""Ground-truth encoding and training loss on the device (include/byolo.h: byolo_encode_gt, byolo_loss; SURVEY.md
section 8 row f4).  `lib_yolo/tfdata.py` and `lib_yolo/layers.py` of this package put the reference's names on top.

PyTorch is plumbing here as everywhere: device buffers and the current HIP stream; the arithmetic is in libbyolo.so."""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib, check


def _torch():
    import torch
    return torch


def _handle(engine):
    return engine._h if engine is not None else None


def _layers_arrays(det_layers):
    """det_layers: objects with .h, .w, .priors (lib_yolo.model.DetLayer / DetLayerBlueprint, lib_yolo.data.DetLayerInfo)."""
    n = len(det_layers)
    hw = (ctypes.c_int32 * (2 * n))()
    pr = (ctypes.c_double * (6 * n))()
    total = 0
    for l, dl in enumerate(det_layers):
        assert len(dl.priors) == 3, 'exactly 3 priors per detection layer'
        hw[2 * l], hw[2 * l + 1] = int(dl.h), int(dl.w)
        for k, p in enumerate(dl.priors):
            pr[(l * 3 + k) * 2], pr[(l * 3 + k) * 2 + 1] = float(p.h), float(p.w)
        total += int(dl.h) * int(dl.w) * 3
    return n, hw, pr, total


class GroundTruth:
    """Encoded ground truth of a batch: loc [B,N,4], obj [B,N], cls [B,N] (int32), ign [B,N] on the device, the prior boxes of
    all detection layers back to back; `layer(k)` = the reference's per-layer dict (views)."""

    def __init__(self, det_layers, loc, obj, cls, ign):
        self.shapes = [(int(dl.h), int(dl.w), 3) for dl in det_layers]
        self.loc, self.obj, self.cls, self.ign = loc, obj, cls, ign
        self.N = int(obj.shape[1])

    def offset(self, k):
        return sum(h * w * b for (h, w, b) in self.shapes[:k])

    def layer(self, k):
        h, w, b = self.shapes[k]
        o, n, B = self.offset(k), h * w * b, self.obj.shape[0]
        return {'loc': self.loc[:, o:o + n].reshape(B, h, w, b, 4), 'cls': self.cls[:, o:o + n].reshape(B, h, w, b),
                'obj': self.obj[:, o:o + n].reshape(B, h, w, b), 'ign': self.ign[:, o:o + n].reshape(B, h, w, b),
                '_gt': self, '_layer': k}

    def layers(self):
        return [self.layer(k) for k in range(len(self.shapes))]


def encode_gt(det_layers, boxes, labels, counts=None, ign_thresh=0.7, engine=None, device=None):
    """byolo_encode_gt.  boxes [B,max_boxes,4] (ymin, xmin, ymax, xmax fractions), labels [B,max_boxes], counts [B] or None:
    numpy arrays or CUDA tensors.  Returns a GroundTruth."""
    torch = _torch()
    dev = torch.device('cuda:%d' % (engine.device if engine is not None else (torch.cuda.current_device() if device is None else device)))

    def dev_tensor(a, dtype):
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=dev, dtype=dtype).contiguous()

    boxes = dev_tensor(boxes, torch.float32)
    labels = dev_tensor(labels, torch.int32)
    assert boxes.dim() == 3 and boxes.shape[2] == 4 and tuple(labels.shape) == tuple(boxes.shape[:2]), 'boxes [B,n,4], labels [B,n]'
    B, mb = int(boxes.shape[0]), int(boxes.shape[1])
    cnt = dev_tensor(counts, torch.int32) if counts is not None else None
    n, hw, pr, N = _layers_arrays(det_layers)
    loc = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
    obj = torch.empty((B, N), dtype=torch.float32, device=dev)
    ign = torch.empty((B, N), dtype=torch.float32, device=dev)
    cls = torch.empty((B, N), dtype=torch.int32, device=dev)
    ws_bytes = int(lib.byolo_encode_gt_workspace_bytes(B, mb))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(_handle(engine), lib.byolo_encode_gt(
            _handle(engine), n, hw, pr, ctypes.c_void_p(boxes.data_ptr() if mb else 0), ctypes.c_void_p(labels.data_ptr() if mb else 0),
            ctypes.c_void_p(cnt.data_ptr() if cnt is not None else 0), B, mb, float(ign_thresh),
            ctypes.c_void_p(loc.data_ptr()), ctypes.c_void_p(obj.data_ptr()), ctypes.c_void_p(cls.data_ptr()),
            ctypes.c_void_p(ign.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, ctypes.c_void_p(stream)))
    return GroundTruth(det_layers, loc, obj, cls, ign)


def detection_loss(raw, kind, cls_cnt, gt, aleatoric_loss=False, want_grad=False, engine=None):
    """byolo_loss for one detection layer.  raw: CUDA float32 [S,lh,lw,F] (dense).  gt: a dict of GroundTruth.layer(k) or any
    dict of CUDA tensors loc [S,lh,lw,3,4], obj / ign [S,lh,lw,3], cls [S,lh,lw,3] int32.  Returns {'loc','obj','cls'} as
    0-d float64 CUDA tensors (and 'grad' [S,lh,lw,F])."""
    torch = _torch()
    assert raw.is_cuda and raw.dtype == torch.float32 and raw.dim() == 4
    raw = raw.contiguous()
    S, lh, lw, F = [int(v) for v in raw.shape]
    dev = raw.device
    parent = gt.get('_gt')
    if parent is not None and parent.obj.device == dev and parent.obj.shape[0] == S:
        o, stride = parent.offset(gt['_layer']), parent.N                        # the layer's slice of the batch's arrays, in place
        g_loc, g_obj, g_cls, g_ign = parent.loc, parent.obj, parent.cls, parent.ign
        ptrs = (g_loc.data_ptr() + 16 * o, g_obj.data_ptr() + 4 * o, g_cls.data_ptr() + 4 * o, g_ign.data_ptr() + 4 * o)
    else:
        g_loc = gt['loc'].to(device=dev, dtype=torch.float32).contiguous()
        g_obj = gt['obj'].to(device=dev, dtype=torch.float32).contiguous()
        g_ign = gt['ign'].to(device=dev, dtype=torch.float32).contiguous()
        g_cls = gt['cls'].to(device=dev, dtype=torch.int32).contiguous()
        assert tuple(g_obj.shape) == (S, lh, lw, 3) and tuple(g_loc.shape) == (S, lh, lw, 3, 4), 'ground truth does not match the layer'
        stride = lh * lw * 3
        ptrs = (g_loc.data_ptr(), g_obj.data_ptr(), g_cls.data_ptr(), g_ign.data_ptr())
    out = torch.empty(3, dtype=torch.float64, device=dev)
    grad = torch.empty_like(raw) if want_grad else None
    nbytes = int(lib.byolo_loss_workspace_bytes())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # per call (24 KB from torch's stream-ordered pool): calls on several streams do not share partial sums
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        check(_handle(engine), lib.byolo_loss(
            _handle(engine), int(kind), int(bool(aleatoric_loss)), int(cls_cnt), ctypes.c_void_p(raw.data_ptr()), F, S, lh, lw,
            ctypes.c_void_p(ptrs[0]), ctypes.c_void_p(ptrs[1]), ctypes.c_void_p(ptrs[2]), ctypes.c_void_p(ptrs[3]), int(stride),
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(grad.data_ptr() if grad is not None else 0), F,
            ctypes.c_void_p(ws.data_ptr()), nbytes, ctypes.c_void_p(stream)))
    res = {'loc': out[0], 'obj': out[1], 'cls': out[2]}
    if want_grad:
        res['grad'] = grad
    return res


def l2_regularization(engine, scale=0.0005):
    """`tf.contrib.layers.l2_regularizer(l2_scale)` on every convolution kernel and on the detection layers' biases
    (lib_yolo/model.py:27, lib_yolo/layers.py:553-554, :604, :612) summed as `tf.losses.get_regularization_loss` does
    (lib_yolo/model.py:200): scale * sum(w ** 2) / 2 per tensor.  Host arithmetic on the handle's parameters (a constant of
    the weights, computed once per call)."""
    total = 0.0
    for name, shape in engine.param_shapes().items():
        if name.endswith('/kernel') or name.endswith('/bias'):
            w = engine.get_param(name, shape).astype(np.float64)
            total += scale * float(np.sum(w * w)) / 2.0
    return total
