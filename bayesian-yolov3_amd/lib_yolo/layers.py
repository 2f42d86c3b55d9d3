"""Training-side pieces of `lib_yolo/layers.py`: `split_detection` (:11-38), `split_detection_aleatoric` (:41-84) and `loss_tf`
(:126-188) on top of libbyolo.so (include/byolo.h: byolo_loss; kernel: csrc/train_kernels.hip).

The inference-side functions of the reference's file (conv, dropout, decode_*) have no Python counterpart here: they are
the library's kernels behind `lib_yolo/model.py`'s `ModelBuilder`.  The inactive `aleatoric_obj_loss` / `aleatoric_cls_loss`
(:87-123, commented out at their only call sites :161-163, :172-174) are not built."""
from byolo import loss as _loss, DET_STANDARD, DET_ALEATORIC


def _split(inputs, boxes_per_cell, cls_cnt, aleatoric):
    assert boxes_per_cell == 3, 'exactly 3 priors per detection layer'
    blk = (10 + 2 * cls_cnt) if aleatoric else (5 + cls_cnt)
    b, h, w, F = inputs.shape
    assert F == boxes_per_cell * blk, 'channel count of a detection layer'
    r = inputs.reshape(b, h, w, boxes_per_cell, blk)
    if not aleatoric:
        det = {'loc': r[..., 0:4], 'obj': r[..., 4], 'cls': r[..., 5:5 + cls_cnt]}
    else:
        det = {'loc': r[..., 0:4], 'log_loc_var': r[..., 4:8], 'obj': r[..., 8], 'log_obj_stddev': r[..., 9],
               'cls': r[..., 10:10 + cls_cnt], 'log_cls_stddev': r[..., 10 + cls_cnt:10 + 2 * cls_cnt]}
    det['_raw'], det['_cls_cnt'], det['_kind'] = inputs, cls_cnt, DET_ALEATORIC if aleatoric else DET_STANDARD
    return det


def split_detection(inputs, boxes_per_cell, cls_cnt):
    """lib_yolo/layers.py:11-38: views 'loc' [b,h,w,3,4], 'obj' [b,h,w,3], 'cls' [b,h,w,3,C] of the raw output [b,h,w,3*(5+C)]."""
    return _split(inputs, boxes_per_cell, cls_cnt, False)


def split_detection_aleatoric(inputs, boxes_per_cell, cls_cnt):
    """lib_yolo/layers.py:41-84: additionally 'log_loc_var', 'log_obj_stddev', 'log_cls_stddev'."""
    return _split(inputs, boxes_per_cell, cls_cnt, True)


def loss_tf(det, gt, aleatoric_loss=False, want_grad=False, engine=None):
    """lib_yolo/layers.py:126-188.  det: the dict of split_detection(_aleatoric) above (views of ONE raw tensor: the fused
    kernel reads that tensor once); gt: dict 'loc' [b,h,w,3,4], 'obj', 'ign' [b,h,w,3], 'cls' [b,h,w,3] (e.g. one entry of
    tfdata.encode_boxes_batch(...).layers()).  Returns {'loc', 'obj', 'cls'} (0-d float64 CUDA tensors; want_grad: also
    'grad' = d(loc + obj + cls) / d(raw output), what `optimizer.minimize` sends into the network, lib_yolo/train.py:88)."""
    if '_raw' not in det:
        raise TypeError('loss_tf: det must come from lib_yolo.layers.split_detection / split_detection_aleatoric')
    if aleatoric_loss and det['_kind'] != DET_ALEATORIC:
        raise KeyError('log_loc_var')                              # what the reference's dict lookup would raise (layers.py:151)
    return _loss.detection_loss(det['_raw'], det['_kind'], det['_cls_cnt'], gt, aleatoric_loss=aleatoric_loss,
                                want_grad=want_grad, engine=engine)
