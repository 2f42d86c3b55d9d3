"""Ground-truth encoding -- mirror of `lib_yolo/tfdata.py` (`encode_boxes` :77-171, `logit` :7-11) on top of libbyolo.so
(include/byolo.h: byolo_encode_gt; kernel: csrc/train_kernels.hip).

The reference encodes ONE image per call inside the training dataset's map stage (lib_yolo/dataset_utils.py:58-63) and lets
`tf.data` batch the results; `encode_boxes` keeps that signature, `encode_boxes_batch` encodes a whole batch in one launch."""
import numpy as np

from byolo import loss as _loss


def logit(x):
    """inverse of the sigmoid function (lib_yolo/tfdata.py:7-11) -- host helper, float32 like the graph's."""
    x = np.asarray(x, dtype=np.float32)
    return -np.log((np.float32(1.) / x) - np.float32(1.))


def encode_boxes_batch(bboxes, labels, det_layers, ign_thresh, counts=None, engine=None):
    """bboxes [B,n,4] (ymin, xmin, ymax, xmax as image fractions), labels [B,n], counts [B] (boxes of each image; None: n).
    Returns a byolo.loss.GroundTruth; `.layers()` = per detection layer the reference's dict, batched:
    'loc' [B,h,w,3,4], 'cls' [B,h,w,3] (int32), 'obj', 'ign' [B,h,w,3]."""
    return _loss.encode_gt(det_layers, bboxes, labels, counts=counts, ign_thresh=ign_thresh, engine=engine)


def encode_boxes(bboxes, labels, det_layers, ign_thresh, engine=None):
    """`tfdata.encode_boxes` (lib_yolo/tfdata.py:77-171): one image, bboxes [n,4], labels [n].  Returns the reference's list of
    dicts, one per detection layer: 'loc' [h,w,3,4], 'cls' [h,w,3], 'obj' [h,w,3], 'ign' [h,w,3] (CUDA tensors)."""
    import torch
    if torch.is_tensor(bboxes):
        bb, lab = bboxes.reshape(1, -1, 4), labels.reshape(1, -1)
    else:
        bb, lab = np.asarray(bboxes, np.float32).reshape(1, -1, 4), np.asarray(labels, np.int32).reshape(1, -1)
    gt = encode_boxes_batch(bb, lab, det_layers, ign_thresh, engine=engine)
    return [{k: v[0] for k, v in d.items() if not k.startswith('_')} for d in gt.layers()]
