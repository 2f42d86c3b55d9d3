"""Host side of the entry points, native (csrc/host_io.cpp through the C-ABI): the PNG decode pool of the input feed
(`lib_yolo/dataset_utils.py:196`, `map(..., num_parallel_calls=cpu_thread_cnt)`) and the ECP-JSON text of the writer
(`inference_epistemic.py:84-92`).  Both calls drop the GIL, so a feeder thread, the device loop and writer threads overlap."""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib

KIND_OF_VARIANT = {'yolov3': _lib.DET_STANDARD, 'yolov3_aleatoric': _lib.DET_ALEATORIC, 'bayesian_yolov3_aleatoric': _lib.DET_EPISTEMIC}


def decode_png_batch(encoded, shape, out=None, threads=1):
    """encoded: list of PNG byte strings; shape (H, W, C).  Decodes into `out` (uint8 [n,H,W,C], C-contiguous; allocated if
    None) and returns (out, status int32 [n], found int32 [n,3]): status `_lib.PNG_*` per record."""
    n = len(encoded)
    h, w, c = [int(v) for v in shape]
    if out is None:
        out = np.empty((n, h, w, c), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.flags['C_CONTIGUOUS'] and out.shape[0] >= n and tuple(out.shape[1:]) == (h, w, c)
    status = np.full(n, _lib.PNG_CORRUPT, dtype=np.int32)
    found = np.zeros((max(n, 1), 3), dtype=np.int32)
    if n:
        keep = [e if isinstance(e, bytes) else bytes(e) for e in encoded]            # alive for the duration of the call
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(e), ctypes.c_void_p).value for e in keep])
        sizes = (ctypes.c_size_t * n)(*[len(e) for e in keep])
        rc = lib.byolo_png_decode_batch(ptrs, sizes, n, h, w, c, ctypes.c_void_p(out.ctypes.data), int(max(1, threads)),
                                        status.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                        found.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        if rc < 0:
            raise ValueError('byolo_png_decode_batch: bad argument')
    return out, status, found[:n]


class EcpJsonFormatter:
    """`json.dumps({'children': [to_ecp(row) for row in rows]})` for the stock `bbox_to_ecp_format` of a variant, as bytes."""

    def __init__(self, variant, img_size, cls_cnt, obj_idx, cls_start_idx, implicit_background_class, label_names):
        self.kind = KIND_OF_VARIANT[variant]
        self.h, self.w = int(img_size[0]), int(img_size[1])
        self.C, self.obj, self.cs = int(cls_cnt), int(obj_idx), int(cls_start_idx)
        self.bg = int(bool(implicit_background_class))
        n = (max(label_names) + 1) if label_names else 0
        self._names = [None] * n
        for k, v in (label_names or {}).items():
            if not (isinstance(k, int) and k >= 0 and isinstance(v, str) and v.isascii() and v.isprintable() and '"' not in v and '\\' not in v):
                raise ValueError('label table entry %r: %r cannot be written by the native formatter' % (k, v))
            self._names[k] = v.encode('ascii')
        self._labels = (ctypes.c_char_p * max(n, 1))(*(self._names or [None]))
        self._n_labels = n

    def format(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n, D = (rows.shape if rows.ndim == 2 else (0, 0))
        if n == 0:
            return b'{"children": []}'
        cap = 64 + n * (420 + 26 * (20 + self.C))
        while True:
            buf = ctypes.create_string_buffer(cap)
            got = lib.byolo_format_ecp_json(self.kind, ctypes.c_void_p(rows.ctypes.data), n, D, self.h, self.w, self.C, self.obj,
                                            self.cs, self.bg, self._labels, self._n_labels, buf, cap)
            if got >= 0:
                return buf.raw[:got]
            if got > -16:
                raise ValueError('byolo_format_ecp_json: rows of %d columns do not hold this variant (error %d)' % (D, got))
            cap = -got - 16 + 64
