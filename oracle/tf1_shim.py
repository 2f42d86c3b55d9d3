"""Oracle (test infrastructure): eager stand-in for the TensorFlow-1.x symbols the reference's
inference path touches, so that the reference's OWN graph-construction code
(`/root/reference/lib_yolo/{yolov3,model,layers,darknet}.py`, `inference_*.py` helpers) can be
imported and executed unmodified in this container to generate golden fixtures.

It is NOT TensorFlow.  The *structure* of every result produced through it (layer order, routing
indices, channel splits, decode formulas, T-reduction, box order, JSON mapping) is the reference's
own code; the *primitive arithmetic* (conv2d, batch-norm, dropout, softmax, det, NMS, ...) is
restated here from TensorFlow's documented semantics.  TensorFlow itself is a
third-party dependency of the reference with no pinned version (TF 1.x API, written around
1.8-1.12) and is not installable here, and the reference ships no tests => results are
"parity unpinned" at this boundary.  Used only by `oracle/make_golden.py` and tests that run in
this container; it never travels to the GPU box as anything but dead weight (it needs
/root/reference to be useful).

Usage:
    import oracle.tf1_shim as shim
    tf = shim.install(dtype=torch.float32, param_provider=fn, seed=42)   # sys.modules['tensorflow']
    sys.path.insert(0, '/root/reference'); from lib_yolo import yolov3 ...
"""
import contextlib
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import rng as _rng
from . import nms_ref as _nms_ref


# --------------------------------------------------------------------------------------------
# global state (one "graph")
# --------------------------------------------------------------------------------------------
class _State:
    def __init__(self):
        self.reset()

    def reset(self, dtype=torch.float32, param_provider=None, seed=0, drop_form="div"):
        self.dtype = dtype
        self.param_provider = param_provider
        self.seed = seed
        self.scope = []            # current variable-scope path components
        self.used = {}             # parent path -> set(names) for default_name uniquification
        self.variables = {}        # name (with ':0') -> Variable
        self.dropout_calls = []    # (ordinal, shape) of every *active* dropout call
        self.drop_form = drop_form  # 'div': (x / keep) * mask (TF<=1.12) | 'mul': x * (1/keep) * mask
        self.taps = []             # (name, tensor) of every op output worth recording
        self.sample_offset = 0     # first MC-sample index of this run (batch-1 loops over images)
        self.masks = None          # injected keep-masks, one per dropout call (None: the build-defined stream)


STATE = _State()


def _scope_path():
    return "/".join(STATE.scope)


def _opname(op):
    p = _scope_path()
    return (p + "/" if p else "") + op + ":0"


# --------------------------------------------------------------------------------------------
# tensor wrapper
# --------------------------------------------------------------------------------------------
class TensorShape:
    def __init__(self, dims):
        self.dims = list(dims)

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __repr__(self):
        return "TensorShape(%r)" % (self.dims,)


def _raw(x):
    if isinstance(x, Tensor):
        return x.t
    return x


def _idx(i):
    if isinstance(i, Tensor):
        return int(i.t.item())
    if isinstance(i, tuple):
        return tuple(_idx(j) for j in i)
    if isinstance(i, slice):
        return slice(_idx(i.start) if i.start is not None else None,
                     _idx(i.stop) if i.stop is not None else None,
                     _idx(i.step) if i.step is not None else None)
    return i


class Tensor:
    __array_priority__ = 1000

    def __init__(self, t, name="op:0"):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        self.t = t
        self.name = name

    # -- TF surface
    @property
    def shape(self):
        return TensorShape(self.t.shape)

    def get_shape(self):
        return self.shape

    def set_shape(self, shape):
        pass

    @property
    def dtype(self):
        return self.t.dtype

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __getitem__(self, i):
        return Tensor(self.t[_idx(i)], self.name)

    def __bool__(self):
        return bool(self.t.item())

    def __int__(self):
        return int(self.t.item())

    def __index__(self):
        return int(self.t.item())

    def __float__(self):
        return float(self.t.item())

    def __len__(self):
        return self.t.shape[0]

    # -- arithmetic
    def _b(self, other, fn, rev=False):
        o = _raw(other)
        if not isinstance(o, torch.Tensor):
            # python scalar: torch keeps the tensor dtype (== TF's scalar->tensor dtype conversion)
            return Tensor(fn(o, self.t) if rev else fn(self.t, o))
        return Tensor(fn(o, self.t) if rev else fn(self.t, o))

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._b(o, lambda a, b: a / b, True)
    def __floordiv__(self, o): return self._b(o, lambda a, b: a // b)
    def __pow__(self, o): return self._b(o, lambda a, b: a ** b)
    def __neg__(self): return Tensor(-self.t)
    def __lt__(self, o): return self._b(o, lambda a, b: a < b)
    def __le__(self, o): return self._b(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._b(o, lambda a, b: a > b)
    def __ge__(self, o): return self._b(o, lambda a, b: a >= b)


class Variable(Tensor):
    pass


def _T(x, dtype=None):
    if isinstance(x, Tensor):
        return x.t
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t


def _tap(name, tensor):
    STATE.taps.append((name, tensor))
    return tensor


# --------------------------------------------------------------------------------------------
# scopes / variables
# --------------------------------------------------------------------------------------------
@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, **kw):
    parent = _scope_path()
    used = STATE.used.setdefault(parent, set())
    if name_or_scope is None:
        name = default_name
        k = 0
        while name in used:           # conv, conv_1, conv_2, ... (tf unique_name)
            k += 1
            name = "%s_%d" % (default_name, k)
    else:
        name = name_or_scope
    used.add(name)
    STATE.scope.append(name)
    try:
        yield name
    finally:
        STATE.scope.pop()


@contextlib.contextmanager
def name_scope(name, *a, **kw):
    yield name


def _get_variable(layer, vname, shape):
    full = "%s/%s/%s:0" % (_scope_path(), layer, vname)
    if full in STATE.variables:
        raise ValueError("variable %s already exists" % full)
    if STATE.param_provider is None:
        raise RuntimeError("tf1_shim: no param_provider installed")
    val = STATE.param_provider(full[:-2], tuple(shape))
    val = np.asarray(val)
    assert tuple(val.shape) == tuple(shape), (full, val.shape, shape)
    v = Variable(torch.as_tensor(val).to(STATE.dtype), full)
    STATE.variables[full] = v
    return v


def global_variables():
    return list(STATE.variables.values())


def assign(var, value, validate_shape=True):
    value = np.asarray(value)
    if validate_shape:
        assert tuple(value.shape) == tuple(var.t.shape), (var.name, value.shape, tuple(var.t.shape))
    var.t = torch.as_tensor(value).to(STATE.dtype)
    return var


# --------------------------------------------------------------------------------------------
# tf.layers
# --------------------------------------------------------------------------------------------
def _conv2d(inputs, filters, kernel_size, strides=1, activation=None, padding="SAME", use_bias=True,
            trainable=True, kernel_regularizer=None, bias_regularizer=None, **kw):
    x = _T(inputs)                      # NHWC
    k = int(kernel_size)
    s = int(strides)
    cin = x.shape[3]
    w = _get_variable("conv2d", "kernel", (k, k, cin, filters))       # HWIO
    b = _get_variable("conv2d", "bias", (filters,)) if use_bias else None
    if padding == "SAME":
        assert s == 1, "shim: SAME only restated for stride 1 (layers.py:533-540 pads explicitly for s=2)"
        pad = (k - 1) // 2
    else:
        pad = 0
    # cross-correlation, NHWC x HWIO  (tf.nn.conv2d)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.t.permute(3, 2, 0, 1), None, stride=s, padding=pad)
    y = y.permute(0, 2, 3, 1).contiguous()
    if b is not None:
        y = y + b.t
        return _tap(_opname("conv2d/BiasAdd"), Tensor(y, _opname("conv2d/BiasAdd")))
    return _tap(_opname("conv2d/Conv2D"), Tensor(y, _opname("conv2d/Conv2D")))


def _batch_normalization(inputs, training=False, trainable=True, epsilon=1e-3, **kw):
    assert not training, "shim: inference only"
    x = _T(inputs)
    c = x.shape[-1]
    gamma = _get_variable("batch_normalization", "gamma", (c,))
    beta = _get_variable("batch_normalization", "beta", (c,))
    mean = _get_variable("batch_normalization", "moving_mean", (c,))
    var = _get_variable("batch_normalization", "moving_variance", (c,))
    eps = torch.tensor(epsilon, dtype=STATE.dtype)
    # y = (x - mean) * (gamma * rsqrt(var + eps)) + beta        
    inv = gamma.t * torch.rsqrt(var.t + eps)
    y = (x - mean.t) * inv + beta.t
    n = _opname("batch_normalization/FusedBatchNorm")
    return _tap(n, Tensor(y, n))


def _dropout(inputs, rate=0.5, training=False, **kw):
    x = _T(inputs)
    if not training:
        return Tensor(x, _opname("dropout/Identity"))
    ordinal = len(STATE.dropout_calls)
    STATE.dropout_calls.append((ordinal, tuple(x.shape)))
    off = STATE.sample_offset * int(np.prod(x.shape[1:]))
    if STATE.masks is not None:
        keep = np.asarray(STATE.masks[ordinal], dtype=bool).reshape(tuple(x.shape))
    else:
        keep = _rng.keep_mask(STATE.seed, ordinal, tuple(x.shape), drop_prob=rate, offset=off)
    m = torch.as_tensor(keep).to(STATE.dtype)
    keep_prob = torch.tensor(1.0 - rate, dtype=STATE.dtype)
    if STATE.drop_form == "div":
        y = (x / keep_prob) * m                                     # TF <= 1.12
    else:
        y = x * (torch.tensor(1.0, dtype=STATE.dtype) / keep_prob) * m   # TF >= 1.13
    n = _opname("dropout/mul")
    return _tap(n, Tensor(y, n))


def _flatten(x):
    t = _T(x)
    return Tensor(t.reshape(t.shape[0], -1))


# --------------------------------------------------------------------------------------------
# free functions
# --------------------------------------------------------------------------------------------
def _leaky_relu(x, alpha=0.2, name=None):
    t = _T(x)
    y = torch.maximum(t, t * alpha)            # max(x, alpha*x)
    n = _opname("LeakyRelu")
    return _tap(n, Tensor(y, n))


def _softmax(x, axis=-1):
    return Tensor(torch.softmax(_T(x), dim=axis))


def _pad(x, paddings, mode="CONSTANT"):
    t = _T(x)
    assert mode == "CONSTANT" and len(paddings) == t.dim()
    flat = []
    for lo, hi in reversed(paddings):
        flat += [int(lo), int(hi)]
    return Tensor(F.pad(t, flat))


def _shape(x):
    return Tensor(torch.tensor(list(_T(x).shape), dtype=torch.int64))


def _resize_nearest_neighbor(images, size, align_corners=False):
    t = _T(images)
    oh, ow = int(size[0]), int(size[1])
    ih, iw = t.shape[1], t.shape[2]
    # align_corners=False: src = floor(dst * in/out)
    ys = torch.floor(torch.arange(oh, dtype=torch.float64) * (ih / oh)).long().clamp(max=ih - 1)
    xs = torch.floor(torch.arange(ow, dtype=torch.float64) * (iw / ow)).long().clamp(max=iw - 1)
    y = t[:, ys][:, :, xs]
    n = _opname("ResizeNearestNeighbor")
    return Tensor(y.contiguous(), n)


def _non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, score_threshold=None, name=None):
    b = _T(boxes).detach().cpu().numpy().astype(np.float32)
    s = _T(scores).detach().cpu().numpy().astype(np.float32)
    keep = _nms_ref.nms_tf(b, s, int(max_output_size), float(iou_threshold))
    return Tensor(torch.as_tensor(keep.astype(np.int32)))


def _concat(values, axis=0, name=None):
    return Tensor(torch.cat([_T(v) for v in values], dim=axis), _opname("concat"))


def _identity(x, name=None):
    return Tensor(_T(x), _opname("Identity"))


def _split(value, num_or_size_splits, axis=0):
    t = _T(value)
    if isinstance(num_or_size_splits, int):
        assert t.shape[axis] % num_or_size_splits == 0
        parts = torch.split(t, t.shape[axis] // num_or_size_splits, dim=axis)
    else:
        parts = torch.split(t, list(num_or_size_splits), dim=axis)
    return [Tensor(p) for p in parts]


def _stack(values, axis=0):
    return Tensor(torch.stack([_T(v) for v in values], dim=axis))


def _squeeze(x, axis=None):
    t = _T(x)
    if axis is None:
        return Tensor(t.squeeze())
    if isinstance(axis, int):
        axis = [axis]
    nd = t.dim()
    for a in sorted([(a + nd) % nd for a in axis], reverse=True):
        assert t.shape[a] == 1
        t = t.squeeze(a)
    return Tensor(t)


def _expand_dims(x, axis):
    return Tensor(_T(x).unsqueeze(axis))


def _range(n, dtype=None):
    return Tensor(torch.arange(int(n)).to(_dt(dtype) if dtype is not None else torch.int64))


def _meshgrid(x, y):
    X, Y = torch.meshgrid(_T(x), _T(y), indexing="xy")
    return Tensor(X), Tensor(Y)


def _reduce(fn):
    def f(x, axis=None, keepdims=False):
        t = _T(x)
        if axis is None:
            return Tensor(fn(t))
        return Tensor(fn(t, dim=axis, keepdim=keepdims))
    return f


def _reduce_prod(x, axis=None, keepdims=False):
    t = _T(x)
    return Tensor(torch.prod(t, dim=axis, keepdim=keepdims)) if axis is not None else Tensor(torch.prod(t))


def _dt(d):
    if d is None:
        return STATE.dtype
    if d is _FLOAT32:
        return STATE.dtype            # "float32" of the graph == the oracle's working precision
    if d is _INT32:
        return torch.int32
    if d is _INT64:
        return torch.int64
    return d


def _ones(shape, dtype=None):
    return Tensor(torch.ones([int(s) for s in shape], dtype=_dt(dtype)))


def _ones_like(x):
    return Tensor(torch.ones_like(_T(x)))


def _zeros_like(x, dtype=None):
    return Tensor(torch.zeros_like(_T(x)))


def _reshape(x, shape):
    return Tensor(_T(x).reshape([int(s) for s in shape]))


def _gather(params, indices, axis=0):
    idx = _T(indices).long()
    if idx.dim() > 1:
        assert axis == 0
        return Tensor(_T(params)[idx])
    return Tensor(torch.index_select(_T(params), axis, idx))


# ---- symbols used only by vis_uncertainty.py (colour mapping of the uncertainty maps) ----------------
def _percentile(x, q, interpolation="nearest", **kw):
    """tf.contrib.distributions.percentile, default 'nearest': sorted[round((n-1) * q/100)]."""
    t = _T(x).reshape(-1)
    s, _ = torch.sort(t)
    i = int(torch.round(torch.tensor((t.numel() - 1) * (q / 100.0), dtype=torch.float64)).item())
    return Tensor(s[i])


def _convert_image_dtype(image, dtype, saturate=False):
    t = _T(image)
    if dtype is _UINT8 and t.is_floating_point():
        # scale = dtype.max + 0.5 (avoids rounding problems in the cast), then saturating cast
        return Tensor(torch.clamp(torch.floor(t * 255.5), 0, 255).to(torch.uint8))
    if dtype is _FLOAT32 and t.dtype == torch.uint8:
        return Tensor(t.to(STATE.dtype) * torch.tensor(1.0 / 255.0, dtype=STATE.dtype))
    raise NotImplementedError("convert_image_dtype %s -> %s" % (t.dtype, dtype))


def _constant(v, dtype=None):
    t = torch.as_tensor(v)
    if dtype is not None and (t.is_floating_point() or dtype is _FLOAT32):
        t = t.to(_dt(dtype))          # tf.constant(x, dtype=tf.float32): "float32" of the graph = the working precision
    return Tensor(t)


# ---- symbols used only by the training side (lib_yolo/tfdata.py encode_boxes, lib_yolo/layers.py loss_tf; row f4) ----
def _zeros(shape, dtype=None):
    return Tensor(torch.zeros([int(s) for s in shape], dtype=_dt(dtype)))


def _where(cond, x, y):
    return Tensor(torch.where(_T(cond), _T(x), _T(y)))


def _binary(fn):
    def f(a, b, name=None):
        a, b = _T(a), _T(b)
        if a.is_floating_point() and not b.is_floating_point():
            b = b.to(a.dtype)
        elif b.is_floating_point() and not a.is_floating_point():
            a = a.to(b.dtype)
        elif a.is_floating_point() and b.is_floating_point() and a.dtype != b.dtype:
            d = a.dtype if a.dim() >= b.dim() else b.dtype       # a Python scalar follows the tensor it meets
            a, b = a.to(d), b.to(d)
        return Tensor(fn(a, b))
    return f


def _sigmoid_cross_entropy_with_logits(labels=None, logits=None, name=None):
    """tf.nn.sigmoid_cross_entropy_with_logits (documented formula): max(x, 0) - x * z + log(1 + exp(-|x|))."""
    x, z = _T(logits), _T(labels)
    return Tensor(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x))))


def _sparse_softmax_cross_entropy_with_logits(labels=None, logits=None, name=None):
    """tf.nn.sparse_softmax_cross_entropy_with_logits: -log_softmax(logits)[label] along the last axis."""
    x, l = _T(logits), _T(labels).long()
    ls = torch.log_softmax(x, dim=-1)
    return Tensor(-torch.gather(ls, -1, l.unsqueeze(-1)).squeeze(-1))


class _Losses:
    """tf.losses: the LOSSES collection of the default graph."""

    def __init__(self):
        self.items = []

    def add_loss(self, loss, loss_collection=None):
        self.items.append(loss)

    def get_total_loss(self, add_regularization_losses=True, name=None):
        t = self.items[0]
        for l in self.items[1:]:
            t = t + l
        return t

    def get_regularization_loss(self, name=None):
        return Tensor(torch.zeros((), dtype=STATE.dtype))


def _while_loop(cond, body, loop_vars, shape_invariants=None, **kw):
    vars_ = list(loop_vars)
    while bool(cond(*vars_)):
        vars_ = list(body(*vars_))
    return vars_


def _cast(x, dtype):
    return Tensor(_T(x).to(_dt(dtype)))


def _det(x):
    return Tensor(torch.linalg.det(_T(x)))


def _diag_part(x):
    return Tensor(torch.diagonal(_T(x), dim1=-2, dim2=-1))


class _DType:
    def __init__(self, n):
        self.n = n

    def __repr__(self):
        return "tf." + self.n


_UINT8 = _DType("uint8")
_FLOAT32 = _DType("float32")
_INT32 = _DType("int32")
_INT64 = _DType("int64")


def build_module():
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "1.12-shim"
    tf.float32, tf.int32, tf.int64 = _FLOAT32, _INT32, _INT64
    tf.variable_scope, tf.name_scope = variable_scope, name_scope
    tf.global_variables, tf.assign = global_variables, assign
    tf.Tensor = Tensor
    tf.TensorShape = TensorShape

    tf.contrib = types.SimpleNamespace(layers=types.SimpleNamespace(l2_regularizer=lambda scale: ("l2", scale)))
    tf.layers = types.SimpleNamespace(conv2d=_conv2d, batch_normalization=_batch_normalization,
                                      dropout=_dropout, flatten=_flatten)
    tf.nn = types.SimpleNamespace(leaky_relu=_leaky_relu, softmax=_softmax)
    tf.image = types.SimpleNamespace(resize_nearest_neighbor=_resize_nearest_neighbor,
                                     non_max_suppression=_non_max_suppression)
    tf.linalg = types.SimpleNamespace(det=_det, diag_part=_diag_part)
    tf.pad, tf.shape = _pad, _shape
    tf.concat, tf.identity, tf.split, tf.stack = _concat, _identity, _split, _stack
    tf.squeeze, tf.expand_dims = _squeeze, _expand_dims
    tf.sigmoid = lambda x: Tensor(torch.sigmoid(_T(x)))
    tf.exp = lambda x: Tensor(torch.exp(_T(x)))
    tf.log = lambda x: Tensor(torch.log(_T(x)))
    tf.range, tf.meshgrid = _range, _meshgrid
    tf.reduce_mean = _reduce(torch.mean)
    tf.reduce_sum = _reduce(torch.sum)
    tf.reduce_prod = _reduce_prod
    tf.ones, tf.ones_like, tf.zeros_like = _ones, _ones_like, _zeros_like
    tf.reshape, tf.gather, tf.constant, tf.cast = _reshape, _gather, _constant, _cast
    tf.while_loop = _while_loop
    # lib_yolo/tfdata.py, lib_yolo/layers.py loss_tf
    tf.zeros, tf.where = _zeros, _where
    tf.greater_equal, tf.less_equal = _binary(lambda a, b: a >= b), _binary(lambda a, b: a <= b)
    tf.greater, tf.less = _binary(lambda a, b: a > b), _binary(lambda a, b: a < b)
    tf.logical_and = lambda a, b: Tensor(torch.logical_and(_T(a), _T(b)))
    tf.maximum, tf.minimum = _binary(torch.maximum), _binary(torch.minimum)
    tf.div, tf.add = _binary(lambda a, b: a / b), _binary(lambda a, b: a + b)
    tf.reduce_max = lambda x, axis=None: Tensor(torch.max(_T(x))) if axis is None else Tensor(torch.amax(_T(x), dim=axis))
    tf.nn.sigmoid_cross_entropy_with_logits = _sigmoid_cross_entropy_with_logits
    tf.nn.sparse_softmax_cross_entropy_with_logits = _sparse_softmax_cross_entropy_with_logits
    tf.losses = _Losses()
    tf.summary = types.SimpleNamespace(scalar=lambda name, value: None)
    # vis_uncertainty.py
    tf.uint8 = _UINT8
    tf.contrib.distributions = types.SimpleNamespace(percentile=_percentile)
    tf.reduce_min = lambda x, axis=None: Tensor(torch.min(_T(x))) if axis is None else Tensor(torch.amin(_T(x), dim=axis))
    tf.clip_by_value = lambda x, clip_value_min, clip_value_max: Tensor(torch.clamp(_T(x), clip_value_min, clip_value_max))
    tf.to_int32 = lambda x: Tensor(_T(x).to(torch.int32))
    tf.round = lambda x: Tensor(torch.round(_T(x)))                  # half to even, like tf.round
    tf.image.convert_image_dtype = _convert_image_dtype

    class _Errors:
        class OutOfRangeError(Exception):
            pass
    tf.errors = _Errors
    return tf


def install(dtype=torch.float32, param_provider=None, seed=0, drop_form="div", sample_offset=0, masks=None):
    """(Re)initialise the shim state and register it as ``tensorflow`` (plus a bare ``cv2`` stub
    for detect.py).  Returns the module."""
    STATE.reset(dtype=dtype, param_provider=param_provider, seed=seed, drop_form=drop_form)
    STATE.sample_offset = sample_offset
    STATE.masks = masks
    tf = sys.modules.get("tensorflow")
    if tf is None or getattr(tf, "__version__", "") != "1.12-shim":
        tf = build_module()
        sys.modules["tensorflow"] = tf
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    tf.losses.items = []                  # a fresh LOSSES collection per graph
    return tf


def input_tensor(array):
    """Wrap an NHWC numpy image batch as the graph input."""
    return Tensor(torch.as_tensor(np.asarray(array)).to(STATE.dtype), "input:0")
