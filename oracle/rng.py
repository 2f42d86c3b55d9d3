"""Oracle (test infrastructure): numpy restatement of the build-defined MC-dropout RNG.

The reference never seeds its dropout (`lib_yolo/layers.py:521-524` calls
``tf.layers.dropout(rate, training=True)`` with no seed anywhere in the tree), so its MC
samples are irreproducible.  The build therefore DEFINES the Bernoulli stream as a pure
function of ``(seed, dropout_layer_ordinal, element_index)`` and implements the same function
bit-exactly here (numpy, uint32 wrap-around arithmetic) and on the device
(``bayesian-yolov3_amd/csrc/byolo_rng.h``).

  element_index i = linear NHWC index into the dropout input tensor [S, h, w, cout]
                  (S = images*T, sample s = img*T + t), as uint64
  dropout_layer_ordinal = 0..14, order of the dropout calls in one forward
                  (`lib_yolo/yolov3.py:544-548`, `:575-579`, `:606-610`)
  h0(g)    one lowbias32 round over the GROUP index g = i >> 2, keyed at both ends (pair_hash)
  h1(g)    next_word(h0): y = h0 * 0x9E3779B1; y ^= y >> 16
  keep(i)  <=>  16-bit field (i & 3) of (h0, h1)  <  min(round((1 - drop_prob) * 2^16), 65535)
(round 5 / ABI 6; rounds 1 - 4 hashed every pair of channels: csrc/byolo_rng.h says why this is cheaper on the device)
"""
import numpy as np

_M1 = np.uint32(0x21F0AAAD)
_M2 = np.uint32(0x735A2D97)
_GOLD = np.uint32(0x9E3779B9)
_M3 = np.uint32(0x9E3779B1)


def mix32(x):
    """lowbias32 avalanche hash on uint32 arrays (wrap-around arithmetic); key derivation."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(15)
    return x


def layer_keys(seed, layer):
    """Per-(seed, dropout layer) key pair; computed on the host in the product too."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    layer = int(layer) & 0xFFFFFFFF
    lo = np.uint32(seed & 0xFFFFFFFF)
    hi = np.uint32(seed >> 32)
    with np.errstate(over="ignore"):
        k0 = mix32(np.uint32(lo ^ (_GOLD * np.uint32(layer + 1))))
        k1 = mix32(np.uint32(hi + k0 + np.uint32(layer)))
    return np.uint32(k0), np.uint32(k1)


def keep_threshold(drop_prob):
    # drop_prob travels through the C-ABI as float32 (byolo_cfg.drop_prob); the 16-bit threshold is
    # computed from that float32 value in double precision (csrc/byolo_rng.h: byolo_layer_keys)
    return np.uint32(min(65535, int((1.0 - float(np.float32(drop_prob))) * 65536.0 + 0.5)))


def is_identity(drop_prob):
    """A rate whose 16-bit threshold rounds to 2^16 keeps everything (csrc/byolo_rng.h byolo_drop_is_identity): rate 0 is the
    identity, as tf.layers.dropout(rate=0) is."""
    return (1.0 - float(np.float32(drop_prob))) * 65536.0 + 0.5 >= 65536.0


def next_word(h0):
    """The second 32 mask bits of a group from its first."""
    with np.errstate(over="ignore"):
        y = np.asarray(h0, dtype=np.uint32) * _M3
        y ^= y >> np.uint32(16)
    return y


def pair_hash(g, k0, k1):
    """The first 32 mask bits of every group index in the uint64 array g."""
    g = np.asarray(g, dtype=np.uint64)
    lo = (g & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (g >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = lo + np.uint32(k0)
        x ^= x >> np.uint32(16)
        x *= _M1
        x ^= np.uint32(k1) + hi * _GOLD
        x ^= x >> np.uint32(15)
        x *= _M2
        x ^= x >> np.uint32(15)
    return x


def keep_mask(seed, layer, shape, drop_prob=0.1, offset=0):
    """Boolean keep-mask for a dropout input of NHWC `shape` (element order = C order)."""
    n = int(np.prod(shape))
    if is_identity(drop_prob):
        return np.ones(shape, dtype=bool)
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    k0, k1 = layer_keys(seed, layer)
    h = pair_hash(idx >> np.uint64(2), k0, k1)
    h = np.where((idx & np.uint64(2)).astype(bool), next_word(h), h)
    u = np.where((idx & np.uint64(1)).astype(bool), h >> np.uint32(16), h & np.uint32(0xFFFF))
    return (u < keep_threshold(drop_prob)).reshape(shape)


def keep_mask_torch(seed, layer, shape, drop_prob=0.1, offset=0, chunk=1 << 24):
    """Same function evaluated with torch (multi-threaded; int64 lanes holding 32-bit values) for
    the large tensors of the CPU baseline.  Bit-identical to keep_mask (tests/test_oracle.py)."""
    import torch
    n = int(np.prod(shape))
    if is_identity(drop_prob):
        return torch.ones(shape, dtype=torch.bool)
    k0, k1 = (int(v) for v in layer_keys(seed, layer))
    thr = int(keep_threshold(drop_prob))
    M = 0xFFFFFFFF
    out = torch.empty(n, dtype=torch.bool)
    for lo_i in range(0, n, chunk):
        hi_i = min(n, lo_i + chunk)
        idx = torch.arange(offset + lo_i, offset + hi_i, dtype=torch.int64)
        g = idx >> 2
        x = ((g & M) + k0) & M
        x = x ^ (x >> 16)
        x = (x * 0x21F0AAAD) & M
        x = x ^ ((k1 + ((g >> 32) * 0x9E3779B9)) & M)
        x = x ^ (x >> 15)
        x = (x * 0x735A2D97) & M
        x = x ^ (x >> 15)
        y = (x * 0x9E3779B1) & M
        y = y ^ (y >> 16)
        x = torch.where((idx & 2).bool(), y, x)
        u = torch.where((idx & 1).bool(), x >> 16, x & 0xFFFF)
        out[lo_i:hi_i] = u < thr
    return out.reshape(shape)
