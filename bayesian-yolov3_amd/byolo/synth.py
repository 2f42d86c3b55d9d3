"""Synthetic (random-init) weights for benchmarks and parity tests -- there are no checkpoints in
this environment.  Recipe of SURVEY.md section 8d:

  * conv kernels  ~ N(0, 1/(k*k*cin)) (HWIO), BN gamma = 1, beta = 0, moving_mean = 0, moving_var = 1
    (the BN statistics are then calibrated on data: Engine.calibrate_bn on the device, or
    oracle.cpu_ref.forward(calibrate=True) for the CPU-generated golden fixtures);
  * detection kernels ~ N(0, 0.25/cin), bias: objectness -3, everything else 0, so logits stay within
    about +-6 (no exp overflow, no NaN entropies) and a few per cent of the boxes score non-trivially.

Every variable gets its own generator seeded by (seed, crc32(name)), so values do not depend on
creation order.  numpy only."""
import zlib

import numpy as np


def base_params(shapes, variant, cls_cnt, seed=7):
    """shapes: ordered {tf variable name: shape} (Engine.param_shapes() or the oracle's
    variable_shapes()).  Returns {name: float32 array}."""
    out = {}
    std_head = variant == "yolov3"
    blk = (5 + cls_cnt) if std_head else 2 * (5 + cls_cnt)
    obj_pos = 4 if std_head else 8
    for name, shp in shapes.items():
        g = np.random.default_rng([int(seed), zlib.crc32(name.encode())])
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            k, _, cin, _ = shp
            var = (0.25 / cin) if "/detection/" in name else 1.0 / (k * k * cin)
            v = g.standard_normal(shp) * np.sqrt(var)
        elif leaf == "bias":
            v = np.zeros(shp)
            v[obj_pos::blk] = -3.0
        elif leaf in ("gamma", "moving_variance"):
            v = np.ones(shp)
        elif leaf in ("beta", "moving_mean"):
            v = np.zeros(shp)
        else:
            raise ValueError("unexpected variable " + name)
        out[name] = v.astype(np.float32)
    return out


def synthetic_images(B, H, W, C=3, seed=1234, first_index=0):
    """i.i.d. U[0,1) frames, seed 1234 + image index."""
    imgs = np.empty((B, H, W, C), dtype=np.float32)
    for b in range(B):
        imgs[b] = np.random.default_rng(seed + first_index + b).random((H, W, C), dtype=np.float32)
    return imgs


ADVERSARIAL_KINDS = ("scales", "tails")


def adversarial_params(shapes, variant, cls_cnt, kind, seed=7):
    """Weights a TRAINED checkpoint may hold and the friendly recipe above never does (tests/test_robustness.py).  Still a
    sane network once its BN statistics are calibrated on data -- what changes is the dynamic range INSIDE a layer:

      'scales'  every output channel's filter is multiplied by 2^U(-10, 10): the conv output's variance spans 1e-6 .. 1e6
                across the channels of one layer (BN's moving variance absorbs it), gamma is log-uniform in [0.05, 50],
                beta ~ N(0, 1);
      'tails'   Student-t (3 degrees of freedom) filters -- a few weights 10..50 sigma out -- and gamma = 250 on every
                fourth BN'd layer, which drives activations to 1e3 .. 1e4 (the split-f16 storage ends at 16376).

    The layer in front of a detection head keeps gamma 1 / beta 0 and the detection kernels / biases keep the friendly
    recipe, so logits stay in a range where float32 evaluations agree.  Returns {name: float32}."""
    assert kind in ADVERSARIAL_KINDS, kind
    out = base_params(shapes, variant, cls_cnt, seed=seed)
    names = list(shapes)
    kernels = [n for n in names if n.endswith("/conv2d/kernel") and "/detection/" not in n]
    # a conv layer feeds a detection head iff its scope is the last BN'd one of its det_net_* scope
    last_of_head = set()
    for head in ("det_net_1", "det_net_2", "det_net_3"):
        hk = [n for n in kernels if n.startswith(head + "/")]
        if hk:
            last_of_head.add(hk[-1].rsplit("/conv2d/kernel", 1)[0])
    for li, kn in enumerate(kernels):
        scope = kn.rsplit("/conv2d/kernel", 1)[0]
        g = np.random.default_rng([int(seed), zlib.crc32(("adv:" + kn).encode())])
        k, _, cin, cout = shapes[kn]
        w = out[kn].astype(np.float64)
        gam = np.ones(cout)
        bet = np.zeros(cout)
        if kind == "scales":
            w = w * np.exp2(g.uniform(-10.0, 10.0, size=cout))[None, None, None, :]
            if scope not in last_of_head:
                gam = np.exp(g.uniform(np.log(0.05), np.log(50.0), size=cout))
                bet = g.standard_normal(cout)
        else:
            t = g.standard_t(3.0, size=w.shape) / np.sqrt(3.0)            # unit variance
            w = t * np.sqrt(1.0 / (k * k * cin))
            if scope not in last_of_head and li % 4 == 1:
                gam = np.full(cout, 250.0)
        out[kn] = w.astype(np.float32)
        out[scope + "/batch_normalization/gamma"] = gam.astype(np.float32)
        out[scope + "/batch_normalization/beta"] = bet.astype(np.float32)
    return out
