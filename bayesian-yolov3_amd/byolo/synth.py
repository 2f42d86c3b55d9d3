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
