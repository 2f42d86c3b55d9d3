"""Darknet-53 topology and Darknet ``.weights`` import -- mirror of `lib_yolo/darknet.py`.

`darknet53(model_builder, training, trainable)` issues the same builder calls in the same order as
`lib_yolo/darknet.py:7-39` (52 convs + 23 residual adds, layer indices 0-74).
`load_darknet_weights(net_layers, weightfile)` follows `lib_yolo/darknet.py:42-122`: 5 x int32 header,
then per conv layer [beta, gamma, moving_mean, moving_variance] followed by the kernel stored
[cout, cin, kh, kw] (-> HWIO); biased (detection) convs store the bias first and have no BN."""
import numpy as np

_STAGES = ((64, 32, 1), (128, 64, 2), (256, 128, 8), (512, 256, 8), (1024, 512, 4))


def darknet53(model_builder, training, trainable):
    mb = model_builder
    mb.make_darknet_conv_layer(32, 3, training, trainable)                      # 0
    for down_filters, block_filters, repeats in _STAGES:                        # strides 2,4,8,16,32
        mb.make_darknet_downsample_layer(down_filters, 3, training, trainable)  # 1, 5, 12, 37, 62
        for _ in range(repeats):
            mb.make_darknet_residual_block(block_filters, training, trainable)  # .. 4, 11, 36, 61, 74


def load_darknet_weights(net_layers, weightfile):
    """net_layers: `model.layers[:darknet53_layer_cnt]` (LayerRef objects).  Assigns the variables
    of every conv layer in order and returns the list of assigned variable names (the reference
    returns tf.assign ops to be run by the caller; here assignment is immediate and the engine is
    re-finalised by the caller)."""
    with open(weightfile, "rb") as f:
        np.fromfile(f, dtype=np.int32, count=5)          # header: major, minor, revision, seen(x2)
        weights = np.fromfile(f, dtype=np.float32)

    ptr = 0
    assigned = []
    for l in net_layers:
        if 'LeakyRelu' not in l.name:                    # only conv layers carry variables (darknet.py:56)
            continue
        eng = l.model_builder.engine
        shapes = l.model_builder.param_shapes()
        scope = '/'.join(l.name.split('/')[:2])
        batch_norm = 'detection' not in l.name
        if batch_norm:
            for v in ('beta', 'gamma', 'moving_mean', 'moving_variance'):    # darknet order (darknet.py:114)
                name = '{}/batch_normalization/{}'.format(scope, v)
                n = int(np.prod(shapes[name]))
                eng.set_param(name, weights[ptr:ptr + n].reshape(shapes[name]))
                ptr += n
                assigned.append(name)
        else:
            name = scope + '/conv2d/bias'
            n = int(np.prod(shapes[name]))
            eng.set_param(name, weights[ptr:ptr + n].reshape(shapes[name]))
            ptr += n
            assigned.append(name)
        name = scope + '/conv2d/kernel'
        h, w, c, n_out = shapes[name]
        cnt = h * w * c * n_out
        kernel = weights[ptr:ptr + cnt].reshape([n_out, c, h, w]).transpose(2, 3, 1, 0)    # -> HWIO
        eng.set_param(name, kernel)
        ptr += cnt
        assigned.append(name)

    assert ptr == len(weights)                           # darknet.py:66
    return assigned
