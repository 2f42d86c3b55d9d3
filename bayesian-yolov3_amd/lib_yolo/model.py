"""Graph builder -- mirror of `lib_yolo/model.py` (ModelBuilder :20-185, Model :188-225, DetLayer
:228-254, blueprints :257-268, img_size_and_priors_if_crop :6-17) on top of libbyolo.so.

The reference's ModelBuilder appends TF tensors to a layer list and routes by index; this one issues
one C-ABI graph call per `make_*` (include/byolo.h) and keeps the same list (as `LayerRef`s that carry
the TF tensor *names*, which the Darknet weight loader keys on).  Nothing is computed until
`Model.run(img)` -- the counterpart of `sess.run`.  `training=True` (batch-statistics BN, back-propagation, the optimiser) is
rejected: out of scope.  Ground truth IS accepted (`gt=` of the `make_detection_layer*` calls, as in the reference): the
loss terms of `lib_yolo/layers.py:126-188` are then evaluated on the device from the raw outputs of the last run
(`DetLayer.loc_loss` ..., `Model.total_loss` ...; csrc/train_kernels.hip) -- the validation-loss half of
`lib_yolo/train.py:69-72`."""
import contextlib

from byolo import Engine, NORM_BN, NORM_DROPOUT, DET_STANDARD, DET_ALEATORIC, DET_EPISTEMIC, NMS_AGNOSTIC
from byolo._lib import ByoloError, ERR_RANGE

from lib_yolo import data


def img_size_and_priors_if_crop(config):
    """`lib_yolo/model.py:6-17`, including its in-place rescale of the (shared) prior table when
    cropping."""
    img_size = config['crop_img_size'] if config['crop'] else config['full_img_size']
    priors = config['priors']

    if config['crop']:
        scale_h = config['full_img_size'][0] / float(config['crop_img_size'][0])
        scale_w = config['full_img_size'][1] / float(config['crop_img_size'][1])
        for stride, prs in priors.items():
            priors[stride] = [data.Prior(h=p.h * scale_h, w=p.w * scale_w) for p in prs]

    return img_size, priors


class Placeholder:
    """Stand-in for `tf.placeholder(tf.float32, shape=...)` (detect.py:93): a shape without data."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class LayerRef:
    """One entry of `Model.layers`: what the reference holds as a TF tensor."""

    def __init__(self, model_builder, index, name, kind):
        self.model_builder = model_builder
        self.index = index
        self.name = name            # e.g. 'darknet53/conv_3/LeakyRelu:0'
        self.kind = kind

    def __repr__(self):
        return '<LayerRef {} {}>'.format(self.index, self.name)


def _shape_of(inputs):
    shape = getattr(inputs, 'shape', None)
    if shape is None:
        raise TypeError('inputs must have a shape (tensor, array or lib_yolo.model.Placeholder)')
    if hasattr(shape, 'as_list'):
        shape = shape.as_list()
    return [None if s is None else int(s) for s in shape]


class ModelBuilder:
    def __init__(self, inputs, cls_cnt, l2_scale=0.0005, engine_options=None):
        self.l2_scale = l2_scale                  # lib_yolo/model.py:27 (tf.contrib.layers.l2_regularizer)
        self.__layers = []
        self.inputs = inputs
        self.__cls_cnt = cls_cnt
        self.__current_downsample = 1
        self.__det_layers = []

        shape = _shape_of(inputs)
        assert len(shape) == 4
        self.__input_size = [shape[1], shape[2]]  # h x w

        self.__scope = []
        self.__used = {}
        opts = dict(engine_options or {})
        self.engine = Engine(img_size=shape[1:], cls_cnt=cls_cnt, **opts)
        self.T = 1

    # -- tf.variable_scope(name) / tf.variable_scope(None, default_name=...) ---------------------
    @contextlib.contextmanager
    def variable_scope(self, name=None, default_name=None):
        parent = '/'.join(self.__scope)
        used = self.__used.setdefault(parent, set())
        if name is None:
            name, k = default_name, 0
            while name in used:
                k += 1
                name = '{}_{}'.format(default_name, k)
        used.add(name)
        self.__scope.append(name)
        try:
            yield '/'.join(self.__scope)
        finally:
            self.__scope.pop()

    def param_shapes(self):
        return self.engine.param_shapes()

    def layer_cnt(self):
        return len(self.__layers)

    def get_model(self, obj_idx, cls_start_idx):
        assert obj_idx < cls_start_idx
        return Model(self.__layers, self.__det_layers, self.__cls_cnt, obj_idx, cls_start_idx, builder=self)

    def __update_layers(self, index, name, kind):
        assert index == len(self.__layers), 'layer numbering out of sync with libbyolo'
        self.inputs = LayerRef(self, index, name, kind)
        self.__layers.append(self.inputs)

    @staticmethod
    def __norm_flags(normalizer):
        # lib_yolo/layers.py:556-571
        if isinstance(normalizer, dict):
            normalizer = [normalizer]
        flags = 0
        drop_prob = None
        for n in normalizer:
            if n['type'] in ('bn', 'darknet_bn'):
                if n.get('training', False):
                    raise NotImplementedError('training=True batch norm: training is out of scope (inference path)')
                flags |= NORM_BN
            elif n['type'] == 'dropout':
                if not n.get('standard_test_dropout', False):     # quirk: result discarded (layers.py:567-568)
                    flags |= NORM_DROPOUT
                    drop_prob = n['drop_prob']
            elif n['type'] is not None:
                raise ValueError('Invalid regularizer type: {}'.format(n['type']))
        return flags, drop_prob

    def __conv_layer(self, filters, kernel_size, strides, normalizer, variable_scope):
        assert strides in [1, 2]
        assert kernel_size in [1, 3], 'invalid kernel size'
        flags, drop_prob = self.__norm_flags(normalizer)
        if drop_prob is not None:
            assert abs(drop_prob - self.engine.cfg.drop_prob) < 1e-7, 'one drop_prob per model'
        with self.variable_scope(None, default_name=variable_scope) as scope:
            idx = self.engine.add_conv(scope, filters, kernel_size, strides, flags)
            self.__update_layers(idx, scope + '/LeakyRelu:0', 'conv')

    def make_conv_layer(self, filters, kernel_size, normalizer):
        self.__conv_layer(filters, kernel_size, 1, normalizer, 'conv')

    def make_downsample_layer(self, filters, kernel_size, normalizer):
        self.__current_downsample *= 2
        self.__conv_layer(filters, kernel_size, 2, normalizer, 'downsample')

    def __darknet_conv_layer(self, filters, kernel_size, strides, training, trainable, variable_scope):
        assert strides in [1, 2]
        if not trainable:
            assert not training
        self.__conv_layer(filters, kernel_size, strides, {'type': 'darknet_bn', 'training': training}, variable_scope)

    def make_darknet_conv_layer(self, filters, kernel_size, training, trainable):
        self.__darknet_conv_layer(filters, kernel_size, 1, training, trainable, 'conv')

    def make_darknet_downsample_layer(self, filters, kernel_size, training, trainable):
        self.__current_downsample *= 2
        self.__darknet_conv_layer(filters, kernel_size, 2, training, trainable, 'downsample')

    def make_route_layer(self, routes):
        assert len(routes) < 3, 'too many routes'
        assert len(routes), 'too few routes'
        with self.variable_scope(None, default_name='route') as scope:
            idx = self.engine.add_route(routes)
            self.__update_layers(idx, scope + ('/concat:0' if len(routes) > 1 else '/Identity:0'), 'route')

    def make_stack_feature_map_layer(self, layer, T):
        with self.variable_scope(None, default_name='stack_feature_map') as scope:
            assert self.T in (1, T), 'one T per model'
            self.T = T
            idx = self.engine.add_stack(layer)
            self.__update_layers(idx, scope + '/concat:0', 'stack')

    def make_residual_layer(self, shortcut):
        with self.variable_scope(None, default_name='residual') as scope:
            idx = self.engine.add_residual(shortcut)
            self.__update_layers(idx, scope + '/add:0', 'residual')

    def make_residual_block(self, filters, normalizer):
        self.make_conv_layer(filters, 1, normalizer)
        self.make_conv_layer(2 * filters, 3, normalizer)
        self.make_residual_layer(-3)

    def make_darknet_residual_block(self, filters, training, trainable):
        self.make_darknet_conv_layer(filters, 1, training, trainable)
        self.make_darknet_conv_layer(2 * filters, 3, training, trainable)
        self.make_residual_layer(-3)

    def make_upsample_layer(self):
        self.__current_downsample //= 2
        with self.variable_scope(None, default_name='upsample') as scope:
            idx = self.engine.add_upsample()
            self.__update_layers(idx, scope + '/ResizeNearestNeighbor:0', 'upsample')

    def __detection(self, all_priors, kind, gt, aleatoric_loss=False):
        if gt is not None and kind == DET_EPISTEMIC:
            # lib_yolo/model.py:171-173 would feed the decode_epistemic dict (no 'loc' key) to loss_tf: a KeyError there
            raise NotImplementedError('the loss exists outside inference mode only (lib_yolo/model.py:166-173)')
        priors = all_priors[self.__current_downsample]
        assert len(priors) == 3, 'exactly 3 priors per detection layer'
        with self.variable_scope('detection') as scope:
            idx = self.engine.add_detection(scope, kind, [(p.h, p.w) for p in priors])
            self.__update_layers(idx, scope + '/conv2d/BiasAdd:0', 'detection')
        self.__det_layers.append(DetLayer(
            input_img_size=self.__input_size,
            downsample_factor=self.__current_downsample,
            priors=priors,
            loss=None,
            det=None,
            bbox=None,
            raw_output=self.inputs,
            layer_id=len(self.__det_layers),
            kind=kind,
            gt=gt,
            aleatoric_loss=aleatoric_loss,
        ))
        return self.inputs

    def make_detection_layer(self, all_priors, gt=None):
        return self.__detection(all_priors, DET_STANDARD, gt)

    def make_detection_layer_aleatoric(self, all_priors, aleatoric_loss, gt=None):
        return self.__detection(all_priors, DET_ALEATORIC, gt, aleatoric_loss)

    def make_detection_layer_aleatoric_epistemic(self, all_priors, aleatoric_loss, gt=None, inference_mode=False):
        # model.py:166-170: the T-reduction only exists in inference mode
        return self.__detection(all_priors, DET_EPISTEMIC if inference_mode else DET_ALEATORIC, gt, aleatoric_loss)


class Model:
    def __init__(self, layers, det_layers, cls_cnt, obj_idx, cls_start_idx, builder=None):
        assert len(det_layers) > 0
        self.layers = layers
        self.det_layers = det_layers
        self.cls_cnt = cls_cnt
        self.obj_idx = obj_idx
        self.cls_start_idx = cls_start_idx
        self.builder = builder
        self.engine = builder.engine if builder is not None else None
        self.T = builder.T if builder is not None else 1
        self.last = None
        self.range_fallback = True          # run(): BYOLO_ERR_RANGE -> switch this model to the fp32 mode and re-run
        self.precision_switches = 0
        base = 0
        for dl in det_layers:
            dl._model = self
            dl._box_base = base
            base += 3 * dl.h * dl.w
        self.n_boxes = base

    # ---- the counterpart of sess.run --------------------------------------------------------------
    def finalize(self):
        self.engine.finalize()
        self._reg = None                # weights may have been restored / calibrated since the last access
        return self

    def run(self, img, seed=0, dropout_on=True, want_boxes=True, want_nms=True, first_image=0, out=None, mask_bits=None, slot=0,
            precision=None, t_shard=None):
        """img: float32 CUDA tensor [B,H,W,C] in [0,1).  Returns the dict of Engine.forward and keeps
        it as `self.last`, which the DetLayer accessors read.  `first_image`: position of img[0] in the logical
        (multi-GPU / split) batch; `out`: preallocated rows / kept / count tensors; `mask_bits`: injected dropout
        masks (Engine.pack_masks) instead of the build-defined stream.

        The reference computes in float32 (`lib_yolo/layers.py:550`), which holds any activation a trained checkpoint
        produces.  The default split-f16 arithmetic holds |activation| <= 16376: the library detects anything beyond
        (BYOLO_ERR_RANGE), and this wrapper then re-runs THAT BATCH in the fp32 mode, with a warning, instead of handing out
        rows the reference would not produce -- on a second handle that holds the same parameters packed for fp32
        (Engine.twin: both packs stay resident), so the batches after it run in the default precision again (round 4 switched
        the model for the rest of the run and re-packed in mid-stream).  `self.precision_switches` counts the switches in BOTH
        directions (two per such batch) and the returned dict says which arithmetic produced it ('precision').  This is the
        ONE-process behaviour (detect.py, vis_uncertainty.py, tests); `self.range_fallback = False` turns it into a plain
        ByoloError.  The multi-GPU driver (byolo/inference.py) runs the engine asynchronously -- the wrapper then never
        decides by itself -- lets all ranks agree through the status words its all-gather carries and asks for the re-run
        with `precision='f32'`.
        `slot`: workspace arena of the call (forwards in flight on different HIP streams must not share one).
        `t_shard` = (t0, t1): run samples t0 .. t1 - 1 of the image's T only and return their per-box SUMS in 'boxes' (the T axis
        sharded over ranks, Engine.forward / include/byolo.h byolo_set_tshard; one image, no NMS)."""
        if not self.engine.finalized:
            self.engine.finalize()
        kw = dict(T=self.T, seed=seed, dropout_on=dropout_on, want_boxes=want_boxes, want_nms=want_nms,
                  first_image=first_image, out=out, mask_bits=mask_bits, slot=slot)
        if t_shard is not None:
            kw.update(T=int(t_shard[1]) - int(t_shard[0]), t_shard=(int(t_shard[0]), self.T))
        if precision is not None and precision != self.engine.precision:
            eng = self.engine.twin(precision)
            if getattr(self.engine, '_async', False):
                eng.set_async(True)
            self.last = eng.forward(img, **kw)
            self.last['precision'] = eng.precision
            self.last['engine'] = eng
            return self.last
        try:
            self.last = self.engine.forward(img, **kw)
        except ByoloError as e:
            if e.code != ERR_RANGE or self.engine.precision != 'split' or getattr(self.engine, '_async', False) or not self.range_fallback:
                raise
            import logging
            logging.warning('%s -- this batch is re-run in the fp32 mode', e)
            self.engine.clear_status()
            eng = self.engine.twin('f32')
            self.precision_switches += 2                  # to fp32 and back
            self.last = eng.forward(img, **kw)
            self.last['precision'] = 'f32'
            self.last['engine'] = eng
            return self.last
        self.last['precision'] = self.engine.precision
        self.last['engine'] = self.engine
        return self.last

    def matches_blueprint(self, blueprint):
        try:
            for dl, bpdl in zip(self.det_layers, blueprint.det_layers):
                assert dl.matches_blueprint(bpdl)
            assert self.cls_cnt == blueprint.cls_cnt
        except AssertionError:
            return False
        return True

    # ---- losses of the last run (lib_yolo/model.py:197-216); None without ground truth, like the reference's attributes ----
    def set_ground_truth(self, gt):
        """gt: a byolo.loss.GroundTruth (lib_yolo.tfdata.encode_boxes_batch) or one dict per detection layer -- the counterpart
        of the dataset iterator feeding gt1, gt2, gt3 into the graph (lib_yolo/train.py:35-36).  Must match the batch of the run."""
        per_layer = gt.layers() if hasattr(gt, 'layers') else list(gt)
        assert len(per_layer) == len(self.det_layers)
        for dl, g in zip(self.det_layers, per_layer):
            dl.gt = g
            dl._loss_of = None

    def _sum(self, key):
        if self.det_layers[0].gt is None:
            return None
        total = None
        for dl in self.det_layers:
            v = dl.loss()[key]
            total = v if total is None else total + v
        return total

    @property
    def loc_loss(self):
        return self._sum('loc')

    @property
    def obj_loss(self):
        return self._sum('obj')

    @property
    def cls_loss(self):
        return self._sum('cls')

    @property
    def detection_loss(self):
        """`tf.losses.get_total_loss(add_regularization_losses=False)`: every term every detection layer added."""
        if self.det_layers[0].gt is None:
            return None
        return self.loc_loss + self.obj_loss + self.cls_loss

    @property
    def regularization_loss(self):
        if self.det_layers[0].gt is None:
            return None
        if getattr(self, '_reg', None) is None:
            from byolo import loss as _loss
            self._reg = _loss.l2_regularization(self.engine, self.builder.l2_scale)
        return self._reg

    @property
    def total_loss(self):
        if self.det_layers[0].gt is None:
            return None
        return self.detection_loss + self.regularization_loss


class DetLayer:
    def __init__(self, input_img_size, downsample_factor, priors, loss, det, bbox, raw_output, layer_id=0, kind=0, gt=None,
                 aleatoric_loss=False):
        self.h = input_img_size[0] // downsample_factor
        self.w = input_img_size[1] // downsample_factor
        self.downsample = downsample_factor
        self.priors = priors
        self.gt = gt if gt else None            # dict 'loc' [B,h,w,3,4], 'obj', 'ign', 'cls' [B,h,w,3] (CUDA tensors), or None
        self.aleatoric_loss = aleatoric_loss
        self._loss_of = None

        self.det = det
        self.layer_id = layer_id
        self.kind = kind
        self._raw_ref = raw_output
        self._model = None
        self._box_base = 0

    def loss(self, want_grad=False):
        """`layers.loss_tf(det, gt, aleatoric_loss)` of this layer (lib_yolo/model.py:118, :143, :173) on the raw output of the
        last `Model.run`: {'loc', 'obj', 'cls'} as 0-d float64 CUDA tensors (want_grad: + 'grad', the derivative with respect to the raw
        output).  Evaluated once per run."""
        from lib_yolo import layers
        if self.gt is None:
            return None
        last = self._model.last
        if last is None:
            raise RuntimeError('DetLayer.loss: call Model.run(img) first')
        if self._loss_of is not None and self._loss_of[0] is last and (not want_grad or 'grad' in self._loss_of[1]):
            return self._loss_of[1]
        raw = self.raw_output
        split = layers.split_detection if self.kind == DET_STANDARD else layers.split_detection_aleatoric
        det = split(raw, boxes_per_cell=len(self.priors), cls_cnt=self._model.cls_cnt)
        res = layers.loss_tf(det, self.gt, aleatoric_loss=self.aleatoric_loss, want_grad=want_grad, engine=self._model.engine)
        self._loss_of = (last, res)
        return res

    @property
    def loc_loss(self):
        return self.loss()['loc'] if self.gt is not None else None

    @property
    def obj_loss(self):
        return self.loss()['obj'] if self.gt is not None else None

    @property
    def cls_loss(self):
        return self.loss()['cls'] if self.gt is not None else None

    @property
    def bbox(self):
        """List of the 3 per-prior box tensors of the last `Model.run` -- [B,lh,lw,D] views into the
        concat_bbox-ordered box tensor (the reference's `det_layers[i].bbox`, model.py:237; the
        epistemic reference has no batch axis because it asserts batch 1)."""
        last = self._model.last if self._model is not None else None
        if last is None or last.get('boxes') is None:
            raise RuntimeError('DetLayer.bbox: call Model.run(img, want_boxes=True) first')
        boxes = last['boxes']
        n = self.h * self.w
        out = []
        for p in range(len(self.priors)):
            lo = self._box_base + p * n
            out.append(boxes[:, lo:lo + n, :].reshape(boxes.shape[0], self.h, self.w, boxes.shape[2]))
        return out

    @property
    def det(self):
        """The dict of `layers.decode_epistemic` (`lib_yolo/layers.py:397-411`) for the last run, image 0 (the
        reference's tensors have no batch axis: it asserts batch 1) -- every key of the reference's dict with the
        reference's shapes: `ev_loc` [lh,lw,3,4], `epi_covar_loc` [lh,lw,3,4,4] (full covariance), `ale_var_loc`
        [lh,lw,3,4], `obj_samples` [T,lh,lw,3], `obj_mean`, `obj_mutual_info`, `obj_entropy` [lh,lw,3], `cls_samples`
        [T,lh,lw,3,C], `cls_mean` [lh,lw,3,C], `cls_mutual_info`, `cls_entropy` [lh,lw,3].  The reduced statistics
        are views of the decoded box rows; `ev_loc`, the off-diagonal covariances and the per-sample tensors come
        from the layer's raw output through byolo_epistemic_stats (same one-pass sums as the decode kernel)."""
        import torch
        if self.kind != DET_EPISTEMIC:
            raise AttributeError('det statistics exist for epistemic detection layers only')
        per_prior = [b[0] for b in self.bbox]                          # 3 x [lh, lw, D]
        rows = torch.stack(per_prior, dim=2)                           # [lh, lw, 3, D]
        C = self._model.cls_cnt
        T = self._model.T
        raw = self.raw_output                                          # [B*T, lh, lw, F]
        st = self._model.engine.epistemic_stats(raw[:T].contiguous(), 1, T)
        return {
            'ev_loc': st['ev_loc'][0], 'epi_covar_loc': st['epi_covar_loc'][0],
            'ale_var_loc': rows[..., 8:12],
            'obj_samples': st['obj_samples'],
            'obj_mean': rows[..., 14], 'obj_mutual_info': rows[..., 15], 'obj_entropy': rows[..., 16],
            'cls_samples': st['cls_samples'],
            'cls_mean': rows[..., 17:17 + C], 'cls_mutual_info': rows[..., 17 + C], 'cls_entropy': rows[..., 18 + C],
        }

    @det.setter
    def det(self, value):
        pass

    @property
    def raw_output(self):
        """Raw detection-conv output [S,lh,lw,F] of the last run (of the handle that ran it: a batch beyond the split-f16 range
        ran on the fp32 twin, Model.run)."""
        eng = (self._model.last or {}).get('engine') or self._model.engine
        return eng.layer_output(self._raw_ref.index)

    def matches_blueprint(self, blueprint):
        try:
            assert self.h == blueprint.h
            assert self.w == blueprint.w
            assert self.downsample == blueprint.downsample
            assert len(self.priors) == len(blueprint.priors)
            for p, bpp in zip(self.priors, blueprint.priors):
                assert p.h == bpp.h
                assert p.w == bpp.w
        except AssertionError:
            return False
        return True


class ModelBlueprint:
    def __init__(self, det_layers, cls_cnt):
        self.det_layers = det_layers
        self.cls_cnt = cls_cnt


class DetLayerBlueprint:
    def __init__(self, input_img_size, downsample_factor, priors):
        self.h = input_img_size[0] // downsample_factor
        self.w = input_img_size[1] // downsample_factor
        self.downsample = downsample_factor
        self.priors = priors
