"""Model factories -- mirror of `lib_yolo/yolov3.py`: the five prior tables (:6-173) and the three
model classes `yolov3` (:176), `yolov3_aleatoric` (:313), `bayesian_yolov3_aleatoric` (:454) with the
reference's constructor config keys, attributes (`img_size`, `blueprint`, `cls_cnt`, `obj_idx`,
`cls_start_idx`) and methods (`init_model`, `get_model`, `load_darknet53_weights`).

The three classes share one topology (Darknet-53 + three heads); they differ in
the normaliser list of the head convs, the detection-layer kind and -- for the Bayesian model in
inference mode -- the `stack_feature_map` layers that fold the T MC samples into the batch axis
(`lib_yolo/yolov3.py:538-541`, `:569-571`, `:600-602`)."""
from lib_yolo import darknet, model, data

# (h, w) per prior, largest first; rows 0-2 -> stride 32, 3-5 -> stride 16, 6-8 -> stride 8.
_CITY_PERSONS_PX = ((495.27, 203.83), (297.84, 122.19), (197.44, 81.48), (141.07, 58.5), (102.72, 43.1),
                    (75.78, 31.66), (54.24, 23.19), (37.55, 16.15), (22.55, 10.09))     # px @ 1024 x 2048
_PRIOR_TABLES = {
    'ECP_9_PRIORS': (
        (0.56643243, 0.13731691), (0.41022839, 0.09028599), (0.30508716, 0.06047965),
        (0.20774711, 0.04376083), (0.15475611, 0.02996197), (0.10878717, 0.02149197),
        (0.07694039, 0.01488527), (0.05248527, 0.01007212), (0.03272104, 0.00631827)),
    'ECP_NIGHT_9_PRIORS': (
        (0.6197282176953125, 0.14694562146874998), (0.4243941425683594, 0.09687759120833334),
        (0.3103862368359375, 0.06362734035416667), (0.23494613041992188, 0.043568554453125),
        (0.1634832566796875, 0.03293052755208333), (0.12444031231445313, 0.023274527578125),
        (0.08800429220703125, 0.016930080526041665), (0.06101826478515625, 0.011638404229166668),
        (0.03925641140625, 0.007475639645833334)),
    'ECP_DAY_NIGHT_9_PRIORS': (
        (0.5728529907421875, 0.13943622409895834), (0.41761617583007815, 0.09156660707291667),
        (0.3015263176855469, 0.06248444700520834), (0.22101856140625, 0.042888710765625),
        (0.1533158565527344, 0.031196821406250002), (0.11255495265625, 0.021566710822916668),
        (0.07823327209960937, 0.015212825187500001), (0.0533416983203125, 0.010216603067708333),
        (0.0332035418359375, 0.006413999807291667)),
    'ECP_BIC_9_PRIORS': (
        (0.5541169062011718, 0.15767184942708334), (0.3872792363671875, 0.08849276056770834),
        (0.27297898112304686, 0.05552458755208333), (0.18570756796875, 0.034849724458333335),
        (0.13080457012695312, 0.052510955223958336), (0.12203939466796875, 0.02422101765625),
        (0.083340965234375, 0.01635016602083333), (0.055563667021484374, 0.010672233619791667),
        (0.03409191838867188, 0.006481136984375)),
    'CITY_PERSONS_9_PRIORS': tuple((h / 1024., w / 2048.) for h, w in _CITY_PERSONS_PX),
}


def _by_stride(rows):
    pr = [data.Prior(h=h, w=w) for h, w in rows]
    return {32: pr[0:3], 16: pr[3:6], 8: pr[6:9]}


CITY_PERSONS_9_PRIORS = _by_stride(_PRIOR_TABLES['CITY_PERSONS_9_PRIORS'])
ECP_9_PRIORS = _by_stride(_PRIOR_TABLES['ECP_9_PRIORS'])
ECP_NIGHT_9_PRIORS = _by_stride(_PRIOR_TABLES['ECP_NIGHT_9_PRIORS'])
ECP_DAY_NIGHT_9_PRIORS = _by_stride(_PRIOR_TABLES['ECP_DAY_NIGHT_9_PRIORS'])
ECP_BIC_9_PRIORS = _by_stride(_PRIOR_TABLES['ECP_BIC_9_PRIORS'])


class _Yolo:
    """Shared machinery of the three public classes."""
    variant = None
    obj_idx = None
    cls_start_idx = None

    def __init__(self, config):
        self._model = None
        self.img_size, self._priors = model.img_size_and_priors_if_crop(config)
        self._darknet53_layer_cnt = 0
        self._freeze_darknet53 = config.get('freeze_darknet53', True)
        self.cls_cnt = config['cls_cnt']
        self._engine_options = dict(config.get('engine_options', {}))     # build-specific (nms_mode, max_out, ...)

        self.blueprint = model.ModelBlueprint(det_layers=[
            model.DetLayerBlueprint(input_img_size=self.img_size, downsample_factor=s, priors=self._priors[s])
            for s in (32, 16, 8)
        ], cls_cnt=self.cls_cnt)

        # input size must be a multiple of the biggest stride (yolov3.py:207-211)
        assert config['full_img_size'][0] % 32 == 0
        assert config['full_img_size'][1] % 32 == 0
        if config['crop']:
            assert config['crop_img_size'][0] % 32 == 0
            assert config['crop_img_size'][1] % 32 == 0

    def set_engine_option(self, key, value):
        """Build-specific: an engine option (e.g. the device of this process under torchrun) set before init_model."""
        assert self._model is None, 'engine options are fixed once the model is built'
        self._engine_options[key] = value

    def get_model(self):
        """
        call init_model first!
        """
        assert self._model is not None, 'Call init_model first.'
        return self._model

    def load_darknet53_weights(self, weightfile):
        assert self._model is not None, 'Call init_model first.'
        return darknet.load_darknet_weights(self._model.layers[:self._darknet53_layer_cnt], weightfile)

    def init_model(self, inputs, training, gt1=None, gt2=None, gt3=None):
        if self._model is not None:
            raise Exception('model can only be initialized once!')
        if training:
            raise NotImplementedError('training=True (batch-statistics BN, back-propagation, the optimiser) is out of scope; '
                                      'ground truth with training=False evaluates the loss of the inference graph')
        self._gt = [gt1, gt2, gt3]              # lib_yolo/train.py:35-36: the dataset's encoded ground truth, one dict per layer

        self._build_model(inputs, training)
        assert self._model.matches_blueprint(self.blueprint), 'Model does not match blueprint'
        return self

    # -- per-variant hooks ------------------------------------------------------------------------
    def _head_normalizers(self, training):
        bn = {'type': 'bn', 'training': training}
        return bn, bn                                    # (first five convs of a head, bn-only convs)

    def _detection(self, mb, gt=None):
        raise NotImplementedError

    def _stack(self, mb, layer):
        return False                                     # no T-stacking

    def _build_model(self, inputs, training):
        mb = model.ModelBuilder(inputs=inputs, cls_cnt=self.cls_cnt, engine_options=self._engine_options)
        drop, bn = self._head_normalizers(training)

        with mb.variable_scope('darknet53'):
            darknet53_training = False if self._freeze_darknet53 else training
            darknet53_trainable = not self._freeze_darknet53
            darknet.darknet53(mb, training=darknet53_training, trainable=darknet53_trainable)   # 0 - 74
        dn_out = mb.inputs
        self._darknet53_layer_cnt = mb.layer_cnt()
        mb.engine.mark_backbone_end()

        self._stack(mb, -1)                               # Bayesian inference: tile dn_out T x (yolov3.py:541)

        outs = []
        for scope, filters, skip in (('det_net_1', 512, None), ('det_net_2', 256, 61), ('det_net_3', 128, 36)):
            with mb.variable_scope(scope):
                if skip is not None:
                    # -3 instead of -4: the YOLO layer is not in the layer list (yolov3.py:563)
                    mb.make_route_layer([-3])
                    mb.make_conv_layer(filters, 1, bn)
                    mb.make_upsample_layer()
                    if self._stack(mb, skip):
                        mb.make_route_layer([-2, -1])    # [upsampled, stacked skip]  (yolov3.py:571)
                    else:
                        mb.make_route_layer([-1, skip])  # [upsampled, skip]          (yolov3.py:269)
                for i in range(3):
                    mb.make_conv_layer(filters, 1, drop)
                    mb.make_conv_layer(2 * filters, 3, drop if i < 2 else bn)
                self._detection(mb, gt=self._gt[len(outs)])   # gt1 / gt2 / gt3 (yolov3.py:254, :278, :302)
                outs.append(mb.inputs)

        self._model = mb.get_model(self.obj_idx, self.cls_start_idx)
        self._model.dn_out = dn_out
        self._model.det_net_1_out, self._model.det_net_2_out, self._model.det_net_3_out = outs


class yolov3(_Yolo):
    variant = 'yolov3'
    obj_idx = 4
    cls_start_idx = 5

    def _detection(self, mb, gt=None):
        mb.make_detection_layer(all_priors=self._priors, gt=gt)


class yolov3_aleatoric(_Yolo):
    variant = 'yolov3_aleatoric'
    obj_idx = 9
    cls_start_idx = 11

    def __init__(self, config):
        self._aleatoric_loss = config['aleatoric_loss']          # required key (yolov3.py:315)
        super().__init__(config)

    def _detection(self, mb, gt=None):
        mb.make_detection_layer_aleatoric(all_priors=self._priors, aleatoric_loss=self._aleatoric_loss, gt=gt)


class bayesian_yolov3_aleatoric(_Yolo):
    variant = 'bayesian_yolov3_aleatoric'
    obj_idx = 14
    cls_start_idx = 17

    def __init__(self, config):
        self._aleatoric_loss = config['aleatoric_loss']          # required keys (yolov3.py:456, :461)
        self._inference_mode = config['inference_mode']
        # yolov3.py:462 hard-codes 0.1; the build lets engine_options['drop_prob'] override it (0 = tf.layers.dropout(rate=0) = identity)
        self._drop_prob = float(config.get('engine_options', {}).get('drop_prob', 0.1))
        if self._inference_mode:
            self._T = config['T']
        self._standard_test_dropout = config.get('standard_test_dropout', False)
        super().__init__(config)
        self._engine_options.setdefault('drop_prob', self._drop_prob)

    def _head_normalizers(self, training):
        bn = {'type': 'bn', 'training': training}
        dropout_bn = [
            {'type': 'dropout', 'drop_prob': self._drop_prob, 'standard_test_dropout': self._standard_test_dropout},
            bn,
        ]  # batch norm after dropout (yolov3.py:524-528)
        return dropout_bn, bn

    def _stack(self, mb, layer):
        if not self._inference_mode:
            return False
        mb.make_stack_feature_map_layer(layer, self._T)
        return True

    def _detection(self, mb, gt=None):
        mb.make_detection_layer_aleatoric_epistemic(all_priors=self._priors, aleatoric_loss=self._aleatoric_loss, gt=gt,
                                                    inference_mode=self._inference_mode)
