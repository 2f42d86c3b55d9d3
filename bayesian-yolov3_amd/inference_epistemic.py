"""
Inference script for the yolov3.bayesian_yolov3_aleatoric class (MC-dropout epistemic + aleatoric
uncertainty) on MI355X -- drop-in for the reference's `inference_epistemic.py`.

Produces one ECP-format .json detection file per input image (usable by the ECP evaluation code).
Same entry points and helpers as the reference: `main()`, `inference(config)`, `Inference`, `nms`,
`concat_bbox`, `bbox_to_ecp_format`; same config keys (edit them in `main()`).

Differences, all additive: `batch_size > 1` is allowed (every image is reduced over its own T samples;
the reference asserts 1, `inference_epistemic.py:193`), and optional keys `weights='synthetic'`,
`seed`, `engine_options={'nms_mode': 1}` (the paper's per-class NMS, reference :104-126).
"""
import json
import logging
import os
import time

import numpy as np

from byolo import inference as _inf
from lib_yolo import yolov3

VARIANT = 'bayesian_yolov3_aleatoric'


def nms(boxes, model):
    # nms ignoring classes: tf.image.non_max_suppression(boxes[:, :4], boxes[:, model.obj_idx], 1000) + gather
    return _inf.nms(boxes, model, batched=False)


def nms_per_class(boxes, model):
    # the variant used for the paper (ped iff cls0 > cls1, rider iff cls1 > cls0; NMS 1000 each; ped then rider)
    return _inf.nms(boxes, model, batched=False, two_class=True)


def bbox_to_ecp_format(bbox, img_size, model, config):
    return _inf.bbox_to_ecp_format(bbox, img_size, model, config, VARIANT)


def concat_bbox(net_out):
    return _inf.concat_bbox(net_out, batched=False)


class Inference(_inf.InferenceLoop):
    def __init__(self, yolo, config):
        assert config['inference_mode']
        super().__init__(yolo, config, VARIANT, bbox_to_ecp_format, batched=False)


def inference(config):
    assert not config['crop']

    logging.info(json.dumps(config, indent=4, default=lambda x: str(x)))
    logging.info('----- START -----')
    start = time.time()

    yolo = yolov3.bayesian_yolov3_aleatoric(config)
    stats = Inference(yolo, config).run()

    elapsed = int(time.time() - start)
    logging.info('----- FINISHED in {:02d}:{:02d}:{:02d} -----'.format(elapsed // 3600, (elapsed // 60) % 60, elapsed % 60))
    return stats            # (the reference returns nothing) feed / device / writer waits of the loop: byolo/inference.py


def main():
    config = {
        'checkpoint_path': './checkpoints',  # edit
        'run_id': 'epi_ale',  # edit
        'step': 'last',  # edit: int or 'last'
        'full_img_size': [1024, 1920, 3],  # edit if not ECP dataset
        'cls_cnt': 2,  # edit if not ECP dataset
        'batch_size': 1,
        'T': 50,  # 288 GB of HBM: no need to lower it
        'inference_mode': True,
        'cpu_thread_cnt': 24,  # decode threads of the input feed (per process)
        'crop': False,
        'training': False,
        'aleatoric_loss': False,
        'priors': yolov3.ECP_9_PRIORS,  # edit
        'implicit_background_class': True,
        'data': {
            'path': '$HOME/data/ecp/tfrecords',  # edit
            'file_pattern': 'ecp-day-val-*-of-*',  # edit
        }
    }
    config['data']['file_pattern'] = os.path.join(os.path.expandvars(config['data']['path']),
                                                  config['data']['file_pattern'])
    config['out_path'] = os.path.join('./inference', config['run_id'])  # edit
    inference(config)


if __name__ == '__main__':
    np.set_printoptions(suppress=True, formatter={'float_kind': '{:5.3}'.format})
    logging.basicConfig(level=logging.DEBUG,
                        format='%(asctime)s, pid: %(process)d, %(levelname)-8s %(message)s',
                        datefmt='%a, %d %b %Y %H:%M:%S')
    main()
