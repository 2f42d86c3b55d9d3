"""
Inference script for the yolov3.yolov3 class on MI355X -- drop-in for the reference's `inference_standard_yolov3.py`.

Produces one ECP-format .json detection file per input image (usable by the ECP evaluation code).
Same entry points and helpers as the reference: `main()`, `inference(config)`, `Inference`, `nms`,
`concat_bbox`, `bbox_to_ecp_format`; same config keys (edit them in `main()`).

Differences, all additive: `nms` returns a list of per-image row tensors instead of concatenating them
(the reference's concat needs equal kept counts per image, `inference_standard_yolov3.py:137-143`); optional keys
`weights='synthetic'`, `engine_options={'nms_mode': 1}`.
"""
import json
import logging
import os
import time

import numpy as np

from byolo import inference as _inf
from lib_yolo import yolov3

VARIANT = 'yolov3'


def nms(boxes, model):
    # nms ignoring classes: tf.image.non_max_suppression(boxes[:, :4], boxes[:, model.obj_idx], 1000) + gather
    return _inf.nms(boxes, model, batched=True)


def nms_per_class(boxes, model):
    # the variant used for the paper (ped iff cls0 > cls1, rider iff cls1 > cls0; NMS 1000 each; ped then rider)
    return _inf.nms(boxes, model, batched=True, two_class=True)


def bbox_to_ecp_format(bbox, img_size, model, config):
    return _inf.bbox_to_ecp_format(bbox, img_size, model, config, VARIANT)


def concat_bbox(net_out):
    return _inf.concat_bbox(net_out, batched=True)


class Inference(_inf.InferenceLoop):
    def __init__(self, yolo, config):
        super().__init__(yolo, config, VARIANT, bbox_to_ecp_format, batched=True)


def inference(config):
    assert not config['crop']

    logging.info(json.dumps(config, indent=4, default=lambda x: str(x)))
    logging.info('----- START -----')
    start = time.time()

    yolo = yolov3.yolov3(config)
    stats = Inference(yolo, config).run()

    elapsed = int(time.time() - start)
    logging.info('----- FINISHED in {:02d}:{:02d}:{:02d} -----'.format(elapsed // 3600, (elapsed // 60) % 60, elapsed % 60))
    return stats            # (the reference returns nothing) feed / device / writer waits of the loop: byolo/inference.py


def main():
    config = {
        'checkpoint_path': './checkpoints',  # edit
        'run_id': 'yolov3',  # edit
        'step': 'last',  # edit: int or 'last'
        'full_img_size': [1024, 1920, 3],  # edit if not ECP dataset
        'cls_cnt': 2,  # edit if not ECP dataset
        'batch_size': 11,  # edit
        'cpu_thread_cnt': 24,  # decode threads of the input feed (per process)
        'crop': False,
        'training': False,
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
