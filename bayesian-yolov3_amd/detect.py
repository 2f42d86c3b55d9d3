"""Image-list front end -- drop-in for the reference's `detect.py` (functions `box_op_*`,
`filter_boxes`, `preproces_boxes`, `load_img`, `load_model`, `do_it`, `main`).

Runs the chosen model on image files and logs the detections above an objectness threshold.  The
reference additionally opens a matplotlib window per image (`detect.py:66-73`, `:133-134`); that
GUI part is out of scope -- pass `show=True` to `do_it` to get the same windows if matplotlib/cv2 are
available, or `save_dir=` to write annotated PNGs instead.
"""
import glob
import logging
import os

import numpy as np

import inference_aleatoric
import inference_epistemic
import inference_standard_yolov3
from byolo import inference as _inf
from lib_yolo import yolov3, model as _model


def box_op_standard(model):
    bbox = inference_standard_yolov3.concat_bbox([det_layer.bbox for det_layer in model.det_layers])
    return inference_standard_yolov3.nms(bbox, model)[0]


def box_op_aleatoric(model):
    bbox = inference_aleatoric.concat_bbox([det_layer.bbox for det_layer in model.det_layers])
    return inference_aleatoric.nms(bbox, model)[0]


def box_op_bayes(model):
    bbox = inference_epistemic.concat_bbox([det_layer.bbox for det_layer in model.det_layers])
    return inference_epistemic.nms(bbox, model)


def filter_boxes(boxes, obj_idx, thresh):
    return [box for box in boxes if box[obj_idx] > thresh]


def preproces_boxes(img_size, boxes, obj_idx, cls_start_idx, cls_cnt, config, cls_mapping=None):
    """`detect.py:40-63`.  Kept as is, including the reference's quirk that with
    `implicit_background_class` the class score is read one column to the right of the winning class
    (`cls_idx` is incremented before `box[cls_idx + cls_start_idx]`, :43-51)."""
    out = []
    for box in boxes:
        cls_idx = np.argmax(box[cls_start_idx:cls_start_idx + cls_cnt])
        if config['implicit_background_class']:
            cls_idx += 1
        cls = cls_mapping[cls_idx] if cls_mapping else cls_idx
        cls_score = box[cls_idx + cls_start_idx]
        out.append({
            'cls': cls,
            'score': box[obj_idx] * cls_score,
            'obj_score': box[obj_idx],
            'cls_score': cls_score,
            'y0': np.clip(box[0], 0, 1) * img_size[0],
            'x0': np.clip(box[1], 0, 1) * img_size[1],
            'y1': np.clip(box[2], 0, 1) * img_size[0],
            'x1': np.clip(box[3], 0, 1) * img_size[1],
        })
    return out


def load_img(config, img_size, filename):
    """`detect.py:76-85`: matplotlib semantics -- PNGs come back as float32 in [0,1], other formats
    (JPEG) as uint8 0-255 which the reference feeds unscaled (App. D.11); optional centre crop."""
    from PIL import Image
    img = np.asarray(Image.open(filename))
    if filename.lower().endswith('.png'):
        img = img.astype(np.float32) / np.float32(255.0 if img.dtype == np.uint8 else 65535.0)
    else:
        img = img.astype(np.float32)
    if img.ndim == 2:
        img = np.stack([img] * 3, axis=-1)
    img = img[:, :, :3]
    if config['crop']:
        y = (img.shape[0] - img_size[0]) // 2
        x = (img.shape[1] - img_size[1]) // 2
        img = img[y:y + img_size[0], x:x + img_size[1], :]
    return np.ascontiguousarray(img[None], dtype=np.float32)


def load_model(config, model_cls):
    if model_cls == yolov3.bayesian_yolov3_aleatoric:
        config['inference_mode'] = True
    yolo = model_cls(config)
    img_tensor = _model.Placeholder((1, *yolo.img_size))
    model = yolo.init_model(inputs=img_tensor, training=False).get_model()
    if config.get('weights') == 'synthetic':
        import torch
        from byolo import synth
        eng = model.engine
        eng.set_params(synth.base_params(eng.param_shapes(), model_cls.variant, model.cls_cnt, seed=7))
        eng.finalize()
        h, w, c = yolo.img_size
        eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(2, h, w, c, seed=999)).cuda())
    else:
        _inf.restore(model, _inf.find_checkpoint(config))
    return model, img_tensor


def do_it(files, thresh, config, model_cls, cls_mapping, show=False, save_dir=None):
    import torch
    box_op = {
        yolov3.yolov3: box_op_standard,
        yolov3.yolov3_aleatoric: box_op_aleatoric,
        yolov3.bayesian_yolov3_aleatoric: box_op_bayes,
    }[model_cls]

    model, img_tensor = load_model(config, model_cls)
    img_size = list(img_tensor.shape[1:])
    results = {}
    for file in files:
        img = load_img(config, img_size, file)
        model.run(torch.from_numpy(img).cuda(), seed=int(config.get('seed', 0)), want_nms=False)
        boxes = box_op(model).cpu().numpy()
        boxes = filter_boxes(boxes, model.obj_idx, thresh)
        boxes = preproces_boxes(img_size, boxes, model.obj_idx, model.cls_start_idx, model.cls_cnt,
                                config, cls_mapping=cls_mapping)
        logging.info('{}: {}'.format(os.path.basename(file), boxes))
        results[file] = boxes
        if show or save_dir:
            _draw(img[0], boxes, file, show, save_dir)
    return results


def _draw(img, boxes, file, show, save_dir):
    import matplotlib
    if not show:
        matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    ax.imshow(np.clip(img, 0, 1))
    for b in boxes:
        ax.add_patch(plt.Rectangle((b['x0'], b['y0']), b['x1'] - b['x0'], b['y1'] - b['y0'], fill=False,
                                   color=(43 / 255., 219 / 255., 216 / 255.), linewidth=1))
        ax.text(b['x0'], b['y0'], '{} {:4.3f}'.format(b['cls'], b['score']), fontsize=6,
                color=(43 / 255., 219 / 255., 216 / 255.))
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        fig.savefig(os.path.join(save_dir, os.path.splitext(os.path.basename(file))[0] + '.png'), dpi=150)
    if show:
        plt.show()
    plt.close(fig)


def main():
    config = {
        'checkpoint_path': './checkpoints/',
        'run_id': 'epi_ale',  # edit
        'step': 'last',  # edit: int or 'last'
        'crop_img_size': [768, 1440, 3],
        'full_img_size': [1024, 1920, 3],  # edit if not ecp
        'cls_cnt': 2,  # edit if not ecp
        'T': 35,  # only relevant for bayesian model
        'cpu_thread_cnt': 10,
        'freeze_darknet53': False,  # actual value irrelevant
        'crop': False,  # edit
        'training': False,
        'aleatoric_loss': True,  # actual value irrelevant
        'priors': yolov3.ECP_9_PRIORS,  # actual value irrelevant
        'out_path': './uncertainty_visualization',  # edit
        'implicit_background_class': True,  # whether the label ids start at 1 or 0. True = 1, False = 0
    }
    class_name_mapping_implicit_background_cls = {1: 'ped', 2: 'rider'}   # edit, or None
    thresh = 0.1  # edit
    files = glob.glob('./test_images/*')  # edit
    # EDIT: chose appropriate model class (yolov3.yolov3 / yolov3.yolov3_aleatoric / yolov3.bayesian_yolov3_aleatoric)
    model_cls = yolov3.bayesian_yolov3_aleatoric
    do_it(files, thresh, config, model_cls, class_name_mapping_implicit_background_cls)


if __name__ == '__main__':
    logging.basicConfig(level=logging.DEBUG,
                        format='%(asctime)s, pid: %(process)d, %(levelname)-8s %(message)s',
                        datefmt='%a, %d %b %Y %H:%M:%S')
    main()
