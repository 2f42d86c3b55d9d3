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


def _box_op(driver, first_only):
    """One of the reference's three `box_op_*` (`detect.py:21-33`): concat the detection layers' rows and run the
    driver module's NMS; the two non-Bayesian drivers return a batch and the reference takes image 0."""
    def op(model):
        kept = driver.nms(driver.concat_bbox([d.bbox for d in model.det_layers]), model)
        return kept[0] if first_only else kept
    return op


box_op_standard = _box_op(inference_standard_yolov3, True)
box_op_aleatoric = _box_op(inference_aleatoric, True)
box_op_bayes = _box_op(inference_epistemic, False)


def filter_boxes(boxes, obj_idx, thresh):
    """Rows whose objectness exceeds `thresh`, order kept (`detect.py:36-37`)."""
    boxes = np.asarray(boxes)
    if boxes.size == 0:
        return []
    return list(boxes[boxes[:, obj_idx] > thresh])


def preproces_boxes(img_size, boxes, obj_idx, cls_start_idx, cls_cnt, config, cls_mapping=None):
    """Box rows -> the dicts `detect.py:40-63` hands to its drawing code (cls, score, obj_score, cls_score and the
    clipped pixel corners).  With `implicit_background_class` the reference shifts the winning class index by one
    BEFORE it reads the class score, so the score comes from the column right of the winner (:43-51); reproduced."""
    rows = np.asarray(boxes, dtype=np.float64).reshape(len(boxes), -1) if len(boxes) else np.zeros((0, cls_start_idx + cls_cnt + 1))
    winner = rows[:, cls_start_idx:cls_start_idx + cls_cnt].argmax(axis=1) if len(rows) else np.zeros(0, dtype=np.int64)
    winner = winner + (1 if config['implicit_background_class'] else 0)
    corners = np.clip(rows[:, :4], 0, 1) * np.asarray([img_size[0], img_size[1], img_size[0], img_size[1]], dtype=np.float64)
    out = []
    for k, box in enumerate(boxes):
        c = int(winner[k])
        cls_score = box[cls_start_idx + c]
        rec = {'cls': cls_mapping[c] if cls_mapping else c, 'score': box[obj_idx] * cls_score,
               'obj_score': box[obj_idx], 'cls_score': cls_score}
        rec.update(zip(('y0', 'x0', 'y1', 'x1'), (box.dtype.type(v) if hasattr(box, 'dtype') else v for v in corners[k])))
        out.append(rec)
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
    box_op = {'yolov3': box_op_standard, 'yolov3_aleatoric': box_op_aleatoric,
              'bayesian_yolov3_aleatoric': box_op_bayes}[model_cls.variant]
    model, img_tensor = load_model(config, model_cls)
    img_size = list(img_tensor.shape[1:])
    results = {}
    # One image per `sess.run` (detect.py:112-135): the device buffers of the loop are allocated ONCE, so that every forward after the
    # second has the arguments of the one before it and the library replays its launch graph instead of enqueueing ~85 launches
    # (include/byolo.h byolo_plan_opts.graphs; a forward that fills the chip by itself -- the Bayesian model at T = 35 -- runs eagerly)
    n_boxes, row_len = model.engine.num_boxes()
    x_dev = torch.empty((1,) + tuple(img_size), dtype=torch.float32, device=model.engine.torch_device)
    out = {'boxes': torch.empty((1, n_boxes, row_len), dtype=torch.float32, device=x_dev.device)}
    for file in files:
        img = load_img(config, img_size, file)
        x_dev.copy_(torch.from_numpy(img))
        model.run(x_dev, seed=int(config.get('seed', 0)), want_nms=False, out=out)
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


# What the reference's `main()` hard-codes and asks the user to edit (`detect.py:138-167`); every key can be
# overridden on the command line here instead.
DEFAULTS = dict(
    checkpoint_path='./checkpoints/', run_id='epi_ale', step='last',
    full_img_size=[1024, 1920, 3], crop_img_size=[768, 1440, 3], crop=False,
    cls_cnt=2, implicit_background_class=True,      # label ids start at 1 (True) or 0 (False)
    T=35,                                           # MC-dropout samples, Bayesian model only
    out_path='./uncertainty_visualization',
    # read by the model classes but without effect on inference:
    cpu_thread_cnt=10, training=False, freeze_darknet53=False, aleatoric_loss=True,
)
MODELS = {'standard': yolov3.yolov3, 'aleatoric': yolov3.yolov3_aleatoric, 'bayesian': yolov3.bayesian_yolov3_aleatoric}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('files', nargs='*', default=None, help="images (default: ./test_images/*)")
    ap.add_argument('--model', choices=sorted(MODELS), default='bayesian')
    ap.add_argument('--thresh', type=float, default=0.1)
    ap.add_argument('--run-id'), ap.add_argument('--step'), ap.add_argument('--checkpoint-path')
    ap.add_argument('--T', type=int), ap.add_argument('--crop', action='store_true')
    ap.add_argument('--save-dir', default=None, help="write annotated PNGs here")
    ap.add_argument('--show', action='store_true', help="open a window per image like the reference")
    args = ap.parse_args(argv)
    config = dict(DEFAULTS, priors=yolov3.ECP_9_PRIORS)
    for key in ('run_id', 'step', 'checkpoint_path', 'T'):
        if getattr(args, key) is not None:
            config[key] = getattr(args, key)
    config['crop'] = args.crop
    names = {1: 'ped', 2: 'rider'} if config['implicit_background_class'] else None
    return do_it(args.files or glob.glob('./test_images/*'), args.thresh, config, MODELS[args.model], names,
                 show=args.show, save_dir=args.save_dir)


if __name__ == '__main__':
    logging.basicConfig(level=logging.DEBUG,
                        format='%(asctime)s, pid: %(process)d, %(levelname)-8s %(message)s',
                        datefmt='%a, %d %b %Y %H:%M:%S')
    main()
