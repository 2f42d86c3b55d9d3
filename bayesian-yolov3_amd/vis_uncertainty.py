"""Per-cell uncertainty heat maps -- counterpart of the reference's `vis_uncertainty.py` for the
yolov3.bayesian_yolov3_aleatoric class (functions `colorize`, `color_map`, `Inference`,
`save_uncertainty_maps`, `worker`, `do_it`, `main`; same config keys, same output file names
`<image>_prior<k>_<ucty>.png`, k = 0..8 over strides 32/16/8 x 3 priors).

The reference builds the model and runs the T-sample network once per uncertainty kind (11 separate
processes per file list, `vis_uncertainty.py:178-202`); here ONE forward per image yields all kinds, since
every statistic is part of the decoded box row (`lib_yolo/model.py` `DetLayer.det`).  Colour mapping is a
few KB of host work: plasma colour map, vmin = 0, vmax = 99th percentile (nearest), nearest upsample by
the layer stride, alpha blend 0.7, uint8 conversion as `tf.image.convert_image_dtype`."""
import glob
import logging
import os
import time

import numpy as np

from byolo import inference as _inf
from lib_yolo import yolov3, model as _model

UCTY_KINDS = ([('epi_covar_loc', i, 'epi_' + n) for i, n in enumerate('xywh')] +
              [('ale_var_loc', i, 'ale_' + n) for i, n in enumerate('xywh')] +
              [(k, -1, k) for k in ('cls_mutual_info', 'obj_mean', 'obj_mutual_info')])


def colorize(img, vmin=None, vmax=None, cmap='plasma'):
    """`vis_uncertainty.py:15-32`: img [h,w,1] -> RGB float [h,w,3] via a 256-entry colour map."""
    import matplotlib
    img = np.asarray(img, dtype=np.float32)
    vmin = img.min() if vmin is None else np.float32(vmin)
    # tf.contrib.distributions.percentile(img, 99.): 'nearest' interpolation on the flattened values
    vmax = np.percentile(img.reshape(-1), 99., method='nearest').astype(np.float32) if vmax is None else np.float32(vmax)
    img = (img - vmin) / (vmax - vmin)
    img = img[..., 0]
    indices = np.clip(np.rint(img * np.float32(255)).astype(np.int32), 0, 255)       # tf.round: half to even
    cm = matplotlib.colormaps[cmap if cmap is not None else 'gray']
    colors = np.asarray(cm.colors, dtype=np.float32)
    return colors[indices]


def color_map(img, uncertainty, stride, vmin, vmax, alpha=0.7):
    """`vis_uncertainty.py:35-47`: img [1,H,W,3] float in [0,1], uncertainty [h,w,1] -> uint8 [H,W,3]."""
    u = colorize(uncertainty, vmin, vmax)
    u = np.repeat(np.repeat(u, stride, axis=0), stride, axis=1)                       # resize_nearest_neighbor
    blended = np.float32(alpha) * np.asarray(img, dtype=np.float32)[0] + np.float32(1 - alpha) * u
    # tf.image.convert_image_dtype(float -> uint8): saturate_cast(x * (255 + 0.5))
    return np.clip(np.floor(blended * np.float32(255.5)), 0, 255).astype(np.uint8)


class Inference:
    def __init__(self, yolo, config):
        self.batch_size = config['batch_size']
        self.img_size = yolo.img_size
        self.config = config
        assert config['inference_mode']
        self.model = yolo.init_model(inputs=_model.Placeholder((1, *self.img_size)), training=False).get_model()
        if config.get('weights') == 'synthetic':
            import torch
            from byolo import synth
            eng = self.model.engine
            eng.set_params(synth.base_params(eng.param_shapes(), 'bayesian_yolov3_aleatoric', self.model.cls_cnt, seed=7))
            eng.finalize()
            h, w, c = self.img_size
            eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(2, h, w, c, seed=999)).cuda())
        else:
            self.checkpoint = _inf.find_checkpoint(config)
            _inf.restore(self.model, self.checkpoint)

    def load_img(self, filename):
        from PIL import Image
        img = np.array(Image.open(filename)).astype(np.float32)
        if self.config['crop']:
            y = (img.shape[0] - self.img_size[0]) // 2
            x = (img.shape[1] - self.img_size[1]) // 2
            img = img[y:y + self.img_size[0], x:x + self.img_size[1], :]
        img = np.expand_dims(img[:, :, :3], axis=0)
        img /= 255.
        return np.ascontiguousarray(img)

    def uncertainty_grids(self, img_data, uncertainty_key, ucty_idx=-1):
        """The 9 blended maps (strides 32, 16, 8 x priors 0..2) of one uncertainty kind for the last run."""
        grids = []
        for l in self.model.det_layers:
            d = l.det[uncertainty_key]
            if 'obj' in uncertainty_key or 'cls' in uncertainty_key:
                u = d
            elif 'epi' in uncertainty_key:
                u = d[..., ucty_idx, ucty_idx]
            else:
                u = d[..., ucty_idx]
            u = u.cpu().numpy()
            for p in range(u.shape[-1]):
                grids.append(color_map(img_data, u[..., p:p + 1], l.downsample, 0, None))
        return grids

    def make_color_map(self, filename, config, kinds=None):
        import torch
        img_data = self.load_img(filename)
        self.model.run(torch.from_numpy(img_data).cuda(), seed=int(config.get('seed', 0)), want_nms=False)
        for key, idx, name in (kinds or [(config['uncertainty_key'], config.get('ucty_idx', -1), config['ucty'])]):
            cfg = dict(config, ucty=name)
            save_uncertainty_maps(self.uncertainty_grids(img_data, key, idx), os.path.basename(filename), cfg)


def save_uncertainty_maps(grids, file_name, config):
    from PIL import Image
    file_name = os.path.basename(file_name)
    for idx, img in enumerate(grids):
        path = os.path.join(config['out_path'],
                            '{}_prior{}_{}.png'.format(os.path.splitext(file_name)[0], idx, config['ucty']))
        Image.fromarray(img).save(path)


def worker(files, config, kinds=None):
    os.makedirs(config['out_path'], exist_ok=True)
    yolo = yolov3.bayesian_yolov3_aleatoric(config)
    inference = Inference(yolo, config)
    for file in files:
        logging.info('Processing file: {}'.format(file))
        inference.make_color_map(file, config, kinds)


def do_it(files, config):
    # all 11 uncertainty kinds from ONE forward per image (the reference spawns one process per kind)
    worker(files, config, kinds=UCTY_KINDS)


def main():
    config = {
        'checkpoint_path': './checkpoints/',
        'run_id': 'epi_ale',  # edit
        'step': 'last',  # edit, int or 'last'
        'crop_img_size': [768, 1440, 3],
        'full_img_size': [1024, 1920, 3],  # edit if not ecp
        'cls_cnt': 2,
        'batch_size': 1,
        'T': 30,
        'inference_mode': True,
        'cpu_thread_cnt': 10,
        'freeze_darknet53': False,  # actual value irrelevant
        'crop': False,  # edit
        'training': False,
        'aleatoric_loss': True,  # actual value irrelevant
        'priors': yolov3.ECP_9_PRIORS,  # actual value irrelevant
        'out_path': './uncertainty_visualization',  # edit
    }
    # NOTE: only works for bayesian_yolov3_aleatoric class
    assert config['batch_size'] == 1
    assert config['inference_mode']
    files = glob.glob('./test_images/*')  # edit
    logging.info('----- START -----')
    start = time.time()
    do_it(files, config)
    elapsed = int(time.time() - start)
    logging.info('----- FINISHED in {:02d}:{:02d}:{:02d} -----'.format(elapsed // 3600, (elapsed // 60) % 60, elapsed % 60))


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO,
                        format='%(asctime)s, pid: %(process)d, %(levelname)-8s %(message)s',
                        datefmt='%a, %d %b %Y %H:%M:%S')
    main()
