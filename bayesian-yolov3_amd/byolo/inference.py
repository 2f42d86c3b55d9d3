"""Shared machinery of the entry points `inference_{standard_yolov3,aleatoric,epistemic}.py` and
`detect.py`: checkpoint restore, the driver loop with its one-deep asynchronous JSON writer
(`inference_epistemic.py:56-92`), ECP-JSON row mapping (`:131-170` and the two siblings, including
their index quirks -- SURVEY.md App. D 6/7), eager `concat_bbox` / `nms` helpers on device tensors.
"""
import glob
import json
import logging
import os
import threading

import numpy as np

LABEL_TO_CLS_NAME = {1: 'pedestrian', 2: 'rider'}      # edit if not ECP dataset (starts at 0 without implicit background)


# ---------------------------------------------------------------------------------------------------
# helpers the three scripts re-export under the reference's names
# ---------------------------------------------------------------------------------------------------
def concat_bbox(net_out, batched):
    """`inference_epistemic.py:173-184` / `inference_aleatoric.py:181-192`: flatten the per-prior box
    tensors in the order layer (stride 32,16,8) -> prior -> row -> col.  Works on torch tensors
    ([B,lh,lw,D] per prior, as `DetLayer.bbox` returns them); `batched=False` drops the batch axis of a
    batch-1 result like the epistemic reference."""
    import torch
    parts = []
    for det_layer in net_out:
        for prior in det_layer:
            parts.append(prior.reshape(prior.shape[0], -1, prior.shape[-1]))
    bbox = torch.cat(parts, dim=1).contiguous()
    if not batched:
        assert bbox.shape[0] == 1, 'the un-batched concat is defined for batch 1 (inference_epistemic.py:193)'
        return bbox[0]
    return bbox


def nms(boxes, model, batched, two_class=False, max_out=1000):
    """`tf.image.non_max_suppression(boxes[:, :4], boxes[:, model.obj_idx], 1000)` + `tf.gather`
    (`inference_epistemic.py:99-102`; 2-class variant `:104-126`) through byolo_sort_nms.
    Returns the kept rows: [k, D] (un-batched) or a list of per-image [k_i, D] tensors.  (The
    reference's batched version concatenates per-image results, which only works when every image
    keeps the same number of boxes -- App. D.8 -- so a list is returned instead.)"""
    b = boxes if boxes.dim() == 3 else boxes[None]
    res = model.engine.sort_nms(b.contiguous(), model.obj_idx, model.cls_start_idx,
                                nms_mode=1 if two_class else 0, max_out=max_out)
    counts = res['count'][:, 0].cpu().tolist()
    rows = [res['rows'][i, :n] for i, n in enumerate(counts)]
    return rows if batched else rows[0]


def bbox_to_ecp_format(bbox, img_size, model, config, variant):
    """Row -> ECP dict.  Coordinates are scaled in float32 and only then converted to Python floats,
    `score = obj * cls[argmax]`, `identity` via the label table (+1 with an implicit background
    class).  Quirks reproduced on purpose: the aleatoric script reads `cls_entropy`, `layer_id` and
    `prior_id` all from column cls_start+C (`inference_aleatoric.py:174-176`); the epistemic script
    hard-codes `ped_score`/`rider_score` to columns 17/18 (`inference_epistemic.py:163-164`)."""
    img_height, img_width = img_size[:2]
    cs, C, obj = model.cls_start_idx, model.cls_cnt, model.obj_idx
    cls_scores = bbox[cs:cs + C]
    cls = np.argmax(cls_scores)
    cls_idx = cls
    if config['implicit_background_class']:
        cls += 1
    out = {
        'y0': float(bbox[0] * img_height),
        'x0': float(bbox[1] * img_width),
        'y1': float(bbox[2] * img_height),
        'x1': float(bbox[3] * img_width),
    }
    score = float(bbox[obj]) * float(bbox[cs + cls_idx])
    if variant == 'yolov3':
        out.update({'score': score, 'cls_scores': cls_scores})
    elif variant == 'yolov3_aleatoric':
        for i, k in enumerate(('x_var', 'y_var', 'w_var', 'h_var', 'total_var')):
            out[k] = float(bbox[4 + i])
        out.update({'score': score, 'obj_entropy': float(bbox[obj + 1]), 'cls_scores': cls_scores,
                    'cls_entropy': float(bbox[cs + C]), 'layer_id': float(bbox[cs + C]), 'prior_id': float(bbox[cs + C])})
    else:
        for i, k in enumerate(('x_var_epi', 'y_var_epi', 'w_var_epi', 'h_var_epi', 'x_var_ale', 'y_var_ale',
                               'w_var_ale', 'h_var_ale', 'total_var_epi', 'total_var_ale')):
            out[k] = float(bbox[4 + i])
        out.update({'score': score, 'obj_mutual_info': float(bbox[obj + 1]), 'obj_entropy': float(bbox[obj + 2]),
                    'cls_scores': cls_scores, 'ped_score': float(bbox[17]), 'rider_score': float(bbox[18]),
                    'cls_mutual_info': float(bbox[cs + C]), 'cls_entropy': float(bbox[cs + C + 1]),
                    'layer_id': float(bbox[cs + C + 2]), 'prior_id': float(bbox[cs + C + 3])})
    out['identity'] = LABEL_TO_CLS_NAME.get(cls, cls)
    return out


# ---------------------------------------------------------------------------------------------------
# checkpoints
# ---------------------------------------------------------------------------------------------------
def find_checkpoint(config):
    """`inference_epistemic.py:27-38`: <checkpoint_path>/<run_id>, step 'last' or an explicit step.
    Accepted formats: a TF checkpoint prefix (`model-<step>.index` + `.data-*`, read by
    byolo.tf_checkpoint), or `model-<step>.npz` holding the same variables by TF name."""
    ckpt_dir = os.path.join(config['checkpoint_path'], config['run_id'])
    cands = []
    for f in sorted(glob.glob(os.path.join(ckpt_dir, '*'))):
        base, ext = os.path.splitext(f)
        if ext in ('.index', '.npz') and '-' in os.path.basename(base):
            cands.append((int(base.rsplit('-', 1)[1]) if base.rsplit('-', 1)[1].isdigit() else -1, f))
    if config['step'] == 'last':
        state = os.path.join(ckpt_dir, 'checkpoint')         # tf.train.latest_checkpoint reads this file
        if os.path.exists(state):
            for line in open(state):
                if line.startswith('model_checkpoint_path:'):
                    name = line.split(':', 1)[1].strip().strip('"')
                    for ext in ('.index', '.npz'):
                        p = os.path.join(ckpt_dir, os.path.basename(name) + ext)
                        if os.path.exists(p):
                            return p
        checkpoint = max(cands)[1] if cands else None
    else:
        checkpoint = None
        for step, f in cands:
            if step == int(config['step']):
                checkpoint = f
                break
    assert checkpoint is not None, 'could not find checkpoint'
    return checkpoint


def restore(model, checkpoint):
    """tf.train.Saver().restore(sess, checkpoint) (`inference_epistemic.py:58`)."""
    if checkpoint.endswith('.npz'):
        data = np.load(checkpoint)
        params = {k: data[k] for k in data.files}
    else:
        from byolo import tf_checkpoint
        params = tf_checkpoint.read(os.path.splitext(checkpoint)[0])
    model.engine.set_params(params, strict=True)
    model.finalize()


def step_of(checkpoint):
    return os.path.splitext(os.path.basename(checkpoint))[0].split('-')[-1]


# ---------------------------------------------------------------------------------------------------
# the driver loop
# ---------------------------------------------------------------------------------------------------
class InferenceLoop:
    """`Inference` of the three scripts: dataset -> model.run -> one-deep asynchronous ECP-JSON writer.
    Build extensions (all optional config keys): `weights='synthetic'` (random-init + device BN
    calibration instead of a checkpoint), `seed`, `nms_mode`.

    Multi-GPU (`torchrun --nproc-per-node N inference_epistemic.py`, one process per GPU; the reference pins one
    device, `inference_epistemic.py:57`): `batch_size` stays the GLOBAL batch.  Rank r builds its engine on
    cuda:LOCAL_RANK, decodes and runs its contiguous block of every global batch with `first_image` = the block's
    position (the N-GPU job draws the dropout masks one GPU would draw on the whole batch), ONE all-gather
    (byolo.dist.allgather_boxes: RCCL over xGMI) assembles the final box list, and rank 0 alone creates the output
    directory and writes the JSON files.  Ranks other than 0 never copy results to the host."""

    def __init__(self, yolo, config, variant, to_ecp, batched):
        from lib_yolo import dataset_utils
        from byolo import dist as bdist
        self.batch_size = config['batch_size']
        self.variant = variant
        self.to_ecp = to_ecp
        self.batched = batched
        self.config = config
        self.img_size = config['full_img_size']
        assert not config['crop']
        # the process group exists BEFORE any rank-specific side effect: a rank that fails below (rank 0 refusing an
        # existing output directory) tells the others, and all of them stop together instead of waiting for a dead peer
        self.rank, self.local_rank, self.world = bdist.init()
        if self.world > 1:                                # one engine per process, on this process's GPU
            yolo.set_engine_option('device', self.local_rank)

        self.dataset = dataset_utils.TestingDataset(config)
        self.model = yolo.init_model(inputs=self.dataset.placeholder, training=False).get_model()
        self.device = self.model.engine.device
        if config.get('weights') == 'synthetic':
            self.checkpoint = 'synthetic-0'
        else:
            self.checkpoint = find_checkpoint(config)
        self.out_path = '{}_{}'.format(config['out_path'], step_of(self.checkpoint))
        err = None
        if self.rank == 0:
            try:
                os.makedirs(self.out_path)                # like the reference: refuses to overwrite an existing run
            except OSError as e:
                err = e
        err = bdist.agree_on_error(err)                   # every rank learns of rank 0's failure (one broadcast)
        if err is not None:
            raise err
        self.worker_thread = None

    def _load_weights(self):
        import torch
        if self.config.get('weights') == 'synthetic':
            from byolo import synth
            eng = self.model.engine
            eng.set_params(synth.base_params(eng.param_shapes(), self.variant, self.model.cls_cnt, seed=7))
            eng.finalize()
            h, w, c = self.img_size
            # the same calibration frames on every rank -> identical weights
            eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(2, h, w, c, seed=999)).to(getattr(eng, 'torch_device', 'cuda:%d' % self.device)))
        else:
            restore(self.model, self.checkpoint)

    def run(self):
        import torch
        from byolo import dist as bdist
        rank, _, world = bdist.init()
        pg = torch.distributed.is_initialized()           # world > 1, or a forced one-rank group (BYOLO_DIST_FORCE=1)
        self._load_weights()
        eng = self.model.engine
        dev = getattr(eng, 'torch_device', 'cuda:%d' % self.device)      # (a CPU stand-in engine in the gloo tests names its own)
        _, D = eng.num_boxes()
        cap = eng.out_cap
        step = 0
        seed = int(self.config.get('seed', 0))
        for imgs, files, lo in self.dataset.iter_shards(rank, world):     # ends like tf.errors.OutOfRangeError
            step += 1
            n_glob, n_loc = len(files), int(imgs.shape[0])
            bl = bdist.padded_block(n_glob, world)        # images per rank in the gathered buffer
            out = {'rows': torch.zeros((bl, cap, D), dtype=torch.float32, device=dev),
                   'kept': torch.full((bl, cap), -1, dtype=torch.int32, device=dev),
                   'count': torch.zeros((bl, 2), dtype=torch.int32, device=dev)}
            if n_loc:
                x = torch.from_numpy(imgs).to(dev)
                self.model.run(x, seed=seed + step, want_boxes=False, first_image=lo,
                               out={k: v[:n_loc] for k, v in out.items()})
            g = (out['rows'], out['kept'], out['count'])
            if pg:                                        # ONE collective per global batch
                g = bdist.allgather_boxes(*g, world)
            if rank == 0:                                 # only the writer copies anything to the host
                boxes = [r.cpu().numpy() for r in bdist.unpack_global(*g, n_glob, world)[0]]
                if self.worker_thread:
                    self.worker_thread.join()
                self.worker_thread = threading.Thread(target=self.write_to_disc, args=(boxes, files))
                self.worker_thread.start()
            if step % 15 == 0:
                logging.info('Processed {} images.'.format(step * self.batch_size))
        logging.info('Processed {} batches.'.format(step))
        if self.worker_thread:
            self.worker_thread.join()
        if pg:
            torch.distributed.barrier()                   # the files exist when any rank returns

    def write_to_disc(self, boxes, files):
        for bxs, filename in zip(boxes, files):
            self.write_ecp_json(bxs, filename)

    def write_ecp_json(self, boxes, img_name):
        out_name = '{}.json'.format(os.path.splitext(os.path.basename(img_name))[0])
        out_file = os.path.join(self.out_path, out_name)
        with open(out_file, 'w') as f:
            json.dump({
                'children': [self.to_ecp(bbox, self.img_size, self.model, self.config) for bbox in boxes],
            }, f, default=lambda x: x.tolist())
