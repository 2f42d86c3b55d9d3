"""Shared machinery of the entry points `inference_{standard_yolov3,aleatoric,epistemic}.py` and
`detect.py`: checkpoint restore, the driver loop with its one-deep asynchronous JSON writer
(`inference_epistemic.py:56-92`), ECP-JSON row mapping (`:131-170` and the two siblings, including
their index quirks -- SURVEY.md App. D 6/7), eager `concat_bbox` / `nms` helpers on device tensors.
"""
import glob
import json
import logging
import os

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


def _is_stock(fn, variant):
    """True iff `fn` is an entry script's unedited `bbox_to_ecp_format`: its body is exactly the delegate into this module."""
    import inspect
    import sys
    try:
        src = inspect.getsource(fn)
    except (OSError, TypeError):
        return False
    body = [l.strip() for l in src.splitlines()[1:] if l.strip() and not l.strip().startswith('#')]
    g = getattr(fn, '__globals__', {})
    return (body == ['return _inf.bbox_to_ecp_format(bbox, img_size, model, config, VARIANT)'] and g.get('VARIANT') == variant
            and g.get('_inf') is sys.modules[__name__])


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
class _Slot:
    """Everything one batch in flight owns: a HIP stream, the uint8 / float32 image buffers, the packed send buffer
    [rows | kept | count | status words] the forward writes its outputs INTO (so the all-gather needs no packing pass), the
    gathered buffer, its pinned host mirror and the event that says the mirror is complete.  Workspace arena `ws_slot`."""

    def __init__(self, loop, k, bl, world):
        import torch
        dev, cuda = loop.dev, loop.cuda
        h, w, c = loop.img_size
        cap, D = loop.cap, loop.D
        self.ws_slot = 1 + k
        self.stream = torch.cuda.Stream(device=dev) if cuda else None
        self.event = torch.cuda.Event() if cuda else None
        self.u8 = torch.empty((bl, h, w, c), dtype=torch.uint8, device=dev)
        self.img = torch.empty((bl, h, w, c), dtype=torch.float32, device=dev)
        self.words = bl * cap * D + bl * cap + bl * 2 + 2
        self.send = torch.zeros(self.words, dtype=torch.float32, device=dev)
        self.recv = torch.empty(world * self.words, dtype=torch.float32, device=dev) if loop.pg else self.send
        self.host = torch.empty(self.recv.numel(), dtype=torch.float32, pin_memory=cuda)

    def views(self, t, bl, cap, D):
        """(rows [bl,cap,D] f32, kept [bl,cap] i32, count [bl,2] i32, status [2] i32) of one rank's packed buffer `t`."""
        import torch
        n_r, n_k = bl * cap * D, bl * cap
        return (t[:n_r].view(bl, cap, D), t[n_r:n_r + n_k].view(torch.int32).view(bl, cap),
                t[n_r + n_k:n_r + n_k + 2 * bl].view(torch.int32).view(bl, 2), t[n_r + n_k + 2 * bl:n_r + n_k + 2 * bl + 2].view(torch.int32))


class InferenceLoop:
    """`Inference` of the three scripts (`inference_epistemic.py:40-92`): dataset -> model.run -> asynchronous ECP-JSON writer.
    Build extensions (all optional config keys): `weights='synthetic'` (random-init + device BN calibration instead of a
    checkpoint), `seed`, `engine_options`, `writer_threads` (default 4), `data.prefetch` (default 2).

    The reference overlaps three things: tf.data decodes (`cpu_thread_cnt` threads) and prefetches one batch while the session
    runs, and one writer thread dumps the previous batch's JSON.  Same stages here, sized for a device that finishes a
    608 x 608, T=30 image in 3 ms:

      feed     lib_yolo.dataset_utils.TestingDataset.iter_shards_u8: native decode pool + prefetch, uint8 frames in pinned memory;
      device   TWO batches in flight, each on its own HIP stream / workspace arena (`_Slot`): H2D of the bytes, * (1/255) on the
               device, the forward with no host wait inside (byolo_set_async), the all-gather, D2H of the box list into pinned
               memory, an event.  The loop enqueues batch i, then waits for batch i-1's event: the device always has work queued;
      writer   `writer_threads` threads, one image per task: the JSON text comes from the native formatter (byolo_format_ecp_json,
               byte-identical to json.dump of `to_ecp`'s dicts, no GIL held) when `to_ecp` is the script's stock function, from
               json.dumps otherwise.

    Multi-GPU (`torchrun --nproc-per-node N inference_epistemic.py`, one process per GPU; the reference pins one device,
    `inference_epistemic.py:57`): `batch_size` stays the GLOBAL batch.  Rank r builds its engine on cuda:LOCAL_RANK, decodes and
    runs its contiguous block of every global batch with `first_image` = the block's position (the N-GPU job draws the dropout
    masks one GPU would draw on the whole batch), ONE all-gather per batch (RCCL over xGMI) assembles the final box list -- and
    every rank's range status -- on every rank, and every rank writes the files of ITS images (rank 0 creates the directory).

    BYOLO_ERR_RANGE (an activation beyond what split-f16 holds): all ranks read the same gathered status words, so they act
    TOGETHER: the offending global batch -- that batch only -- is re-run in the fp32 mode on every rank (on a second handle that
    holds the same parameters packed for fp32: both packs stay resident, nothing is re-packed in mid-stream), the batches that
    were in flight behind it are re-run in the default precision (the status word is sticky: their own verdict was lost), and the
    run carries on in split-f16.  `stats['precision_switches']` counts both directions (two per such batch), `stats['fp32_batches']`
    lists the batches; one JSON file per image regardless (`inference_epistemic.py:84-92`).  (Round 4 switched the whole job to
    fp32 -- 0.49 x the throughput -- for the rest of the run.)

    A feed failure on ONE rank (a corrupt record, a bad PNG in its block; every rank reads only its own records): the rank sends
    a 'feed failed' status word in the batch's all-gather instead of rows, every rank sees it and all of them raise together
     -- none is left waiting in a collective for a peer that is gone."""


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
            yolo.set_engine_option('device', bdist.local_device(self.local_rank))

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
        self.stats = {}

    def _load_weights(self):
        import torch
        if self.config.get('weights') == 'synthetic':
            from byolo import synth
            eng = self.model.engine
            eng.set_params(synth.base_params(eng.param_shapes(), self.variant, self.model.cls_cnt, seed=7))
            eng.finalize()
            h, w, c = self.img_size
            # the same calibration frames on every rank -> identical weights
            eng.calibrate_bn(torch.from_numpy(synth.synthetic_images(2, h, w, c, seed=999)).to(eng.torch_device))
        else:
            restore(self.model, self.checkpoint)

    # ---- one batch: enqueue everything, wait for nothing ----------------------------------------------------------
    FEED_FAILED = 1 << 30            # status flag of a rank whose feed raised: travels in the batch's all-gather like the range flags

    def _enqueue(self, shard, step, slot, precision=None):
        """A shard with `error` set: this rank's feed failed at this batch -- it still takes part in the batch's collective, with the
        FEED_FAILED word instead of rows; every rank (this one included) raises when it retires the batch."""
        import contextlib
        import torch
        from byolo import dist as bdist
        eng = self.model.engine
        failed = getattr(shard, 'error', None) is not None
        n_loc = 0 if failed else int(shard.u8.shape[0])
        bl = bdist.padded_block(shard.n_global, self.world)       # images per rank in the gathered buffer
        words = bl * self.cap * self.D + bl * self.cap + bl * 2 + 2
        send = slot.send[:words]
        rows, kept, count, status = slot.views(send, bl, self.cap, self.D)
        with (torch.cuda.stream(slot.stream) if self.cuda else contextlib.nullcontext()):
            ran = eng
            if n_loc:
                slot.u8[:n_loc].copy_(torch.from_numpy(shard.u8), non_blocking=True)          # pinned -> device, 1 byte per value
                x = eng.normalize_u8(slot.u8[:n_loc], out=slot.img[:n_loc])                   # decode_img's * (1/255), on the device
                kw = {} if precision is None else {'precision': precision}
                res = self.model.run(x, seed=self.seed + step, want_boxes=False, first_image=shard.lo, slot=slot.ws_slot,
                                     out={'rows': rows[:n_loc], 'kept': kept[:n_loc], 'count': count[:n_loc]}, **kw)
                ran = (res or {}).get('engine') or eng
            if n_loc < bl:
                count[n_loc:].zero_()                             # padding images of a short block: nothing kept
            if failed:
                status[0] = self.FEED_FAILED; status[1] = -1
            else:
                ran.copy_status(status)                           # this rank's range status rides in the same buffer
            recv = send
            if self.pg:                                           # ONE collective per global batch
                recv = slot.recv[:self.world * words]
                bdist.all_gather_flat(recv, send)
            host = slot.host[:recv.numel()]
            host.copy_(recv, non_blocking=True)
            if self.cuda:
                slot.event.record(slot.stream)
        return dict(shard=shard, step=step, slot=slot, bl=bl, words=words, n_loc=n_loc, host=host, precision=precision)

    # ---- ... and its completion: the only place the host waits for the device ---------------------------------------
    def _complete(self, job):
        """Waits for the batch; returns False if some rank's forward left the split-f16 range (the rows are then not used)."""
        import time
        t0 = time.perf_counter()
        if self.cuda:
            job['slot'].event.synchronize()
        self.stats['wait_device_s'] += time.perf_counter() - t0
        slot, bl, words = job['slot'], job['bl'], job['words']
        per_rank = job['host'].view(-1, words)
        for r in range(per_rank.shape[0]):
            flags = int(slot.views(per_rank[r], bl, self.cap, self.D)[3][0])
            if flags & self.FEED_FAILED:
                err = getattr(job['shard'], 'error', None)
                if err is not None:
                    raise err                                     # this rank's own feed error, at the batch it belongs to
                raise RuntimeError('rank %d could not read its records of batch %d (see that rank\'s error): all ranks stop' % (r, job['step']))
            if flags:
                self._range_rank = r
                return False
        rows, _, count, _ = slot.views(per_rank[self.rank if self.pg else 0], bl, self.cap, self.D)
        counts = count[:job['n_loc'], 0].tolist()
        # copies: the pinned mirror is overwritten two batches from now, the writer may be slower than that
        boxes = [rows[j, :n].numpy().copy() for j, n in enumerate(counts)]
        job['shard'].release()                                    # the frames have left the feed's buffer
        self._write_async(boxes, job['shard'].names)
        self.stats['images'] += job['n_loc']
        self.stats['steady'].append((time.perf_counter(), self.stats['images']))
        return True

    def _redo_out_of_range(self, jobs):
        """Every rank runs this on the same batch (they all read the same gathered status words).  jobs[0] is the batch whose
        status words came back raised; the others were enqueued behind it."""
        import torch
        eng = self.model.engine
        logging.warning('rank %d reported BYOLO_ERR_RANGE in batch %d: every rank re-runs THAT batch in the fp32 mode; the run stays in %s',
                        self._range_rank, jobs[0]['step'], getattr(eng, 'precision', '?'))
        if self.cuda:
            for j in jobs:                                        # whatever is still in flight ran with the sticky words raised
                j['slot'].event.synchronize()
        eng.clear_status()                                        # on torch's current stream ...
        if self.cuda:
            torch.cuda.synchronize(self.dev)                      # ... while the re-runs' copy_status sits on the slots' streams
        self.stats['precision_switches'] += 1                     # -> fp32
        # the fp32 twin handle (a second weight pack + workspace arena) is built HERE, by every rank at the same point, and the ranks
        # agree that it exists before any of them enters the re-run's collective: a rank whose allocation fails would otherwise raise
        # alone and leave its peers waiting in the all-gather
        from byolo import dist as bdist
        err = None
        try:
            if hasattr(eng, 'twin'):
                eng.twin('f32')
        except Exception as e:
            err = e
        err = bdist.agree_on_any_error(err)
        if err is not None:
            raise err
        redo = self._enqueue(jobs[0]['shard'], jobs[0]['step'], jobs[0]['slot'], precision='f32')
        if not self._complete(redo):
            raise RuntimeError('BYOLO_ERR_RANGE in the fp32 mode: a raw detection output is inf / NaN (batch %d)' % jobs[0]['step'])
        self.stats['precision_switches'] += 1                     # -> back to the default precision
        self.stats['fp32_batches'].append(jobs[0]['step'])
        for j in jobs[1:]:
            redo = self._enqueue(j['shard'], j['step'], j['slot'])
            if not self._complete(redo):
                self._redo_out_of_range([redo])

    # ---- the T axis sharded over the ranks: the latency path at the reference's batch_size = 1 -----------------------------
    def _run_t_sharded(self):
        """config['shard'] = 'T' (the alternative).  The reference's own default workload is ONE
        image per step with T = 50 MC samples (inference_epistemic.py:193, :220-221): sharding the batch axis gives rank 0 the image
        and the other N - 1 GPUs nothing.  Here EVERY rank reads every frame, runs the backbone on it and the heads on ITS share of
        the T samples (byolo.dist.shard_range(T, rank, world): samples t0 .. t1 - 1, drawing exactly their masks of the image's T),
        and hands out the per-box sums of what lib_yolo/layers.py:377-395 averages (21 + C floats per box: sum l, the upper triangle
        of sum l l^T, sum e^logvar, sum sigma(obj), sum H(obj), sum softmax, sum H(cls)); ONE all-reduce per image adds them (RCCL over
        xGMI; 2.1 MB at 608 x 608, 11 MB at 1024 x 1920 -- it replaces the batch path's all-gather), every rank finishes the rows
        (byolo_finish_tshard) and runs the NMS, and image i's file is written by rank i % world.  Summation order differs from the
        one-GPU reduction: rows agree within float32 rounding (the contract's 1e-4), kept indices where scores are not tied.
        A forward beyond the split-f16 range on ANY rank is seen by all (a flag word rides in the all-reduced buffer): that image is
        re-run in fp32 everywhere.  Synchronous per image: this is the latency mode, throughput is the batch path's business."""
        import time
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from byolo import dist as bdist
        rank, _, world = bdist.init()
        self.pg = torch.distributed.is_initialized()
        self._load_weights()
        eng = self.model.engine
        was_async = getattr(eng, '_async', False)
        eng.set_async(True)
        self.dev = eng.torch_device
        self.cuda = str(self.dev).startswith('cuda')
        N, D = eng.num_boxes()
        T = self.model.T
        t0, t1 = bdist.shard_range(T, rank, world)              # this rank's samples (empty when there are more ranks than samples)
        self.seed = int(self.config.get('seed', 0))
        self.stats = dict(images=0, batches=0, wait_feed_s=0.0, wait_device_s=0.0, wait_writer_s=0.0, precision_switches=0, fp32_batches=[],
                          native_json=False, device=getattr(eng, 'device', None), rank=rank, world=world, shard='T', samples=[t0, t1],
                          latency_ms=[])
        self._writer_setup()
        h, w, c = self.img_size
        buf = torch.zeros(N * D + 2, dtype=torch.float32, device=self.dev)          # [sums | range flag | feed flag]: ONE collective
        st_words = torch.zeros(2, dtype=torch.int32, device=self.dev)
        t_start = time.perf_counter()
        feed = self.dataset.iter_shards_u8(0, 1, errors='yield')                    # every rank reads every record
        step = n_img = 0

        def one(x, j, precision):
            """Image j of the batch: this rank's sums (+ flags) -> all-reduce -> rows, kept rows; returns (kept rows on the host or None, flag)."""
            buf.zero_()
            ran = eng
            if t1 > t0 and x is not None:
                kw = {} if precision is None else {'precision': precision}
                res = self.model.run(x, seed=self.seed + step, want_boxes=True, want_nms=False, first_image=j, t_shard=(t0, t1),
                                     out={'boxes': buf[:N * D].view(1, N, D)}, **kw)          # the sums land in the collective's buffer
                if res['boxes'].data_ptr() != buf.data_ptr():                                 # (a stand-in engine that returns its own tensor)
                    buf[:N * D].copy_(res['boxes'].reshape(-1))
                ran = (res or {}).get('engine') or eng
            if x is None:
                buf[N * D + 1] = 1.0                                                 # this rank could not read the frame
            else:
                ran.copy_status(st_words)
                buf[N * D] = (st_words[0] != 0).to(torch.float32)
            if self.pg:
                bdist.all_reduce_flat(buf)
            flags = buf[N * D:].cpu().tolist()                                       # the one host wait per image
            if flags[0] or flags[1]:
                return None, flags
            rows = eng.finish_tshard(buf[:N * D].view(1, N, D), T)
            res = eng.sort_nms(rows, self.model.obj_idx, self.model.cls_start_idx)
            if n_img % world != rank:
                return False, flags
            n = int(res['count'][0, 0])
            return res['rows'][0, :n].cpu().numpy(), flags

        with ThreadPoolExecutor(max_workers=self.writer_threads, thread_name_prefix='byolo-writer') as self._pool:
            try:
                for shard in feed:
                    step += 1
                    err = getattr(shard, 'error', None)
                    n = self.batch_size if err is not None else int(shard.u8.shape[0])
                    xs = None if err is not None else self.model.engine.normalize_u8(torch.from_numpy(shard.u8).to(self.dev))
                    for j in range(n):
                        t_img = time.perf_counter()
                        x = None if xs is None else xs[j:j + 1]
                        kept, flags = one(x, j, None)
                        if flags[1]:
                            raise err if err is not None else RuntimeError('a rank could not read its records of batch %d: all ranks stop' % step)
                        if flags[0]:                                                 # some rank left the split-f16 range: THIS image in fp32, everywhere
                            logging.warning('BYOLO_ERR_RANGE in batch %d, image %d: every rank re-runs it in the fp32 mode', step, j)
                            eng.clear_status()
                            if self.cuda:
                                torch.cuda.synchronize(self.dev)
                            self.stats['precision_switches'] += 2
                            self.stats['fp32_batches'].append(step)
                            kept, flags = one(x, j, 'f32')
                            if flags[0]:
                                raise RuntimeError('BYOLO_ERR_RANGE in the fp32 mode: a raw detection output is inf / NaN (batch %d)' % step)
                        if kept is not False:
                            self._write_async([kept], [shard.names[j]])
                            self.stats['images'] += 1
                        n_img += 1
                        self.stats['latency_ms'].append(1e3 * (time.perf_counter() - t_img))
                    shard.release()
            finally:
                feed.close()
                eng.set_async(was_async)
            self._writer_drain(0)
        self.stats['batches'] = step
        self.stats['loop_s'] = time.perf_counter() - t_start
        self.stats['precision'] = getattr(eng, 'precision', None)
        lat = sorted(self.stats.pop('latency_ms'))
        if lat:
            self.stats['latency_ms_median'] = lat[len(lat) // 2]
            self.stats['latency_ms_max'] = lat[-1]
        if self.pg:
            torch.distributed.barrier()
        return self.stats

    def run(self):
        import collections
        import time
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from byolo import dist as bdist
        if self.config.get('shard') == 'T':
            return self._run_t_sharded()
        rank, _, world = bdist.init()
        self.pg = torch.distributed.is_initialized()      # world > 1, or a forced one-rank group (BYOLO_DIST_FORCE=1)
        self._load_weights()
        eng = self.model.engine
        was_async = getattr(eng, '_async', False)
        eng.set_async(True)                               # no host wait inside forward(): the status words travel with the rows
        self.dev = eng.torch_device                       # (a CPU stand-in engine in the gloo tests names its own)
        self.cuda = str(self.dev).startswith('cuda')
        _, self.D = eng.num_boxes()
        self.cap = eng.out_cap
        self.seed = int(self.config.get('seed', 0))
        self.stats = dict(images=0, batches=0, wait_feed_s=0.0, wait_device_s=0.0, wait_writer_s=0.0, precision_switches=0,
                          fp32_batches=[], steady=[], native_json=False, device=getattr(eng, 'device', None), rank=rank, world=world)
        bl_max = bdist.padded_block(self.batch_size, world)
        slots = [_Slot(self, k, bl_max, world if self.pg else 1) for k in range(2)]
        self._writer_setup()

        def pinned(shape):
            return torch.empty(shape, dtype=torch.uint8, pin_memory=self.cuda).numpy()

        inflight = collections.deque()
        t_start = time.perf_counter()
        feed = self.dataset.iter_shards_u8(rank, world, alloc=pinned, extra_buffers=len(slots) + 1, errors='yield')
        step = 0
        with ThreadPoolExecutor(max_workers=self.writer_threads, thread_name_prefix='byolo-writer') as self._pool:
            try:
                while True:
                    t0 = time.perf_counter()
                    # (a block this rank cannot read / check / decode arrives as a Shard with `error` set: the rank takes part in
                    # that batch's all-gather with the FEED_FAILED word, and EVERY rank raises when it retires the batch -- ADVICE r4)
                    shard = next(feed, None)              # ends like tf.errors.OutOfRangeError
                    self.stats['wait_feed_s'] += time.perf_counter() - t0
                    if shard is None:
                        break
                    step += 1
                    inflight.append(self._enqueue(shard, step, slots[step % len(slots)]))
                    if len(inflight) == len(slots):
                        self._retire(inflight)
                    if step % 15 == 0:
                        logging.info('Processed {} images.'.format(step * self.batch_size))
                while inflight:
                    self._retire(inflight)
            finally:
                feed.close()
                eng.set_async(was_async)                  # a later Model.run() on this engine checks its status again
            self._writer_drain(0)
        self.stats['batches'] = step
        self.stats['loop_s'] = time.perf_counter() - t_start
        self.stats['precision'] = getattr(eng, 'precision', None)
        # steady-state rate: images completed after the pipeline's fill (the first quarter of the batches, at least two) / the time
        # they took -- the loop's wall time also holds the fill: the first decode, the first H2D, the first forward's plan
        pts = self.stats.pop('steady')
        k = min(len(pts) - 1, max(2, len(pts) // 4))
        if k >= 1 and len(pts) > k and pts[-1][0] > pts[k][0]:
            self.stats['steady_img_s'] = (pts[-1][1] - pts[k][1]) / (pts[-1][0] - pts[k][0])
            self.stats['steady_images'] = pts[-1][1] - pts[k][1]
        logging.info('Processed {} batches.'.format(step))
        if self.pg:
            torch.distributed.barrier()                   # the files exist when any rank returns
        return self.stats

    def _retire(self, inflight):
        if self._complete(inflight[0]):
            inflight.popleft()
            return
        jobs = list(inflight)
        inflight.clear()
        self._redo_out_of_range(jobs)

    # ---- writer ---------------------------------------------------------------------------------------------------------
    def _writer_setup(self):
        import collections
        self.writer_threads = max(1, int(self.config.get('writer_threads', 4)))
        self._pending = collections.deque()
        self._formatter = None
        # the native formatter writes what the SCRIPT'S OWN, UNEDITED bbox_to_ecp_format returns.  The reference invites users to
        # edit that function (inference_epistemic.py:131 ff.: extra fields, other keys): `_is_stock` reads the function's SOURCE
        # -- the stock one is a one-line delegate to this module -- so an edited or replaced function sends its dicts through
        # json.dumps (the check used to compare the script's attribute with itself and was always true).
        # config['native_json'] = False forces json.dumps as well.
        if _is_stock(self.to_ecp, self.variant) and self.config.get('native_json', True):
            try:
                from byolo import hostio
                self._formatter = hostio.EcpJsonFormatter(self.variant, self.img_size, self.model.cls_cnt, self.model.obj_idx,
                                                          self.model.cls_start_idx, self.config['implicit_background_class'],
                                                          LABEL_TO_CLS_NAME)
            except ValueError as e:                       # a label table the native formatter does not write
                logging.info('ECP JSON through json.dumps: %s', e)
        self.stats['native_json'] = self._formatter is not None
        logging.info('ECP JSON writer: %s', 'native formatter (byolo_format_ecp_json)' if self._formatter is not None else 'json.dumps of to_ecp dicts')

    def _write_async(self, boxes, files):
        for bxs, filename in zip(boxes, files):
            self._pending.append(self._pool.submit(self.write_ecp_json, bxs, filename))
        self._writer_drain(16 * self.writer_threads)      # back-pressure: at most this many images wait for the writer

    def _writer_drain(self, keep):
        import time
        t0 = time.perf_counter()
        while len(self._pending) > keep:
            self._pending.popleft().result()              # re-raises a writer's exception
        self.stats['wait_writer_s'] += time.perf_counter() - t0

    def write_to_disc(self, boxes, files):
        for bxs, filename in zip(boxes, files):
            self.write_ecp_json(bxs, filename)

    def write_ecp_json(self, boxes, img_name):
        out_name = '{}.json'.format(os.path.splitext(os.path.basename(img_name))[0])
        out_file = os.path.join(self.out_path, out_name)
        formatter = getattr(self, '_formatter', None)
        if formatter is not None:
            text = formatter.format(boxes)
        else:                                             # json.dump(..., f) writes exactly these characters
            text = json.dumps({
                'children': [self.to_ecp(bbox, self.img_size, self.model, self.config) for bbox in boxes],
            }, default=lambda x: x.tolist()).encode('ascii')
        with open(out_file, 'wb') as f:
            f.write(text)
