"""Engine: thin object wrapper over the C-ABI handle (include/byolo.h).

PyTorch-ROCm is used here only as plumbing: device buffers (workspace, outputs), the current HIP
stream and torch.distributed.  All compute happens inside libbyolo.so.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import lib, check, Cfg


def _torch():
    import torch
    return torch


class Engine:
    """One handle == one graph on one device (not thread-safe; one per device/thread)."""

    def __init__(self, img_size, cls_cnt, drop_prob=0.1, max_out=1000, iou_thresh=0.5,
                 nms_mode=_lib.NMS_AGNOSTIC, keep_all_outputs=False, device=0):
        h, w, c = [int(v) for v in img_size]
        self.cfg = Cfg(h, w, c, int(cls_cnt), float(drop_prob), int(max_out), float(iou_thresh),
                       int(nms_mode), int(bool(keep_all_outputs)))
        self.device = int(device)
        self._h = ctypes.c_void_p()
        rc = lib.byolo_create(ctypes.byref(self.cfg), self.device, ctypes.byref(self._h))
        if rc < 0:
            msg = lib.byolo_last_error(None)
            raise _lib.ByoloError(rc, msg.decode() if msg else "?")
        self._ws = None
        self._ws_key = None
        self.finalized = False
        self._graph = []                 # the builder calls in order: twin() replays them on a second handle
        self._twin = None

    # ---- arithmetic of the convolution stack (include/byolo.h: BYOLO_PREC_*) ----------------------------
    @property
    def precision(self):
        return "split" if check(self._h, lib.byolo_get_precision(self._h)) == 1 else "f32"

    def set_precision(self, precision):
        """'f32' (fp32 matrix instruction) or 'split' (hi + lo fp16 pairs, three fp16 products per fp32 product).
        Call before finalize()."""
        check(self._h, lib.byolo_set_precision(self._h, {"f32": 0, "split": 1}[precision]))
        self.finalized = False

    @property
    def precision_note(self):
        """Why finalize() chose the fp32 mode although split-f16 was asked for ('' if it did not)."""
        return lib.byolo_precision_note(self._h).decode()

    # ---- the plan of this handle (include/byolo.h byolo_plan_opts) --------------------------------------------
    def plan_opts(self):
        """The handle's plan options as a dict (defaults + the BYOLO_* environment at creation, or what set_plan_opts left)."""
        o = _lib.PlanOpts()
        check(self._h, lib.byolo_get_plan_opts(self._h, ctypes.byref(o)))
        return {n: getattr(o, n) for n, _ in _lib.PlanOpts._fields_ if n != "struct_bytes"}

    def set_plan_opts(self, **kw):
        """Change plan options of THIS handle (byolo_set_plan_opts): e.g. set_plan_opts(graphs=0, wino_split=2).  Options that change
        what finalize() packs (dedup, lowmain, kx3, p1, wshift_per_layer) un-finalize the handle: call finalize() again."""
        o = _lib.PlanOpts()
        check(self._h, lib.byolo_get_plan_opts(self._h, ctypes.byref(o)))
        for k, v in kw.items():
            if k == "struct_bytes" or not hasattr(o, k):
                raise KeyError("byolo_plan_opts has no field %r" % k)
            setattr(o, k, v)
        check(self._h, lib.byolo_set_plan_opts(self._h, ctypes.byref(o)))
        if any(k in kw for k in ("dedup", "lowmain", "kx3", "p1", "wshift_per_layer")):
            self.finalized = False

    def set_graphs(self, on):
        """Launch-graph replay of forwards that do not fill the chip: True = the default rule (1), False = never, 2 = every forward."""
        self.set_plan_opts(graphs=int(on) if on in (0, 1, 2) and not isinstance(on, bool) else (1 if on else 0))

    def graph_stats(self):
        n, r, c, u = ctypes.c_int32(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        check(self._h, lib.byolo_graph_stats(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(c), ctypes.byref(u)))
        return dict(graphs=n.value, replays=r.value, captures=c.value, updates=u.value)

    # ---- numeric status of the split-f16 mode (include/byolo.h: BYOLO_ERR_RANGE) -------------------------
    def set_async(self, on=True):
        """on: forward() neither waits for the stream nor checks the status words; the caller asks check_status() where
        it synchronises anyway.  Off (the default): forward() raises ByoloError(ERR_RANGE) itself."""
        check(self._h, lib.byolo_set_async(self._h, int(bool(on))))
        self._async = bool(on)

    def status(self):
        """(flags, layer) after waiting for the current stream: bit 0 = an activation left the split-f16 range in `layer`,
        bit 1 = a raw detection output is inf / NaN.  Does not raise."""
        torch = _torch()
        flags, layer = ctypes.c_uint32(), ctypes.c_int32()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        lib.byolo_status(self._h, ctypes.c_void_p(stream), ctypes.byref(flags), ctypes.byref(layer))
        return int(flags.value), int(layer.value)

    def check_status(self):
        """Wait for the current stream; raise ByoloError(ERR_RANGE) if a forward since the last clear left the range."""
        torch = _torch()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self._h, lib.byolo_status(self._h, ctypes.c_void_p(stream), None, None))

    def copy_status(self, out):
        """The two status words into `out` (2 x int32 / uint32 on the device) on the current stream, without waiting: the
        multi-GPU driver sends them along with the box list (byolo/inference.py)."""
        torch = _torch()
        assert out.is_cuda and out.numel() >= 2 and out.element_size() == 4 and out.is_contiguous()
        check(self._h, lib.byolo_copy_status(self._h, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def clear_status(self):
        torch = _torch()
        check(self._h, lib.byolo_clear_status(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    # ---- injected dropout masks (byolo_forward's d_mask_bits) ---------------------------------------------
    def num_dropout(self):
        return check(self._h, lib.byolo_num_dropout(self._h))

    def mask_layout(self, B, T=1):
        """[(bit_offset, elements)] per dropout layer for a (B, T) call, and the total length in 32-bit words."""
        off, n = ctypes.c_int64(), ctypes.c_int64()
        out = []
        nd = self.num_dropout()
        for k in range(nd + 1):
            check(self._h, lib.byolo_mask_layout(self._h, int(B), int(T), k, ctypes.byref(off), ctypes.byref(n)))
            out.append((int(off.value), int(n.value)))
        return out[:nd], out[nd][0] // 32

    def pack_masks(self, masks, B, T=1):
        """masks: one boolean array per dropout layer (creation order), shaped like its dropout input [S,h,w,cout]
        (True = keep).  Returns the uint32 word array byolo_forward takes (numpy; bit i of a layer at bit_offset + i)."""
        layout, words = self.mask_layout(B, T)
        assert len(masks) == len(layout), "%d masks for %d dropout layers" % (len(masks), len(layout))
        buf = np.zeros(words, dtype=np.uint32)
        for m, (off, n) in zip(masks, layout):
            m = np.ascontiguousarray(m, dtype=bool).reshape(-1)
            assert m.size == n, "mask of %d elements, the layer has %d" % (m.size, n)
            bits = np.packbits(m, bitorder="little")
            bits = np.concatenate([bits, np.zeros((-bits.size) % 4, dtype=np.uint8)])
            buf[off // 32: off // 32 + bits.size // 4] = bits.view("<u4")
        return buf

    def twin(self, precision="f32"):
        """A SECOND handle with the same graph and the same parameters (calibrated BN statistics included), finalized in the other
        arithmetic and kept for the life of this engine: both weight packs stay resident (2 x 246 MB for the reference's models),
        so ONE batch whose activations leave the split-f16 range is re-run in fp32 without re-packing anything, and the batches
        after it stay in the default precision (byolo/inference.py, lib_yolo/model.py Model.run).  Parameters are copied when the
        twin is made: call drop_twin() after changing this engine's parameters."""
        if self._twin is not None and self._twin.precision == precision:
            return self._twin
        self.drop_twin()
        t = Engine((self.cfg.img_h, self.cfg.img_w, self.cfg.img_c), self.cfg.cls_cnt, self.cfg.drop_prob, self.cfg.max_out,
                   self.cfg.iou_thresh, self.cfg.nms_mode, bool(self.cfg.keep_all_outputs), self.device)
        for name, args in self._graph:
            getattr(t, name)(*args)
        t.set_plan_opts(**self.plan_opts())               # the same plan on both handles (per-handle since round 6)
        t.set_precision(precision)
        t.set_params(self.get_params())
        t.finalize()
        self._twin = t
        return t

    def drop_twin(self):
        if self._twin is not None:
            self._twin.close()
            self._twin = None

    def close(self):
        self.drop_twin() if getattr(self, "_twin", None) is not None else None
        if getattr(self, "_h", None) and self._h.value:
            lib.byolo_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- graph construction (lib_yolo/model.py ModelBuilder.make_*) -------------------------
    def add_conv(self, scope, filters, ksize, stride, norm_flags):
        idx = check(self._h, lib.byolo_add_conv(self._h, scope.encode(), filters, ksize, stride, norm_flags))
        self._graph.append(('add_conv', (scope, filters, ksize, stride, norm_flags)))      # (recorded once the native call succeeded)
        return idx

    def add_residual(self, shortcut):
        idx = check(self._h, lib.byolo_add_residual(self._h, shortcut))
        self._graph.append(('add_residual', (shortcut,)))
        return idx

    def add_route(self, routes):
        arr = (ctypes.c_int32 * len(routes))(*[int(r) for r in routes])
        idx = check(self._h, lib.byolo_add_route(self._h, arr, len(routes)))
        self._graph.append(('add_route', (list(routes),)))
        return idx

    def add_upsample(self):
        idx = check(self._h, lib.byolo_add_upsample(self._h))
        self._graph.append(('add_upsample', ()))
        return idx

    def add_stack(self, src):
        idx = check(self._h, lib.byolo_add_stack(self._h, int(src)))
        self._graph.append(('add_stack', (src,)))
        return idx

    def add_detection(self, scope, kind, priors_hw):
        flat = [float(v) for p in priors_hw for v in p]
        assert len(flat) == 6, "exactly 3 priors (h, w) per detection layer"
        arr = (ctypes.c_float * 6)(*flat)
        idx = check(self._h, lib.byolo_add_detection(self._h, scope.encode(), int(kind), arr))
        self._graph.append(('add_detection', (scope, kind, [tuple(p) for p in priors_hw])))
        return idx

    def mark_backbone_end(self):
        check(self._h, lib.byolo_mark_backbone_end(self._h))
        self._graph.append(('mark_backbone_end', ()))

    # ---- parameters ------------------------------------------------------------------------------
    def param_shapes(self):
        """Ordered {tf variable name: shape} in TF creation order."""
        out = {}
        n = check(self._h, lib.byolo_num_params(self._h))
        name = ctypes.c_char_p()
        nd = ctypes.c_int32()
        shp = (ctypes.c_int64 * 4)()
        for i in range(n):
            check(self._h, lib.byolo_param_info(self._h, i, ctypes.byref(name), ctypes.byref(nd), shp))
            out[name.value.decode()] = tuple(int(shp[k]) for k in range(nd.value))
        return out

    def set_param(self, name, value):
        self.drop_twin()                                  # a twin holds a copy of the parameters
        a = np.ascontiguousarray(value, dtype=np.float32)
        check(self._h, lib.byolo_set_param(self._h, name.encode(), a.ctypes.data, a.size))

    def get_param(self, name, shape):
        a = np.empty(shape, dtype=np.float32)
        check(self._h, lib.byolo_get_param(self._h, name.encode(), a.ctypes.data, a.size))
        return a

    def set_params(self, params, strict=True):
        shapes = self.param_shapes()
        for k, shp in shapes.items():
            if k in params:
                v = np.asarray(params[k])
                if tuple(v.shape) != tuple(shp):
                    raise ValueError("variable %s: shape %s != %s" % (k, v.shape, shp))
                self.set_param(k, v)
            elif strict:
                raise KeyError("missing variable %s" % k)

    def get_params(self):
        return {k: self.get_param(k, s) for k, s in self.param_shapes().items()}

    def finalize(self):
        check(self._h, lib.byolo_finalize(self._h))
        self.finalized = True

    # ---- shapes / cost ----------------------------------------------------------------------------
    def num_layers(self):
        return check(self._h, lib.byolo_num_layers(self._h))

    def num_boxes(self):
        n = ctypes.c_int64()
        d = ctypes.c_int32()
        check(self._h, lib.byolo_num_boxes(self._h, ctypes.byref(n), ctypes.byref(d)))
        return int(n.value), int(d.value)

    def workspace_bytes(self, B, T=1):
        s = ctypes.c_size_t()
        check(self._h, lib.byolo_workspace_bytes(self._h, int(B), int(T), ctypes.byref(s)))
        return int(s.value)

    def flops(self, B, T=1):
        f = ctypes.c_double()
        check(self._h, lib.byolo_flops(self._h, int(B), int(T), ctypes.byref(f)))
        return float(f.value)

    @property
    def out_cap(self):
        return self.cfg.max_out * (2 if self.cfg.nms_mode == _lib.NMS_TWO_CLASS else 1)

    # ---- run ------------------------------------------------------------------------------------------
    def _workspace(self, B, T, slot=0):
        """Workspace arena per slot: concurrent forwards on different HIP streams (slot = stream
        index) must not share activations."""
        torch = _torch()
        need = self.workspace_bytes(B, T)
        if slot:
            if not hasattr(self, "_ws_slots"):
                self._ws_slots = {}
            ws = self._ws_slots.get(slot)
            if ws is None or ws.numel() < need:
                ws = self._ws_slots[slot] = torch.empty(need, dtype=torch.uint8, device="cuda:%d" % self.device)
            return ws
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device="cuda:%d" % self.device)
        return self._ws

    def _check_img(self, img):
        torch = _torch()
        if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.float32 and img.is_contiguous()):
            raise TypeError("img must be a contiguous float32 CUDA tensor [B,H,W,C]")
        if tuple(img.shape[1:]) != (self.cfg.img_h, self.cfg.img_w, self.cfg.img_c):
            raise ValueError("img shape %s does not match %s" % (tuple(img.shape), (self.cfg.img_h, self.cfg.img_w, self.cfg.img_c)))
        if img.device.index != self.device:
            raise ValueError("img is on cuda:%s, engine on cuda:%d" % (img.device.index, self.device))

    @property
    def torch_device(self):
        return "cuda:%d" % self.device

    def normalize_u8(self, u8, out=None):
        """decode_img's `convert_image_dtype(uint8 -> float32)` on the device: out = float(u8) * (1/255), fp32, on the current
        stream (byolo_normalize_u8).  u8: contiguous uint8 CUDA tensor; returns the float32 tensor forward() takes."""
        torch = _torch()
        if not (isinstance(u8, torch.Tensor) and u8.is_cuda and u8.dtype == torch.uint8 and u8.is_contiguous()):
            raise TypeError("u8 must be a contiguous uint8 CUDA tensor")
        if out is None:
            out = torch.empty(u8.shape, dtype=torch.float32, device=u8.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == u8.numel()
        check(self._h, lib.byolo_normalize_u8(self._h, ctypes.c_void_p(u8.data_ptr()), u8.numel(), ctypes.c_void_p(out.data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def max_images(self, T=1):
        """Largest batch one byolo_forward call accepts at this T (32-bit source offsets: every activation < 3 GiB)."""
        n = ctypes.c_int32()
        check(self._h, lib.byolo_max_images(self._h, int(T), ctypes.byref(n)))
        return int(n.value)

    def forward(self, img, T=1, seed=0, dropout_on=True, want_boxes=False, want_nms=True, out=None, slot=0, first_image=0,
                mask_bits=None, t_shard=None):
        """One sess.run of the reference (inference_epistemic.py:76).  Returns a dict of device
        tensors: rows [B,cap,D], kept [B,cap] int32, count [B,2] int32 and (want_boxes) boxes [B,N,D].
        Everything is enqueued on torch's current stream.  Host synchronisation: in the split-f16 precision the call WAITS for
        the stream at its end to read the range status and raises ByoloError(ERR_RANGE) itself -- unless set_async(True), under
        which nothing waits and the caller asks check_status() / status() / copy_status() where it synchronises anyway (the
        inference driver and bench.py do).  The fp32 mode never waits.

        first_image: position of img[0] in the logical batch (a shard of a data-parallel batch, a sub-batch): the
        dropout masks are those the unsplit batch would draw.  Batches beyond max_images(T) are run by byolo_forward itself as
        consecutive pieces with exactly that mechanism, so the result does not depend on the cut.

        t_shard = (t0, T_total): this call's T samples are samples t0 .. t0 + T - 1 of the T_total every image has in the whole
        job (the T axis sharded over ranks, include/byolo.h byolo_set_tshard; ONE image per call).  'boxes' then holds the per-box
        SUMS over the call's samples, not rows: add the ranks' tensors and call finish_tshard(); no NMS in this call."""
        torch = _torch()
        self._check_img(img)
        B = int(img.shape[0])
        if t_shard is not None:
            t0, T_total = int(t_shard[0]), int(t_shard[1])
            if B != 1 or want_nms or not want_boxes or mask_bits is not None or not (0 <= t0 and t0 + T <= T_total):
                raise ValueError("a T shard runs ONE image with want_boxes=True, want_nms=False and t0 + T <= T_total")
            check(self._h, lib.byolo_set_tshard(self._h, t0, T_total))
        try:
            return self._forward(img, B, T, seed, dropout_on, want_boxes, want_nms, out, slot, first_image, mask_bits)
        finally:
            if t_shard is not None:
                check(self._h, lib.byolo_set_tshard(self._h, 0, 0))

    def finish_tshard(self, sums, T_total):
        """The ranks' summed T-shard buffers [B, N, 21 + C] -> rows, in place (byolo_finish_tshard); returns `sums`."""
        torch = _torch()
        N, D = self.num_boxes()
        assert sums.is_cuda and sums.dtype == torch.float32 and sums.is_contiguous() and tuple(sums.shape[1:]) == (N, D)
        check(self._h, lib.byolo_finish_tshard(self._h, ctypes.c_void_p(sums.data_ptr()), int(sums.shape[0]), int(T_total),
                                               ctypes.c_void_p(torch.cuda.current_stream(sums.device).cuda_stream)))
        return sums

    def _forward(self, img, B, T, seed, dropout_on, want_boxes, want_nms, out, slot, first_image, mask_bits):
        torch = _torch()
        cap_b = self.max_images(T)
        if cap_b < 1:
            raise ValueError("T=%d at %dx%d: one image's stacked activation exceeds the 3 GiB a convolution source may "
                             "span (32-bit buffer offsets); lower T or the image size" % (T, self.cfg.img_h, self.cfg.img_w))
        if B > cap_b and mask_bits is not None:
            raise ValueError("injected masks describe ONE piece of a forward: keep B <= max_images(T) = %d" % cap_b)
        check(self._h, lib.byolo_set_first_image(self._h, int(first_image)))
        ws = self._workspace(B, T, slot)
        self._last_ws = ws                                # byolo_layer_output points into the last forward's workspace
        N, D = self.num_boxes()
        dev = img.device
        res = out if out is not None else {}
        boxes = rows = kept = count = None
        if want_boxes:
            boxes = res.get("boxes")
            if boxes is None:
                boxes = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        if want_nms:
            cap = self.out_cap
            rows, kept, count = res.get("rows"), res.get("kept"), res.get("count")
            if rows is None:
                rows = torch.empty((B, cap, D), dtype=torch.float32, device=dev)
                kept = torch.empty((B, cap), dtype=torch.int32, device=dev)
                count = torch.empty((B, 2), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        if mask_bits is not None:
            assert mask_bits.is_cuda and mask_bits.dtype in (torch.int32, torch.uint32) and mask_bits.is_contiguous()
            assert mask_bits.numel() >= self.mask_layout(B, T)[1], "mask_bits shorter than byolo_mask_layout's total"
        check(self._h, lib.byolo_forward(self._h, p(img), B, int(T), ctypes.c_uint64(int(seed) & (2**64 - 1)),
                                         int(bool(dropout_on)), p(mask_bits), p(ws), ws.numel(), p(boxes), p(rows), p(kept),
                                         p(count), ctypes.c_void_p(stream)))
        return dict(boxes=boxes, rows=rows, kept=kept, count=count)

    def layer_output(self, idx):
        """Copy of layer `idx`'s output after a forward (needs keep_all_outputs=True)."""
        torch = _torch()
        ptr = ctypes.c_void_p()
        shp = (ctypes.c_int64 * 4)()
        check(self._h, lib.byolo_layer_output(self._h, int(idx), ctypes.byref(ptr), shp))
        shape = tuple(int(s) for s in shp)
        ws = getattr(self, "_last_ws", None)
        n = int(np.prod(shape))
        off = ptr.value - ws.data_ptr() if ws is not None else -1
        if ws is None or off < 0 or off + 4 * n > ws.numel():
            raise RuntimeError("layer_output: the workspace of the last forward is gone")
        # float32 values whatever the handle's precision (split-f16 activations are hi/lo pairs in the workspace)
        out = torch.empty(shape, dtype=torch.float32, device="cuda:%d" % self.device)
        stream = torch.cuda.current_stream(out.device).cuda_stream
        check(self._h, lib.byolo_copy_layer_output(self._h, int(idx), ctypes.c_void_p(out.data_ptr()), n, ctypes.c_void_p(stream)))
        torch.cuda.synchronize(self.device)
        return out

    def calibrate_bn(self, img):
        self.drop_twin()
        self._check_img(img)
        torch = _torch()
        B = int(img.shape[0])
        ws = self._workspace(B, 1)
        self._last_ws = ws
        stream = torch.cuda.current_stream(img.device).cuda_stream
        check(self._h, lib.byolo_calibrate_bn(self._h, ctypes.c_void_p(img.data_ptr()), B,
                                              ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(stream)))

    def set_profiling(self, level, keep=False):
        """0 off, 1 per-stage hipEvents, 2 additionally one hipEvent per conv launch.  keep: the records already taken stay
        readable (byolo_resume_profiling) -- for a run that records every n-th forward."""
        check(self._h, (lib.byolo_resume_profiling if keep else lib.byolo_set_profiling)(self._h, int(level)))

    def set_profile_depth(self, depth):
        """Keep the profile records of the last `depth` forwards (read them with select_profile(age))."""
        check(self._h, lib.byolo_set_profile_depth(self._h, int(depth)))

    def select_profile(self, age):
        """Which profiled forward step_profile() / stage_ms() read: 0 = the last one, 1 = the one before, ..."""
        check(self._h, lib.byolo_select_profile(self._h, int(age)))

    def step_profile(self):
        """Per conv launch of the last forward (profiling level 2): list of dicts
        {layer, variant (tile BN or -1 = direct), M, N, K, ms, flops}."""
        n = check(self._h, lib.byolo_num_steps(self._h))
        out = []
        layer, var, ms, algo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_float(), ctypes.c_double()
        mnk = (ctypes.c_int64 * 3)()
        ks, stl = ctypes.c_int32(), ctypes.c_int32()
        for i in range(n):
            check(self._h, lib.byolo_step_profile(self._h, i, ctypes.byref(layer), ctypes.byref(var), mnk, ctypes.byref(ms),
                                                  ctypes.byref(algo)))
            check(self._h, lib.byolo_step_split(self._h, i, ctypes.byref(ks), ctypes.byref(stl)))
            out.append(dict(layer=layer.value, variant=var.value, M=int(mnk[0]), N=int(mnk[1]), K=int(mnk[2]),
                            ms=float(ms.value), flops=float(algo.value), flops_executed=2.0 * mnk[0] * mnk[1] * mnk[2],
                            ksplit=ks.value, split_tiles=stl.value))
        return out

    def stage_ms(self):
        ms = (ctypes.c_float * 4)()
        check(self._h, lib.byolo_stage_ms(self._h, ms))
        return dict(backbone=ms[0], heads=ms[1], decode=ms[2], sort_nms=ms[3])

    # ---- staged tail (parity tests; also the eager `nms(...)` helpers of inference_*.py) -------------
    def decode(self, kind, raw, B, T, priors_hw, layer_id, boxes, box_base):
        torch = _torch()
        S, lh, lw, F = raw.shape
        assert S == B * T and raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous()
        arr = (ctypes.c_float * 6)(*[float(v) for p in priors_hw for v in p])
        stream = torch.cuda.current_stream(raw.device).cuda_stream
        check(self._h, lib.byolo_decode(self._h, int(kind), ctypes.c_void_p(raw.data_ptr()), int(B), int(T), int(lh),
                                        int(lw), arr, int(layer_id), ctypes.c_void_p(boxes.data_ptr()),
                                        int(boxes.shape[1]), int(box_base), ctypes.c_void_p(stream)))

    def epistemic_stats(self, raw, B, T):
        """decode_epistemic's dict entries outside the box row from a raw epistemic detection output [B*T,lh,lw,F]:
        dict(ev_loc [B,lh,lw,3,4], epi_covar_loc [B,lh,lw,3,4,4], obj_samples [B*T,lh,lw,3], cls_samples [B*T,lh,lw,3,C])."""
        torch = _torch()
        S, lh, lw, F = raw.shape
        C = self.cfg.cls_cnt
        assert S == B * T and F == 3 * 2 * (5 + C) and raw.is_cuda and raw.dtype == torch.float32 and raw.is_contiguous()
        dev = raw.device
        out = dict(ev_loc=torch.empty((B, lh, lw, 3, 4), device=dev), epi_covar_loc=torch.empty((B, lh, lw, 3, 4, 4), device=dev),
                   obj_samples=torch.empty((S, lh, lw, 3), device=dev), cls_samples=torch.empty((S, lh, lw, 3, C), device=dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        check(self._h, lib.byolo_epistemic_stats(self._h, p(raw), int(B), int(T), int(lh), int(lw), p(out["ev_loc"]),
                                                 p(out["epi_covar_loc"]), p(out["obj_samples"]), p(out["cls_samples"]),
                                                 ctypes.c_void_p(stream)))
        return out

    def sort_nms(self, boxes, obj_idx, cls_start_idx, nms_mode=None, max_out=None, iou_thresh=None):
        """tf.image.non_max_suppression + tf.gather per image on boxes [B,N,D] (device tensor)."""
        torch = _torch()
        assert boxes.is_cuda and boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 3
        B, N, D = boxes.shape
        nms_mode = self.cfg.nms_mode if nms_mode is None else nms_mode
        max_out = self.cfg.max_out if max_out is None else max_out
        iou_thresh = self.cfg.iou_thresh if iou_thresh is None else iou_thresh
        cap = max_out * (2 if nms_mode == _lib.NMS_TWO_CLASS else 1)
        wsb = int(lib.byolo_nms_workspace_bytes(B, N))
        ws = torch.empty(wsb, dtype=torch.uint8, device=boxes.device)
        rows = torch.empty((B, cap, D), dtype=torch.float32, device=boxes.device)
        kept = torch.empty((B, cap), dtype=torch.int32, device=boxes.device)
        count = torch.empty((B, 2), dtype=torch.int32, device=boxes.device)
        stream = torch.cuda.current_stream(boxes.device).cuda_stream
        check(self._h, lib.byolo_sort_nms(self._h, ctypes.c_void_p(boxes.data_ptr()), B, N, D, int(obj_idx),
                                          int(cls_start_idx), int(nms_mode), int(max_out), float(iou_thresh),
                                          ctypes.c_void_p(ws.data_ptr()), wsb, ctypes.c_void_p(rows.data_ptr()),
                                          ctypes.c_void_p(kept.data_ptr()), ctypes.c_void_p(count.data_ptr()),
                                          ctypes.c_void_p(stream)))
        return dict(rows=rows, kept=kept, count=count)
