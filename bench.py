#!/usr/bin/env python
"""bench.py -- headline benchmark of BASELINE.json: img/s of epistemic (MC-dropout) inference at
T=30, 608x608 ("configs[3]", the configuration the metric is quoted on; 8 images per GPU).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
Darknet-53 once per image, the three heads on B*T MC samples (dropout masks from the counter RNG),
per-box T-reduction + decode, sort + NMS, and (N > 1) ONE RCCL all-gather of the padded box lists.
Weights are random-init with BN statistics calibrated on the device (no checkpoints, no network).

Prints ONE JSON line on rank 0.  `roofline` follows SURVEY.md section 8(d) for the dominant kernel = the matrix-pipe
kernel with the most device time in the timed region (`by_kernel` lists all of them):

  achieved               ALGORITHMIC fp32-equivalent FLOP/s of its launches: 2*M*N*K of the convolution as written (SURVEY 8(d)'s
                         F(H,W,T) is made of these); a Winograd launch stands for the direct-convolution FLOPs of its samples
                         although its matrix pipe executes 1 / 2.25 of them (`achieved_executed_padded`); tile padding is not
                         work.  Time = hipEvents recorded around every launch ON THE LAUNCH STREAM (byolo_step_profile; the
                         records of all K steps are read AFTER the timed region);
  peak / frac            the dense peak of the matrix instruction the kernel issues, every useful FLOP counted ONCE: 2500
                         TFLOP/s (v_mfma_f32_32x32x16_f16) for the split-f16 kernels of the default precision -- which
                         execute three fp16 products per fp32 product, so a split kernel cannot exceed frac 1/3 -- and 157.3
                         (v_mfma_f32_32x32x2_f32) for the fp32 mode;
  frac_executed_vs_fp16_peak   (split kernels) the executed fp16 products / 2500: the utilisation of the pipe itself;
  traffic                L2<->fabric bytes per launch of the dominant kernel from separate rocprofv3 --pmc FETCH_SIZE /
                         WRITE_SIZE passes over this command (tools/profile_round.sh, tools/pmc_traffic.py ->
                         profiles/traffic_cfgN.json, stamped with the commit they were measured at), reads and writes
                         separately, each beside its algorithmic byte count (`read_amplification`, `write_amplification`).

`fp32_mode`: the same workload timed in the same run under BYOLO_PREC_F32 (the reference's own arithmetic, lib_yolo/layers.py:550)
on a second handle: a few steps after the headline region.  `cpu_baseline` is the oracle's CPU restatement (PyTorch/oneDNN, NOT
TensorFlow) on a bounded sample; the same oracle rows give `parity_note`: the device's distance from the float32 oracle per
column group, in units of the bound 1e-4 * max(1, |ref|), on image 0 of the benchmark's own batch.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))

CONFIGS = {   # BASELINE.json configs[0..4]  (per-GPU batch)
    # configs[0]: the reference's CPU-runnable plumbing case (inference_standard_yolov3.py, one image) -- run on the device as well
    1: dict(variant="yolov3", H=416, W=416, B=1, T=1, nms=0),
    2: dict(variant="yolov3_aleatoric", H=416, W=416, B=8, T=1, nms=0),
    3: dict(variant="bayesian_yolov3_aleatoric", H=416, W=416, B=16, T=10, nms=0),
    4: dict(variant="bayesian_yolov3_aleatoric", H=608, W=608, B=8, T=30, nms=0),
    5: dict(variant="bayesian_yolov3_aleatoric", H=1024, W=1024, B=1, T=50, nms=1),
    # the reference's own default frame (inference_epistemic.py:218, full ECP image), not a BASELINE line
    6: dict(variant="bayesian_yolov3_aleatoric", H=1024, W=1920, B=1, T=50, nms=1),
    # the reference's own default workloads of the two non-epistemic scripts (inference_aleatoric.py:219-227,
    # inference_standard_yolov3.py:210-218): the full ECP frame, batch_size 11, class-agnostic NMS -- not BASELINE lines either
    7: dict(variant="yolov3_aleatoric", H=1024, W=1920, B=11, T=1, nms=0),
    8: dict(variant="yolov3", H=1024, W=1920, B=11, T=1, nms=0),
}
PEAK_FP32_MFMA = 157.3e12      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA = 2500e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (no sparsity)
# split-f16 precision: three fp16 matrix products per fp32 product, so the matrix-pipe ceiling of a split kernel in
# fp32-equivalent FLOP/s is PEAK_F16_MFMA / 3; under the 1300 W socket cap an MFMA-only loop on random operands sustains
# 1756 TFLOP/s fp16 (tools/mfma_f16_power_probe.hip, profiles/r2_probes.md) -- reported beside the nominal peak
SUSTAINED_F16_MFMA = 1756e12


_STREAMS = {}


def pipeline_streams(device, n=2):
    """The HIP streams whole steps alternate over -- ONE set per process, created before the first engine runs and shared by the
    headline loop and every `other_configs` leg.  (A leg that created its own two streams after the headline's engine had run sometimes
    found them on one hardware queue: BASELINE configs[0] then read 630 img/s -- its one-stream rate -- where the same leg reads 1 150
    in a fresh process or when called a second time; profiles/r6_small_configs.md.)"""
    import torch
    key = (int(device), int(n))
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device="cuda:%d" % device) for _ in range(n)]
    return _STREAMS[key]


def build(cfg, device, precision=None, params=None):
    """The benchmark's model: seeded random weights, BN statistics calibrated on the device (or `params`: the complete parameter
    set of an engine built here before -- the other precision of the same configuration computes on the SAME weights)."""
    from lib_yolo import yolov3, model
    from byolo import synth
    import torch
    config = {"full_img_size": [cfg["H"], cfg["W"], 3], "crop": False, "cls_cnt": 2, "priors": yolov3.ECP_9_PRIORS,
              "aleatoric_loss": False, "inference_mode": True, "T": cfg["T"], "implicit_background_class": True,
              "engine_options": {"nms_mode": cfg["nms"], "device": device}}
    yolo = getattr(yolov3, cfg["variant"])(config)
    m = yolo.init_model(inputs=model.Placeholder((None, cfg["H"], cfg["W"], 3)), training=False).get_model()
    eng = m.engine
    if precision is not None:
        eng.set_precision(precision)
    if params is not None:
        eng.set_params(params)
        eng.finalize()
        return m
    eng.set_params(synth.base_params(eng.param_shapes(), cfg["variant"], 2, seed=7))
    eng.finalize()
    calib = torch.from_numpy(synth.synthetic_images(2, cfg["H"], cfg["W"], seed=999)).to("cuda:%d" % device)
    eng.calibrate_bn(calib)           # same calibration frames on every rank -> identical weights
    del calib
    return m


def cpu_baseline(cfg, params, n_img=1):
    """Oracle (CPU restatement, unfused op sequence, same NMS) timed on the host cores."""
    import numpy as np
    import torch
    from oracle import cpu_ref
    from byolo import synth
    # oneDNN stops scaling on these layer sizes well before a 256-thread host is full (and
    # oversubscription hurts): use at most 64 threads and report the number actually used
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    imgs = synth.synthetic_images(n_img, cfg["H"], cfg["W"], seed=1234)
    tp = cpu_ref.to_torch_params(params)
    t0 = time.time()
    with torch.no_grad():
        boxes, _ = cpu_ref.detect_boxes(tp, imgs, cfg["variant"], T=cfg["T"], seed=42)
        kept = cpu_ref.nms_batch(boxes, cfg["variant"], two_class=bool(cfg["nms"]))
    dt = time.time() - t0
    return {"value": n_img / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%d image(s) %dx%d T=%d, CPU restatement (PyTorch/oneDNN fp32, not TensorFlow), %.1f s"
                      % (n_img, cfg["H"], cfg["W"], cfg["T"], dt)}, boxes.numpy(), kept


def parity_check(cfg, eng, x, ref32, ref64=None, oracle_kept=None, seconds=None):
    """THE PARITY CONTRACT (oracle/report.py) on image 0 of a batch, dropout seed 42: one more device forward of the SAME batch
    against the oracle rows `ref32` (float32 CPU restatement) and, where given, `ref64` (the same in float64 = the exact value of
    the reference's graph).  Per column group: the worst value in units of the bound 1e-4 * max(1, |ref|), the largest ABSOLUTE
    error, the largest RELATIVE error over |ref| > 1, the allowance and whether it holds:
        vs_float64: max(1, F(g));   vs_float32: max(1, F(g)) + F(g) -- F = `float32_vs_float64` measured here; 1 where no float64 run.
    Tail: the oracle's NMS on the DEVICE's rows must keep exactly what the device kept.  `ok` = everything holds; bench.py exits
    non-zero after printing its line when any leg's `ok` is false."""
    import numpy as np
    import torch
    from oracle import report, cpu_ref
    r = eng.forward(x, T=cfg["T"], seed=42, want_boxes=True, want_nms=True, first_image=0)
    got = r["boxes"][:1].cpu().numpy()
    n = int(r["count"][0, 0])
    kept = r["kept"][0, :n].cpu().numpy()
    ref32 = np.asarray(ref32)[:1]
    floor = report.rows_report(ref32, np.asarray(ref64)[:1], cfg["variant"]) if ref64 is not None else None
    vs32, ok32 = report.check(report.rows_report(got, ref32, cfg["variant"]), report.allowance(floor, "float32"))
    pat = bool(np.array_equal(np.isnan(got), np.isnan(ref32)) and np.array_equal(np.isinf(got), np.isinf(ref32)))
    # the oracle's NMS on the DEVICE's rows must keep exactly what the device kept; against the oracle's NMS of the ORACLE's rows the
    # greedy visiting order may flip between near-tied scores, so that comparison is a count of differing boxes
    tail = cpu_ref.nms_batch(torch.from_numpy(got), cfg["variant"], two_class=bool(cfg["nms"]))[0][1]
    tail_ok = bool(n == len(tail) and np.array_equal(kept, tail))
    res = {"compared": "image 0 of the batch, all %d pre-NMS rows, dropout seed 42; device (%s)%s"
                       % (got.shape[1], eng.precision, "" if seconds is None else "; %.1f s of host time for the oracle" % seconds),
           "bound": "1e-4 * max(1, |ref|); allowance max(1, F) vs float64, max(1, F) + F vs float32, F = float32_vs_float64 measured here; 1 where there is no float64 run (oracle/report.py)",
           "vs_float32": vs32, "nan_inf_pattern_equal": pat, "kept_indices_bit_exact_vs_oracle_nms_on_device_rows": tail_ok, "kept_boxes": n}
    ok = ok32 and pat and tail_ok
    if ref64 is not None:
        vs64, ok64 = report.check(report.rows_report(got, np.asarray(ref64)[:1], cfg["variant"]), report.allowance(floor))
        res["vs_float64"] = vs64
        res["float32_vs_float64"] = {k: round(v["worst_in_bounds"], 3) for k, v in floor.items()}
        ok = ok and ok64
    res["worst_in_bounds"] = {k: v["worst_in_bounds"] for k, v in vs32.items()}            # (the round-4 field, kept)
    if oracle_kept is not None:
        okk = oracle_kept[0][1]
        res["kept_boxes_oracle_rows"] = int(len(okk))
        res["kept_set_symmetric_difference_vs_oracle_rows"] = int(len(set(kept.tolist()) ^ set(np.asarray(okk).tolist())))
    res["ok"] = bool(ok)
    return res


def oracle_rows(cfg, params, f64=False):
    """Image 0 of the configuration's batch through the CPU restatement with dropout seed 42; (rows, seconds)."""
    import torch
    from oracle import cpu_ref
    from byolo import synth
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    dt = torch.float64 if f64 else torch.float32
    imgs = synth.synthetic_images(1, cfg["H"], cfg["W"], seed=1234)
    t0 = time.time()
    with torch.no_grad():
        ref, _ = cpu_ref.detect_boxes(cpu_ref.to_torch_params(params, dt), imgs, cfg["variant"], T=cfg["T"], seed=42, dtype=dt)
    return ref.numpy(), time.time() - t0


def other_config_leg(num, device, steps=5, warmup=2, oracle="f32"):
    """One of the other BASELINE configs (parity-test cases, not the headline), timed the way the headline is -- whole steps
    alternating over two HIP streams, inputs resident, `steps` timed steps -- so that every configuration BASELINE.json names has
    a driver-run number; `parity` = parity_check on image 0 (oracle "f32": against the float32 oracle at the literal bound; "f64":
    additionally against the float64 run, whose distance from the float32 one is then the float32 comparison's allowance;
    False: the oracle image takes minutes on the host -- tests/test_gpu_bench_shapes.py covers the shape)."""
    import numpy as np
    import torch
    from byolo import synth
    cfg = dict(CONFIGS[num])
    m = build(cfg, device)
    eng = m.engine
    eng.set_async(True)
    B, T = cfg["B"], cfg["T"]
    x = torch.from_numpy(synth.synthetic_images(B, cfg["H"], cfg["W"], seed=1234)).to("cuda:%d" % device)
    N, D = eng.num_boxes()
    cap = eng.out_cap
    pipes = [dict(st=st,
                  out={"rows": torch.empty((B, cap, D), device=x.device), "kept": torch.empty((B, cap), dtype=torch.int32, device=x.device),
                       "count": torch.empty((B, 2), dtype=torch.int32, device=x.device)}) for st in pipeline_streams(device, 2)]

    def step(i):
        pp = pipes[i % 2]
        with torch.cuda.stream(pp["st"]):
            eng.forward(x, T=T, seed=1000 + i, want_boxes=False, want_nms=True, out=pp["out"], slot=1 + i % 2)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a launch-sized configuration (configs[0]: 1.5 ms per step) is not measured by 5 steps: the first replays of its launch graph, the
    # host's first trips through the binding.  Timed again over enough steps for ~0.3 s (round 5 quoted 628 img/s from 5 steps where
    # 200 steps give 1 150: profiles/r6_small_configs.md)
    if dt < 0.25:
        done = warmup + steps
        steps = int(min(400, max(steps, round(steps * 0.3 / max(dt, 1e-4)))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(done + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    flags, layer = eng.status()
    peak = PEAK_F16_MFMA if eng.precision == "split" else PEAK_FP32_MFMA
    res = {"workload": "%s %dx%d T=%d, %d images/GPU, class-%s NMS" % (cfg["variant"], cfg["H"], cfg["W"], T, B, "wise 2-class" if cfg["nms"] else "agnostic"),
           "img_s": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup, "precision": eng.precision,
           "gflop_per_image": eng.flops(1, T) / 1e9, "end_to_end_frac": (B * steps / dt) * eng.flops(1, T) / peak,
           "range_status": "ok" if flags == 0 else "RANGE (layer %d): invalid" % layer,
           "launch_graphs": eng.graph_stats() if hasattr(eng, "graph_stats") else None}
    if oracle:
        try:
            p = eng.get_params()
            ref32, t32 = oracle_rows(cfg, p)
            ref64, t64 = oracle_rows(cfg, p, f64=True) if oracle == "f64" else (None, 0.0)
            res["parity"] = parity_check(cfg, eng, x, ref32, ref64, seconds=t32 + t64)
        except Exception as e:
            res["parity"] = {"error": repr(e), "ok": None}        # the oracle leg itself failed: unknown, not a parity failure
    else:
        res["parity"] = "see profiles/*_parity_table.json (tests/test_gpu_bench_shapes.py at this shape; the oracle image takes minutes on the host)"
    eng.close()
    return res


def t_shard_leg(device, ranks=8, reps=10):
    """The T-sharded latency path (config['shard'] = 'T') at the reference's OWN default workload -- ONE
    1024 x 1920 image per step, T = 50 (inference_epistemic.py:193, :218-221) -- as far as ONE GPU can show it: the device time of
    rank 0's share of an N-rank job (backbone + its T / N samples of the heads + the per-box sums; then byolo_finish_tshard and the
    NMS on the summed buffer), next to the one-GPU forward of all T samples.  NOT an N-GPU measurement: the all-reduce of the
    N * (21 + C) floats (11 MB here) between the two is not in it, and the other ranks are assumed to take as long as rank 0."""
    import torch
    from byolo import synth, dist as bdist
    cfg = dict(CONFIGS[6], nms=0)
    m = build(cfg, device)
    eng = m.engine
    eng.set_async(True)
    T = cfg["T"]
    x = torch.from_numpy(synth.synthetic_images(1, cfg["H"], cfg["W"], seed=1234)).to("cuda:%d" % device)
    t0, t1 = bdist.shard_range(T, 0, ranks)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))

    def whole():
        return eng.forward(x, T=T, seed=7, want_boxes=False, want_nms=True)

    def shard():
        sums = eng.forward(x, T=t1 - t0, seed=7, want_boxes=True, want_nms=False, t_shard=(t0, T))["boxes"]
        e1.record()
        rows = eng.finish_tshard(sums, T)
        return eng.sort_nms(rows, m.obj_idx, m.cls_start_idx)
    out = {}
    for name, fn in (("one_gpu_all_samples_ms", whole), ("rank0_of_%d_ms" % ranks, shard)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        tot = part = 0.0
        for _ in range(reps):
            e0.record(); fn(); e2.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e2)
            if fn is shard:
                part += e0.elapsed_time(e1)
        out[name] = tot / reps
        if fn is shard:
            out["rank0_forward_ms"] = part / reps
            out["rank0_finish_and_nms_ms"] = (tot - part) / reps
    N, D = eng.num_boxes()
    out.update({"workload": "bayesian_yolov3_aleatoric 1024x1920, ONE image, T=%d; rank 0 of %d runs samples [%d, %d)" % (T, ranks, t0, t1),
                "all_reduce_bytes": N * D * 4 + 8, "device_time_ratio": out["one_gpu_all_samples_ms"] / out["rank0_of_%d_ms" % ranks],
                "note": "device time of ONE rank's share, measured on one GPU; the all-reduce between forward and finish and the other ranks are not in it"})
    eng.close()
    return out


def entry_point_leg(cfg, device, n_frames=512, distinct=32, extras=True):
    """The drop-in path a user of the reference runs: `inference_epistemic.inference(config)` (inference_epistemic.py:186-208) over
    a TFRecord shard set generated here -- `distinct` synthetic frames (i.i.d. uniform, quantised to bytes) as PNG
    records, repeated to n_frames with their own file names, two shards -- with the benchmark's batch, T and weights recipe.
    Everything a user pays is inside: record framing + CRC-32C, PNG decode, H2D, the forward, the all-gather (N > 1), D2H,
    ECP-JSON text, file writes.  `img_s` = frames / the driver loop's wall time (first record requested -> last file closed);
    `wall_s` additionally holds building the engine, generating and calibrating the random weights.  `feed_img_s` /
    `writer_img_s`: the feed and the writer ALONE on this host, same thread counts -- what bounds the loop when the device does not."""
    import io
    import shutil
    import tempfile
    import numpy as np
    from PIL import Image
    from byolo import synth, hostio
    from byolo import inference as binf
    from lib_yolo import dataset_utils as du, yolov3
    import inference_epistemic
    H, W, B, T = cfg["H"], cfg["W"], cfg["B"], cfg["T"]
    tmp = tempfile.mkdtemp(prefix="byolo_entry_")
    try:
        t0 = time.perf_counter()
        frames = (synth.synthetic_images(distinct, H, W, seed=1234) * 256.0).astype(np.uint8)
        enc = []
        for f in frames:
            b = io.BytesIO()
            Image.fromarray(f).save(b, format="PNG", compress_level=1)
            enc.append(b.getvalue())
        for k in range(2):                       # records are generated as they are written: a long run does not hold them in memory
            du.write_tfrecords(os.path.join(tmp, "ecp-day-val-%05d-of-00002" % k),
                               (du.make_example({"image/encoded": enc[i % distinct], "image/filename": "frame_%05d.png" % i,
                                                 "image/height": H, "image/width": W}) for i in range(k, n_frames, 2)))
        t_gen = time.perf_counter() - t0
        threads = max(1, min(24, (os.cpu_count() or 1) - 2))             # the reference's default cpu_thread_cnt is 24
        config = {"checkpoint_path": tmp, "run_id": "bench", "step": "last", "weights": "synthetic", "full_img_size": [H, W, 3],
                  "cls_cnt": 2, "batch_size": B, "T": T, "inference_mode": True, "cpu_thread_cnt": threads, "crop": False,
                  "training": False, "aleatoric_loss": False, "priors": yolov3.ECP_9_PRIORS, "implicit_background_class": True,
                  "engine_options": {"nms_mode": cfg["nms"], "device": device}, "seed": 1000, "writer_threads": 4,
                  "data": {"file_pattern": os.path.join(tmp, "ecp-day-val-*-of-*")}, "out_path": os.path.join(tmp, "out", "bench")}
        t0 = time.perf_counter()
        stats = inference_epistemic.inference(config)
        wall = time.perf_counter() - t0
        out_dir = os.path.join(tmp, "out", "bench_0")
        files = sorted(os.listdir(out_dir))
        assert len(files) == n_frames, "the entry point wrote %d files for %d frames" % (len(files), n_frames)
        sample = json.load(open(os.path.join(out_dir, files[0])))["children"]
        json_bytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in files)
        if not extras:
            return {"img_s": n_frames / stats["loop_s"], "steady_img_s": stats.get("steady_img_s"), "steady_images": stats.get("steady_images"),
                    "unit": "img/s", "frames": n_frames, "batch_size": B, "T": T,
                    "img_size": [H, W], "loop_s": stats["loop_s"], "wall_s": wall, "decode_threads": threads,
                    "host_waited_s": {"feed": stats["wait_feed_s"], "device": stats["wait_device_s"], "writer": stats["wait_writer_s"]},
                    "boxes_per_image": len(sample), "precision": stats["precision"]}
        # the feed alone (decode pool + prefetch, frames dropped), then the writer alone (the rows of one written file, n_frames times)
        t0 = time.perf_counter()
        n = 0
        for sh in du.TestingDataset(config).iter_shards_u8(0, 1):
            n += len(sh.names)
            sh.release()
        feed_s = time.perf_counter() - t0
        from concurrent.futures import ThreadPoolExecutor
        D = 23
        rows = np.random.default_rng(0).random((max(1, len(sample)), D), dtype=np.float32)
        fmt = hostio.EcpJsonFormatter("bayesian_yolov3_aleatoric", [H, W, 3], 2, 14, 17, True, binf.LABEL_TO_CLS_NAME)
        wdir = os.path.join(tmp, "w")
        os.makedirs(wdir)

        def one(i):
            with open(os.path.join(wdir, "%05d.json" % i), "wb") as f:
                f.write(fmt.format(rows))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=4) as pool:
            list(pool.map(one, range(n_frames)))
        writer_s = time.perf_counter() - t0
        return {"img_s": n_frames / stats["loop_s"], "unit": "img/s", "frames": n_frames, "batch_size": B, "T": T,
                # the loop's rate after its fill (the first quarter of the batches: first decode, first H2D, the first forward's plan)
                "steady_img_s": stats.get("steady_img_s"), "steady_images": stats.get("steady_images"),
                "entry": "inference_epistemic.inference(config) -- TFRecord shards -> PNG decode -> device -> ECP JSON files",
                "loop_s": stats["loop_s"], "wall_s": wall, "setup_s": wall - stats["loop_s"], "records_generated_in_s": t_gen,
                "feed_img_s": n / feed_s, "writer_img_s": n_frames / writer_s, "cores": os.cpu_count(), "decode_threads": threads,
                "writer_threads": 4, "prefetch_batches": max(2, min(16, -(-threads // B))), "batches_in_flight": 2,
                "host_waited_s": {"feed": stats["wait_feed_s"], "device": stats["wait_device_s"], "writer": stats["wait_writer_s"]},
                "boxes_per_image": len(sample), "json_mb_written": json_bytes / 1e6, "png_mb_read": sum(len(e) for e in enc) / distinct * n_frames / 1e6,
                "native_json": stats["native_json"], "precision": stats["precision"], "precision_switches": stats["precision_switches"]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def training_side_leg(cfg, device, images=8, boxes_per_image=40, reps=50):
    """SURVEY 8(f) item 4 on the benchmark's geometry: byolo_encode_gt (lib_yolo/tfdata.py:77-171) for `images` images with
    `boxes_per_image` boxes each, and byolo_loss (lib_yolo/layers.py:126-188, aleatoric_loss, with the gradient) on the three raw
    detection tensors of a training-mode batch (no T-stacking: S = images).  Device time per call from hipEvents over `reps`
    back-to-back calls; `bytes` = ALGORITHMIC bytes (inputs read once + outputs written once) -> GB/s against the 8 TB/s HBM peak.
    Both are launch-latency-sized at this batch (a few MB per call): the fraction says so."""
    import numpy as np
    import torch
    from lib_yolo import yolov3, tfdata, data
    from byolo import loss as bl, DET_ALEATORIC
    H, W = cfg["H"], cfg["W"]
    layers = [data.DetLayerInfo(h=H // s, w=W // s, priors=yolov3.ECP_9_PRIORS[s]) for s in (32, 16, 8)]
    flat = [p for l in layers for p in l.priors]
    rng = np.random.default_rng(5)
    bb = np.zeros((images, boxes_per_image, 4), np.float32)
    for i in range(images):
        for j in range(boxes_per_image):
            p = flat[rng.integers(len(flat))]
            h, w, cy, cx = p.h * rng.uniform(0.7, 1.4), p.w * rng.uniform(0.7, 1.4), rng.uniform(0.05, 0.95), rng.uniform(0.05, 0.95)
            bb[i, j] = [cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2]
    lab = rng.integers(0, 2, (images, boxes_per_image)).astype(np.int32)
    dev = "cuda:%d" % device
    d_bb, d_lab = torch.from_numpy(bb).to(dev), torch.from_numpy(lab).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    gt = tfdata.encode_boxes_batch(d_bb, d_lab, layers, 0.7)
    N = gt.N
    enc_ms = timed(lambda: tfdata.encode_boxes_batch(d_bb, d_lab, layers, 0.7))
    enc_bytes = images * (N * 7 * 4 + boxes_per_image * 20)
    F = 3 * 14
    raws = [torch.randn((images, l.h, l.w, F), device=dev) for l in layers]
    gts = gt.layers()
    loss_ms = timed(lambda: [bl.detection_loss(r, DET_ALEATORIC, 2, g, aleatoric_loss=True, want_grad=True) for r, g in zip(raws, gts)])
    loss_bytes = images * N // 3 * (2 * F * 4) + images * N * 7 * 4
    return {"what": "byolo_encode_gt + byolo_loss (3 detection layers, aleatoric_loss, with gradient) at %dx%d, %d images, %d boxes each; "
                    "host-side call overhead included" % (H, W, images, boxes_per_image),
            "objects": int(gt.obj.sum().item()), "prior_boxes_per_image": N,
            "encode_ms": enc_ms, "encode_algorithmic_MB": enc_bytes / 1e6, "encode_GBps": enc_bytes / (enc_ms * 1e-3) / 1e9,
            "loss_ms_3_layers": loss_ms, "loss_algorithmic_MB": loss_bytes / 1e6, "loss_GBps": loss_bytes / (loss_ms * 1e-3) / 1e9,
            "hbm_peak_GBps": 8000.0, "bound": "launch latency at this size (a few MB per call); HBM for large batches"}


def time_steps(eng, x, cfg, steps, warmup, first_image=0):
    """warmup + `steps` forwards of the batch on the current stream with per-launch hipEvents; returns (seconds, {variant: [useful
    flops, ms, launches]}).  Used for the fp32_mode leg."""
    import torch
    eng.set_async(True)
    eng.set_profiling(2)
    for i in range(warmup):
        eng.forward(x, T=cfg["T"], seed=2000 + i, want_boxes=False, want_nms=True, first_image=first_image)
    n_sub = -(-int(x.shape[0]) // eng.max_images(cfg["T"]))
    eng.set_profile_depth(steps * n_sub)
    eng.set_profiling(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.forward(x, T=cfg["T"], seed=2000 + warmup + i, want_boxes=False, want_nms=True, first_image=first_image)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = {}
    for age in range(steps * n_sub):
        eng.select_profile(age)
        for s in eng.step_profile():
            a = acc.setdefault(s["variant"], [0.0, 0.0, 0, 0.0])
            a[0] += s["flops"]                                                             # algorithmic (direct-convolution) FLOPs
            a[1] += s["ms"]; a[2] += 1
            a[3] += s["flops"] / 2.25 if s["variant"] in (129, 130, 140) else s["flops"]   # what the matrix pipe needs for them
    eng.select_profile(0)
    eng.set_profiling(0)
    return dt, acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the config's)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the config's images per GPU on every GPU (default); strong: --global-batch images in total, "
                         "global/N per GPU (SURVEY.md 8d: global B = 64 at config 4)")
    ap.add_argument("--global-batch", type=int, default=64, help="strong scaling: images per step over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="no per-launch hipEvents in the timed region")
    ap.add_argument("--profile-every", type=int, default=10,
                    help="per-launch hipEvents on every n-th step of the timed region (a recorded event costs the GPU ~3 us: ~95 per "
                         "step are 1.3 %% of a config-4 step -- and a recorded forward and its successor do not run beside another "
                         "forward's heads, byolo_plan_opts.serialize_heads: 0.4 ms each); 1 = every step")
    ap.add_argument("--streams", type=int, default=1, help="split the per-GPU batch over this many concurrent HIP streams")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="alternate whole steps over this many HIP streams (own workspace and output buffers each): the latency-bound "
                         "tail of step i (decode, sort, NMS) overlaps the convolutions of step i+1.  1 = one stream")
    ap.add_argument("--fp32-steps", type=int, default=5, help="timed steps of the fp32_mode leg (0 = skip it)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 5-step timings of BASELINE configs 2, 3, 5 and the reference's default frame")
    ap.add_argument("--entry-frames", type=int, default=1536,
                    help="frames of the entry_point leg (inference_epistemic.inference over generated TFRecord shards); 0 = skip it")
    ap.add_argument("--quick-parity", action="store_true", help="skip the minute-long float32 oracle image of configs[4] (1024x1024, T=50)")
    ap.add_argument("--no-dropout", action="store_true",
                    help="[experiment, not the metric] skip the dropout masks: isolates the epilogue's RNG cost")
    ap.add_argument("--dump-steps", default=None, help="write the per-launch table (layer, variant, M, N, K, ms, TF/s) here")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from byolo import synth, dist as bdist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (MI355X); there is no CPU path in the product")
    rank, local, world = bdist.init()
    if world != args.gpus:        # a mis-launch must not produce a line that looks like an N-GPU number
        sys.exit("bench.py --gpus %d, but WORLD_SIZE=%d: launch N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                 "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ...` (one rank per GPU), N = 1 as plain `python bench.py`"
                 % (args.gpus, world, args.gpus, args.gpus))
    device = bdist.local_device(local) if world > 1 else 0        # (BYOLO_DIST_SHARE_DEVICE=1: every rank on cuda:0, a one-GPU box)
    if world > 1 and device >= torch.cuda.device_count():
        sys.exit("rank %d: LOCAL_RANK=%d but only %d GPUs are visible" % (rank, local, torch.cuda.device_count()))
    pg = dist.is_initialized()         # world > 1, or a forced one-rank group (BYOLO_DIST_FORCE=1)
    torch.cuda.set_device(device)
    pipeline_streams(device, max(2, args.pipeline))     # before any engine exists
    # preflight of the N > 1 path, before anything is built: what the BACKEND says the job is (not the environment), and one tiny
    # all-reduce over the link the all-gather will use -- a rank on the wrong GPU or a group of the wrong size shows up in the line
    # (`ranks[].nccl_world`, `ranks[].preflight_sum`) instead of as a hang in the timed region
    nccl_world, backend, preflight = 1, None, None
    if pg:
        nccl_world, backend = dist.get_world_size(), dist.get_backend()
        one = torch.ones(1, device="cuda:%d" % device) if backend == "nccl" else torch.ones(1)
        dist.all_reduce(one)
        preflight = float(one.item())
        if nccl_world != world or preflight != world:
            sys.exit("rank %d: the process group reports %d ranks (all-reduce of ones: %g), WORLD_SIZE is %d" % (rank, nccl_world, preflight, world))

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.scaling == "strong":
        assert args.global_batch % world == 0, "--global-batch must divide by the number of GPUs"
        cfg["B"] = args.global_batch // world
    m = build(cfg, device)
    eng = m.engine
    # the timed region must not wait inside byolo_forward: the split-f16 range status (include/byolo.h: byolo_status) is
    # read ONCE, after the run, and a raised status invalidates the line
    eng.set_async(True)
    B, T = cfg["B"], cfg["T"]
    x = torch.from_numpy(synth.synthetic_images(B, cfg["H"], cfg["W"], seed=1234, first_index=rank * B)).to("cuda:%d" % device)
    N, D = eng.num_boxes()
    cap = eng.out_cap
    out = {"rows": torch.empty((B, cap, D), device=x.device), "kept": torch.empty((B, cap), dtype=torch.int32, device=x.device),
           "count": torch.empty((B, 2), dtype=torch.int32, device=x.device)}

    nstreams = max(1, args.streams)
    assert B % nstreams == 0
    streams = [torch.cuda.Stream(device=x.device) for _ in range(nstreams)] if nstreams > 1 else []
    Bs = B // nstreams
    subs = [dict(x=x[k * Bs:(k + 1) * Bs], out={n: t[k * Bs:(k + 1) * Bs] for n, t in out.items()}) for k in range(nstreams)]

    # --pipeline P: whole steps alternate over P streams (own workspace slot and output buffers each), so the
    # latency-bound tail of step i (decode, NMS) and its launch ramps overlap the convolutions of step i+1
    npipe = max(1, args.pipeline)
    pipes = [dict(st=st, out={n: torch.empty_like(t) for n, t in out.items()}) for st in pipeline_streams(device, npipe)] if npipe > 1 else []

    def step(i):
        if npipe > 1:
            pp = pipes[i % npipe]
            with torch.cuda.stream(pp["st"]):
                r = eng.forward(x, T=T, seed=1000 + i, dropout_on=not args.no_dropout, want_boxes=False, want_nms=True,
                                out=pp["out"], slot=1 + i % npipe, first_image=rank * B)
                if pg:
                    return bdist.allgather_boxes(r["rows"], r["kept"], r["count"], world)
            return r["rows"], r["kept"], r["count"]
        if nstreams > 1:
            # the batch as `nstreams` independent sub-batches on separate HIP streams: one sub-batch's
            # kernel tails / launch gaps are filled by the other's kernels (images are independent)
            cur = torch.cuda.current_stream(x.device)
            for k, st in enumerate(streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    eng.forward(subs[k]["x"], T=T, seed=1000 + i, dropout_on=not args.no_dropout, want_boxes=False, want_nms=True,
                                out=subs[k]["out"], slot=k, first_image=rank * B + k * Bs)
            for st in streams:
                cur.wait_stream(st)
            r = out
        else:
            # rank r holds images [r*B, (r+1)*B) of the global batch and draws THEIR dropout masks: the N-GPU job
            # computes what one GPU would compute on the whole batch
            r = eng.forward(x, T=T, seed=1000 + i, dropout_on=not args.no_dropout, want_boxes=False, want_nms=True, out=out,
                            first_image=rank * B)
        if pg:
            return bdist.allgather_boxes(r["rows"], r["kept"], r["count"], world)
        return r["rows"], r["kept"], r["count"]

    # per-launch hipEvents: every forward owns one record slot of the handle's ring (byolo_set_profile_depth) and records its
    # events on ITS stream, so pipelined steps keep their own timings (a launch's time then includes what it shares with the
    # other stream's tail kernels)
    prof = not args.no_profile and nstreams == 1
    eng.set_profiling(2 if prof else 0)          # on during the warm-up as well: the event pools exist before the timed region
    for i in range(args.warmup):
        step(i)
    n_sub = -(-B // eng.max_images(T))           # byolo_forward calls per step (a batch beyond max_images runs as sub-batches)
    every = max(1, args.profile_every)
    n_prof = len(range(0, args.steps, every))       # profiled steps of the timed region: i = 0, every, 2 * every, ...
    if prof:
        eng.set_profile_depth(n_prof * n_sub)       # every profiled step's launch records stay readable until after the run
    eng.set_profiling(2 if prof else 0)
    acc = {}                      # variant -> [algorithmic flops, ms, launches, executed flops, useful flops, algorithmic read bytes, algorithmic write bytes]
    acc_cn = {}                   # Winograd variants -> bytes of the CONVOLUTIONS their launches stand for (input + weights + output)
    per_launch = {}
    stage = {"backbone": 0.0, "heads": 0.0, "decode": 0.0, "sort_nms": 0.0}
    if pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if prof and every > 1:
            eng.set_profiling(2 if i % every == 0 else 0, keep=True)
        step(args.warmup + i)
    torch.cuda.synchronize()
    if pg:
        dist.barrier()
    dt = time.perf_counter() - t0
    eng.set_profiling(2 if prof else 0, keep=True)
    if pg:
        t = torch.tensor([dt], dtype=torch.float64, device=x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if prof and rank == 0:
        # the timed region is over: read the K steps' records (age K-1 = the first timed step)
        WINO = (129, 130, 140, 141)
        for age in range(n_prof * n_sub - 1, -1, -1):
            eng.select_profile(age)
            for j, s in enumerate(eng.step_profile()):
                a = acc.setdefault(s["variant"], [0.0, 0.0, 0, 0.0, 0.0, 0.0, 0.0])
                wino = s["variant"] in WINO
                if s["variant"] == 4256:      # 3x3 + fused follower: {M, N, K} describe the 3x3; the launch's FLOPs hold both (nothing padded)
                    n2 = (s["flops"] - 2.0 * s["M"] * s["N"] * s["K"]) / (2.0 * s["M"] * s["N"])      # the follower's output channels
                    s = dict(s, flops_executed=s["flops"])
                a[0] += s["flops"]; a[1] += s["ms"]; a[2] += 1; a[3] += s["flops_executed"]
                if s["variant"] in (130, 140):      # the convolution's OWN bytes of a Winograd launch: input + 3x3 weights + output, once each
                    px = s["flops"] / (18.0 * s["K"] * s["N"])
                    acc_cn[s["variant"]] = acc_cn.get(s["variant"], 0.0) + 4.0 * (px * s["K"] + 9.0 * s["K"] * s["N"] + px * s["N"])
                a[4] += s["flops"] / (1.5 if s["variant"] == 141 else 2.25) if wino else s["flops"]
                # algorithmic bytes of the launch: A operand once + result once + weights once
                # (a shared-tap 3x3 launch reads its input once: M * K / 9 elements, not the im2col matrix)
                # (input once + weights once | result once)
                a[5] += 4.0 * ((s["M"] * s["K"] // 3 + 4 * s["K"] * s["N"]) if s["variant"] == 141 else       # 1-D: V [4][rows][C] once + U [4][3C][N]
                               (s["M"] * s["K"] + 16 * s["K"] * s["N"]) if s["variant"] in (130, 140) else
                               (s["M"] * s["K"] // 9 + s["K"] * s["N"]) if s["variant"] in (4256, 3256, 3128, 3064) else
                               (s["M"] * s["K"] + (16 if wino else 1) * s["K"] * s["N"]))
                a[6] += 4.0 * ((s["M"] // 2) * s["N"] if s["variant"] == 141 else (s["M"] // 4) * s["N"] if s["variant"] in (130, 140) else s["M"] * (n2 if s["variant"] == 4256 else s["N"]))
                if s["variant"] == 4256:
                    a[5] += 4.0 * s["N"] * n2                # + the follower's weights
                if args.dump_steps:
                    per_launch.setdefault(j, dict(s, ms=0.0))["ms"] += s["ms"] / n_prof
            for k, v in eng.stage_ms().items():
                stage[k] += v
        eng.select_profile(0)
    # per-rank diagnostics for the N > 1 line: which device ran which block of the global batch, and a checksum of the last
    # step's kept indices (a wrong shard or a rank on the wrong GPU shows up as a duplicate / missing first_image or checksum)
    last = step(args.warmup + args.steps)
    torch.cuda.synchronize()
    g_kept, g_count = last[1], last[2]
    cs = [int((g_kept[b].to(torch.int64).clamp(min=0) * torch.arange(1, g_kept.shape[1] + 1, device=g_kept.device)).sum().item() % 1000003)
          for b in range(g_kept.shape[0])]
    mine = {"rank": rank, "device": "cuda:%d (%s)" % (device, torch.cuda.get_device_name(device)), "first_image": rank * B, "images": B,
            "backend": backend, "nccl_world": nccl_world, "preflight_sum": preflight,
            "host_threads": {"cpu_count": os.cpu_count(), "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)}}
    ranks = [None] * world
    if pg:
        dist.all_gather_object(ranks, mine)
    else:
        ranks = [mine]
    range_flags, range_layer = eng.status()
    assert range_flags == 0 or os.environ.get("BYOLO_LIB"), "an activation left the split-f16 range during the benchmark (layer %d): the number would be invalid" % range_layer
    if rank == 0:
        imgs = world * B * args.steps
        flops_img = eng.flops(1, T)
        line = {
            "metric": "img/s at T=%d MC-dropout, %dx%d" % (T, cfg["H"], cfg["W"]),
            "value": imgs / dt, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "ranks": ranks, "kept_checksum_per_image_of_one_more_step": cs,
            "vs_baseline": None, "dtype": "f32" if eng.precision == "f32" else "f32 via split-f16 (hi+lo fp16 pairs, 3 fp16 MFMA products, fp32 accumulate)", "data": "synthetic" + (" [EXPERIMENT: dropout off, invalid]" if args.no_dropout else ""),
            "config": {"workload": "%s: %s %dx%d T=%d, %d images/GPU (global batch %d), "
                                   "class-%s NMS max_out=1000, random-init weights with device-calibrated BN"
                                   % ("BASELINE configs[%d]" % (args.config - 1) if args.config <= 5 else "reference default frame",
                                      cfg["variant"], cfg["H"], cfg["W"], T, B, B * world,
                                      "wise 2-class" if cfg["nms"] else "agnostic"),
                       "images_per_gpu": B, "T": T, "img_size": [cfg["H"], cfg["W"]], "parallelism": "dp%d" % world,
                       "gflop_per_image": flops_img / 1e9, "precision": eng.precision,
                       "precision_note": ("split-f16: every activation / weight as hi + lo fp16 (23 significant bits; fp32 has 24), one exact "
                                          "power-of-two scale per output channel of every filter bank, products hi*hi + hi*lo + lo*hi on "
                                          "v_mfma_f32_32x32x16_f16 into fp32 accumulators.  Narrower than the reference's float32 in RANGE: an "
                                          "activation beyond +-16376 is detected and reported as BYOLO_ERR_RANGE (`range_status`; the entry points "
                                          "then re-run in the fp32 mode), never stored.  Accuracy: output rows within the 1e-4 bound of the float32 "
                                          "oracle at this shape (`parity_note`, measured in this run); deep intermediate taps of small networks sit "
                                          "at 0.97 - 1.08 of that bound from the float32 fixtures while being closer to the float64 oracle than those "
                                          "fixtures are (tests/test_gpu_parity.py, DESIGN.md section 5).  `fp32_mode` times the reference's own "
                                          "arithmetic in this run") if eng.precision == "split" else
                                         "fp32 operands on v_mfma_f32_32x32x2_f32, Winograd F(2x2,3x3) on the large 3x3 layers"},
        }
        SPLIT = (4256, 3256, 3128, 3064, 2128, 2064, 1128, 1064, 1032, 140)
        KERNELS = {4256: "conv_igemm_kernel<128,256,1,8,kx3>+fused_tail (split-f16 3x3/stride-1 on shared-tap stages with the following 1x1 convolution / detection head fused in; BYOLO_B2B)",
                   3256: "conv_igemm_kernel<128,256,1,8,kx3> (split-f16 3x3/stride-1 on shared-tap stages, 8 waves; BYOLO_KX3_WIDE)",
                   3128: "conv_igemm_kernel<128,128,1,4,kx3> (split-f16 3x3/stride-1 on shared-tap stages, v_mfma_f32_32x32x16_f16 x3)",
                   3064: "conv_igemm_kernel<128,64,2,2,kx3> (split-f16 3x3/stride-1 on shared-tap stages)",
                   2128: "conv_igemm_kernel<128,128,1,4,p1> (split-f16 1x1 convolutions on the uniform loop)",
                   2064: "conv_igemm_kernel<128,64,2,2,p1> (split-f16 1x1 convolutions / detection heads on the uniform loop)",
                   1128: "conv_igemm_kernel<128,128,1,4,split> (split-f16 1x1 / stride-2 / two-source convolutions)",
                   1064: "conv_igemm_kernel<128,64,2,2,split>", 1032: "conv_igemm_kernel<128,32,4,1,split>",
                   140: "wino_split_kernel (Winograd F(2x2,3x3) in split-f16: transform-domain GEMM + output transform + epilogue, v_mfma_f32_32x32x16_f16 x3; workgroups walk the unit list)",
                   130: "wino_fused_kernel (Winograd-domain GEMM + output transform + epilogue, fp32 v_mfma_f32_32x32x2_f32)",
                   129: "gemm_stream_kernel<128,0> (Winograd-domain GEMM, fp32 v_mfma_f32_32x32x2_f32)",
                   131: "gemm_stream_kernel<128,1> (row-streaming 1x1 convolution)",
                   132: "gemm_stream_kernel<64,*> (row-streaming 1x1 convolution / detection head, 64-wide tile)",
                   128: "conv_igemm_kernel<128,128,2,2,*> (fp32 v_mfma_f32_32x32x2_f32)",
                   64: "conv_igemm_kernel<128,64,2,2,*>", 32: "conv_igemm_kernel<128,32,4,1,*>"}
        mm = [v for v in acc if v in KERNELS]
        if prof and mm:
            # dominant kernel = the matrix-pipe kernel with the most time in the timed region.  achieved = EXECUTED
            # matrix-pipe FLOPs of its launches / their time.  For the direct convolutions that is the algorithmic 2*M*N*K;
            # a Winograd-domain GEMM launch executes 1/2.25 of the direct-convolution FLOPs it stands for (x tile padding),
            # and its transforms are separate, HBM-bound launches.
            dom = max(mm, key=lambda v: acc[v][1])
            f, ms, n, fx, fu, abr, abw = acc[dom]
            tot_f = sum(a[0] for a in acc.values()); tot_ms = sum(a[1] for a in acc.values())
            # achieved: the ALGORITHMIC FLOPs the launches stand for (direct-convolution 2*M*N*K per layer; a Winograd
            # launch carries the direct-convolution FLOPs of its samples, although its matrix pipe executes 1 / 2.25 of them)
            ach = f / (ms * 1e-3)
            split = dom in SPLIT
            # peak of the matrix instruction actually issued, useful FLOPs counted once
            peak = PEAK_F16_MFMA if split else PEAK_FP32_MFMA
            wino_ms = sum(acc[v][1] for v in (-2, -3, -4, -6) if v in acc)
            # fabric bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes over this same command
            # (tools/profile_round.sh + tools/pmc_traffic.py -> profiles/traffic_cfgN.json); null if absent / another kernel
            traffic = tr_read = tr_write = measured_at = refused = None

            def fresh(tj):
                """A traffic file belongs to the kernel it was measured on: tools/pmc_traffic.py records the SHA-256 of the kernel's
                source files, and a file whose sources have changed since (or that carries none) is REFUSED, not quoted (VERDICT r5
                "missing" 6: the round-5 line quoted the transform traffic of a kernel that had been replaced)."""
                import hashlib
                src = (tj.get("measured_at") or {}).get("sources")
                if not src:
                    return "no source hashes recorded (measured before round 6)"
                for f, h in src.items():
                    try:
                        if hashlib.sha256(open(os.path.join(REPO, f), "rb").read()).hexdigest()[:16] != h:
                            return "%s has changed since the measurement (%s)" % (f, (tj.get("measured_at") or {}).get("commit"))
                    except OSError:
                        return "%s is gone" % f
                return None
            # one file per kernel that has been the dominant one: profiles/traffic_cfg<N>_<tag>.json
            tag = {4256: "b2b", 3128: "kx3", 140: "wino", 130: "wino_fused"}.get(dom, "v%d" % dom)
            tpath = os.path.join(REPO, "profiles", "traffic_cfg%d_%s.json" % (args.config, tag))
            if os.path.exists(tpath) and not args.batch and args.scaling == "weak":
                tj = json.load(open(tpath))
                want = {4256: "conv_igemm_kernel<128, 256, 1, 8, true, true, 1", 3128: "conv_igemm_kernel<128, 128, 1, 4, true, true, 1", 140: "wino_split_kernel", 130: "wino_fused_kernel"}.get(dom)
                if want and (want in tj.get("kernel", "") or tj.get("kernel", "") in want):
                    refused = fresh(tj)
                    if refused is None:
                        traffic = tj.get("traffic_bytes_per_launch")
                        tr_read, tr_write = tj.get("read_bytes_per_launch"), tj.get("write_bytes_per_launch")
                    measured_at = tj.get("measured_at")
            ab_r, ab_w = abr / n, abw / n
            # the CONVOLUTION's own bytes per launch: input once + weights once + output once (for a Winograd
            # launch `algorithmic_bytes_per_launch` counts the 4x-expanded transform-domain V and U as its operands -- what THIS
            # kernel must read -- which flatters the layer: V is a by-product of the method, written and re-read on top of the input)
            conv_bytes = None
            if dom in (130, 140):
                conv_bytes = acc_cn[dom] / n                          # (pixels of a launch = its direct FLOPs / (2 * 9 * C * N))
            tr_in = None                                              # the separate input-transform launch's measured traffic, if profiled
            ipath = os.path.join(REPO, "profiles", "traffic_cfg%d_wino_input.json" % args.config)
            if dom == 140 and os.path.exists(ipath) and traffic:
                ij = json.load(open(ipath))
                tr_in = ij.get("traffic_bytes_per_launch") if fresh(ij) is None else None
            line["roofline"] = {"bound": "mfma", "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                                "frac": ach / peak,
                                "traffic": traffic, "traffic_read": tr_read, "traffic_write": tr_write, "traffic_measured_at": measured_at,
                                "traffic_refused": refused,
                                "algorithmic_bytes_per_launch": ab_r + ab_w,
                                "algorithmic_read_bytes_per_launch": ab_r, "algorithmic_write_bytes_per_launch": ab_w,
                                "traffic_vs_algorithmic": (traffic / (ab_r + ab_w)) if traffic else None,
                                "convolution_bytes_per_launch": conv_bytes,
                                "traffic_vs_convolution_bytes": (traffic / conv_bytes) if traffic and conv_bytes else None,
                                "input_transform_traffic_per_launch": tr_in,
                                "traffic_with_input_transform_vs_convolution_bytes": ((traffic + tr_in) / conv_bytes) if traffic and tr_in and conv_bytes else None,
                                "read_amplification": (tr_read / ab_r) if tr_read else None,
                                "write_amplification": (tr_write / ab_w) if tr_write else None,
                                "kernel": KERNELS[dom],
                                "launches": n, "profiled_steps": "%d of %d (every %d-th step of the timed region records per-launch hipEvents; a recorded forward and the one after it wait for the "
                                                  "other stream's WHOLE convolution stack, so these launches run alone on the device -- the other steps run their heads beside the next "
                                                  "step's backbone (byolo_plan_opts.serialize_heads: faster steps, longer single launches; tools/rocprof_summary.py lists both averages)" % (n_prof, args.steps, every),
                                "avg_launch_ms": ms / n, "share_of_conv_flops": f / tot_f,
                                "definition": "SURVEY.md 8(d): achieved = algorithmic fp32-equivalent FLOPs of the launches (2MNK of the convolution as written; a "
                                              "Winograd launch stands for the direct-convolution FLOPs of its samples; tile padding not counted), each counted ONCE, "
                                              "/ hipEvent time of the launches on their launch stream; peak = dense peak of the matrix instruction issued ("
                                              + ("v_mfma_f32_32x32x16_f16, 2500: a split-f16 kernel executes three fp16 products per useful product, so its "
                                                 "frac cannot exceed 1/3" if split else "v_mfma_f32_32x32x2_f32, 157.3") + ")",
                                # what the matrix pipe executed for them (padded extents; Winograd: 16 transform-domain GEMMs = direct / 2.25)
                                "achieved_executed_padded": fx / (ms * 1e-3) / 1e12,
                                "share_of_conv_time": ms / tot_ms, "winograd_transform_share_of_conv_time": wino_ms / tot_ms,
                                # algorithmic (direct-convolution) FLOPs of the whole conv stack / its time, transforms included
                                "all_conv_algorithmic": tot_f / (tot_ms * 1e-3) / 1e12,
                                "by_kernel": {KERNELS[v].split(" ")[0]: {"launches": acc[v][2], "ms": acc[v][1],
                                                                          "algorithmic_tflops": acc[v][0] / (acc[v][1] * 1e-3) / 1e12,
                                                                          "executed_tflops": acc[v][3] / (acc[v][1] * 1e-3) / 1e12}
                                              for v in sorted(mm, key=lambda v: -acc[v][1])},
                                "transform_kernels_ms": {str(v): acc[v][1] for v in (-2, -3, -4, -6) if v in acc},
                                # 8(d)'s formula for the whole step: img/s * F(H,W,T) / (n_gpu * peak); and against the fp32 MFMA peak the
                                # reference's own arithmetic would be priced at (> 1 = beyond that instruction's ceiling)
                                "end_to_end_frac": (imgs / dt) * flops_img / (world * peak),
                                "end_to_end_vs_fp32_mfma_peak": (imgs / dt) * flops_img / (world * PEAK_FP32_MFMA)}
            if split:
                line["roofline"].update({"frac_executed_vs_fp16_peak": 3.0 * fx / (ms * 1e-3) / PEAK_F16_MFMA,
                                         "fp16_executed_tflops": 3.0 * fx / (ms * 1e-3) / 1e12,
                                         "fp16_sustained_under_power_cap_tflops": SUSTAINED_F16_MFMA / 1e12,
                                         "frac_executed_of_sustained": 3.0 * fu / (ms * 1e-3) / SUSTAINED_F16_MFMA})
            if npipe > 1:
                line["roofline"]["note"] = ("steps alternate over %d HIP streams; the library runs a handle's convolution stacks one after the other "
                                            "(byolo_api.hip ev_convs), so a convolution launch shares the chip only with the previous step's tail "
                                            "kernels (decode, sort, NMS: a few hundred microseconds of small launches)" % npipe)
            line["stage_ms_per_step"] = {k: v / n_prof for k, v in stage.items()}
            if npipe > 1:       # stage events sit on the step's own stream: the first stage includes the wait for the previous step's convolutions
                line["stage_ms_per_step"]["note"] = "pipelined steps: `backbone` includes waiting for the previous step's convolution stack (byolo_api.hip ev_convs)"
        # the reference's own arithmetic (float32, lib_yolo/layers.py:550) timed in the SAME run: a second handle in BYOLO_PREC_F32
        # on the same batch, a few steps on one stream after the headline region (the headline engine stays alive: its weights)
        if world == 1 and args.fp32_steps > 0 and eng.precision != "f32" and not args.batch and args.scaling == "weak":
            try:
                m32 = build(cfg, device, precision="f32")
                dt32, acc32 = time_steps(m32.engine, x, cfg, args.fp32_steps, 2)
                d32 = max((v for v in acc32 if v in KERNELS), key=lambda v: acc32[v][1])
                a32 = acc32[d32][0] / (acc32[d32][1] * 1e-3)
                line["fp32_mode"] = {"value": B * args.fp32_steps / dt32, "unit": "img/s", "ms_per_step": 1e3 * dt32 / args.fp32_steps,
                                     "steps": args.fp32_steps, "warmup": 2, "streams": 1, "precision": m32.engine.precision,
                                     "dtype": "f32 (v_mfma_f32_32x32x2_f32, Winograd F(2x2,3x3) on the large 3x3 layers)",
                                     "dominant_kernel": KERNELS[d32], "launches": acc32[d32][2],
                                     # frac = what the matrix pipe EXECUTED / its peak (a Winograd launch executes 1 / 2.25 of the direct
                                     # convolution it stands for, so the algorithmic rate of such a kernel may exceed the instruction's peak:
                                     # that figure is kept as a named extra, not as a fraction of the roofline)
                                     "achieved": acc32[d32][3] / (acc32[d32][1] * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA / 1e12,
                                     "frac": acc32[d32][3] / (acc32[d32][1] * 1e-3) / PEAK_FP32_MFMA,
                                     "algorithmic_direct_convolution_tflops": a32 / 1e12,
                                     "headline_speedup_over_fp32_mode": (imgs / dt) / (B * args.fp32_steps / dt32)}
                m32.engine.close()
                del m32
            except Exception as e:
                line["fp32_mode"] = {"value": None, "error": repr(e)}
        if world == 1 and not args.no_other_configs and args.config == 4 and not args.batch and args.scaling == "weak":
            line["other_configs"] = {}
            for num, name, orc in ((1, "configs[0]", "f64"), (2, "configs[1]", "f64"), (3, "configs[2]", "f64"), (5, "configs[4]", "f32"),
                                   (6, "reference default frame (inference_epistemic.py:218-221)", False),
                                   (7, "reference default aleatoric workload (inference_aleatoric.py:219-227: 1024x1920, batch_size 11)", False),
                                   (8, "reference default standard workload (inference_standard_yolov3.py:210-218: 1024x1920, batch_size 11)", False)):
                try:
                    line["other_configs"][name] = other_config_leg(num, device, oracle=orc and not args.no_cpu_baseline and (orc if num != 5 or not args.quick_parity else False))
                except Exception as e:
                    line["other_configs"][name] = {"img_s": None, "error": repr(e)}
        if world == 1 and args.entry_frames > 0 and not args.batch and args.scaling == "weak":
            try:
                line["entry_point"] = entry_point_leg(cfg, device, n_frames=args.entry_frames)
                line["entry_point"]["vs_value"] = line["entry_point"]["img_s"] / line["value"]
                if line["entry_point"].get("steady_img_s"):
                    line["entry_point"]["steady_vs_value"] = line["entry_point"]["steady_img_s"] / line["value"]
            except Exception as e:
                line["entry_point"] = {"img_s": None, "error": repr(e)}
            # the reference's own default workload through the same entry point: full ECP frame, T = 50, batch_size = 1
            # (inference_epistemic.py:212-240; class-agnostic NMS as the reference runs it) -- a short run
            if args.config == 4 and not args.no_other_configs:
                try:
                    c6 = dict(CONFIGS[6], nms=0)
                    # 256 frames (round 4: 48 frames = 1.3 s, a fifth of it pipeline fill -- too short to mean anything)
                    ep6 = entry_point_leg(c6, device, n_frames=256, distinct=8, extras=False)
                    dev6 = (line.get("other_configs") or {}).get("reference default frame (inference_epistemic.py:218-221)", {}).get("img_s")
                    ep6["vs_device_only"] = (ep6["img_s"] / dev6) if dev6 else None
                    ep6["steady_vs_device_only"] = (ep6["steady_img_s"] / dev6) if dev6 and ep6.get("steady_img_s") else None
                    line["entry_point_reference_default"] = ep6
                except Exception as e:
                    line["entry_point_reference_default"] = {"img_s": None, "error": repr(e)}
        if world == 1 and not args.no_other_configs and args.config == 4 and not args.batch and args.scaling == "weak":
            try:
                line["t_shard_latency_path"] = t_shard_leg(device)
            except Exception as e:
                line["t_shard_latency_path"] = {"error": repr(e)}
        if world == 1 and not args.no_other_configs and not args.batch and args.scaling == "weak":
            try:
                line["training_side"] = training_side_leg(cfg, device)
            except Exception as e:
                line["training_side"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"], o_rows, o_kept = cpu_baseline(cfg, eng.get_params())
                eng.set_profiling(0)
                line["parity_note"] = parity_check(cfg, eng, x, o_rows, None, o_kept)
            except Exception as e:          # the CPU leg must never cost the GPU number
                line.setdefault("cpu_baseline", {"value": None, "unit": "img/s", "cores": os.cpu_count(), "kind": "port",
                                                 "sample": "failed: %r" % (e,)})
                line["parity_note"] = {"error": repr(e), "ok": None}
        line["range_status"] = "ok (byolo_status after the timed region: no activation left the split-f16 range)" if eng.precision == "split" else "n/a (fp32 mode)"
        # every parity leg of the line against its stated allowance (the line used to print 1.27 and exit 0)
        legs = {"parity_note": line.get("parity_note")}
        legs.update({"other_configs/" + k: v.get("parity") for k, v in (line.get("other_configs") or {}).items() if isinstance(v, dict)})
        failed = sorted(k for k, v in legs.items() if isinstance(v, dict) and v.get("ok") is False)
        checked = sorted(k for k, v in legs.items() if isinstance(v, dict) and v.get("ok") is not None)
        # null, not true, when no leg ran (--no-cpu-baseline: the A/B scripts under tools/): a build that corrupted its activations
        # read "parity_ok": true through such a run once (profiles/r6_wino_small.md); A/Bs are held to tools/rows_digest.py
        line["parity_ok"] = (not failed) if checked else None
        line["parity_legs_checked"] = len(checked)
        if failed:
            line["parity_failed"] = failed
        print(json.dumps(line))
        parity_exit = 3 if failed else 0
        if args.dump_steps and per_launch:
            with open(args.dump_steps, "w") as f:
                f.write("| # | layer | variant | M | N | K | ms | executed TF/s | algorithmic TF/s |\n|---|---|---|---|---|---|---|---|---|\n")
                for j in sorted(per_launch):
                    s = per_launch[j]
                    f.write("| %d | %d | %d | %d | %d | %d | %.4f | %.1f | %.1f |\n" % (
                        j, s["layer"], s["variant"], s["M"], s["N"], s["K"], s["ms"],
                        s["flops_executed"] / (s["ms"] * 1e-3) / 1e12, s["flops"] / (s["ms"] * 1e-3) / 1e12))
    if pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity_exit:
        sys.stderr.write("bench.py: parity beyond the stated allowance in %s (see the line's parity fields)\n" % ", ".join(failed))
        sys.exit(parity_exit)


if __name__ == "__main__":
    main()
