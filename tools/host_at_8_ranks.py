#!/usr/bin/env python
"""The HOST side of an 8-GPU job, loaded the way 8 ranks will load it -- without 8 GPUs.

    python tools/host_at_8_ranks.py [--world 8] [--frames-per-rank 768] [--img-s-per-rank 380] [--size 608] [--out profiles/...md]

Starts `world` processes of the product's driver loop (byolo/inference.py InferenceLoop through inference_epistemic.Inference, the
same code path `torchrun --nproc-per-node 8 inference_epistemic.py` runs) as a gloo job on this host.  Everything on the host side
is REAL: TFRecord shards of 608 x 608 PNG frames (uniform noise: the worst case for inflate), the native decode pool with its
per-rank thread cap (cpu_count / LOCAL_WORLD_SIZE), uint8 frames, two batches in flight, ONE all-gather per global batch carrying
every rank's padded box list (1000 rows x 23 columns per image, the reference's `max_output_size`), the native ECP-JSON formatter
and `writer_threads` file writers per rank (~0.8 MB of JSON per frame).  Only the GPU is a stand-in: an engine that hands out 1000
finished rows per image and takes `8 / img_s_per_rank` seconds per 8-image block, the pace of one MI355X at BASELINE configs[3].

Reported per rank and in aggregate: images / s of the loop (steady state, fill excluded), seconds the loop waited for the feed / the
device / the writer, PNG MB/s decoded, JSON MB/s written; then the feed ALONE and the writer ALONE at 8 concurrent ranks.  The
first stage that cannot keep up with world x img_s_per_rank is the job's host-side bottleneck.  What the stand-in cannot show: RCCL
over xGMI (gloo over loopback moves the same bytes through the host's memory instead, which is MORE host load than the real job
has), PCIe H2D / D2H, and the GPUs themselves.
"""
import argparse
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "bayesian-yolov3_amd")


# ---------------------------------------------------------------------------------------------------------------------
# the stand-in for one GPU
# ---------------------------------------------------------------------------------------------------------------------
class PacedEngine:
    torch_device = "cpu"
    out_cap = 1000
    precision = "split"

    def __init__(self, device, block_s):
        self.device = device
        self.block_s = block_s
        self.next_free = 0.0

    def num_boxes(self):
        return 22743, 23

    def param_shapes(self):
        return {}

    def set_params(self, *a, **k):
        pass

    def finalize(self):
        pass

    def calibrate_bn(self, x):
        pass

    def set_async(self, on=True):
        self._async = on

    def normalize_u8(self, u8, out=None):        # on the device in the real job: free for the host
        return u8 if out is None else out

    def copy_status(self, out):
        out[0] = 0
        out[1] = -1

    def clear_status(self):
        pass


class PacedModel:
    cls_cnt, obj_idx, cls_start_idx, T = 2, 14, 17, 30

    def __init__(self, device, block_s):
        import numpy as np
        import torch
        self.engine = PacedEngine(device, block_s)
        rng = np.random.default_rng(7)
        rows = rng.random((1000, 23), dtype=np.float32)
        rows[:, 0:2] *= 0.5
        rows[:, 2:4] = rows[:, 0:2] + 0.1 + 0.4 * rows[:, 2:4]
        rows[:, 14] = np.sort(rows[:, 14])[::-1]
        self.rows = torch.from_numpy(rows.copy())

    def finalize(self):
        pass

    def run(self, x, seed=0, want_boxes=True, first_image=0, out=None, precision=None, **kw):
        """One block of images: the rows of 1000 kept boxes per image appear in `out`; the call returns when the stand-in device
        would have finished the block -- a device is a resource with a throughput: blocks are `block_s` apart, whatever else the
        host did in between."""
        eng = self.engine
        n = int(x.shape[0])
        out["rows"][:n] = self.rows
        out["kept"][:n] = 0
        out["count"][:n] = 1000
        now = time.perf_counter()
        eng.next_free = max(eng.next_free, now) + eng.block_s * n / 8.0
        ahead = eng.next_free - eng.block_s * n / 8.0 - now        # the device is busy until then with what was enqueued before
        if ahead > 0:
            time.sleep(ahead)
        return {"engine": eng}


class PacedYolo:
    def __init__(self, block_s):
        self.block_s = block_s
        self.options = {}
        self.model = None

    def set_engine_option(self, k, v):
        self.options[k] = v

    def init_model(self, inputs=None, training=False):
        self.model = PacedModel(self.options.get("device", 0), self.block_s)
        return self

    def get_model(self):
        return self.model


def worker(rank, world, port, data_dir, out_path, size, batch, img_s, threads, writers, mode):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    sys.path.insert(0, REPO)
    sys.path.insert(0, PKG)
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from lib_yolo import dataset_utils as du, yolov3
    cfg = {"batch_size": batch, "full_img_size": [size, size, 3], "crop": False, "cls_cnt": 2, "implicit_background_class": True, "weights": "synthetic",
           "seed": 3, "inference_mode": True, "T": 30, "cpu_thread_cnt": threads, "writer_threads": writers, "out_path": out_path,
           "priors": yolov3.ECP_9_PRIORS, "data": {"file_pattern": os.path.join(data_dir, "val-*")}}
    res = {"rank": rank}
    if mode == "loop":
        import inference_epistemic as ie
        loop = ie.Inference(PacedYolo(8.0 / img_s), cfg)
        st = loop.run()
        res.update(images=st["images"], loop_s=st["loop_s"], steady_img_s=st.get("steady_img_s"), batches=st["batches"],
                   waited={"feed": st["wait_feed_s"], "device": st["wait_device_s"], "writer": st["wait_writer_s"]}, native_json=st["native_json"],
                   decode_threads=loop.dataset.threads)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
    elif mode == "feed":                                       # the feed alone: this rank's block of every global batch, frames dropped
        t0 = time.perf_counter()
        n = 0
        ds = du.TestingDataset(cfg)
        for sh in ds.iter_shards_u8(rank, world):
            n += len(sh.names)
            sh.release()
        res.update(images=n, loop_s=time.perf_counter() - t0, decode_threads=ds.threads)
    else:                                                      # the writer alone: this rank's share of the files, 1000 boxes each
        from concurrent.futures import ThreadPoolExecutor
        from byolo import hostio
        from byolo import inference as binf
        fmt = hostio.EcpJsonFormatter("bayesian_yolov3_aleatoric", [size, size, 3], 2, 14, 17, True, binf.LABEL_TO_CLS_NAME)
        rows = PacedModel(0, 0.0).rows.numpy()
        wdir = os.path.join(out_path + "_w", "r%d" % rank)
        os.makedirs(wdir, exist_ok=True)
        n = int(mode.split(":")[1])

        def one(i):
            with open(os.path.join(wdir, "%06d.json" % i), "wb") as f:
                f.write(fmt.format(rows))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=writers) as pool:
            list(pool.map(one, range(n)))
        res.update(images=n, loop_s=time.perf_counter() - t0, json_mb=sum(os.path.getsize(os.path.join(wdir, f)) for f in os.listdir(wdir)) / 1e6)
    json.dump(res, open(os.path.join(data_dir, "res_%s_%d.json" % (mode.split(":")[0], rank)), "w"))


def launch(world, mode, args, data_dir, out_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), str(port), data_dir, out_path, str(args.size),
                               str(args.batch), str(args.img_s_per_rank), str(args.threads), str(args.writers), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=args.timeout)[0] for p in procs]
    wall = time.perf_counter() - t0
    for p, o in zip(procs, outs):
        if p.returncode:
            raise RuntimeError("a %s worker failed (rc %d):\n%s" % (mode, p.returncode, o[-3000:]))
    return [json.load(open(os.path.join(data_dir, "res_%s_%d.json" % (mode.split(":")[0], r)))) for r in range(world)], wall


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        a = sys.argv[2:]
        return worker(int(a[0]), int(a[1]), int(a[2]), a[3], a[4], int(a[5]), int(a[6]), float(a[7]), int(a[8]), int(a[9]), a[10])
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames-per-rank", type=int, default=768)
    ap.add_argument("--img-s-per-rank", type=float, default=380.0)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--threads", type=int, default=24, help="config['cpu_thread_cnt'] (the reference's default: 24); the product caps it per rank")
    ap.add_argument("--writers", type=int, default=4)
    ap.add_argument("--timeout", type=int, default=1200)
    ap.add_argument("--tmp", default=None)
    ap.add_argument("--out", default=None, help="markdown report")
    args = ap.parse_args()
    args.batch = 8 * args.world                              # BASELINE configs[3]: 8 images per GPU
    sys.path.insert(0, REPO)
    sys.path.insert(0, PKG)
    import numpy as np
    from PIL import Image
    from byolo import synth
    from lib_yolo import dataset_utils as du
    tmp = tempfile.mkdtemp(prefix="byolo_host8_", dir=args.tmp)
    try:
        n_frames = args.frames_per_rank * args.world
        t0 = time.perf_counter()
        frames = (synth.synthetic_images(args.distinct, args.size, args.size, seed=1234) * 256.0).astype(np.uint8)
        enc = []
        for f in frames:
            b = io.BytesIO()
            Image.fromarray(f).save(b, format="PNG", compress_level=1)
            enc.append(b.getvalue())
        n_shards = 8
        for k in range(n_shards):
            du.write_tfrecords(os.path.join(tmp, "val-%05d-of-%05d" % (k, n_shards)),
                               (du.make_example({"image/encoded": enc[i % args.distinct], "image/filename": "frame_%06d.png" % i,
                                                 "image/height": args.size, "image/width": args.size}) for i in range(k, n_frames, n_shards)))
        gen_s = time.perf_counter() - t0
        png_mb = sum(len(enc[i % args.distinct]) for i in range(n_frames)) / 1e6
        rep = {"host": {"cores": os.cpu_count(), "tmp": tmp, "free_gb": shutil.disk_usage(tmp).free / 1e9},
               "workload": {"world": args.world, "frames": n_frames, "global_batch": args.batch, "img_size": args.size, "png_mb": png_mb,
                            "target_img_s": args.world * args.img_s_per_rank, "records_generated_in_s": gen_s}}
        res, wall = launch(args.world, "loop", args, tmp, os.path.join(tmp, "out", "run"))
        out_dir = os.path.join(tmp, "out", "run_0")
        files = os.listdir(out_dir)
        assert len(files) == n_frames, "%d files for %d frames" % (len(files), n_frames)
        json_mb = sum(os.path.getsize(os.path.join(out_dir, f)) for f in files) / 1e6
        loop_s = max(r["loop_s"] for r in res)
        rep["loop"] = {"ranks": res, "wall_s": wall, "loop_s_max": loop_s, "img_s": n_frames / loop_s,
                       "steady_img_s_sum": sum(r["steady_img_s"] or 0.0 for r in res), "json_mb": json_mb, "json_mb_s": json_mb / loop_s, "png_mb_s": png_mb / loop_s}
        res, wall = launch(args.world, "feed", args, tmp, os.path.join(tmp, "out2", "run"))
        t = max(r["loop_s"] for r in res)
        rep["feed_alone"] = {"img_s": n_frames / t, "png_mb_s": png_mb / t, "decode_threads_per_rank": res[0]["decode_threads"], "seconds": t}
        res, wall = launch(args.world, "writer:%d" % args.frames_per_rank, args, tmp, os.path.join(tmp, "out3", "run"))
        t = max(r["loop_s"] for r in res)
        rep["writer_alone"] = {"img_s": n_frames / t, "json_mb_s": sum(r["json_mb"] for r in res) / t, "writer_threads_per_rank": args.writers, "seconds": t}
        print(json.dumps(rep))
        if args.out:
            tgt = rep["workload"]["target_img_s"]
            with open(args.out, "w") as f:
                f.write("# The host at %d ranks (tools/host_at_8_ranks.py; stand-in engines paced at %.0f img/s each, everything else real)\n\n" % (args.world, args.img_s_per_rank))
                f.write("Host: %d hardware threads.  Workload: %d frames of %d x %d (PNG, uniform noise: %.0f MB), global batch %d, target %.0f img/s.\n\n"
                        % (os.cpu_count(), n_frames, args.size, args.size, png_mb, args.batch, tgt))
                f.write("| stage | img/s | of the target | MB/s | note |\n|---|---|---|---|---|\n")
                L = rep["loop"]
                f.write("| the %d driver loops together (gloo all-gather of every rank's padded box list per batch) | %.0f (steady state, summed: %.0f) | %.2f | PNG %.0f in, JSON %.0f out | %d JSON files, %.0f MB |\n"
                        % (args.world, L["img_s"], L["steady_img_s_sum"], L["steady_img_s_sum"] / tgt, L["png_mb_s"], L["json_mb_s"], n_frames, json_mb))
                f.write("| the feeds alone (%d decode threads per rank) | %.0f | %.2f | PNG %.0f | |\n" % (rep["feed_alone"]["decode_threads_per_rank"], rep["feed_alone"]["img_s"], rep["feed_alone"]["img_s"] / tgt, rep["feed_alone"]["png_mb_s"]))
                f.write("| the writers alone (%d threads per rank, 1000 boxes per file) | %.0f | %.2f | JSON %.0f | |\n\n" % (args.writers, rep["writer_alone"]["img_s"], rep["writer_alone"]["img_s"] / tgt, rep["writer_alone"]["json_mb_s"]))
                f.write("| rank | images | loop s | steady img/s | waited for feed s | device s | writer s |\n|---|---|---|---|---|---|---|\n")
                for r in L["ranks"]:
                    f.write("| %d | %d | %.2f | %.0f | %.3f | %.3f | %.3f |\n" % (r["rank"], r["images"], r["loop_s"], r["steady_img_s"] or 0.0, r["waited"]["feed"], r["waited"]["device"], r["waited"]["writer"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
