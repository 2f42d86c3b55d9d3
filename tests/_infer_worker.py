"""Worker for tests/test_dist_cpu.py::test_inference_epistemic_world2_*: `inference_epistemic.Inference` -- the driver loop of the
entry point, unmodified (dataset sharding, first_image, padded blocks, two batches in flight, the ONE all-gather that also carries
the range status, every rank writing the files of its own images, error agreement)
-- as rank `rank` of a gloo job on CPU tensors, with a stand-in for the GPU engine that returns rows which are a pure function
of (pixels, position of the image in the GLOBAL batch, seed): any mistake in the sharding shows up in the JSON files."""
import os
import sys


class FakeEngine:
    torch_device = "cpu"
    device = 0
    out_cap = 6
    precision = "split"

    def __init__(self):
        self.raise_in = None          # (rank, forward call number): that forward "leaves the split-f16 range"
        self.flags = 0
        self.forwards = 0
        self.log = []

    def num_boxes(self):
        return 50, 23

    def param_shapes(self):
        return {}

    def set_params(self, *a, **k):
        pass

    def finalize(self):
        self.log.append("finalize:" + self.precision)

    def calibrate_bn(self, x):
        pass

    def set_async(self, on=True):
        self.log.append("async")

    def normalize_u8(self, u8, out=None):           # byolo_normalize_u8
        import torch
        y = u8.to(torch.float32) * torch.tensor(1.0 / 255.0, dtype=torch.float32)
        if out is None:
            return y
        out.copy_(y)
        return out

    # ---- the T-sharded path (byolo/inference.py _run_t_sharded): sums of small integers are exact in float32, so any cut of the
    # samples over any number of ranks adds up to the same bits -- a wrong shard, a missing rank or a double count changes them
    def finish_tshard(self, sums, T_total):
        sums /= float(T_total)
        return sums

    def sort_nms(self, rows, obj_idx, cls_start_idx, **kw):
        import torch
        k = 1 + int(rows[0, 0, 0].item() * 3) % self.out_cap
        out = torch.zeros((1, self.out_cap, rows.shape[2]))
        out[0, :k] = rows[0, :k]
        return {"rows": out, "kept": torch.arange(self.out_cap, dtype=torch.int32)[None], "count": torch.tensor([[k, k]], dtype=torch.int32)}

    def copy_status(self, out):                     # byolo_copy_status: sticky until cleared
        out[0] = self.flags
        out[1] = 7 if self.flags else -1

    def clear_status(self):
        self.flags = 0

    def set_precision(self, p):
        self.precision = p
        self.log.append("precision:" + p)


class FakeModel:
    cls_cnt, obj_idx, cls_start_idx = 2, 14, 17

    def __init__(self):
        self.engine = FakeEngine()
        self.calls = []

    def finalize(self):
        self.engine.finalize()

    T = 3

    def run(self, x, seed=0, want_boxes=True, first_image=0, out=None, precision=None, t_shard=None, **kw):
        import torch
        eng = self.engine
        eng.forwards += 1
        if t_shard is not None:                      # per-box SUMS over samples t0 .. t1 - 1 of image `first_image` of the batch
            t0, t1 = t_shard
            self.calls.append((int(x.shape[0]), int(first_image), int(seed), int(t0), int(t1)))
            if precision is not None:
                eng.log.append("run:%s:%d" % (precision, seed))
            if precision is None and eng.precision == "split" and eng.raise_in == eng.forwards:
                eng.flags = 1
                return {"boxes": torch.full((1, 50, 23), float("nan")), "engine": eng}
            base = int(x.sum().item() * 7) % 13
            b = torch.arange(50, dtype=torch.float32)[:, None]
            d = torch.arange(23, dtype=torch.float32)[None, :]
            sums = torch.zeros((1, 50, 23))
            for t in range(t0, t1):
                sums[0] += ((b * 5 + d + base + 3 * t + first_image + seed) % 17) + 1.0
            return {"boxes": sums, "engine": eng}
        self.calls.append((int(x.shape[0]), int(first_image), int(seed)))
        if precision is not None:                    # Model.run(precision='f32'): THIS batch on the fp32 twin handle
            eng.log.append("run:%s:%d" % (precision, seed))
        if precision is None and eng.precision == "split" and eng.raise_in == eng.forwards:
            eng.flags = 1                            # what the epilogues do on the device; the rows are then garbage
            out["rows"].fill_(float("nan"))
            out["count"].fill_(3)
            return out
        for j in range(x.shape[0]):
            g = first_image + j                                    # position in the global batch
            k = 1 + (int(x[j].sum().item() * 7) + g) % self.engine.out_cap
            base = x[j].mean() + 0.001 * seed
            for b in range(k):
                out["rows"][j, b] = torch.arange(23, dtype=torch.float32) * 0.01 + base + 0.1 * b + g
                out["rows"][j, b, 17:19] = torch.tensor([0.7, 0.3]) if (g + b) % 2 else torch.tensor([0.2, 0.8])
            out["kept"][j, :k] = torch.arange(k, dtype=torch.int32) + 100 * g
            out["count"][j] = k
        return out


class FakeYolo:
    def __init__(self):
        self.model = FakeModel()
        self.options = {}

    def set_engine_option(self, k, v):
        self.options[k] = v

    def init_model(self, inputs=None, training=False):
        return self

    def get_model(self):
        return self.model


def main(rank, world, port, data_dir, out_path, raise_rank=-1, raise_call=0, shard_t=0):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "bayesian-yolov3_amd"))
    import torch
    import inference_epistemic as ie
    cfg = {"batch_size": 5, "full_img_size": [32, 32, 3], "crop": False, "cls_cnt": 2, "implicit_background_class": True,
           "weights": "synthetic", "seed": 3, "inference_mode": True, "T": 3, "out_path": out_path, "data": {"file_pattern": os.path.join(data_dir, "val-*")}}
    if shard_t:
        cfg["shard"] = "T"
    yolo = FakeYolo()
    if rank == raise_rank:
        yolo.model.engine.raise_in = raise_call
    try:
        loop = ie.Inference(yolo, cfg)
    except OSError as e:                                   # rank 0's own failure
        print("RANK%d OSERROR %s" % (rank, e)); sys.exit(7)
    except RuntimeError as e:                              # the other ranks learn of it (byolo.dist.agree_on_error)
        print("RANK%d AGREED %s" % (rank, e)); sys.exit(7)
    loop.run()
    import json
    json.dump({"calls": yolo.model.calls, "options": yolo.options, "log": yolo.model.engine.log, "stats": loop.stats,
               "precision": yolo.model.engine.precision}, open(os.path.join(data_dir, "calls_w%d_r%d.json" % (world, rank)), "w"))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], *[int(a) for a in sys.argv[6:9]])
