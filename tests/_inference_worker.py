"""Worker for tests/test_entry_points.py: `inference_epistemic.inference(config)` as a torchrun rank
(`python -m torch.distributed.run ... _inference_worker.py <records pattern> <checkpoint dir> <out path> <batch> [<stats file prefix> [<shard: T | -> [<T>]]]`)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "bayesian-yolov3_amd"))


def main(pattern, ckpt, out_path, batch, stats_prefix=None, shard=None, T=3):
    import inference_epistemic as mod
    from lib_yolo import yolov3
    cfg = {"full_img_size": [64, 96, 3], "crop": False, "cls_cnt": 2, "priors": yolov3.ECP_9_PRIORS, "aleatoric_loss": False,
           "inference_mode": True, "T": int(T), "implicit_background_class": True, "batch_size": int(batch),
           "checkpoint_path": ckpt, "run_id": "run", "step": "last", "seed": 10, "data": {"file_pattern": pattern},
           "out_path": out_path}
    if shard and shard != "-":
        cfg["shard"] = shard
    stats = mod.inference(cfg)
    if stats_prefix:
        import json
        json.dump(stats, open("%s_rank%s.json" % (stats_prefix, os.environ.get("RANK", "0")), "w"))


if __name__ == "__main__":
    main(*sys.argv[1:8])
