#!/usr/bin/env python
"""Where a small model's set-up time goes (build, byolo_finalize, first / second forward): most -m gpu tests are dominated by it.
    gpurun -- 'python tools/time_small.py; BYOLO_FINALIZE_THREADS=1 python tools/time_small.py'
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def once(v, H, W, T, B):
    import torch
    from conftest import build_model, golden_params, golden_images
    t0 = time.time()
    _, m = build_model(v, H, W, T=T, params=golden_params(v), engine_options={"keep_all_outputs": True})
    t1 = time.time()
    m.finalize()
    torch.cuda.synchronize()
    t2 = time.time()
    x = torch.from_numpy(golden_images(B)).cuda() if (H, W) == (64, 96) else torch.rand(B, H, W, 3).cuda()
    m.run(x, seed=42)
    torch.cuda.synchronize()
    t3 = time.time()
    m.run(x, seed=42)
    torch.cuda.synchronize()
    t4 = time.time()
    print("%-28s %4d x %-4d threads %-4s build %.2f  finalize %.2f  first run %.3f  second run %.4f s" % (
        v, H, W, os.environ.get("BYOLO_FINALIZE_THREADS", "all"), t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)


if __name__ == "__main__":
    import torch
    torch.zeros(1).cuda()
    for _ in range(2):
        once("yolov3", 64, 96, 1, 2)
        once("bayesian_yolov3_aleatoric", 64, 96, 3, 2)
    once("bayesian_yolov3_aleatoric", 608, 608, 30, 1)
