#!/usr/bin/env python
"""Long run of the drop-in entry point (bench.py's entry_point leg at N frames): sustained throughput, host memory before / after
(nothing may accumulate per batch: pinned buffers, writer futures, decode scratch), and -- with --gpus2 -- the same through
`bench.py --gpus 2` with two real engines on one device (gloo, host-staged collective), the closest thing to a scale run a one-GPU
box allows.

    python tools/entry_point_soak.py [--frames 4096] > gpurun_out/entry_point_soak.json"""
import argparse
import json
import os
import resource
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "bayesian-yolov3_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--gpus2", action="store_true")
    a = ap.parse_args()
    import bench
    cfg = dict(bench.CONFIGS[4])
    import gc

    def rss_now():                                # resident set NOW (ru_maxrss is a high-water mark: it cannot show that a run gave memory back)
        return int(open("/proc/self/statm").read().split()[1]) * 4096 / 1e6
    rss = {"start": rss_now()}
    short = bench.entry_point_leg(cfg, 0, n_frames=512, extras=False)          # warms every pool: engine, pinned buffers, allocator arenas
    gc.collect(); rss["after_512_frames"] = rss_now()
    long = bench.entry_point_leg(cfg, 0, n_frames=a.frames, extras=False)
    gc.collect(); rss["after_%d_more_frames" % a.frames] = rss_now()
    again = bench.entry_point_leg(cfg, 0, n_frames=a.frames, extras=False)
    gc.collect(); rss["after_another_%d_frames" % a.frames] = rss_now()
    out = {"short_run": short, "long_run": long, "long_run_again": again, "rss_mb": rss,
           "max_rss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0}
    if a.gpus2:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BYOLO_DIST_BACKEND="gloo", BYOLO_DIST_SHARE_DEVICE="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29571", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3",
               "--no-cpu-baseline", "--fp32-steps", "0", "--entry-frames", "0", "--no-other-configs", "--no-profile", "--batch", "4"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out["bench_gpus2_two_engines_one_device"] = json.loads(lines[-1]) if lines else {"error": r.stderr[-2000:]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
