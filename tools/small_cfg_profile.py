#!/usr/bin/env python
"""Where the time of a launch-sized forward goes (BASELINE configs[0..2]: 65 GFLOP .. 3.4 TFLOP per step in ~85 launches).

    python tools/small_cfg_profile.py --cfg 1 [--steps 200] [--md out.md]
    rocprofv3 --kernel-trace --stats -d DIR -o small -- python tools/small_cfg_profile.py --cfg 1 --steps 50 --no-table

Per configuration (bench.py CONFIGS numbering: 1 = configs[0] ...):
  * wall time per step, two streams alternating (the way bench.py's `other_configs` legs run) and ONE stream;
  * host time per `Engine.forward` call (what the CPU spends enqueueing one forward, nothing waited for);
  * the per-launch table of one forward (byolo_step_profile: hipEvents around every launch on the launch stream) and its sum:
    wall (one stream) - sum = what the launches do not account for (gaps between dependent launches, the tail, the host);
  * the same with the forward replayed from a launch graph (Engine.set_graphs(True)) when the library has it.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "bayesian-yolov3_amd"))


def timed(eng, x, T, steps, warmup, streams, out, seeds_vary=True):
    import torch
    host = 0.0

    def step(i):
        nonlocal host
        k = i % len(streams)
        with torch.cuda.stream(streams[k]):
            t = time.perf_counter()
            eng.forward(x, T=T, seed=1000 + (i if seeds_vary else 0), want_boxes=False, want_nms=True, out=out[k], slot=1 + k)
            host += time.perf_counter() - t
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return 1e3 * dt / steps, 1e3 * host / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--md", default=None)
    ap.add_argument("--no-table", action="store_true")
    ap.add_argument("--graph", type=int, default=-1, help="-1: both (if the library has graphs), 0 eager only, 1 graph only")
    args = ap.parse_args()
    import torch
    import bench
    from byolo import synth
    cfg = dict(bench.CONFIGS[args.cfg])
    m = bench.build(cfg, 0)
    eng = m.engine
    eng.set_async(True)
    B, T = cfg["B"], cfg["T"]
    x = torch.from_numpy(synth.synthetic_images(B, cfg["H"], cfg["W"], seed=1234)).to("cuda:0")
    N, D = eng.num_boxes()
    cap = eng.out_cap
    mk = lambda: {"rows": torch.empty((B, cap, D), device=x.device), "kept": torch.empty((B, cap), dtype=torch.int32, device=x.device),
                  "count": torch.empty((B, 2), dtype=torch.int32, device=x.device)}
    out = [mk(), mk()]
    streams = [torch.cuda.Stream(device=x.device) for _ in range(2)]
    res = {"cfg": args.cfg, "workload": "%s %dx%d B=%d T=%d" % (cfg["variant"], cfg["H"], cfg["W"], B, T), "gflop_per_step": eng.flops(B, T) / 1e9}
    has_graph = hasattr(eng, "set_graphs")
    modes = [0, 1] if args.graph < 0 else [args.graph]
    for g in modes:
        if g and not has_graph:
            continue
        if has_graph:
            eng.set_graphs(bool(g))
        tag = "graph" if g else "eager"
        ms2, host2 = timed(eng, x, T, args.steps, args.warmup, streams, out)
        ms1, host1 = timed(eng, x, T, args.steps, args.warmup, streams[:1], out)
        res[tag] = {"ms_per_step_two_streams": ms2, "ms_per_step_one_stream": ms1, "host_ms_per_forward": host1,
                    "img_s_two_streams": B / ms2 * 1e3, "img_s_one_stream": B / ms1 * 1e3}
    if not args.no_table:
        if has_graph:
            eng.set_graphs(False)
        eng.set_profiling(2)
        with torch.cuda.stream(streams[0]):
            for i in range(3):
                eng.forward(x, T=T, seed=7 + i, want_boxes=False, want_nms=True, out=out[0], slot=1)
        torch.cuda.synchronize()
        prof = eng.step_profile()
        st = eng.stage_ms()
        eng.set_profiling(0)
        tot = sum(p["ms"] for p in prof)
        res["per_launch"] = {"launches": len(prof), "sum_ms": tot, "stages_ms": st,
                             "note": "hipEvents between launches on the launch stream: a launch's figure includes the gap to the next one"}
        if args.md:
            with open(args.md, "w") as f:
                f.write("| # | layer | variant | M | N | K | ms | TF/s |\n|---|---|---|---|---|---|---|---|\n")
                for i, p in enumerate(prof):
                    f.write("| %d | %d | %d | %d | %d | %d | %.4f | %.1f |\n" % (i, p["layer"], p["variant"], p["M"], p["N"], p["K"], p["ms"],
                                                                              p["flops"] / max(p["ms"], 1e-9) / 1e9))
                f.write("\nsum %.4f ms over %d launches; stages %s\n" % (tot, len(prof), json.dumps(st)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
