#!/usr/bin/env python
"""Launch-graph replay under load: several handles with split-K / stream-K plans replay their captured forwards a few hundred times.

    python tools/graph_stress.py h      # K slices + stream-K + 2 x planner, 500 iterations (modes a .. j: see the bottom)

Why this exists (round 6): with the forward's ticket words zeroed by `hipMemsetAsync` inside the captured region, a process holding
THREE or more executable graphs produced inf rows in every split-K / stream-K handle from replay 274 on (206 with four graphs:
~8 192 graph operations in total), deterministically, with or without concurrency, again 273 replays after a re-capture -- and never with
eager launches, one or two graphs, or handles without tickets.  The runtime's memset NODE went wrong, not a kernel: zeroing the words
with a kernel of the library (csrc/conv_kernels.hip zero_words_kernel) ends it (0 mismatches in every mode; tools/stress_handoff.py 1000
iterations clean with graphs on).  tests/test_gpu_layers.py runs mode h past the old onset.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian-yolov3_amd"))
import torch
import stress_handoff as sh

LAST_BAD = 0


def run(which, iters, sync_each=False, one_stream=False, regraph_at=None):
    plans = [("K slices forced", {"BYOLO_KSPLIT": "3"}), ("stream-K forced", {"BYOLO_STREAMK": "2"}), ("planner", {})]
    plans = [plans[i] for i in which]
    x = torch.from_numpy(np.random.default_rng(1).random((6, 96, 96, 64), dtype=np.float32)).cuda()
    engs, first = [], []
    for name, env in plans:
        e = sh.build(env)
        out = e.forward(x, T=1, seed=0, want_boxes=True, want_nms=False)
        torch.cuda.synchronize()
        engs.append(e); first.append(out["boxes"].clone())
    streams = [torch.cuda.Stream() for _ in engs]
    if one_stream: streams = [streams[0]] * len(engs)
    outs = [dict(boxes=torch.empty_like(f)) for f in first]
    onset = {}
    nbad = 0
    for it in range(iters):
        if regraph_at and it == regraph_at:
            for e in engs: e.set_graphs(False); e.set_graphs(True)
        for k, (e, st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                e.forward(x, T=1, seed=0, want_boxes=True, want_nms=False, out=outs[k], slot=1 + k)
            if sync_each: torch.cuda.synchronize()
        torch.cuda.synchronize()
        for k in range(len(engs)):
            if not torch.equal(outs[k]["boxes"], first[k]):
                nbad += 1
                if (plans[k][0], k) not in onset:
                    d = (outs[k]["boxes"] - first[k]).abs()
                    onset[(plans[k][0], k)] = (it, float(d.max()), int((d > 0).sum()), int(d.numel()))
    global LAST_BAD
    LAST_BAD = nbad
    print("handles %s iters %d sync_each %s one_stream %s regraph_at %s: bad %d onset %s stats %s" % (which, iters, sync_each, one_stream, regraph_at, nbad, onset, [e.graph_stats() for e in engs]), flush=True)

if __name__ != "__main__":
    sys.argv = [sys.argv[0], ""]
mode = sys.argv[1]
if mode == "a": run([0], 1200)
if mode == "b": run([0, 1], 700)
if mode == "c": run([0, 1, 2], 500, sync_each=True)
if mode == "d": run([0, 1, 2], 500, one_stream=True)
if mode == "e": run([0, 1, 2], 700, regraph_at=400)
if mode == "f": run([2, 2, 2], 500)
if mode == "g": run([0, 1, 0], 500)
if mode == "h": run([0, 1, 2, 2], 500)
if mode == "i": run([0, 0, 0], 500)
if mode == "j": run([0, 2, 2], 500)
