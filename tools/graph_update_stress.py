#!/usr/bin/env python
"""Is it safe to update an executable launch graph in place (hipGraphExecUpdate: another dropout seed, csrc/byolo_api.hip forward_graph)
while earlier launches of it are still queued or running?  The small Bayesian model (64 x 96, T = 3), a new seed on every call:
  1. four graphs updated and launched back to back without a host wait, every result against the eager forward of its seed;
  2. one graph, a new seed every call;
  3. THE hazard: a launch with seed s1 queued behind 10 ms of other work, its result snapshotted in stream order, then an in-place
     update to s2 and a second launch -- the first launch must still compute s1's rows.
Measured on ROCm 7.2 / MI355X (round 6): 0 mismatches in 1 600 + 85 + 200 x 2 checks: a launch takes its arguments when it is enqueued.
    python tools/graph_update_stress.py
"""
import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "bayesian-yolov3_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch, numpy as np
from conftest import build_model, golden_params, golden_images
v = "bayesian_yolov3_aleatoric"
_, m = build_model(v, 64, 96, T=3, params=golden_params(v)); m.finalize(); eng = m.engine; eng.set_async(True)
x = torch.from_numpy(golden_images(1)).cuda()
N, D = eng.num_boxes()
eng.set_graphs(False)
refs = {s: eng.forward(x, T=3, seed=s, want_boxes=True, want_nms=False)["boxes"].clone() for s in range(4)}
eng.set_graphs(True)
outs = [{"boxes": torch.empty((1, N, D), device="cuda")} for _ in range(4)]      # four buffers -> four graphs, results kept until checked
bad = 0
for it in range(400):
    seeds = [(it + k) % 4 for k in range(4)]
    for k in range(4):                       # four back-to-back forwards without any sync: every one an in-place update of its graph
        eng.forward(x, T=3, seed=seeds[k], want_boxes=True, want_nms=False, out=outs[k])
    torch.cuda.synchronize()
    for k in range(4):
        if not torch.equal(outs[k]["boxes"], refs[seeds[k]]): bad += 1
print("hipGraphExecUpdate with replays in flight: %d mismatches in 1600 forwards; stats %s" % (bad, eng.graph_stats()))
# the same graph updated twice in a row with its previous replay still running
out = {"boxes": torch.empty((1, N, D), device="cuda")}
bad = 0
for it in range(600):
    eng.forward(x, T=3, seed=it % 4, want_boxes=True, want_nms=False, out=out)
    if it % 7 == 6:
        torch.cuda.synchronize()
        if not torch.equal(out["boxes"], refs[it % 4]): bad += 1
print("one graph, a new seed every call: %d mismatches at 85 checks; stats %s" % (bad, eng.graph_stats()))
# the hazard proper: graph launch with seed s1 (in flight), then an in-place update to s2 of the SAME executable graph + launch; the
# first launch's result is snapshotted in stream order between the two
bad1 = bad2 = 0
big = torch.randn(8192, 8192, device="cuda")
for it in range(200):
    s1, s2 = it % 4, (it + 1) % 4
    tmp = big @ big                          # ~10 ms of device work in front: the first launch is still QUEUED when the update happens
    eng.forward(x, T=3, seed=s1, want_boxes=True, want_nms=False, out=out)
    snap = out["boxes"].clone()
    eng.forward(x, T=3, seed=s2, want_boxes=True, want_nms=False, out=out)
    torch.cuda.synchronize()
    bad1 += int(not torch.equal(snap, refs[s1])); bad2 += int(not torch.equal(out["boxes"], refs[s2]))
print("update of an executable graph whose previous launch is in flight: first launch wrong %d, second wrong %d of 200 (first launch queued behind 10 ms of work); stats %s" % (bad1, bad2, eng.graph_stats()))
