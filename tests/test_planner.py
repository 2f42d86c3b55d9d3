"""The workspace planner (csrc/byolo_plan.hip) against the launches it plans for -- on the CPU, through the C-ABI's introspection
entry points (include/byolo.h: byolo_plan_num / byolo_plan_step / byolo_plan_tensor).

THE INVARIANT: two tensors whose byte ranges in the workspace overlap are never alive at the same time -- where a tensor is alive
from the launch that writes it to the last launch that reads it, a launch that reads one tensor and writes another INCLUDED
(round 4's one data-corruption bug: the back-to-back fused launch wrote the follower's output into the memory its own 3x3
convolution was still reading; it was caught by the range sentinel at full size, not by a test -- VERDICT r4 item 10).  A fused
pair is ONE launch: it reads the first step's operands, writes the second step's output, and the first step's output never exists.
What a step reads comes from its operand description, not from the liveness table the planner releases by.

Checked for the three reference models at small and at the benchmark's sizes, both precisions, every planning knob, masked calls,
keep_all_outputs -- and for a few hundred random graphs (residuals of views, concats, upsampling, stacks, several heads).
The reference has no counterpart (TensorFlow owns its tensors: lib_yolo/model.py only lists layers)."""
import ctypes
import itertools
import os

import numpy as np
import pytest

from conftest import build_model

INF = 1 << 30


def _plan(eng, B, T, inject=0):
    from byolo._lib import lib, check
    ns, nt, arena = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    check(eng._h, lib.byolo_plan_num(eng._h, B, T, inject, ctypes.byref(ns), ctypes.byref(nt), ctypes.byref(arena)))
    steps, tensors = [], []
    out, fuse, n = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    reads = (ctypes.c_int32 * 8)()
    for s in range(ns.value):
        check(eng._h, lib.byolo_plan_step(eng._h, s, ctypes.byref(out), ctypes.byref(fuse), reads, ctypes.byref(n)))
        steps.append((out.value, bool(fuse.value), [reads[k] for k in range(n.value)]))
    off, size, after = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    for t in range(nt.value):
        check(eng._h, lib.byolo_plan_tensor(eng._h, t, ctypes.byref(off), ctypes.byref(size), ctypes.byref(after)))
        tensors.append((off.value, size.value, bool(after.value)))
    return steps, tensors, arena.value


def _launches(steps):
    """[(reads, writes)] per LAUNCH: a step that fuses its successor and that successor are one launch."""
    out, si = [], 0
    while si < len(steps):
        o, fuse, reads = steps[si]
        if fuse:
            o2, fuse2, reads2 = steps[si + 1]
            assert not fuse2, "a fused follower cannot fuse again"
            assert o in reads2, "the follower of a fused pair reads the first step's output"
            out.append((sorted(set(reads) | (set(reads2) - {o})), [o2], (si, si + 1)))
            si += 2
        else:
            out.append((sorted(set(reads)), [o], (si,)))
            si += 1
    return out


def check_plan(steps, tensors, arena, what):
    launches = _launches(steps)
    first, last = {}, {}
    for li, (reads, writes, _) in enumerate(launches):
        for t in reads:
            assert t in first and first[t] < li, "%s: launch %d (steps %s) reads tensor %d before any launch wrote it" % (what, li, launches[li][2], t)
            last[t] = li
        for t in writes:
            first.setdefault(t, li)
            last[t] = max(last.get(t, li), li)
    for t, (off, size, after) in enumerate(tensors):
        if t in first and after:
            last[t] = INF                                   # a detection layer's raw output: read by the decode launch after the last step
    live = sorted(first)
    for t in live:
        off, size, _ = tensors[t]
        assert off >= 0 and size > 0 and off + size <= arena, "%s: tensor %d is used but has no memory inside the arena (%d + %d of %d)" % (what, t, off, size, arena)
    # pairwise: overlapping memory -> disjoint lifetimes (sweep over tensors sorted by offset)
    order = sorted(live, key=lambda t: tensors[t][0])
    checked = 0
    for i, a in enumerate(order):
        a_off, a_size, _ = tensors[a]
        for b in order[i + 1:]:
            b_off, b_size, _ = tensors[b]
            if b_off >= a_off + a_size:
                break
            checked += 1
            ok = last[a] < first[b] or last[b] < first[a]
            assert ok, ("%s: tensors %d [%d, +%d) alive over launches %d..%s and %d [%d, +%d) alive over %d..%s share memory while both are alive"
                        % (what, a, a_off, a_size, first[a], last[a], b, b_off, b_size, first[b], last[b]))
    return len(launches), checked


KNOBS = {"BYOLO_B2B": ("0", "1", "2"), "BYOLO_LOWMAIN": ("0", "1"), "BYOLO_NO_DEDUP": ("0", "1"), "BYOLO_WINO_SPLIT": ("0", "1", "2"),
         "BYOLO_WINOGRAD": ("0", "1"), "BYOLO_KX3_WIDE": ("0", "2")}


@pytest.mark.parametrize("precision", ["split", "f32"])
@pytest.mark.parametrize("variant,H,W,B,T", [("bayesian_yolov3_aleatoric", 608, 608, 8, 30), ("bayesian_yolov3_aleatoric", 1024, 1920, 1, 50),
                                             ("bayesian_yolov3_aleatoric", 64, 96, 2, 3), ("yolov3_aleatoric", 416, 416, 8, 1), ("yolov3", 416, 416, 1, 1)])
def test_reference_models_plans_never_alias_a_live_tensor(variant, H, W, B, T, precision, monkeypatch):
    """The reference's three topologies (lib_yolo/yolov3.py:232-310, :370-451, :518-628) at the benchmark's sizes: the default plan
    and every planning knob one at a time and a few together, masked and unmasked calls."""
    monkeypatch.setenv("BYOLO_PRECISION", precision)
    combos = [{}] + [{k: v} for k, vals in KNOBS.items() for v in vals] + [{"BYOLO_B2B": "2", "BYOLO_WINO_SPLIT": "0", "BYOLO_LOWMAIN": "0"},
                                                                           {"BYOLO_B2B": "2", "BYOLO_WINO_SPLIT": "2", "BYOLO_NO_DEDUP": "1"}]
    fused_seen = 0
    for env in combos:
        for k in KNOBS:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        yolo, m = build_model(variant, H, W, T=T)
        for inject in (0, 1):
            steps, tensors, arena = _plan(m.engine, B, T, inject)
            n_launch, n_pairs = check_plan(steps, tensors, arena, "%s %dx%d B=%d T=%d %s %s inject=%d" % (variant, H, W, B, T, precision, env, inject))
            fused_seen += sum(1 for s in steps if s[1])
            assert n_launch >= 70
        m.engine.close()
    if variant.startswith("bayes") and (H, W) == (608, 608) and precision == "split":
        assert fused_seen >= 3, "the benchmark's plan fuses the three 76x76 pairs: the fused case must be part of what was checked"


def test_the_checker_sees_round_4s_bug():
    """The invariant, turned on a plan doctored the way round 4's planner was wrong: the fused launch's OUTPUT placed on the memory
    of the 3x3 convolution's INPUT (dead for an unfused follower, alive for the fused launch)."""
    os.environ.pop("BYOLO_PRECISION", None)
    yolo, m = build_model("bayesian_yolov3_aleatoric", 608, 608, T=30)
    steps, tensors, arena = _plan(m.engine, 8, 30)
    si = next(i for i, s in enumerate(steps) if s[1])
    src = steps[si][2][0]                                    # what the 3x3 convolution reads
    dst = steps[si + 1][0]                                   # what the fused launch writes
    check_plan(steps, tensors, arena, "as planned")
    bad = list(tensors)
    bad[dst] = (tensors[src][0], tensors[dst][1], tensors[dst][2])
    with pytest.raises(AssertionError, match="share memory while both are alive"):
        check_plan(steps, bad, arena, "doctored")
    m.engine.close()


def _random_graph(rng):
    """A random layer graph through the builder half of the C-ABI (no device): 3 .. 16 layers of 1x1 / 3x3 / stride-2 convolutions,
    residuals, identity and concat routes, upsampling, the T-fold stack, one to three detection heads."""
    from byolo import Engine
    H, W = int(rng.integers(1, 6)) * 32, int(rng.integers(1, 6)) * 32
    T = int(rng.integers(1, 5))
    bayes = T > 1
    eng = Engine((H, W, 3), 2, keep_all_outputs=bool(rng.random() < 0.15))
    L = []                                                   # (h, w, c, stacked, readable)

    def add(h, w, c, st, readable=True):
        L.append((h, w, c, st, readable))
    n_conv = [0]

    def conv(f, k, s, drop=False):
        h, w, _, st, _ = L[-1] if L else (H, W, 3, False, True)
        eng.add_conv("c%d" % n_conv[0], f, k, s, 1 | (2 if drop else 0))
        n_conv[0] += 1
        add(h // s, w // s, f, st)
    conv(int(rng.choice([16, 32, 64])), 3, 1)
    n_det, stacked = 0, False
    for _ in range(int(rng.integers(3, 16))):
        h, w, c, st, _ = L[-1]
        op = rng.choice(["conv", "conv", "conv", "res", "up", "route", "stack", "det"])
        if op == "conv":
            k = int(rng.choice([1, 3]))
            s = 2 if (k == 3 and rng.random() < 0.3 and h % 2 == 0 and w % 2 == 0 and min(h, w) >= 4) else 1
            conv(int(rng.choice([32, 64, 96, 128, 256])), k, s, drop=bool(rng.random() < 0.3))
        elif op == "res":
            c_ = [j for j, r in enumerate(L[:-1]) if r[4] and r[:4] == (h, w, c, st)]
            if c_:
                eng.add_residual(int(rng.choice(c_)))
                add(h, w, c, st)
        elif op == "up" and max(h, w) <= 80:
            eng.add_upsample()
            add(2 * h, 2 * w, c, st)
        elif op == "route":
            c_ = [j for j, r in enumerate(L) if r[4] and r[3] == st]
            j = int(rng.choice(c_))
            same = [q for q in c_ if q != j and L[q][:2] == L[j][:2]]
            if same and rng.random() < 0.7:
                q = int(rng.choice(same))
                eng.add_route([j, q])
                add(L[j][0], L[j][1], L[j][2] + L[q][2], st)
            else:
                eng.add_route([j])
                add(*L[j][:4])
        elif op == "stack" and bayes and not stacked:
            c_ = [j for j, r in enumerate(L) if r[4] and not r[3]]
            j = int(rng.choice(c_))
            eng.add_stack(j)
            add(L[j][0], L[j][1], L[j][2], True)
            stacked = True
        elif op == "det" and n_det < 3 and (stacked or not bayes):
            eng.add_detection("d%d/detection" % n_det, 2 if bayes else int(rng.integers(0, 2)), [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])
            add(h, w, 42, L[-1][3], readable=False)
            n_det += 1
            c_ = [j for j, r in enumerate(L) if r[4] and r[3] == L[-1][3]]
            j = int(rng.choice(c_))
            eng.add_route([j])
            add(*L[j][:4])
    if n_det == 0:
        if bayes and not stacked:
            eng.add_stack(len(L) - 1)
            add(L[-1][0], L[-1][1], L[-1][2], True)
        eng.add_detection("d0/detection", 2 if bayes else 1, [(0.1, 0.2), (0.3, 0.1), (0.5, 0.5)])
    return eng, T


@pytest.mark.parametrize("precision", ["split", "f32"])
def test_random_graphs_plans_never_alias_a_live_tensor(precision, monkeypatch):
    """300 random graphs per precision under random planning knobs and batch sizes: the lowering materialises views, fuses
    residuals, splits concat convolutions into T-invariant halves -- whatever it makes of a graph, the plan must keep live tensors
    apart.  Graphs the builder or the lowering refuses (an error code, tools/fuzz_builder.py's business) are skipped and counted."""
    from byolo import ByoloError
    monkeypatch.setenv("BYOLO_PRECISION", precision)
    rng = np.random.default_rng(20260930 + (precision == "f32"))
    done = refused = launches = pairs = 0
    while done < 300:
        for k, vals in KNOBS.items():
            v = str(rng.choice(("",) + vals))
            if v:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        try:
            eng, T = _random_graph(rng)
        except ByoloError:
            refused += 1
            continue
        B = int(rng.integers(1, 5))
        try:
            for inject in (0, 1):
                steps, tensors, arena = _plan(eng, B, T, inject)
                a, b = check_plan(steps, tensors, arena, "random graph %d (%s) B=%d T=%d inject=%d" % (done, precision, B, T, inject))
                launches += a
                pairs += b
            done += 1
        except ByoloError:
            refused += 1
        finally:
            eng.close()
    print("%s: %d graphs planned (%d refused), %d launches, %d memory-sharing tensor pairs checked" % (precision, done, refused, launches, pairs))
    assert pairs > 1000 and refused < 3 * done
