#!/usr/bin/env python
"""fuzz_builder.py -- random call sequences against the graph-builder half of the C-ABI (no GPU needed): whatever the
arguments, a call returns a status (ByoloError on the Python side) -- it never crashes the process, and a graph that
lowers reports consistent sizes.  Each sequence runs in a child process (no device, 32 GB address-space limit) so that a crash or a
runaway allocation is reported, not suffered.

    python tools/fuzz_builder.py --runs 300 --seed 1
"""
import argparse
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, random
sys.path.insert(0, %(pkg)r)
from byolo import Engine, ByoloError
rnd = random.Random(%(seed)d)
def pick(*xs): return rnd.choice(xs)
def small(): return pick(-3, -1, 0, 1, 2, 3, 5, 7, 32, 64, 255, 1 << 20, (1 << 31) - 1)
ok = err = 0
for g in range(20):
    try:
        e = Engine((pick(0, 32, 64, 96, 100, 608, 1 << 16), pick(32, 64, 96, 352), pick(1, 3, 4)), pick(-1, 0, 1, 2, 80, 128, 129),
                   drop_prob=pick(0.0, 0.1, 0.999, 1.0, -0.5), max_out=pick(-1, 0, 1, 1000, 1 << 22), iou_thresh=pick(-1.0, 0.0, 0.5, 2.0))
    except ByoloError:
        err += 1
        continue
    names = 0
    for step in range(rnd.randint(0, 40)):
        op = rnd.randint(0, 7)
        try:
            if op == 0:
                names += 1
                e.add_conv(pick("s%%d" %% names, "s1", "", "x/y"), pick(small(), 8, 16, 32, 64, 256), pick(1, 3, small()), pick(1, 2, small()), pick(0, 1, 2, 3, small()))
            elif op == 1:
                e.add_residual(small() if rnd.random() < 0.5 else -rnd.randint(1, 4))
            elif op == 2:
                e.add_route([pick(small(), -1, -2, -3) for _ in range(rnd.randint(0, 3))])
            elif op == 3:
                e.add_upsample()
            elif op == 4:
                e.add_stack(pick(small(), -1))
            elif op == 5:
                names += 1
                e.add_detection("d%%d/detection" %% names, pick(0, 1, 2, small()), [(rnd.random(), rnd.random())] * pick(0, 1, 3, 3, 3, 4))
            elif op == 6:
                e.mark_backbone_end()
            else:
                b, t = pick(0, 1, 2, 8, small()), pick(1, 1, 3, 30, small())
                ws = e.workspace_bytes(b, t)
                fl = e.flops(max(b, 1), max(t, 1))
                n, d = e.num_boxes()
                assert ws > 0 and fl >= 0 and n > 0 and d > 0, (ws, fl, n, d)
                assert e.max_images(max(t, 1)) >= 0
            ok += 1
        except ByoloError:
            err += 1
        except (TypeError, ValueError, OverflowError, AssertionError) as ex:   # refused by the Python wrapper / ctypes: no call
            if isinstance(ex, AssertionError) and 'priors' not in str(ex): raise
            err += 1
    e.close()
print("OK %%d calls succeeded, %%d refused" %% (ok, err))
'''


def _limit():
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (32 << 30, 32 << 30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    bad = 0
    tot_ok = tot_err = 0
    for r in range(a.runs):
        code = CHILD % {"pkg": os.path.join(REPO, "bayesian-yolov3_amd"), "seed": a.seed * 100003 + r}
        # no device (the builder half needs none) and an address-space limit: a call that tries to allocate the world is a reported
        # crash of the child, not the end of the machine (round 4: filters = INT_MAX was a 232 GB vector)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, preexec_fn=_limit,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        if p.returncode != 0 or not line.startswith("OK"):
            bad += 1
            print("run %d (seed %d): exit %d\n%s" % (r, a.seed * 100003 + r, p.returncode, (p.stderr or p.stdout)[-1500:]))
        else:
            _, n_ok, _, _, n_err, _ = line.split()
            tot_ok += int(n_ok); tot_err += int(n_err)
    print("%d runs, %d crashed or inconsistent; %d calls succeeded, %d refused with an error" % (a.runs, bad, tot_ok, tot_err))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
