#!/usr/bin/env python
"""Summarise a `rocprofv3 --kernel-trace --stats` run (rocpd sqlite database) of bench.py into the
markdown table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_rN/bench_results.db --steps K --launches-per-step L > profiles/...

Besides the whole-process per-kernel table it isolates the TIMED REGION of bench.py for the dominant
kernel (the last steps*L dispatches of conv_igemm_kernel<128,128,2,2>: warm-up and the one-off BN
calibration come earlier), which is the set bench.py's `roofline.avg_launch_ms` averages over.
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--launches-per-step", type=int, default=66)
    ap.add_argument("--kernel", default="conv_igemm_kernel<128, 128")
    ap.add_argument("--profile-every", type=int, default=10, help="bench.py --profile-every of the profiled command: the steps whose launches "
                    "carry per-launch hipEvents run in a quiet window (include/byolo.h byolo_plan_opts.serialize_heads)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                     "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("## per-kernel totals (whole process: calibration + warm-up + timed steps)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for n, cnt, tot, avg, mn, mx, vg, ag, sg, lds in rows:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s |"
              % (n.split("(")[0], cnt, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds))
    k = a.steps * a.launches_per_step
    d = c.execute("select duration from kernels where name like ? order by start desc limit ?",
                  ("%" + a.kernel + "%", k)).fetchall()
    if d:
        ds = [x[0] for x in d]
        print("\n## dominant kernel in the timed region (last %d dispatches of `%s...`)\n" % (len(ds), a.kernel))
        print("- launches: %d\n- total: %.3f ms\n- average launch duration: %.4f ms" % (len(ds), sum(ds) / 1e6, sum(ds) / len(ds) / 1e6))
        # the steps bench.py records per-launch hipEvents on (i = 0, every, 2 * every, ...): their forwards wait for the other stream's
        # WHOLE convolution stack and so does the forward after them -- the launches of these steps run alone on the device, the
        # others beside the next step's backbone (which is what makes the step faster and a single launch longer)
        ds_t = ds[::-1]                                   # oldest first
        L = a.launches_per_step
        quiet = [d for st in range(0, a.steps, max(1, a.profile_every)) for d in ds_t[st * L:(st + 1) * L]]
        rest = [d for st in range(a.steps) if st % max(1, a.profile_every) for d in ds_t[st * L:(st + 1) * L]]
        if quiet and rest:
            print("- of these, the %d launches of the profiled step(s) (quiet window; what bench.py's hipEvents time): average %.4f ms" % (len(quiet), sum(quiet) / len(quiet) / 1e6))
            print("- the other %d launches (their step's heads share the device with the next step's backbone): average %.4f ms" % (len(rest), sum(rest) / len(rest) / 1e6))


if __name__ == "__main__":
    main()
