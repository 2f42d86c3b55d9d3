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
    # bench.py runs, after the K timed steps, ONE more step (the kept-index checksum of the line) with per-launch recording on: the last
    # (K + 1) * L dispatches of the kernel are steps 0 .. K - 1 of the timed region and that extra step.
    L, K, every = a.launches_per_step, a.steps, max(1, a.profile_every)
    d = c.execute("select duration from kernels where name like ? order by start desc limit ?",
                  ("%" + a.kernel + "%", (K + 1) * L)).fetchall()
    if len(d) == (K + 1) * L:
        ds = [x[0] for x in d][::-1]                     # oldest first
        timed = ds[:K * L]
        print("\n## dominant kernel in the timed region (the %d dispatches of `%s...` in its %d steps)\n" % (len(timed), a.kernel, K))
        print("- launches: %d\n- total: %.3f ms\n- average launch duration: %.4f ms" % (len(timed), sum(timed) / 1e6, sum(timed) / len(timed) / 1e6))
        # A step's HEAD launches run alone on the device when the NEXT forward waits for the whole convolution stack, i.e. when this
        # step or the next one records per-launch hipEvents (include/byolo.h byolo_plan_opts.serialize_heads: a recorded forward and
        # its successor wait as a whole): the profiled steps i = 0, every, ... -- what bench.py's `roofline` times -- and the last
        # timed step (the checksum step behind it is recorded).  The other steps' heads share the device with the next step's
        # backbone: that is what makes the STEP faster and a single launch longer.
        prof = [i for i in range(K) if i % every == 0]
        quiet = sorted(set(prof) | {K - 1})
        pq = [x for i in prof for x in timed[i * L:(i + 1) * L]]
        qq = [x for i in quiet if i not in prof for x in timed[i * L:(i + 1) * L]]
        rest = [x for i in range(K) if i not in quiet for x in timed[i * L:(i + 1) * L]]
        print("- the %d launches of the profiled step(s) %s (bench.py's per-launch hipEvents time exactly these): average %.4f ms" % (len(pq), prof, sum(pq) / len(pq) / 1e6))
        if qq:
            print("- the %d launches of the last timed step (quiet as well: the checksum step behind it is recorded): average %.4f ms" % (len(qq), sum(qq) / len(qq) / 1e6))
        if rest:
            print("- the other %d launches (their step's heads share the device with the next step's backbone): average %.4f ms" % (len(rest), sum(rest) / len(rest) / 1e6))

if __name__ == "__main__":
    main()
