#!/usr/bin/env python
"""Turn `rocprofv3 --pmc` passes over bench.py into the per-launch HBM-traffic figure of the dominant
kernel (bench.py's `roofline.traffic`) and a small markdown summary for profiles/.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/FETCH_SIZE -o pmc -- python bench.py --steps 1 --warmup 1 --no-profile --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ...           (separate pass: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2)
    python tools/pmc_traffic.py <dir> --launches 66 --out profiles/rN_traffic.json

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
in KiB (bytes = value * 1024); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
(16 B/lane) coalesced read stream, so the read side is doubled.  Infinity-Cache hits are counted, not
excluded: read it as fabric traffic, an upper bound of HBM traffic.
"""
import argparse
import collections
import csv
import json
import os


def last_launches(path, kernel, counter, n):
    rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows[-n:]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--kernel", default="conv_igemm_kernel<128, 128")
    ap.add_argument("--launches", type=int, default=66, help="launches of the kernel in ONE bench step")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tag", default=None, help="profile round the passes belong to (recorded in the output)")
    ap.add_argument("--sources", nargs="*", default=None, help="source files of the kernel (repo-relative): their SHA-256 is recorded, and bench.py "
                    "refuses the file once one of them has changed (a traffic figure belongs to the kernel it was measured on)")
    a = ap.parse_args()
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        p = os.path.join(a.dir, c, "pmc_counter_collection.csv")
        v = last_launches(p, a.kernel, c, a.launches)
        res[c] = {"launches": len(v), "sum_KiB": sum(v)}
    n = res["FETCH_SIZE"]["launches"]
    read_b = 2.0 * res["FETCH_SIZE"]["sum_KiB"] * 1024.0          # gfx950: x2
    write_b = res["WRITE_SIZE"]["sum_KiB"] * 1024.0
    out = {"kernel": a.kernel, "launches": n, "read_bytes_per_launch": read_b / n, "write_bytes_per_launch": write_b / n,
           "traffic_bytes_per_launch": (read_b + write_b) / n,
           "note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB units; last step of bench.py; includes Infinity-Cache hits"}
    # when and on which source state the counters were collected: bench.py copies this next to `roofline.traffic`
    import datetime
    import subprocess
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except Exception:
        commit = ""
    out["measured_at"] = {"commit": commit, "date": datetime.date.today().isoformat(), "profile": a.tag,
                          "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_round.sh"}
    if a.sources:
        import hashlib
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out["measured_at"]["sources"] = {f: hashlib.sha256(open(os.path.join(repo, f), "rb").read()).hexdigest()[:16] for f in a.sources}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
