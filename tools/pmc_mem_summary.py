#!/usr/bin/env python
"""Sum the counters of tools/pmc_mem_probe.sh over the last `--launches` dispatches of a kernel (one bench step) and print them
per elapsed GPU cycle (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8), i.e. as average busy / stalled units.

    python tools/pmc_mem_summary.py gpurun_out/prof_TAG --kernel wino_split_kernel --launches 6
"""
import argparse, csv, glob, json, os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("--kernel", default="wino_split_kernel"); ap.add_argument("--launches", type=int, default=6)
    a = ap.parse_args()
    res = {"kernel": a.kernel, "launches": a.launches, "passes": {}}
    for sub in sorted(glob.glob(os.path.join(a.dir, "MEM_*"))):
        if not os.path.isdir(sub):
            continue
        raw = {}
        for path in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            by = {}
            for r in csv.DictReader(open(path)):
                if a.kernel in r["Kernel_Name"]:
                    by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            for c, v in by.items():
                v.sort()
                raw[c] = sum(x for _, x in v[-a.launches:])
        g = raw.get("GRBM_GUI_ACTIVE")
        out = {"raw": raw}
        if g:
            cyc = g / 8.0
            out["per_gpu_cycle"] = {k: v / cyc for k, v in raw.items() if k != "GRBM_GUI_ACTIVE"}
        res["passes"][os.path.basename(sub)] = out
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
