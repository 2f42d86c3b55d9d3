#!/usr/bin/env python
"""Summarise the SQ counter passes of tools/profile_round.sh for the dominant convolution kernel.

    python tools/pmc_sq_summary.py gpurun_out/prof_TAG --launches 31 > profiles/TAG_pmc_sq_summary.json

Counters are summed over the last `--launches` dispatches of the kernel (one bench step).  Conventions measured on
this part: SQ_VALU_MFMA_BUSY_CYCLES counts 64 per fp32 32x32x2 MFMA issued -> divide by
(4 SIMDs x CU-busy cycles); SQ_INSTS_VALU includes the MFMAs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import argparse
import csv
import glob
import json
import os


def read(dirname, kernel, n):
    out, dur = {}, None
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in by.items():
            v.sort()
            out[c] = sum(x for _, x in v[-n:])
    for path in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-n:]) * 1e-6
    return out, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--kernel", default="conv_igemm_kernel<128, 128")
    ap.add_argument("--launches", type=int, default=31)
    a = ap.parse_args()
    raw, ms = {}, None
    for sub in ("SQ_A", "SQ_B"):
        r, d = read(os.path.join(a.dir, sub), a.kernel, a.launches)
        raw.update(r)
        if sub == "SQ_B":
            ms = d
    res = {"kernel": a.kernel, "launches": a.launches, "time_ms_pass_B": ms, "raw": raw}
    g = raw.get("GRBM_GUI_ACTIVE")
    if g and ms:
        res["clock_GHz"] = g / 8.0 / (ms * 1e-3) / 1e9
    if g and raw.get("SQ_VALU_MFMA_BUSY_CYCLES"):     # of the kernel's elapsed cycles x 256 CUs x 4 SIMDs
        res["mfma_busy_frac"] = raw["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * 256.0 * g / 8.0)
    if g and raw.get("SQ_BUSY_CU_CYCLES"):
        res["cu_busy_frac"] = raw["SQ_BUSY_CU_CYCLES"] / (256.0 * g / 8.0)
    if raw.get("SQ_INSTS_MFMA") and raw.get("SQ_INSTS_VALU"):
        res["valu_insts_per_mfma"] = (raw["SQ_INSTS_VALU"] - raw["SQ_INSTS_MFMA"]) / raw["SQ_INSTS_MFMA"]
    if raw.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if raw.get(k):
                res[k.lower() + "_frac_of_wave_cycles"] = raw[k] / raw["SQ_WAVE_CYCLES"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
