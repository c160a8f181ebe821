#!/usr/bin/env python
"""Reduce the two PMC passes of scripts/pmc_probe.py to per-launch HBM traffic of the merge kernel (JSON)."""
import csv
import glob
import json
import sys


def per_kernel(dirpath, counter):
    path = glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
copy_bytes = int(sys.argv[3])
alg = int(sys.argv[4])


def pick(d, key):
    ks = [k for k in d if key in k]
    vals = [v for k in ks for v in d[k]]
    return sorted(vals)[len(vals) // 2] if vals else None


cf, cw = pick(fetch, "neg_kernel"), pick(write, "neg_kernel")
mf, mw = pick(fetch, "merge_co_kernel"), pick(write, "merge_co_kernel")
res = {"counters_raw_KiB": {"copy_fetch": cf, "copy_write": cw, "merge_fetch": mf, "merge_write": mw},
       "calibration": {"known_copy_bytes_each_way": copy_bytes,
                       "fetch_bytes_per_count": copy_bytes / cf if cf else None,
                       "write_bytes_per_count": copy_bytes / cw if cw else None,
                       "note": "gfx950 FETCH_SIZE reads ~1/2 of a wide coalesced stream (MI355X_MICROARCH.md); "
                               "both counters calibrated on a same-footprint 16 B/lane copy in the same run"}}
if cf and cw and mf and mw:
    rd, wr = mf * copy_bytes / cf, mw * copy_bytes / cw
    res["merge_per_launch"] = {"hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_total_bytes": round(rd + wr),
                               "algorithmic_bytes": alg, "traffic_over_algorithmic": round((rd + wr) / alg, 4)}
alg_step = int(sys.argv[5]) if len(sys.argv) > 5 else None
sf, sw = pick(fetch, "merge_step_kernel"), pick(write, "merge_step_kernel")
if cf and cw and sf and sw and alg_step:
    rd, wr = sf * copy_bytes / cf, sw * copy_bytes / cw
    res["merge_step_per_launch"] = {"hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr), "hbm_total_bytes": round(rd + wr),
                                    "algorithmic_bytes": alg_step, "traffic_over_algorithmic": round((rd + wr) / alg_step, 4)}
print(json.dumps(res, indent=1))
