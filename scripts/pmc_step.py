#!/usr/bin/env python
"""Whole-step HBM bytes and MFMA utilisation of the benchmarked training step from rocprofv3 PMC passes.

Workload of every pass: `python bench.py --mode eager --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary`
(eager, so every step is the same sequence of dispatches; LORA_AMD_TUNE_CACHE re-uses the attention choices of a warm
run).  Passes (separate, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; --kernel-trace only besides --pmc):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/step_f -o p -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/step_w -o p -- python bench.py ...
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace ... -d $OUT/step_m ...
    python scripts/pmc_step.py reduce $OUT/step_f $OUT/step_w $OUT/step_m <fetch_bytes_per_count> <write_bytes_per_count> > profiles/r03_step_pmc.json

A step = the dispatches between two consecutive `clip_adamw` launches (one per optimiser step); the LAST `nsteps`
steps of the run are averaged.  Byte calibration: bytes per counter unit from the same-run 16 B/lane copy of the merge
PMC file (profiles/r03_merge_pmc.json): FETCH_SIZE under-reports wide coalesced streams by 2x on gfx950.
MFMA: SQ_INSTS_VALU_MFMA_MOPS_BF16 counts matrix math operations / 512 -> flops = 512 x count; utilisation of a kernel =
flops / duration / 2.5e15 (dense bf16 peak), durations from the kernel trace of the same pass."""
import csv
import glob
import json
import sys
from collections import defaultdict

MFMA_PEAK = 2.5e15


def load_counters(dirpath):
    """dispatch id -> {counter: value}, dispatch id -> kernel name (rows of one dispatch summed over dimensions)."""
    path = glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True)[0]
    vals, names = defaultdict(lambda: defaultdict(float)), {}
    with open(path) as f:
        for row in csv.DictReader(f):
            d = int(row["Dispatch_Id"])
            vals[d][row["Counter_Name"]] += float(row["Counter_Value"])
            names[d] = row["Kernel_Name"]
    return vals, names


def load_durations(dirpath):
    path = glob.glob(dirpath + "/**/*kernel_trace.csv", recursive=True)[0]
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            out[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    return out


def step_window(names, nsteps):
    ids = sorted(names)
    marks = [d for d in ids if "clip_adamw" in names[d]]
    if len(marks) < nsteps + 1:
        raise SystemExit(f"only {len(marks)} optimiser steps in the trace, need {nsteps + 1}")
    lo, hi = marks[-nsteps - 1], marks[-1]
    return [d for d in ids if lo < d <= hi]


def short(name):
    name = name.replace("void ", "")
    return name[:110]


def reduce(fdir, wdir, mdir, fetch_bpc, write_bpc, nsteps=2):
    res = {"what": "rocprofv3 --pmc passes over `python bench.py --mode eager` (see scripts/pmc_step.py); averages over the "
                   f"last {nsteps} optimiser steps", "calibration": {"fetch_bytes_per_count": fetch_bpc,
                                                                     "write_bytes_per_count": write_bpc}}
    fv, fn = load_counters(fdir)
    wv, wn = load_counters(wdir)
    fw, ww = step_window(fn, nsteps), step_window(wn, nsteps)
    rd = sum(fv[d]["FETCH_SIZE"] for d in fw) * fetch_bpc / nsteps
    wr = sum(wv[d]["WRITE_SIZE"] for d in ww) * write_bpc / nsteps
    res.update({"dispatches_per_step": len(fw) / nsteps, "hbm_read_bytes_per_step": round(rd),
                "hbm_write_bytes_per_step": round(wr), "hbm_bytes_per_step": round(rd + wr)})
    per = defaultdict(lambda: [0, 0.0, 0.0])
    for d in fw:
        k = per[short(fn[d])]
        k[0] += 1
        k[1] += fv[d]["FETCH_SIZE"] * fetch_bpc / nsteps
    for d in ww:
        per[short(wn[d])][2] += wv[d]["WRITE_SIZE"] * write_bpc / nsteps
    res["top_kernels_by_bytes"] = [{"kernel": k, "calls_per_step": v[0] / nsteps, "read_MB": round(v[1] / 1e6, 2),
                                    "write_MB": round(v[2] / 1e6, 2)}
                                   for k, v in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:25]]
    if mdir:
        mv, mn = load_counters(mdir)
        dur = load_durations(mdir)
        win = step_window(mn, nsteps)
        agg = defaultdict(lambda: defaultdict(float))
        for d in win:
            a = agg[short(mn[d])]
            a["calls"] += 1
            a["seconds"] += dur.get(d, 0.0)
            for c, v in mv[d].items():
                a[c] += v
        tot_s = sum(a["seconds"] for a in agg.values())
        rows = []
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["seconds"])[:30]:
            row = {"kernel": k, "calls_per_step": a["calls"] / nsteps, "ms_per_step": round(a["seconds"] * 1e3 / nsteps, 3),
                   "share_of_gpu_time": round(a["seconds"] / tot_s, 4)}
            if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in a:
                fl = a["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
                row["bf16_mfma_tflops"] = round(fl / a["seconds"] / 1e12, 1) if a["seconds"] else None
                row["mfma_frac_of_2.5PF"] = round(fl / a["seconds"] / MFMA_PEAK, 4) if a["seconds"] else None
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("GRBM_GUI_ACTIVE"):
                row["mfma_busy_over_gui_active"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["GRBM_GUI_ACTIVE"], 2)
            if "SQ_INSTS_MFMA" in a:
                row["mfma_insts"] = a["SQ_INSTS_MFMA"] / a["calls"]
            rows.append(row)
        fl_all = sum(a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) for a in agg.values()) * 512.0
        res["mfma"] = {"peak": "2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md)",
                       "counters": sorted({c for a in agg.values() for c in a if c not in ("calls", "seconds")}),
                       "gpu_kernel_ms_per_step": round(tot_s * 1e3 / nsteps, 3),
                       "whole_step_bf16_mfma_tflops": round(fl_all / tot_s / 1e12, 1) if tot_s and fl_all else None,
                       "whole_step_mfma_frac_of_peak": round(fl_all / tot_s / MFMA_PEAK, 4) if tot_s and fl_all else None,
                       "top_kernels_by_time": rows}
    return res


if __name__ == "__main__":
    if len(sys.argv) >= 7 and sys.argv[1] == "reduce":
        mdir = sys.argv[4] if sys.argv[4] not in ("-", "") else None
        print(json.dumps(reduce(sys.argv[2], sys.argv[3], mdir, float(sys.argv[5]), float(sys.argv[6])), indent=1))
    else:
        raise SystemExit(__doc__)
