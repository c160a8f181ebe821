#!/usr/bin/env python
"""Per-kernel SQ / TCC counters of the factor-gradient passes (scripts/kbench.py --what fm under rocprofv3 --pmc):

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d gpurun_out/fm_pmc_X -o p -- python scripts/kbench.py --what fm --iters 2
    python scripts/fm_pmc.py gpurun_out/fm_pmc_X [more dirs]     -> one JSON line per kernel family: mean counters, mean duration

(one counter group per pass; --kernel-trace only besides --pmc)."""
import csv
import glob
import json
import sys
from collections import defaultdict

FAMILIES = ("factors_reg_kernel<lora_amd::bf16_t, false, 6, 3, 2>", "factors_reg_kernel<lora_amd::bf16_t, false, 6, 3, 4>",
            "factors_reg_kernel<lora_amd::bf16_t, false, 10, 2, 4>", "factors_mfma_engine_kernel", "factors_mfma_kernel",
            "linear_bwd_factors_self_ragged_kernel", "factor_pack_kernel", "reduce_batched_kernel", "merge_step_kernel",
            "merge_co_kernel")


def main(dirs):
    out = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        cpath = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        tpath = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        dur = {}
        if tpath:
            with open(tpath[0]) as f:
                for row in csv.DictReader(f):
                    dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
        if not cpath:
            continue
        per = defaultdict(lambda: defaultdict(float))
        names = {}
        with open(cpath[0]) as f:
            for row in csv.DictReader(f):
                k = int(row["Dispatch_Id"])
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                names[k] = row["Kernel_Name"]
        for k, cs in per.items():
            fam = next((fm for fm in FAMILIES if fm in names[k]), None)
            if fam is None:
                continue
            grid = names[k]
            for c, v in cs.items():
                out[fam][c].append(v)
            if k in dur:
                out[fam]["duration_us"].append(dur[k])
    for fam, cs in out.items():
        print(json.dumps({"kernel": fam, "dispatches": max(len(v) for v in cs.values()),
                          **{c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}}))


if __name__ == "__main__":
    main(sys.argv[1:])
