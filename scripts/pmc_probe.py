#!/usr/bin/env python
"""Workload for the PMC (FETCH_SIZE / WRITE_SIZE) passes: a calibration copy of known size, then the K3 merge.

Run under `rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...` (separate passes); scripts/pmc_reduce.py
turns the two counter CSVs into per-launch HBM bytes with the gfx950 corrections of MI355X_MICROARCH.md §HBM
(calibrate on a known byte count in the same access pattern: the 16-byte-per-lane copy below).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C  # noqa: E402
from scripts.kbench import merge_plan, mstep_plan, mstep_tensors  # noqa: E402

DEV = "cuda:0"
n = 192_634_880  # = total elements of the 144 SD1.5 sites: same footprint as the merge (beyond the 256 MiB L3)
a = torch.randn(n // 64, device=DEV).to(torch.bfloat16).repeat(64)
b = torch.empty_like(a)
for _ in range(3):
    torch.neg(a, out=b)  # vectorized_elementwise_kernel<8, neg>: reads n*2 bytes, writes n*2 bytes, 16 B per lane
torch.cuda.synchronize()
plan = merge_plan(False)
for _ in range(3):
    plan.launch(0.7)
torch.cuda.synchronize()
splan = mstep_plan(mstep_tensors())   # the in-step merge: W_eff and W_eff^T from one read of W (csrc/merge_step.hip)
for _ in range(3):
    splan.launch(0.7, _C.ROUND_DITHER)
torch.cuda.synchronize()
print("copy_bytes_each_way", n * 2, "merge_algorithmic_bytes", plan.bytes_algorithmic, "merge_step_algorithmic_bytes",
      splan.bytes_algorithmic)
