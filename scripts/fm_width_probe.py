#!/usr/bin/env python
"""Does the 3-3-2-2 split of a 320-wide operand's ten column groups over the four waves cost the factor pass?  The same table
(30 sites, M = 16384, rank 4, one (G, X) per site) at widths 192 (2-2-1-1), 256 (2-2-2-2) and 320 (3-3-2-2); 384 is register class 2: time and bytes / us
of ONE launch of lora_amd_linear_bwd_factors_mfma_ragged.  Run on the GPU box: python scripts/fm_width_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C  # noqa: E402
from scripts.kbench import timeit  # noqa: E402

DEV = "cuda:0"
dt, r, M, NS = torch.bfloat16, 4, 16384, 30
out = {}
for W in (192, 256, 320, 192, 256, 320):
    rows, packs = [], []
    for i in range(NS):
        g = torch.randn(M, W, device=DEV).to(dt)
        x = torch.randn(M, W, device=DEV).to(dt)
        down, up = torch.randn(r, W, device=DEV) * 0.25, torch.randn(W, r, device=DEV) * 0.05
        pl = _C.factors_mfma_plan(M, W, W, r, dt)
        assert pl.supported and int(pl.lds_class) == 1 and int(pl.rows_per_block) == 64, (W, pl.lds_class, pl.rows_per_block)
        up_part, down_part = torch.empty(int(pl.up_part_floats), device=DEV), torch.empty(int(pl.down_part_floats), device=DEV)
        pk_down = torch.empty(int(pl.pack_down_elems), dtype=dt, device=DEV)
        pk_up = torch.empty(int(pl.pack_up_elems), dtype=dt, device=DEV)
        packs.append((down, up, pk_down, pk_up))
        rows.append((g, x, pk_down, pk_up, up_part, down_part, 1.0, None, None, r, pl))
    arr, total = _C.factor_pack_table(packs)
    _C.factor_pack(_C.table_to_device(arr, DEV), len(packs), total, dt)
    arr, grid = _C.factors_mfma_table(rows, dt, 1)
    tab = _C.table_to_device(arr, DEV)
    t, _ = timeit(lambda: _C.linear_bwd_factors_mfma_ragged(tab, NS, grid, 1, dt, False, 64), inner=5)
    b = NS * M * 2 * W * 2
    out.setdefault(W, []).append({"us": round(t * 1e6, 1), "GB": round(b / 1e9, 4), "TBs": round(b / t / 1e12, 3),
                                  "blocks": grid, "us_per_block_slot": round(t * 1e6 * 768 / grid, 2)})
    del rows, packs, tab
    torch.cuda.empty_cache()
print(json.dumps(out))
