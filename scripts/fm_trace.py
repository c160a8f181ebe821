#!/usr/bin/env python
"""Cycle stamps of ONE workgroup of the engine factor pass (LORA_AMD_FM_TRACE=1): where a row block's time goes.

One 320 x 320 attention site, M = 16384 (256 row blocks of 64): consumer wave 0 and the loader wave stamp clock64() around
every barrier; printed as per-phase cycles of the traced workgroup's row blocks.  A diagnostic, not a product path."""
import os
import sys

import torch

os.environ["LORA_AMD_FM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lora_amd import _C  # noqa: E402

DEV = "cuda:0"
M, K, N, r = 16384, 320, int(sys.argv[1]) if len(sys.argv) > 1 else 320, 4
dt = torch.bfloat16
nsites = int(sys.argv[2]) if len(sys.argv) > 2 else 30
trace = torch.zeros(1024, dtype=torch.int64, device=DEV)
sites, packs = [], []
for i in range(nsites):
    g, x = torch.randn(M, N, device=DEV).to(dt), torch.randn(M, K, device=DEV).to(dt)
    down, up = torch.randn(r, K, device=DEV) * 0.25, torch.randn(N, r, device=DEV) * 0.05
    pl = _C.factors_mfma_plan(M, K, N, r, dt)
    up_part, down_part = torch.empty(int(pl.up_part_floats), device=DEV), torch.empty(int(pl.down_part_floats), device=DEV)
    pk_down, pk_up = torch.empty(int(pl.pack_down_elems), dtype=dt, device=DEV), torch.empty(int(pl.pack_up_elems), dtype=dt, device=DEV)
    packs.append((down, up, pk_down, pk_up))
    sites.append((g, x, pk_down, pk_up, up_part, down_part, 1.0, None, None, r, pl))
print("plan: class", pl.lds_class, "rows", pl.rows_per_block, "blocks/wg", pl.blocks_per_wg, "a_bufs", pl.a_bufs, "lds", pl.lds_bytes)
arr, total = _C.factor_pack_table(packs)
_C.factor_pack(_C.table_to_device(arr, DEV), len(packs), total, dt)
arr, grid = _C.factors_mfma_table(sites, dt, int(pl.lds_class))
arr[0].offset_dev = trace.data_ptr()  # the trace buffer rides in site 0's (unused: no dropout) offset_dev
tab = _C.table_to_device(arr, DEV)
for _ in range(3):
    trace.zero_()
    _C.linear_bwd_factors_mfma_ragged(tab, len(sites), grid, int(pl.lds_class), dt)
torch.cuda.synchronize()
t = trace.cpu().tolist()


def stamps(base):
    out = []
    for i in range(250):
        tag, cyc = t[base + 2 * i], t[base + 2 * i + 1]
        if cyc == 0:
            break
        out.append((tag, cyc))
    return out


for name, base in (("consumer wave 0", 0), ("loader wave", 512)):
    st = stamps(base)
    print(f"--- {name}: {len(st)} stamps; tag:+cycles since the previous stamp")
    line, prev = [], st[0][1] if st else 0
    for tag, cyc in st:
        line.append(f"{tag}:+{cyc - prev}")
        prev = cyc
    print(" ".join(line))
    if st:
        print("total cycles", st[-1][1] - st[0][1])
