#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel calls / total / avg / share (sorted by total)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
by_grid = len(sys.argv) > 3 and sys.argv[3] == "by-grid"  # separate rows per launch geometry (= per problem shape)
agg = defaultdict(lambda: [0, 0])
with open(path) as f:
    for row in csv.DictReader(f):
        d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        key = row["Kernel_Name"]
        if by_grid:
            key = f'[{row.get("Grid_Size_X", "?")}x{row.get("Grid_Size_Y", "?")}x{row.get("Grid_Size_Z", "?")}] ' + key
        a = agg[key]
        a[0] += 1
        a[1] += d
tot = sum(v[1] for v in agg.values())
print(f"# {path}: {sum(v[0] for v in agg.values())} dispatches, {tot / 1e6:.3f} ms GPU kernel time")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'pct':>6}  kernel")
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.2f} {100 * t / tot:6.2f}  {name[:150]}")
