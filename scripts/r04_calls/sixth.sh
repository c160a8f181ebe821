#!/bin/bash
# Round 4, sixth GPU call: new tests, merge_step variants, the bench line with the graph-capture fix, configs[3] paths + per-shape trace
set -u
OUT=gpurun_out
TAG=r04j
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -q -k "selection_modes or merge_step or consecutive" > $OUT/${TAG}_pytest.log 2>&1
tail -5 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what mstep > $OUT/${TAG}_kbench_mstep.log 2>&1
grep merge_step $OUT/${TAG}_kbench_mstep.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tile'], d['dither'], round(d['us'], 1), round(d['frac8'], 3))"
timeout 900 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac")) for k, v in d.get("roofline_in_step", {}).items()})
for s in d.get("secondary", []): print("  sec", s.get("tag"), s.get("value"), s.get("execution"), s.get("skipped"))
PY
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS --path-log $OUT/${TAG}_cfg3_paths.json > $OUT/${TAG}_cfg3.json 2> $OUT/${TAG}_cfg3.err
python - <<PY
import json
from collections import Counter
d = json.loads(open("$OUT/${TAG}_cfg3.json").read().strip().splitlines()[-1])
print("cfg3", d["value"], d["ms_per_step"], d["config"]["execution"])
print(json.dumps(d["config"]["kernel_choices"]))
c = Counter(tuple(r) for r in json.load(open("$OUT/${TAG}_cfg3_paths.json")))
for k, v in sorted(c.items()): print("   ", v, k)
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o cfg3 -- python bench.py $ARGS --no-roofline > $OUT/${TAG}_cfg3_traced.json 2> $OUT/${TAG}_cfg3_traced.err
python scripts/prof_summary.py $(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1) 120 by-grid | grep -E "lora_amd|^#|calls" > $OUT/${TAG}_cfg3_by_grid.txt
rm -rf $OUT/${TAG}_trace
head -70 $OUT/${TAG}_cfg3_by_grid.txt | cut -c1-170
