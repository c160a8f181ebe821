#!/bin/bash
# Round 4: the full GPU suite + the driver-style bench line + smoke, on one box
set -u
OUT=gpurun_out
TAG=${1:-r04i}
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_full.log 2>&1
tail -6 $OUT/${TAG}_pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log
timeout 1500 python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", json.dumps(d.get("roofline_in_step"))[:600])
print("adapter_path", {k: d["adapter_path"][k] for k in ("gpu_ms_per_step", "frac", "sites_fwd_library_gemm_on_merged_weight")})
for k, v in list(d["adapter_path"]["kernels"].items())[:8]: print("   ", k[:70], v)
print("cpu", d.get("cpu_baseline", {}).get("value"))
for s in d.get("secondary", []): print("  sec", s.get("tag"), s.get("value"), s.get("execution"), s.get("skipped"))
PY
