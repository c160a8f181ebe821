#!/bin/bash
# Round 4, final evidence on the final tree: full GPU suite, smoke, then scripts/r04_profiles.sh (bench line, kernel trace,
# merge / merge_step PMC, configs[3] trace, step PMC)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -m gpu -q > $OUT/r04_pytest_full.log 2>&1
tail -4 $OUT/r04_pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r04_smoke.log 2>&1; tail -1 $OUT/r04_smoke.log
bash scripts/r04_profiles.sh "bench trace mergepmc cfg3trace steppmc" > $OUT/r04_profiles_stdout.log 2>&1
tail -30 $OUT/r04_profiles_stdout.log | cut -c1-300
python - <<PY
import json
d = json.loads(open("$OUT/r04_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"].get("traffic"))
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac")) for k, v in d.get("roofline_in_step", {}).items()})
print("cpu", d.get("cpu_baseline", {}).get("value"))
for s in d.get("secondary", []): print("  sec", s.get("tag"), s.get("value"), s.get("execution"), s.get("skipped"))
PY
