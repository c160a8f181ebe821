#!/bin/bash
# Round 6, evidence on the final tree: full GPU suite (with durations), smoke, then scripts/r06_profiles.sh (the driver's bench
# command, kernel trace + by-grid summary, merge / merge_step PMC, step PMC)
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -m gpu -q --durations=25 > $OUT/r06_pytest_full.log 2>&1
tail -34 $OUT/r06_pytest_full.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r06_smoke.log 2>&1; tail -1 $OUT/r06_smoke.log
bash scripts/r06_profiles.sh "${1:-bench trace mergepmc steppmc cfg3trace cfg3twin svdtrace}" > $OUT/r06_profiles_stdout.log 2>&1
tail -30 $OUT/r06_profiles_stdout.log | cut -c1-300
wc -c $OUT/r06_bench_line.json
python - <<PY
import json
d = json.loads(open("$OUT/r06_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "traffic", d["roofline"].get("traffic"))
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac")) for k, v in d.get("roofline_in_step", {}).items()})
print("gemm", [(e["kernel"][:40], e["site"], e["avg_launch_us"], e["frac"]) for e in d.get("roofline_fused_gemm", [])])
print("cpu", d.get("cpu_baseline", {}).get("value"), "overhead ms: cfg1", d.get("lora_overhead_ms"), "cfg2", d.get("lora_overhead_ms_cfg2"), "cfg3", d.get("lora_overhead_ms_cfg3"))
for s in d.get("secondary", []): print("  sec", s.get("tag"), s.get("value"), s.get("execution"), s.get("skipped"))
PY
