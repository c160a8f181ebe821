#!/bin/bash
# Round 4, eleventh GPU call: one-launch register-resident factor pass with streaming loads
set -u
OUT=gpurun_out
TAG=${1:-r04q}
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 300 python -m pytest tests/test_gpu_parity_r4.py -q -x -k "factors_mfma or selection_modes" > $OUT/${TAG}_pytest.log 2>&1
tail -3 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what fm > $OUT/${TAG}_kbench_fm.log 2>&1
tail -1 $OUT/${TAG}_kbench_fm.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if not isinstance(v, dict)})"
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac"), v.get("launches")) for k, v in d.get("roofline_in_step", {}).items()})
print("adapter_path", d["adapter_path"]["gpu_ms_per_step"])
for k, v in list(d["adapter_path"]["kernels"].items())[:4]: print("   ", k[:70], v)
PY
