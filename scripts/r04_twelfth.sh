#!/bin/bash
# Round 4, twelfth GPU call: several row blocks per workgroup in the register-resident factor pass (next block's pieces prefetched)
set -u
OUT=gpurun_out
TAG=r04s
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_parity_r4.py -q -x -k "factors_mfma or selection_modes" > $OUT/${TAG}_pytest.log 2>&1
tail -3 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what fm > $OUT/${TAG}_kbench_fm.log 2>&1
grep -E "reg_pass_us|blocks_per_wg\"" $OUT/${TAG}_kbench_fm.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print({k: v for k, v in d.items() if k in ('factors_mfma_blocks_per_wg', 'reg_pass_us', 'reg_one_launch_us', 'mfma_fold_us', 'valu_pass_us', 'max_rel_diff_valu_vs_matrix_core_last_run')})"
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac")) for k, v in d.get("roofline_in_step", {}).items()})
for k, v in list(d["adapter_path"]["kernels"].items())[:3]: print("   ", k[:70], v)
PY
