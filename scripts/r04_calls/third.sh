#!/bin/bash
# Round 4, third GPU call: the engine kernel of the factor pass (parity, sweeps), dropout in the pass, bench A/B.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -x -k "factors_mfma" > $OUT/r04c_pytest_fm.log 2>&1
tail -6 $OUT/r04c_pytest_fm.log
for V in "" "LORA_AMD_FM_NB=1" "LORA_AMD_FM_NB=4" "LORA_AMD_FM_NB=8" "LORA_AMD_FM_ROWS=32" "LORA_AMD_FM_ENGINE=0"; do
  tag=$(echo "$V" | tr '=' '_' | tr -d ' ')
  env $V timeout 300 python scripts/kbench.py --what fm 2>&1 | tail -1 > $OUT/r04c_kbench_fm_${tag:-default}.log
  echo "== $V"; cut -c1-1100 $OUT/r04c_kbench_fm_${tag:-default}.log
done
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS > $OUT/r04c_bench_default.json 2> $OUT/r04c_bench_default.err
LORA_AMD_FACTORS_MFMA=0 timeout 400 python bench.py $ARGS > $OUT/r04c_bench_valu.json 2> $OUT/r04c_bench_valu.err
python - <<'PY'
import json
for t in ("default", "valu"):
    try:
        d = json.loads(open(f"gpurun_out/r04c_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], json.dumps(d.get("roofline_in_step"))[:900])
    except Exception as e:
        print(t, "failed", e)
PY
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -k "consecutive or extended" > $OUT/r04c_pytest_steps.log 2>&1
tail -5 $OUT/r04c_pytest_steps.log
