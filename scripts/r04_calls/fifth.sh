#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -k "factors_mfma or merge_step" > $OUT/r04f_pytest_fm.log 2>&1
tail -3 $OUT/r04f_pytest_fm.log
for V in "" "LORA_AMD_FM_NB=1" "LORA_AMD_FM_NB=4" "LORA_AMD_FM_ENGINE=0"; do
  tag=$(echo "$V" | tr '=' '_' | tr -d ' ')
  env $V timeout 300 python scripts/kbench.py --what fm 2>&1 | tail -1 > $OUT/r04f_kbench_fm_${tag:-default}.log
  echo "== $V"; cut -c1-1300 $OUT/r04f_kbench_fm_${tag:-default}.log
done
timeout 300 python scripts/kbench.py --what merge > $OUT/r04f_kbench_merge.log 2>&1; tail -4 $OUT/r04f_kbench_merge.log | cut -c1-600
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS > $OUT/r04f_bench_default.json 2> $OUT/r04f_bench_default.err
LORA_AMD_FACTORS_MFMA=0 timeout 400 python bench.py $ARGS > $OUT/r04f_bench_valu.json 2> $OUT/r04f_bench_valu.err
python - <<'PY'
import json
for t in ("default", "valu"):
    try:
        d = json.loads(open(f"gpurun_out/r04f_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], json.dumps(d.get("roofline_in_step"))[:700])
    except Exception as e:
        print(t, "failed", e)
PY
