#!/bin/bash
# Round 4, second GPU call: pipelined factor pass (blocks per workgroup sweep), in-step merge kernel, concatenated groups.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -s -k "tr16 or factors_mfma or philox or merge_step" > $OUT/r04b_pytest_kernels.log 2>&1
tail -4 $OUT/r04b_pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -q -k "merged or merge or factors" > $OUT/r04b_pytest_r3_merged.log 2>&1
tail -4 $OUT/r04b_pytest_r3_merged.log
for NB in 1 2 4 8; do
  LORA_AMD_FM_NB=$NB timeout 300 python scripts/kbench.py --what fm 2>&1 | tail -1 > $OUT/r04b_kbench_fm_nb$NB.log
  cat $OUT/r04b_kbench_fm_nb$NB.log | cut -c1-900
done
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS > $OUT/r04b_bench_default.json 2> $OUT/r04b_bench_default.err
LORA_AMD_FACTORS_MFMA=0 timeout 400 python bench.py $ARGS > $OUT/r04b_bench_valu.json 2> $OUT/r04b_bench_valu.err
LORA_AMD_AB=CONCAT_GROUPS=0 timeout 400 python bench.py $ARGS > $OUT/r04b_bench_noconcat.json 2> $OUT/r04b_bench_noconcat.err
LORA_AMD_MERGE_ROUNDING=once timeout 400 python bench.py $ARGS > $OUT/r04b_bench_once.json 2> $OUT/r04b_bench_once.err
python - <<'PY'
import json
for t in ("default", "valu", "noconcat", "once"):
    try:
        d = json.loads(open(f"gpurun_out/r04b_bench_{t}.json").read().strip().splitlines()[-1])
        ap = d.get("adapter_path") or {}
        print(t, d["value"], d["ms_per_step"], {k: ap.get(k) for k in ("device_ms", "merged_gemm_ms", "merged_gemm_calls", "merge_ms", "factor_pass_ms")})
    except Exception as e:
        print(t, "failed", e)
PY
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py -q -s -k "consecutive or trajectory" > $OUT/r04b_pytest_steps.log 2>&1
grep -n "from-zero" $OUT/r04b_pytest_steps.log | cut -c1-1500
tail -5 $OUT/r04b_pytest_steps.log
