#!/bin/bash
# Round 4, first GPU call: the matrix-core factor pass (probe, parity, kbench variants), a same-box headline A/B and the
# new whole-step parity tests.  Writes gpurun_out/r04a_*.
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -q -s -k "tr16 or factors_mfma or philox" > $OUT/r04a_pytest_fm.log 2>&1
tail -5 $OUT/r04a_pytest_fm.log
timeout 300 python scripts/kbench.py --what fm > $OUT/r04a_kbench_fm.log 2>&1
LORA_AMD_FM_ROWS=32 timeout 300 python scripts/kbench.py --what fm > $OUT/r04a_kbench_fm_rows32.log 2>&1
LORA_AMD_FM_GATHER=1 timeout 300 python scripts/kbench.py --what fm > $OUT/r04a_kbench_fm_gather.log 2>&1
tail -2 $OUT/r04a_kbench_fm.log $OUT/r04a_kbench_fm_rows32.log $OUT/r04a_kbench_fm_gather.log
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
LORA_AMD_FACTORS_MFMA=1 timeout 400 python bench.py $ARGS > $OUT/r04a_bench_mfma.json 2> $OUT/r04a_bench_mfma.err
LORA_AMD_FACTORS_MFMA=0 timeout 400 python bench.py $ARGS > $OUT/r04a_bench_valu.json 2> $OUT/r04a_bench_valu.err
python - <<'PY'
import json
for t in ("mfma", "valu"):
    try:
        d = json.loads(open(f"gpurun_out/r04a_bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d.get("adapter_path", {}).get("device_ms"))
    except Exception as e:
        print(t, "failed", e)
PY
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py -q -s -k "consecutive or trajectory or clip or extended" > $OUT/r04a_pytest_steps.log 2>&1
tail -30 $OUT/r04a_pytest_steps.log
