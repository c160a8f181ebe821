#!/bin/bash
# configs[3] (extended, rank 16, 768^2, dropout 0.1): same-box A/B of the deferred masked factor pass + a kernel trace
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
timeout 400 python bench.py $ARGS > $OUT/r04h_cfg3_default.json 2> $OUT/r04h_cfg3_default.err
LORA_AMD_AB=DEFER_MASKED_FACTORS=0 timeout 400 python bench.py $ARGS > $OUT/r04h_cfg3_nodefer.json 2> $OUT/r04h_cfg3_nodefer.err
python - <<'PY'
import json
for t in ("default", "nodefer"):
    try:
        d = json.loads(open(f"gpurun_out/r04h_cfg3_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["config"]["execution"], json.dumps(d["config"]["kernel_choices"])[:600])
    except Exception as e:
        print(t, "failed", e)
PY
export LORA_AMD_TUNE_CACHE=/tmp/lora_amd_tune_cfg3.json
timeout 400 python bench.py $ARGS --steps 5 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04h_trace -o cfg3 -- python bench.py $ARGS --steps 10 > $OUT/r04h_cfg3_traced.json 2> $OUT/r04h_cfg3_traced.err
python scripts/prof_summary.py $(find $OUT/r04h_trace -name "*kernel_trace.csv" | head -1) 60 > $OUT/r04h_cfg3_kernel_trace_summary.txt
rm -rf $OUT/r04h_trace
head -30 $OUT/r04h_cfg3_kernel_trace_summary.txt | cut -c1-150
