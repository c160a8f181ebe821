#!/bin/bash
# Round 4, eighth GPU call: rank-16 kernels, second form (column-split rowdot, LDS-shared T fragments in rank_update)
set -u
OUT=gpurun_out
TAG=r04l
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_rank16.py -q -x > $OUT/${TAG}_pytest.log 2>&1
tail -4 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what r16 > $OUT/${TAG}_kbench_r16.log 2>&1
grep kernel $OUT/${TAG}_kbench_r16.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kernel'], d['M'], d['C'], 'mfma', d['mfma_us'], d['mfma_GBs'], 'valu', d['valu_us'], d['valu_GBs'])"
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS > $OUT/${TAG}_cfg3.json 2> $OUT/${TAG}_cfg3.err
LORA_AMD_AB=RANK16_MFMA=0,DEFER_MASKED_FACTORS=0 timeout 400 python bench.py $ARGS --no-roofline > $OUT/${TAG}_cfg3_r3.json 2> $OUT/${TAG}_cfg3_r3.err
python - <<PY
import json
for t in ("cfg3", "cfg3_r3"):
    try:
        d = json.loads(open("$OUT/${TAG}_%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["config"]["execution"], d["config"]["adapter_options"].get("ab_overrides"))
        if d.get("adapter_path"):
            print("   adapter_path", d["adapter_path"]["gpu_ms_per_step"], "of", d["adapter_path"]["all_kernels_gpu_ms_per_step"])
            for k, v in list(d["adapter_path"]["kernels"].items())[:16]: print("      ", k[:90], v)
    except Exception as e:
        print(t, "failed", e)
PY
