#!/bin/bash
# Round 4, seventh GPU call: rank-16 matrix-core kernels (parity, per-kernel A/B, configs[3] A/B), merge_step default 128x128
set -u
OUT=gpurun_out
TAG=r04k
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_rank16.py tests/test_gpu_parity_r4.py -q -x -k "rank16 or rowdot16 or rank_update16 or bwd_g16 or linear_fwd16 or merge_step or extended_rank16" > $OUT/${TAG}_pytest.log 2>&1
tail -15 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what r16 > $OUT/${TAG}_kbench_r16.log 2>&1
grep kernel $OUT/${TAG}_kbench_r16.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kernel'], d['M'], d['C'], 'mfma', d['mfma_us'], d['mfma_GBs'], 'valu', d['valu_us'], d['valu_GBs'])"
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 400 python bench.py $ARGS --path-log $OUT/${TAG}_cfg3_paths.json > $OUT/${TAG}_cfg3.json 2> $OUT/${TAG}_cfg3.err
LORA_AMD_AB=RANK16_MFMA=0 timeout 400 python bench.py $ARGS --no-roofline > $OUT/${TAG}_cfg3_valu.json 2> $OUT/${TAG}_cfg3_valu.err
LORA_AMD_AB=RANK16_MFMA=0,DEFER_MASKED_FACTORS=0 timeout 400 python bench.py $ARGS --no-roofline > $OUT/${TAG}_cfg3_r3.json 2> $OUT/${TAG}_cfg3_r3.err
python - <<PY
import json
from collections import Counter
for t in ("cfg3", "cfg3_valu", "cfg3_r3"):
    try:
        d = json.loads(open("$OUT/${TAG}_%s.json" % t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["config"]["execution"], d["config"].get("adapter_options"))
        if d.get("adapter_path"):
            print("   adapter_path", d["adapter_path"]["gpu_ms_per_step"], "of", d["adapter_path"]["all_kernels_gpu_ms_per_step"])
            for k, v in list(d["adapter_path"]["kernels"].items())[:14]: print("      ", k[:90], v)
            print(json.dumps(d["config"]["kernel_choices"]))
    except Exception as e:
        print(t, "failed", e)
PY
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac")) for k, v in d.get("roofline_in_step", {}).items()})
PY
