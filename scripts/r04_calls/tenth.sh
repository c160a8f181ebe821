#!/bin/bash
# Round 4, tenth GPU call: register-resident factor pass as the default for every deferred site (maskless + dropout)
set -u
OUT=gpurun_out
TAG=r04p
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r3.py -q -x -k "factors_mfma or selection_modes or consecutive or sd15_size_step or merged_weight or extended_rank16 or sd15_unet_plus_clip" > $OUT/${TAG}_pytest.log 2>&1
tail -5 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what fm > $OUT/${TAG}_kbench_fm.log 2>&1
tail -1 $OUT/${TAG}_kbench_fm.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if not isinstance(v, dict)}); print(d.get('mfma_class1'), d.get('mfma_class2'))"
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
LORA_AMD_FACTORS_MFMA=masked timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-roofline > $OUT/${TAG}_bench_valu.json 2> $OUT/${TAG}_bench_valu.err
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
timeout 400 python bench.py $ARGS > $OUT/${TAG}_cfg3.json 2> $OUT/${TAG}_cfg3.err
python - <<PY
import json
d = json.loads(open("$OUT/${TAG}_bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "K3", d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("in-step", {k: (v.get("avg_launch_us"), v.get("frac"), v.get("kernel", "")[:60]) for k, v in d.get("roofline_in_step", {}).items()})
print("adapter_path", d["adapter_path"]["gpu_ms_per_step"])
for k, v in list(d["adapter_path"]["kernels"].items())[:6]: print("   ", k[:70], v)
for t in ("bench_valu", "cfg3"):
    e = json.loads(open("$OUT/${TAG}_%s.json" % t).read().strip().splitlines()[-1])
    print(t, e["value"], e["ms_per_step"], e["config"]["execution"])
PY
