#!/bin/bash
# round 6, call 29: reduce_batched with four outputs per thread (16-byte loads of the slabs)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py tests/test_gpu_kernels.py tests/test_gpu_conv_nhwc.py -q -x -k "reduce or factors or factor_pass or conv or consecutive" > $O/c29_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c29_tests.log | head -8
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c29_kbench_fm.log 2> $O/c29_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c29_kbench_fm.log").read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if "fold" in k or k.startswith("mfma_pass") or k.startswith("max_rel")})
PY
timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c29_bench.json 2> $O/c29_bench.err
python - <<PY
import json
d=json.loads(open("$O/c29_bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline_in_step"]["factor_pass"])
PY
