#!/bin/bash
# round 6, call 3: the factor pass per register class (class 1 on its three kernels), parity of the narrow kernel, the step with
# one launch (rounds 4-5) against one launch per class, the cfg2 frozen-twin leg that crashed in c02
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r6.py tests/test_gpu_parity_r4.py -x -q -k "factor_pass or factors_mfma" > $O/c03_fm_tests.log 2>&1; echo "fm tests rc=$?"; tail -2 $O/c03_fm_tests.log
timeout 400 python scripts/kbench.py --what fm > $O/c03_kbench_fm.log 2>&1; tail -1 $O/c03_kbench_fm.log
LORA_AMD_FM_SHARED=1 timeout 400 python scripts/kbench.py --what fm > $O/c03_kbench_fm_shared.log 2>&1; tail -1 $O/c03_kbench_fm_shared.log
for ab in "FM_TWO_CLASSES=1" "FM_TWO_CLASSES=0"; do
  LORA_AMD_AB=$ab timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c03_bench_$ab.json 2> $O/c03_bench_$ab.err
  python - <<PY
import json
d=json.loads(open("$O/c03_bench_$ab.json").read().strip().splitlines()[-1])
print("$ab", d["value"], d["ms_per_step"], d.get("roofline_in_step",{}).get("factor_pass"))
PY
done
timeout 300 python bench.py --adapters none --text-encoder 1 --rank 8 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > $O/c03_frozen_cfg2.json 2> $O/c03_frozen_cfg2.err; echo "frozen cfg2 rc=$?"; cut -c1-300 $O/c03_frozen_cfg2.json; tail -3 $O/c03_frozen_cfg2.err | cut -c1-300
