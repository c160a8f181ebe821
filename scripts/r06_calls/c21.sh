#!/bin/bash
# round 6, call 21: shared-x groups of the factor pass (q / k / v read their x once), the four-per-CU class-1 kernel; tables
# longest-first in the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -k "factors_mfma or factor_pass or consecutive_optimizer_steps or extended_rank16 or shared_x" > $O/c21_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c21_tests.log | head -12
timeout 900 python scripts/kbench.py --what fm > $O/c21_kbench_fm.log 2> $O/c21_kbench_fm.err; echo "kbench rc=$?"; tail -3 $O/c21_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c21_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("part_") or k.startswith("mfma_") or k.startswith("class1_") or k=="max_rel_diff_valu_vs_matrix_core_last_run": print(k, v)
PY
for ab in "FM_SHARE_X=1" "FM_SHARE_X=0" "FM_SHARE_X=1"; do
  LORA_AMD_AB=$ab timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c21_bench.json 2> $O/c21_bench.err
  python - <<PY
import json
d=json.loads(open("$O/c21_bench.json").read().strip().splitlines()[-1]); f=d["roofline_in_step"]["factor_pass"]; print("$ab", d["value"], d["ms_per_step"], f["avg_launch_us"], f["frac"])
PY
done
LORA_AMD_AB=FM_SHARE_X=0 LORA_AMD_FM_NARROW=6 timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c21_bench.json 2> $O/c21_bench.err
python - <<PY
import json
d=json.loads(open("$O/c21_bench.json").read().strip().splitlines()[-1]); f=d["roofline_in_step"]["factor_pass"]; print("no share, narrow 6", d["value"], d["ms_per_step"], f["avg_launch_us"], f["frac"])
PY
