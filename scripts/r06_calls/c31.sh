#!/bin/bash
# round 6, call 31: the factor pass as a PERSISTENT launch (one workgroup per resident slot, further blocks drawn from the table's
# counter): parity tests of the pass, then kbench with a workgroup per block / one / two workgroups per slot
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py tests/test_gpu_parity_r3.py -q -x -k "factors_mfma or factor_pass or factors_self" > $O/c31_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c31_tests.log | head -8
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c31_kbench_fm.log 2> $O/c31_kbench_fm.err; echo "kbench rc=$?"; tail -3 $O/c31_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c31_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("persist") or k.startswith("mfma_") or k.startswith("class1_") or k.startswith("part_") or k=="max_rel_diff_valu_vs_matrix_core_last_run": print(k, v)
PY
