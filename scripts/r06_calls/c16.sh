#!/bin/bash
# round 6, call 16: the factor pass with a workgroup walking `span` consecutive row blocks (launch + site lookup paid once per run):
# every table at span 1 / 2 / 4 / 8, class 1 by site type, class 2 as the step's one mixed launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -q -k "factors_mfma or factor_pass" > $O/c16_tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/c16_tests.log
timeout 900 python scripts/kbench.py --what fm > $O/c16_kbench_fm.log 2> $O/c16_kbench_fm.err; echo "kbench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/c16_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("span_") or k.startswith("mfma_") or k.startswith("class1_") or k=="max_rel_diff_valu_vs_matrix_core_last_run": print(k, v)
PY
