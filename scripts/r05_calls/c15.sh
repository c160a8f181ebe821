#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python scripts/kbench.py --what fm --iters 10 > $O/c15_kbench_fm.log 2>&1; grep "^{" $O/c15_kbench_fm.log | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print({k:d[k] for k in d if k in ('valu_fold_us','mfma_fold_us','mfma_partial_MB','reg_one_launch_us','reg_one_launch_frac8','pack_us','max_rel_diff_up','max_rel_diff_down') or 'diff' in k})"
timeout 300 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r3.py tests/test_gpu_kernels.py -q -k "factors or reduce or training_steps or merged_weight" > $O/c15_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/c15_tests.log
