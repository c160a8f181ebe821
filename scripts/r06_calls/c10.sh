#!/bin/bash
# round 6, call 10: class 1 of the factor pass on deeper B rings (6 / 8 / 12 units at two workgroups per CU); the flaky selection test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python scripts/kbench.py --what fm > $O/c10_kbench_fm.log 2>&1; tail -1 $O/c10_kbench_fm.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'class1_kernel' in k or k in ('reg_class1_us','reg_class2_us','reg_pass_us')})"
timeout 300 python -m pytest tests/test_gpu_parity_r4.py -q -k "selection_modes" > $O/c10_sel.log 2>&1; echo "selection rc=$?"; tail -1 $O/c10_sel.log
