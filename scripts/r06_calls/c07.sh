#!/bin/bash
# round 6, call 7: the t = 0 diagnosis of the batch-4 fixture; where the extra copy / fill / sum launches of configs[3] come from;
# the svd tests after the fixes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python scripts/r06_bracket_t0.py > $O/c07_bracket_t0.log 2>&1; echo "t0 rc=$?"; grep -E "^\[t0\]|Error" $O/c07_bracket_t0.log | cut -c1-250
timeout 600 python scripts/op_profile.py cfg3 > $O/c07_op_profile_cfg3.txt 2>&1; echo "op profile rc=$?"; sed -n '/copy \/ fill \/ sum launches/,$p' $O/c07_op_profile_cfg3.txt | cut -c1-230 | head -50
timeout 600 python -m pytest tests/test_gpu_svd_small.py -q > $O/c07_svd_tests.log 2>&1; echo "svd tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c07_svd_tests.log | cut -c1-200 | head
