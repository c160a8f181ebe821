#!/bin/bash
# round 6, call 1: the fp16 tests (factor pass pre-scaling, VALU routing at ranks 9..16, the fp16 run from up = 0), the
# factor-pass / rank-16 tests of rounds 4-5 on the rebuilt library (ABI 6), smoke, the moved input-stationary experiment
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r6.py -x -q -s > $O/c01_r6tests.log 2>&1; echo "r6 tests rc=$?"
tail -15 $O/c01_r6tests.log
timeout 600 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_rank16.py -x -q -k "factors_mfma or factor_pass or rank16 or rowdot16 or rank_update16 or bwd_g16" > $O/c01_r4tests.log 2>&1; echo "r4/rank16 tests rc=$?"
tail -3 $O/c01_r4tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/c01_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/c01_smoke.log
timeout 600 python -m pytest scripts/gemm_xs/check_gemm_xs.py -x -q -m gpu > $O/c01_xs.log 2>&1; echo "xs rc=$?"; tail -3 $O/c01_xs.log
