#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_svd_small.py tests/test_cli_svd.py tests/test_cli_gpu.py -q > $O/c13_svd.log 2>&1; echo "svd rc=$?"; tail -4 $O/c13_svd.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q > $O/c13_pytest_full.log 2>&1; echo "full rc=$?"; tail -3 $O/c13_pytest_full.log | cut -c1-200
