#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r3.py -q -s -k "bench_configuration" > $O/c10_r3.log 2>&1; echo "rc=$?"; grep -E "worst cosines|passed|failed|^E " $O/c10_r3.log | head -20
