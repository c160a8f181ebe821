#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python scripts/kbench.py --what merge --iters 10 > $O/c14_kbench_merge.log 2>&1; grep "^{" $O/c14_kbench_merge.log | cut -c1-400
