#!/bin/bash
# round 6, call 34: operator profile of configs[3] after the head-padded weight-stationary route: which ATen copies / fills / sums
# does the adapter step still issue that its twin does not
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python scripts/op_profile.py cfg3 > $O/c34_op_profile_cfg3.txt 2>&1; echo "op profile rc=$?"; sed -n '/pointwise \/ copy operators/,$p' $O/c34_op_profile_cfg3.txt | cut -c1-230 | head -90
