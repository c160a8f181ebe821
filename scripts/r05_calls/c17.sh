#!/bin/bash
# round 5, call 17: bench.py with NO flags (50 timed steps, secondaries, CPU baseline) on the last tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
( time timeout 420 python bench.py > $O/c17_default_line.json 2> $O/c17_default.err ) 2>&1 | grep real; echo "rc=$?"; wc -c $O/c17_default_line.json; cut -c1-700 $O/c17_default_line.json
