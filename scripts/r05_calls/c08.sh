#!/bin/bash
# round 5, call 8: the three re-run tests + the whole GPU suite once more on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=12 > $O/c08_pytest_full.log 2>&1; echo "full rc=$?"; tail -22 $O/c08_pytest_full.log | cut -c1-200
