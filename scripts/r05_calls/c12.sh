#!/bin/bash
# round 5, call 12: the final tree once more — whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > $O/c12_pytest_full.log 2>&1; echo "full rc=$?"; tail -14 $O/c12_pytest_full.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c12_bench_line.json 2> $O/c12_bench.err; echo "bench rc=$?"; wc -c $O/c12_bench_line.json; cat $O/c12_bench_line.json
