#!/bin/bash
# round 5, call 1: the new round-5 test, the driver's bench command (compact line), K1 baseline table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity_r5.py -x -q > $O/c01_r5tests.log 2>&1; echo "r5 tests rc=$?"
tail -3 $O/c01_r5tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c01_bench_line.json 2> $O/c01_bench.err; echo "bench rc=$?"
wc -c $O/c01_bench_line.json; cat $O/c01_bench_line.json
timeout 200 python scripts/kbench.py --what ws > $O/c01_kbench_ws.log 2>&1; tail -30 $O/c01_kbench_ws.log
