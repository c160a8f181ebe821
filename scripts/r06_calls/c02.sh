#!/bin/bash
# round 6, call 2: the frozen-twin test, the driver's bench command with the per-config overhead legs (first r06 line),
# a kernel trace of the same command (baseline for the factor-pass and conv-adapter work of this round)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r6.py -x -q -k "frozen_twins" > $O/c02_twins.log 2>&1; echo "twins rc=$?"; tail -3 $O/c02_twins.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/c02_bench_line.json 2> $O/c02_bench.err; echo "bench rc=$?"
wc -c $O/c02_bench_line.json; cat $O/c02_bench_line.json; tail -5 $O/c02_bench.err | cut -c1-400
