#!/bin/bash
# round 5, call 16 (measurement only, no product change): where the three long whole-step tests spend their time (cold box,
# then the 768^2 one again in a new process with the box's caches warm), and the kernel trace of the final tree's bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
export LORA_AMD_TEST_LAPS=1
T="tests/test_gpu_parity_r3.py::test_sd15_size_step_in_bench_configuration_matches_oracle tests/test_gpu_parity_r4.py::test_extended_rank16_768_step_with_dropout_matches_oracle"
( time timeout 400 python -m pytest $T -q -s --durations=5 ) > $O/c16_laps_cold.log 2>&1; echo "cold rc=$?"; grep -E "^\[lap\]|passed|failed|^real" $O/c16_laps_cold.log
( time timeout 300 python -m pytest tests/test_gpu_parity_r4.py::test_extended_rank16_768_step_with_dropout_matches_oracle -q -s ) > $O/c16_laps_warm.log 2>&1; echo "warm rc=$?"; grep -E "^\[lap\]|passed|failed|^real" $O/c16_laps_warm.log
unset LORA_AMD_TEST_LAPS
bash scripts/r05_profiles.sh trace > $O/c16_trace.log 2>&1; head -30 $O/r05_bench_kernel_trace_summary.txt | cut -c1-170; cat $O/r05_bench_traced.json | cut -c1-400
