#!/bin/bash
# round 6, call 22: kernel trace of the step with the tables longest-first (per-launch times of the two factor-pass launches)
cd $GRAFT_REPO_ROOT
bash scripts/r06_profiles.sh trace > gpurun_out/c22.log 2>&1
grep -E "factors_reg|factor_pack|reduce_batched|merge_step|merge_co" gpurun_out/r06_bench_kernel_trace_by_grid.txt | cut -c1-200
cat gpurun_out/r06_bench_traced.json | cut -c1-300
