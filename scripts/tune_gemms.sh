#!/bin/bash
# PyTorch TunableOp over the step's library GEMMs (the merged-weight sites' X W_eff^T / G W_eff and the host model's own
# projections): every GEMM shape of one eager step is timed against all hipBLASLt / rocBLAS solutions once, the picks go to
# a CSV that later runs only READ (PYTORCH_TUNABLEOP_TUNING=0).  Usage: bash scripts/tune_gemms.sh [out.csv] [bench flags]
set -u
OUT=${1:-gpurun_out/tunableop_mi355x.csv}
shift || true
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$OUT
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-15} PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=${TUNE_ITERS:-30}
export PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 PYTORCH_TUNABLEOP_VERBOSE=0
python bench.py --mode eager --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary "$@"
