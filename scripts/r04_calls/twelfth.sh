#!/bin/bash
# Round 4, last GPU call: cli_svd's big products on the matrix cores over (hi, lo) planes — parity and configs[4] timing
set -u
OUT=gpurun_out
TAG=r04z
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 500 python -m pytest tests/test_gpu_rank16.py tests/test_gpu_parity_r3.py tests/test_cli_svd.py tests/test_gpu_parity_r2.py -m gpu -q -x -k "planes or svd or distill or quantile" > $OUT/${TAG}_pytest.log 2>&1
tail -6 $OUT/${TAG}_pytest.log
timeout 300 python bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > $OUT/${TAG}_bench_svd.json 2> $OUT/${TAG}_bench_svd.err
tail -c 700 $OUT/${TAG}_bench_svd.json
