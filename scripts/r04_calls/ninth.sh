#!/bin/bash
# Round 4, ninth GPU call: the register-resident form of the matrix-core factor pass (parity on the existing tests, kbench fm)
set -u
OUT=gpurun_out
TAG=r04o
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -q -x -k "factors_mfma" > $OUT/${TAG}_pytest.log 2>&1
tail -6 $OUT/${TAG}_pytest.log
timeout 300 python scripts/kbench.py --what fm > $OUT/${TAG}_kbench_fm.log 2>&1
tail -2 $OUT/${TAG}_kbench_fm.log
