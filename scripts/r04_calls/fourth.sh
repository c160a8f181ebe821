#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -x -k "factors_mfma" > $OUT/r04d_pytest_fm.log 2>&1
tail -3 $OUT/r04d_pytest_fm.log
for V in "" "LORA_AMD_FM_NB=1" "LORA_AMD_FM_NB=2" "LORA_AMD_FM_NB=4" "LORA_AMD_FM_ENGINE=0"; do
  tag=$(echo "$V" | tr '=' '_' | tr -d ' ')
  env $V timeout 300 python scripts/kbench.py --what fm 2>&1 | tail -1 > $OUT/r04d_kbench_fm_${tag:-default}.log
  echo "== $V"; cut -c1-1300 $OUT/r04d_kbench_fm_${tag:-default}.log
done
