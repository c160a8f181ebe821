#!/bin/bash
# round 6, call 44: K5's passes over the residual planes with TWO slabs per wave (a k-step's 2 KB of factor fragments against 2 KB of
# hi-plane data instead of 1 KB): parity, then bench.py --svd with this library and with the previous commit's, alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_rank16.py tests/test_cli_svd.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -q -x -k "svd or distill or planes or spectrum or adaptive or rowdot16" > $O/c44_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c44_tests.log | head -8
for lib in lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so; do
  for it in "" 4; do
    LORA_AMD_SVD_ITERS=$it timeout 300 python scripts/ab/run_with_lib.py $lib bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'iters', '$it' or 'adaptive', d['value'], d['ms_per_step'], d['config']['power_iterations'], d['roofline']['frac'])"
  done
done
