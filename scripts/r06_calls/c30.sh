#!/bin/bash
# round 6, call 30: K5's adaptive loop decides one iteration ahead (no repeated pass): tests, bench both ways
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_rank16.py tests/test_cli_svd.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -q -k "svd or distill or planes or spectrum or adaptive" > $O/c30_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c30_tests.log | head -8
for it in "" "4" "" "4"; do LORA_AMD_SVD_ITERS=$it timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iters', '$it' or 'adaptive', d['value'], d['ms_per_step'], d['config']['power_iterations'], d['roofline']['frac'])"; done
