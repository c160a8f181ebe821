#!/bin/bash
# round 6, call 27: K5's passes over the residual planes with static slots and row-contiguous loads (+ ds_bpermute to operand order)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_rank16.py tests/test_cli_svd.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -q -k "svd or distill or planes or spectrum or adaptive or rank16" > $O/c27_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c27_tests.log | head -8
for it in "" "4"; do LORA_AMD_SVD_ITERS=$it timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | cut -c1-330; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c27_trace -o svd -- python bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2> $O/c27_traced.err
python scripts/prof_summary.py $(find $O/c27_trace -name "*kernel_trace.csv" | head -1) 12 | cut -c1-150
rm -rf $O/c27_trace
