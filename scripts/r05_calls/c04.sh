#!/bin/bash
# round 5, call 4: the input-stationary K1 kernel (tests + kbench), K5 after the write-through hand-off, r5 parity tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_xs.py -q -x > $O/c04_xs.log 2>&1; echo "xs tests rc=$?"; tail -12 $O/c04_xs.log | cut -c1-300
timeout 300 python scripts/kbench.py --what xs > $O/c04_kbench_xs.log 2>&1; cat $O/c04_kbench_xs.log | cut -c1-600
timeout 300 python -m pytest tests/test_gpu_svd_small.py -q -x > $O/c04_svdtests.log 2>&1; echo "svd tests rc=$?"; tail -3 $O/c04_svdtests.log
timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $O/c04_svd.json 2> $O/c04_svd.err; echo "svd rc=$?"; cat $O/c04_svd.json | cut -c1-700
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/svdtrace -o svd -- python $GRAFT_REPO_ROOT/bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/svdtrace -name "*kernel_stats.csv" | head -1); python scripts/stats_top.py "$f" 25 > $O/c04_svd_kernel_stats.txt; head -14 $O/c04_svd_kernel_stats.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_parity_r5.py tests/test_gpu_parity_r3.py tests/test_gpu_parity_r4.py -q -s --durations=12 > $O/c04_parity.log 2>&1; echo "parity rc=$?"; grep -E "ratio|tensor [0-9]+ \(|passed|failed|Error|^FAILED|s call|s setup" $O/c04_parity.log | head -50
