#!/bin/bash
# round 5, call 3: K5 after the finish speed-ups + fused residual split; the bf16-bracket parity test; full GPU suite durations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_svd_small.py -q -x > $O/c03_svdtests.log 2>&1; echo "svd tests rc=$?"; tail -5 $O/c03_svdtests.log
timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $O/c03_svd.json 2> $O/c03_svd.err; echo "svd rc=$?"; cat $O/c03_svd.json
LORA_AMD_SVD_ITERS=4 timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2>/dev/null | tee $O/c03_svd_fixed4.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/svdtrace -o svd -- python $GRAFT_REPO_ROOT/bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/svdtrace -name "*kernel_stats.csv" | head -1); python scripts/stats_top.py "$f" 25 > $O/c03_svd_kernel_stats.txt; head -22 $O/c03_svd_kernel_stats.txt | cut -c1-170
timeout 600 python -m pytest tests/test_gpu_parity_r5.py -q -x -s > $O/c03_r5.log 2>&1; echo "r5 rc=$?"; grep -E "ratio|passed|failed|Error|assert" $O/c03_r5.log | head -20
timeout 1100 python -m pytest tests -m gpu -q --durations=45 > $O/c03_full.log 2>&1; echo "full rc=$?"; tail -60 $O/c03_full.log
