#!/bin/bash
# round 5, call 5: input-stationary kernel with the panel loop (tests + sweep); whole-step tests after the oracle changes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm_xs.py -q -x > $O/c05_xs.log 2>&1; echo "xs tests rc=$?"; tail -12 $O/c05_xs.log | cut -c1-300
timeout 600 python scripts/kbench.py --what xs --iters 10 > $O/c05_kbench_xs.log 2>&1; cat $O/c05_kbench_xs.log | cut -c1-1500
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r5.py -q -s --durations=8 -k "extended or bracketed or dither or consecutive" > $O/c05_parity.log 2>&1; echo "parity rc=$?"; grep -E "ratio|passed|failed|Error|^FAILED|s call|s setup" $O/c05_parity.log | head -30
