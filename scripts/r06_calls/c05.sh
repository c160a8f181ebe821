#!/bin/bash
# round 6, call 5: kernel traces of configs[3] and of its frozen twin on the current tree; the batch-4 bracket tests (r3 fixture
# at batch 4 in three configurations, r5 tightened, the r6 bisect under the reference's precision policy); K5 with hi-plane passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
bash scripts/r06_profiles.sh "cfg3trace cfg3twin" > $O/c05_profiles.log 2>&1; tail -30 $O/c05_profiles.log | cut -c1-160
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r5.py tests/test_gpu_parity_r6.py -x -q -s -k "sd15_size_step or bracket" > $O/c05_bracket.log 2>&1; echo "bracket tests rc=$?"
grep -E "^\[bracket|passed|failed|Error|assert" $O/c05_bracket.log | cut -c1-260 | tail -40
timeout 600 python -m pytest tests/test_gpu_svd_small.py tests/test_cli_svd.py tests/test_gpu_parity_r2.py -x -q -k "svd or distill or quantile or spectrum" > $O/c05_svd_tests.log 2>&1; echo "svd tests rc=$?"; tail -3 $O/c05_svd_tests.log
for it in "" "4"; do
  LORA_AMD_SVD_ITERS=$it timeout 300 python bench.py --svd --warmup 1 --no-cpu-baseline > $O/c05_svd_iters_$it.json 2> $O/c05_svd_$it.err; cut -c1-400 $O/c05_svd_iters_$it.json
done
