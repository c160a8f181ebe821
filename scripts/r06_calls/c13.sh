#!/bin/bash
# round 6, call 13: the three SVD tests that failed in the full suite (last dW Qz on both planes now), svd bench; the factor-pass
# launches of a flush on forked streams: parity of the whole step + same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_rank16.py tests/test_cli_svd.py -q -k "svd or distill or planes or spectrum or quantile or adaptive or hand_off" > $O/c13_svd_tests.log 2>&1; echo "svd tests rc=$?"; grep -E "passed|failed|^FAILED" $O/c13_svd_tests.log | head
for it in "" "4"; do LORA_AMD_SVD_ITERS=$it timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_parity_r4.py -q -k "consecutive_optimizer_steps or extended_rank16 or selection_modes" > $O/c13_step_tests.log 2>&1; echo "step tests rc=$?"; tail -1 $O/c13_step_tests.log
for ab in "CONCURRENT_FACTOR_LAUNCHES=1" "CONCURRENT_FACTOR_LAUNCHES=0" "CONCURRENT_FACTOR_LAUNCHES=1"; do
  LORA_AMD_AB=$ab timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c13_bench_$ab.json 2> $O/c13_bench_$ab.err
  python - <<PY
import json
d=json.loads(open("$O/c13_bench_$ab.json").read().strip().splitlines()[-1]); print("$ab", d["value"], d["ms_per_step"], d["roofline_in_step"]["factor_pass"]["avg_launch_us"], d["roofline_in_step"]["factor_pass"]["frac"], d["config"].get("execution"))
PY
done
