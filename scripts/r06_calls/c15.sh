#!/bin/bash
# round 6, call 15: class 2 of the factor pass as ONE launch for both block heights; the selection test on loss-equal steps; K5 with
# the adaptive loop's repeated last pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -k "factors_mfma or factor_pass or consecutive_optimizer_steps or extended_rank16" > $O/c15_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|skipped|^FAILED|^E   " $O/c15_tests.log | head -6
timeout 600 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_rank16.py tests/test_cli_svd.py -q -k "svd or distill or planes or spectrum or adaptive" > $O/c15_svd_tests.log 2>&1; echo "svd tests rc=$?"; tail -1 $O/c15_svd_tests.log
for it in "" "4"; do LORA_AMD_SVD_ITERS=$it timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | cut -c1-330; done
timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c15_bench.json 2> $O/c15_bench.err
python - <<PY
import json
d=json.loads(open("$O/c15_bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline_in_step"]["factor_pass"], d["config"].get("execution"))
PY
