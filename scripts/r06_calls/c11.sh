#!/bin/bash
# round 6, call 11: the factor pass with static ring slots (no copies of in-flight registers): parity, kbench per class / ring,
# the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -k "factors_mfma or factor_pass" > $O/c11_fm_tests.log 2>&1; echo "fm tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c11_fm_tests.log | head -8
timeout 600 python scripts/kbench.py --what fm > $O/c11_kbench_fm.log 2>&1; tail -1 $O/c11_kbench_fm.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'class' in k or k in ('mfma_pass_us','mfma_frac8','mfma_fold_us','pack_us')})"
for m in 1 2 3 4; do
LORA_AMD_FM_NARROW=$m timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c11_bench_$m.json 2> $O/c11_bench_$m.err
python - <<PY
import json
d=json.loads(open("$O/c11_bench_$m.json").read().strip().splitlines()[-1]); print("narrow mode $m", d["value"], d["ms_per_step"], d["roofline_in_step"]["factor_pass"]["avg_launch_us"], d["roofline_in_step"]["factor_pass"]["frac"])
PY
done
