#!/bin/bash
# round 6, call 45: the fold (lora_amd_reduce_batched) with two neighbouring outputs per thread walked by 8-byte loads: parity of everything
# that folds, kbench's fold leg and the step with this library and the previous commit's
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_r2.py tests/test_gpu_parity_r4.py tests/test_gpu_conv_nhwc.py tests/test_gpu_ws_heads.py -q -x -k "reduce or factor or batched or training_steps or conv or module" > $O/c45_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c45_tests.log | head -8
for lib in "" scripts/ab/liblora_amd_old.so; do
  LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=$lib timeout 900 python scripts/kbench.py --what fm 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product}', {k: v for k, v in d.items() if k in ('mfma_fold_us','valu_fold_us','mfma_pass_us','max_rel_diff_valu_vs_matrix_core_last_run')})"
done
for lib in lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so; do
  timeout 400 python scripts/ab/run_with_lib.py $lib bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$lib', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
