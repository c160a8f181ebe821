#!/bin/bash
# round 6, call 41: the factor pass with full-line loads (a wave owns PAIRS of column groups; a load instruction = 8 rows x 128 bytes):
# every factor-pass parity test, kbench with this library and with one built from the previous commit's factor_mfma.hip, the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py tests/test_gpu_ws_heads.py tests/test_gpu_parity_r3.py -q -x -k "factor_pass or factors_mfma or block_map or merged_weight or consecutive or fp16" > $O/c41_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c41_tests.log | head -8
for lib in "" scripts/ab/liblora_amd_old.so; do
  LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=$lib timeout 900 python scripts/kbench.py --what fm 2> /dev/null > $O/c41_kbench_fm_${lib:+old}.log
  python -c "
import sys,json
d=json.loads(open('$O/c41_kbench_fm_${lib:+old}.log').read().strip().splitlines()[-1])
print('${lib:-product}', {k: (v['us'] if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('mfma_class') or k.startswith('part_c') or k.startswith('class1_kernel') or k in ('mfma_pass_us','max_rel_diff_valu_vs_matrix_core_last_run')})"
done
for lib in lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so; do
  timeout 400 python scripts/ab/run_with_lib.py $lib bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$lib', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
