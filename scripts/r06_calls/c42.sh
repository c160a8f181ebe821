#!/bin/bash
# round 6, call 42: the pair form (full-line loads) for class 2 + the long-stream sites, the group form for class 1's short blocks:
# parity, kbench both libraries, the step (this library with the long-stream sites moved / not moved, and the previous commit's)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py tests/test_gpu_ws_heads.py tests/test_gpu_parity_r3.py -q -x -k "factor_pass or factors_mfma or block_map or merged_weight or consecutive or fp16" > $O/c42_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c42_tests.log | head -8
for lib in "" scripts/ab/liblora_amd_old.so; do
  LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=$lib timeout 900 python scripts/kbench.py --what fm 2> /dev/null > $O/c42_kbench_fm_${lib:+old}.log
  python -c "
import sys,json
d=json.loads(open('$O/c42_kbench_fm_${lib:+old}.log').read().strip().splitlines()[-1])
print('${lib:-product}', {k: (v['us'] if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('mfma_class') or k.startswith('part_') or k in ('mfma_pass_us','max_rel_diff_valu_vs_matrix_core_last_run')})"
done
for spec in "lora_amd/csrc/liblora_amd.so FM_LONG_STREAMS_TO_CLASS2=1" "lora_amd/csrc/liblora_amd.so FM_LONG_STREAMS_TO_CLASS2=0" "scripts/ab/liblora_amd_old.so FM_LONG_STREAMS_TO_CLASS2=0" "lora_amd/csrc/liblora_amd.so FM_LONG_STREAMS_TO_CLASS2=1" "lora_amd/csrc/liblora_amd.so FM_LONG_STREAMS_TO_CLASS2=0" "scripts/ab/liblora_amd_old.so FM_LONG_STREAMS_TO_CLASS2=0"; do
  set -- $spec
  LORA_AMD_AB=$2 timeout 400 python scripts/ab/run_with_lib.py $1 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$1 $2', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
