#!/bin/bash
# round 6, call 43: the factor pass's slab stores as non-temporal stores (a build beside the product's), in the step and in kbench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for lib in "" scripts/ab/liblora_amd_ntstores.so; do
  LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=$lib timeout 900 python scripts/kbench.py --what fm 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product}', {k: (v['us'] if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('mfma_') or k in ('max_rel_diff_valu_vs_matrix_core_last_run',)})"
done
for lib in lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_ntstores.so lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_ntstores.so; do
  timeout 400 python scripts/ab/run_with_lib.py $lib bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$lib', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
