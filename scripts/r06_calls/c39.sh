#!/bin/bash
# round 6, call 39: one slab store per column group at rank tiles 4 / 8 (DPP row shift of the second tile's values into the idle
# lanes): parity, then the step under LORA_AMD_AB=FM_MERGED_STORES=1 / 0
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py tests/test_gpu_ws_heads.py tests/test_gpu_parity_r3.py -q -x -k "factor_pass or factors_mfma or block_map or merged_weight or consecutive or fp16" > $O/c39_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c39_tests.log | head -8
for ab in FM_MERGED_STORES=1 FM_MERGED_STORES=0 FM_MERGED_STORES=1 FM_MERGED_STORES=0; do
  LORA_AMD_AB=$ab timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$ab', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
