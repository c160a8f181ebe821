#!/bin/bash
# round 6, call 36: the factor pass with the block -> site map (one scalar load instead of the LDS prefix search): parity, kbench
# both ways, then the step under LORA_AMD_AB=FM_BLOCK_MAP=1 / 0
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r6.py tests/test_gpu_parity_r4.py tests/test_gpu_ws_heads.py -q -x -k "factor_pass or factors_mfma or block_map" > $O/c36_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c36_tests.log | head -8
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c36_kbench_fm.log 2> $O/c36_kbench_fm.err; echo "kbench rc=$?"; tail -2 $O/c36_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c36_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("mapped") or k.startswith("search") or k.startswith("mfma_"): print(k, v)
PY
for ab in FM_BLOCK_MAP=1 FM_BLOCK_MAP=0 FM_BLOCK_MAP=1 FM_BLOCK_MAP=0; do
  LORA_AMD_AB=$ab timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$ab', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
