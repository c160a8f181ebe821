#!/bin/bash
# round 6, call 38: compact factor fragments (pk[split][c/8][RT][8]: a rank-4 fragment is 256 B, its rows repeated to the 16 of the
# tile by the lanes' addresses): every factor-pass parity test, kbench, the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py tests/test_gpu_ws_heads.py tests/test_gpu_parity_r3.py -q -x -k "factor_pass or factors_mfma or block_map or merged_weight or consecutive or fp16" > $O/c38_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c38_tests.log | head -8
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c38_kbench_fm.log 2> $O/c38_kbench_fm.err; echo "kbench rc=$?"; tail -2 $O/c38_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c38_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("mapped") or k.startswith("search") or k.startswith("mfma_") or k=="pack_us" or k.startswith("max_rel"): print(k, v)
PY
LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=scripts/ab/liblora_amd_old.so timeout 900 python scripts/kbench.py --what fm > $O/c38_kbench_fm_old.log 2> /dev/null
python - <<PY
import json
d=json.loads(open("$O/c38_kbench_fm_old.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("mapped") or k.startswith("mfma_") or k=="pack_us": print("old", k, v)
PY
# scripts/ab/liblora_amd_old.so = the same tree with factor_mfma.hip of the commit before (16-row fragments)
for lib in lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so lora_amd/csrc/liblora_amd.so scripts/ab/liblora_amd_old.so; do
  timeout 400 python scripts/ab/run_with_lib.py $lib bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{}).get('factor_pass',{})
print('$lib', d['value'], d['ms_per_step'], 'factor pass', r.get('avg_launch_us'), r.get('frac'))"
done
