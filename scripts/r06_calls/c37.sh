#!/bin/bash
# round 6, call 37: is staying resident worth anything WITHOUT the counter's atomic in front of a block's loads?  A strided
# persistent launch (k workgroups per slot, blocks b, b + grid, ...) on tables of equal blocks, mapped lookups
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -x -k "factors_mfma or factor_pass or block_map" > $O/c37_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c37_tests.log | head -8
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c37_kbench_fm.log 2> $O/c37_kbench_fm.err; echo "kbench rc=$?"; tail -2 $O/c37_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c37_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("strided") or k.startswith("mapped") or k.startswith("mfma_"): print(k, v)
PY
