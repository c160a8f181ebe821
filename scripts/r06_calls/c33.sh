#!/bin/bash
# round 6, call 33: the dropout sites around the attention core in head-padded rows on the weight-stationary kernel (no pack /
# unpack copies): the new parity tests, the neighbouring suites, then configs[3] with ops.WS_HEADS on / off, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ws_heads.py -q -x > $O/c33_tests_new.log 2>&1; echo "new tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c33_tests_new.log | head -12
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r2.py tests/test_standin.py -q -x -k "dropout or ws or extended or heads or padded" > $O/c33_tests_near.log 2>&1; echo "near tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c33_tests_near.log | head -12
for ab in WS_HEADS=1 WS_HEADS=0 WS_HEADS=1 WS_HEADS=0; do
  LORA_AMD_AB=$ab timeout 500 python bench.py --extended 1 --rank 16 --res 768 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ab', d['value'], d['ms_per_step'])"
done
