#!/bin/bash
# round 5, call 9: configs[3] same-box A/B of the dropout-site routing (the alternative to the weight-stationary kernel is now
# library GEMM + the rank-16 matrix-core primitives of round 4), then the whole GPU suite on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
: > $O/c09_cfg3_ab.txt
for AB in "" "WS_DROPOUT_WIDE=0" "WS_DROPOUT_WIDE_BWD=0" "WS_DROPOUT_WIDE=0,WS_DROPOUT_WIDE_BWD=0" ""; do
  LORA_AMD_AB="$AB" timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LORA_AMD_AB=%r' % '$AB', d['value'], 'steps/s', d['ms_per_step'], 'ms')" >> $O/c09_cfg3_ab.txt
done
cat $O/c09_cfg3_ab.txt
timeout 1200 python -m pytest tests -m gpu -q --durations=8 > $O/c09_pytest_full.log 2>&1; echo "full rc=$?"; tail -16 $O/c09_pytest_full.log | cut -c1-200
