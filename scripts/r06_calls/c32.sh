#!/bin/bash
# round 6, call 32: class-1 table longest blocks first, in the step, with the row-load kernel (call c23 measured it with the
# operand-order loads: no difference; kbench since: 263 against 276 us)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for ab in FM_LONGEST_FIRST_CLASS1=1 FM_LONGEST_FIRST_CLASS1=0 FM_LONGEST_FIRST_CLASS1=1 FM_LONGEST_FIRST_CLASS1=0; do
  LORA_AMD_AB=$ab timeout 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline_in_step',{})
print('$ab', d['value'], d['ms_per_step'], r.get('factor_pass'), r.get('merge'))"
done
