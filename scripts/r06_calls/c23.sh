#!/bin/bash
# round 6, call 23: same-box A/B of the table order (kernel traces: per-launch times of the two factor-pass launches)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2> $O/c23_warm.err
for ab in "FM_LONGEST_FIRST=1" "FM_LONGEST_FIRST=0" "FM_LONGEST_FIRST_CLASS1=0" "FM_LONGEST_FIRST=1" "FM_LONGEST_FIRST=0" "FM_LONGEST_FIRST_CLASS1=0"; do
  rm -rf $O/c23_trace
  LORA_AMD_AB=$ab timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c23_trace -o bench -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline > $O/c23_traced.json 2> $O/c23_traced.err
  echo "== $ab"
  python scripts/prof_summary.py $(find $O/c23_trace -name "*kernel_trace.csv" | head -1) 400 by-grid | grep -E "factors_reg" | cut -c1-120
done
rm -rf $O/c23_trace
