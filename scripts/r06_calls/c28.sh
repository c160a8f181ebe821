#!/bin/bash
# round 6, call 28: K5's passes — static slots with operand-order loads vs row-contiguous loads + ds_bpermute (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in 0 1 0 1; do
  echo "== LORA_AMD_PLANES_ROWLD=$v"
  LORA_AMD_PLANES_ROWLD=$v LORA_AMD_SVD_ITERS=4 timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline 2> /dev/null | cut -c100-260
  LORA_AMD_PLANES_ROWLD=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c28_trace -o svd -- python bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2> $O/c28_traced.err
  python scripts/prof_summary.py $(find $O/c28_trace -name "*kernel_trace.csv" | head -1) 12 | grep planes | cut -c1-130
  rm -rf $O/c28_trace
done
