#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for NT in 0 1 0 1; do
  LORA_AMD_PLANES_NT=$NT LORA_AMD_SVD_ITERS=4 timeout 200 python bench.py --svd --warmup 2 --steps 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NT=$NT', d['ms_per_step'], 'ms', d['value'], 'sites/s')"
done | tee $O/c11_planes_nt_ab.txt
