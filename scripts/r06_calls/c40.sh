#!/bin/bash
# round 6, call 40: what would full-line loads be worth in the factor pass?  A timing build (-DFM_LINE_PROBE: 8 rows x 128 bytes per
# load instruction instead of 16 rows x 64 bytes; garbage results) against the product library, kbench.py --what fm, same call
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for lib in "" scripts/ab/liblora_amd_lineprobe.so "" scripts/ab/liblora_amd_lineprobe.so; do
  LORA_AMD_FM_RINGS=0 LORA_AMD_LIB=$lib timeout 900 python scripts/kbench.py --what fm 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-product}', {k: (v['us'] if isinstance(v, dict) else v) for k, v in d.items() if k.startswith('mfma_class') or k.startswith('part_c') or k in ('mfma_pass_us',)})"
done
