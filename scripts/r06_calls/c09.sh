#!/bin/bash
# round 6, call 9: where the factor pass's time goes (parts of the kernel switched off), svd adaptive test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
LORA_AMD_FM_ATTRIB=1 timeout 600 python scripts/kbench.py --what fm > $O/c09_kbench_fm_attrib.log 2>&1; tail -1 $O/c09_kbench_fm_attrib.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'attrib' in k or k in ('reg_class1_us','reg_class2_us','class1_kernel0_us','class1_kernel1_us')})"
timeout 300 python -m pytest tests/test_gpu_svd_small.py -q -k "adaptive or residual_planes or hi_plane" > $O/c09_svd.log 2>&1; echo "svd rc=$?"; grep -E "passed|failed|^E   " $O/c09_svd.log | head -5
timeout 300 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -k "factors_mfma or factor_pass" > $O/c09_fm.log 2>&1; echo "fm rc=$?"; tail -1 $O/c09_fm.log
