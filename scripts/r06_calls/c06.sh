#!/bin/bash
# round 6, call 6: is the 6.4 x bracket ratio of c05 (r3 fixture, batch 4, seed 123) the data or this round's factor-pass split?
# the r5 bracket test (seed 77) and the r3 one under FM_TWO_CLASSES=1 / 0; the load-shape probe; the svd tests after the edits
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for ab in "FM_TWO_CLASSES=1" "FM_TWO_CLASSES=0"; do
  LORA_AMD_AB=$ab timeout 900 python -m pytest tests/test_gpu_parity_r5.py tests/test_gpu_parity_r3.py -q -s -k "bracketed or (sd15_size_step and bench-)" > $O/c06_bracket_$ab.log 2>&1; echo "$ab rc=$?"
  grep -E "^\[bracket|passed|failed|ratio" $O/c06_bracket_$ab.log | cut -c1-220 | head -12
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ld_shape_probe.hip -o /tmp/ld_shape_probe && (timeout 120 /tmp/ld_shape_probe 1572864 320; timeout 120 /tmp/ld_shape_probe 196608 2560; timeout 120 /tmp/ld_shape_probe 786432 640) > $O/c06_ld_shape_probe.log 2>&1; cat $O/c06_ld_shape_probe.log
timeout 600 python -m pytest tests/test_gpu_svd_small.py -q > $O/c06_svd_tests.log 2>&1; echo "svd tests rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/c06_svd_tests.log | head -12
