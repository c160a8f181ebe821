#!/bin/bash
# round 6, call 4: the fused 3x3 site (one forward launch, Gt fold inside the G pass, per-step batched pack): parity, then
# configs[3] A/B on the same box (LORA_AMD_AB=CONV3_FUSED=0 = rounds 3-5's launch sequence), its frozen twin
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_nhwc.py tests/test_gpu_rank16.py -x -q > $O/c04_conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -4 $O/c04_conv_tests.log
timeout 600 python -m pytest tests/test_gpu_parity_r4.py -x -q -k "extended_rank16" > $O/c04_cfg3_parity.log 2>&1; echo "cfg3 parity rc=$?"; tail -2 $O/c04_cfg3_parity.log
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary"
for ab in "CONV3_FUSED=1" "CONV3_FUSED=0" "CONV3_FUSED=1"; do
  LORA_AMD_AB=$ab timeout 600 python bench.py $CFG3 > $O/c04_cfg3_$ab.json 2> $O/c04_cfg3_$ab.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/c04_cfg3_$ab.json").read().strip().splitlines()[-1]); print("$ab", d["value"], d["ms_per_step"], d["config"].get("execution"))
except Exception as e: print("$ab failed", e)
PY
done
tail -3 "$O/c04_cfg3_CONV3_FUSED=1.err" | cut -c1-600
