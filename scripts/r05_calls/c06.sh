#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm_xs.py -q -x > $O/c06_xs.log 2>&1; echo "xs tests rc=$?"; tail -3 $O/c06_xs.log | cut -c1-300
timeout 600 python scripts/kbench.py --what xs --iters 10 > $O/c06_kbench_xs.log 2>&1; python - <<'P'
import json
for ln in open("gpurun_out/c06_kbench_xs.log"):
    if not ln.startswith("{"): continue
    d=json.loads(ln); sw=d.pop("xs_plain_sweep")
    print({k:d[k] for k in ("M","K","N","floor_us_8TBs","xs_us","xs_drop_us","xs_plain_us","xs_plain_best","xs_plain_rowmajor_us","ws_us","ws_drop_us","lib_gemm_us","lib_gemm_plus_lora_us") if k in d}, {k:d[k] for k in d if k.endswith("dx_us")})
    print("   ", sw)
P
