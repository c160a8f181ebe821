#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gemm_xs.py -q -x > $O/c07_xs.log 2>&1; echo "xs tests rc=$?"; tail -3 $O/c07_xs.log | cut -c1-300
timeout 600 python scripts/kbench.py --what xs --iters 10 > $O/c07_kbench_xs.log 2>&1; python - <<'P'
import json
for ln in open("gpurun_out/c07_kbench_xs.log"):
    if not ln.startswith("{"): continue
    d=json.loads(ln); sw=d.pop("xs_plain_sweep")
    print({k:d[k] for k in ("M","K","N","floor_us_8TBs","xs_us","xs_drop_us","xs_plain_us","xs_plain_best","xs_plain_rowmajor_us","ws_us","ws_drop_us","lib_gemm_us","lib_gemm_plus_lora_us") if k in d}, {k:d[k] for k in d if k.endswith("dx_us")})
    print("   ", sw)
P
timeout 300 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_parity_r3.py tests/test_cli_svd.py -q -x -k "svd or thin or packed or select or residual or distill or quantile or adaptive or choleskyqr or fused" > $O/c07_svdtests.log 2>&1; echo "svd tests rc=$?"; tail -3 $O/c07_svdtests.log
timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $O/c07_svd.json 2> $O/c07_svd.err; echo "svd rc=$?"; cat $O/c07_svd.json | cut -c1-700
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/svdtrace -o svd -- python $GRAFT_REPO_ROOT/bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/svdtrace -name "*kernel_stats.csv" | head -1); python scripts/stats_top.py "$f" 25 > $O/c07_svd_kernel_stats.txt; head -12 $O/c07_svd_kernel_stats.txt | cut -c1-150
