#!/bin/bash
# round 5, call 2: the fused small steps of K5 (tests, bench --svd, kernel trace), r5 parity test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_svd_small.py tests/test_gpu_parity_r5.py -q -x --durations=10 > $O/c02_tests.log 2>&1; echo "tests rc=$?"
tail -25 $O/c02_tests.log
timeout 300 python -m pytest tests/test_cli_svd.py tests/test_gpu_parity_r3.py -q -k "svd or quantile" > $O/c02_oldsvd.log 2>&1; echo "old svd tests rc=$?"; tail -5 $O/c02_oldsvd.log
timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $O/c02_svd.json 2> $O/c02_svd.err; echo "svd rc=$?"; cat $O/c02_svd.json; tail -3 $O/c02_svd.err | cut -c1-400
LORA_AMD_SVD_ITERS=4 timeout 300 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $O/c02_svd_fixed4.json 2>/dev/null; cat $O/c02_svd_fixed4.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/svdtrace -o svd -- python $GRAFT_REPO_ROOT/bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/svdtrace -name "*kernel_stats.csv" | head -1); echo $f; python - "$f" <<'P' > $O/c02_svd_kernel_stats.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("# kernel stats, 1 warm-up + 3 steps of bench.py --svd (plus its setup); total %.3f ms, %d calls"%(tot/1e6,sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:40]:
    print("%6d %10.3f ms %9.2f us %5.1f%%  %s"%(int(r["Calls"]),float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3,100*float(r["TotalDurationNs"])/tot,r["Name"][:110]))
P
head -30 $O/c02_svd_kernel_stats.txt
