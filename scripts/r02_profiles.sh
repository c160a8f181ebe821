#!/bin/bash
# Round-2 evidence batch (run on the GPU box from the repo root): bench lines for the headline and the other BASELINE
# geometries, kernel trace of the headline step, PMC traffic of the adapter kernels.  Outputs under gpurun_out/.
set -u
export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/lora_amd_tune_r02.json   # the first run times the attention candidates, every later run (incl. the traced one) re-uses its choices
OUT=gpurun_out
mkdir -p $OUT
timeout 400 python bench.py > $OUT/r02_bench_line.json 2> $OUT/r02_bench_line.err
timeout 300 python bench.py --text-encoder 1 --rank 8 --no-cpu-baseline --no-roofline > $OUT/r02_bench_cfg2.json 2> /dev/null
timeout 400 python bench.py --extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline > $OUT/r02_bench_cfg3.json 2> $OUT/r02_bench_cfg3.err
timeout 300 python bench.py --extended 1 --rank 16 --res 768 --batch 1 --channels-last 0 --no-cpu-baseline --no-roofline > $OUT/r02_bench_cfg3_nchw.json 2> /dev/null
LORA_AMD_WS_DROPOUT=0 timeout 300 python bench.py --extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline > $OUT/r02_bench_cfg3_unfused_dropout.json 2> /dev/null
timeout 300 python bench.py --with-prior-preservation 1 --no-cpu-baseline --no-roofline > $OUT/r02_bench_prior.json 2> /dev/null
timeout 200 python bench.py --svd --warmup 1 > $OUT/r02_bench_svd.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_trace -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/r02_bench_traced.json 2> $OUT/r02_bench_traced.err
python scripts/prof_summary.py $(find $OUT/r02_trace -name "*kernel_trace.csv" | head -1) 60 > $OUT/r02_bench_kernel_trace_summary.txt
rm -rf $OUT/r02_trace
if [ "${SKIP_PMC:-0}" != "1" ]; then
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r02_pmc_f -o p -- python scripts/pmc_kernels.py run > /dev/null 2> $OUT/r02_pmc_f.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/r02_pmc_w -o p -- python scripts/pmc_kernels.py run > /dev/null 2> $OUT/r02_pmc_w.err
python scripts/pmc_kernels.py reduce $OUT/r02_pmc_f $OUT/r02_pmc_w > $OUT/r02_adapter_pmc.json 2> $OUT/r02_pmc_reduce.err
rm -rf $OUT/r02_pmc_f $OUT/r02_pmc_w
timeout 120 python scripts/kbench.py --what nhwc > $OUT/r02_kbench_nhwc.log 2>&1
fi
for f in r02_bench_line r02_bench_cfg2 r02_bench_cfg3 r02_bench_cfg3_nchw r02_bench_cfg3_unfused_dropout r02_bench_prior r02_bench_svd; do python - <<PY
import json
try:
    d = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["unit"], d.get("ms_per_step"))
except Exception as e: print("$f FAILED", e)
PY
done

