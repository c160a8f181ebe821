#!/bin/bash
# Profiles behind the numbers bench.py prints.  Run on the GPU box from the repo root:
#   bash scripts/profile_bench.sh r01
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out
export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/lora_amd_tune_${TAG}.json   # the warm run times the candidates, the traced run re-uses them
mkdir -p $OUT
ARGS="--steps 5 --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-}"
ONLY=${2:-all}
if [ "$ONLY" != "pmc" ]; then
# 1. un-profiled run first: fills MIOpen's find cache (a cold MIOpen under the profiler falls back to naive convs)
python bench.py $ARGS > $OUT/${TAG}_bench_warm.json 2> $OUT/${TAG}_bench_warm.err
# 2. kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o bench -- python bench.py $ARGS > $OUT/${TAG}_bench_traced.json 2> $OUT/${TAG}_bench_traced.err
python scripts/prof_summary.py $(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1) 60 > $OUT/${TAG}_bench_kernel_summary.txt
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_trace
fi
if [ "$ONLY" == "trace" ]; then head -45 $OUT/${TAG}_bench_kernel_summary.txt | cut -c1-180; exit 0; fi
# 3. PMC passes (each counter alone; no trace domains besides kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o p -- python scripts/pmc_probe.py > $OUT/${TAG}_pmc_probe.txt 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o p -- python scripts/pmc_probe.py > /dev/null 2> $OUT/${TAG}_pmc_write.err
COPYB=$(grep copy_bytes_each_way $OUT/${TAG}_pmc_probe.txt | awk '{print $2}')
ALGB=$(grep copy_bytes_each_way $OUT/${TAG}_pmc_probe.txt | awk '{print $4}')
python scripts/pmc_reduce.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $COPYB $ALGB > $OUT/${TAG}_merge_pmc.json 2> $OUT/${TAG}_pmc_reduce.err
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
head -c 1500 $OUT/${TAG}_merge_pmc.json
head -30 $OUT/${TAG}_bench_kernel_summary.txt | cut -c1-180
