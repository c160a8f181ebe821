#!/bin/bash
# Round-6 evidence batch (run on the GPU box from the repo root).  Outputs under gpurun_out/r06_*; the summaries that are
# cited get copied into profiles/.  Stages can be selected: bash scripts/r06_profiles.sh "bench trace mergepmc steppmc kbench"
set -u
export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/lora_amd_tune_r06.json
OUT=gpurun_out
mkdir -p $OUT
STAGES=${1:-"bench trace mergepmc steppmc"}
EAGER="--mode eager --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary"
for S in $STAGES; do case $S in
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_bench_line.json 2> $OUT/r06_bench_line.err
  tail -c 600 $OUT/r06_bench_line.err ;;
benchshort)
  timeout 400 python bench.py --no-secondary --no-cpu-baseline > $OUT/r06_bench_short.json 2> $OUT/r06_bench_short.err ;;
trace)
  # MIOpen's find cache and the tune cache do not survive between boxes: fill them first, or the trace holds the find pass
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > /dev/null 2> $OUT/r06_trace_warm.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_trace -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/r06_bench_traced.json 2> $OUT/r06_bench_traced.err
  python scripts/prof_summary.py $(find $OUT/r06_trace -name "*kernel_trace.csv" | head -1) 70 > $OUT/r06_bench_kernel_trace_summary.txt
  # the same trace per launch geometry, own kernels only: the ONE in-step merge / factor-pass launch per step is not averaged
  # with the per-site launches of the warm-up
  python scripts/prof_summary.py $(find $OUT/r06_trace -name "*kernel_trace.csv" | head -1) 400 by-grid | grep -E "^#|calls|lora_amd::" | head -60 > $OUT/r06_bench_kernel_trace_by_grid.txt
  rm -rf $OUT/r06_trace ;;
mergepmc)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r06_pmc_fetch -o p -- python scripts/pmc_probe.py > $OUT/r06_pmc_probe.txt 2> $OUT/r06_pmc_fetch.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/r06_pmc_write -o p -- python scripts/pmc_probe.py > /dev/null 2> $OUT/r06_pmc_write.err
  COPYB=$(grep copy_bytes_each_way $OUT/r06_pmc_probe.txt | awk '{print $2}')
  ALGB=$(grep copy_bytes_each_way $OUT/r06_pmc_probe.txt | awk '{print $4}')
  ALGS=$(grep copy_bytes_each_way $OUT/r06_pmc_probe.txt | awk '{print $6}')
  python scripts/pmc_reduce.py $OUT/r06_pmc_fetch $OUT/r06_pmc_write $COPYB $ALGB $ALGS > $OUT/r06_merge_pmc.json 2> $OUT/r06_pmc_reduce.err
  rm -rf $OUT/r06_pmc_fetch $OUT/r06_pmc_write
  head -c 1200 $OUT/r06_merge_pmc.json ;;
steppmc)
  python bench.py $EAGER > /dev/null 2> $OUT/r06_step_warm.err   # fills the tune cache and MIOpen's find cache
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/r06_step_f -o p -- python bench.py $EAGER > /dev/null 2> $OUT/r06_step_f.err
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/r06_step_w -o p -- python bench.py $EAGER > /dev/null 2> $OUT/r06_step_w.err
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/r06_step_m -o p -- python bench.py $EAGER > /dev/null 2> $OUT/r06_step_m.err
  if [ -z "$(find $OUT/r06_step_m -name '*counter_collection.csv' 2>/dev/null | head -1)" ]; then
    rm -rf $OUT/r06_step_m
    timeout 400 rocprofv3 --pmc SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/r06_step_m -o p -- python bench.py $EAGER > /dev/null 2> $OUT/r06_step_m2.err
  fi
  FB=$(python -c "import json;print(json.load(open('$OUT/r06_merge_pmc.json'))['calibration']['fetch_bytes_per_count'])" 2>/dev/null || echo 2048)
  WB=$(python -c "import json;print(json.load(open('$OUT/r06_merge_pmc.json'))['calibration']['write_bytes_per_count'])" 2>/dev/null || echo 1024)
  MD=$OUT/r06_step_m; [ -d $MD ] || MD=-
  python scripts/pmc_step.py reduce $OUT/r06_step_f $OUT/r06_step_w $MD $FB $WB > $OUT/r06_step_pmc.json 2> $OUT/r06_step_reduce.err
  rm -rf $OUT/r06_step_f $OUT/r06_step_w $OUT/r06_step_m
  head -c 1500 $OUT/r06_step_pmc.json; tail -3 $OUT/r06_step_reduce.err ;;
cfg3trace)
  ARGS="--extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
  python bench.py $ARGS --steps 3 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_cfg3_trace -o cfg3 -- python bench.py $ARGS > $OUT/r06_cfg3_traced.json 2> $OUT/r06_cfg3_traced.err
  python scripts/prof_summary.py $(find $OUT/r06_cfg3_trace -name "*kernel_trace.csv" | head -1) 70 > $OUT/r06_cfg3_kernel_trace_summary.txt
  rm -rf $OUT/r06_cfg3_trace
  head -12 $OUT/r06_cfg3_kernel_trace_summary.txt | cut -c1-150 ;;
cfg3twin)
  # the frozen twin of configs[3] (bench.py --adapters none: the merged path's GEMM launches on frozen weights, no LoRA launch):
  # the kernel-time difference to cfg3trace is what the adapters cost, kernel by kernel
  ARGS="--adapters none --extended 1 --rank 16 --res 768 --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline"
  python bench.py $ARGS --steps 3 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_cfg3twin_trace -o cfg3 -- python bench.py $ARGS > $OUT/r06_cfg3twin_traced.json 2> $OUT/r06_cfg3twin_traced.err
  python scripts/prof_summary.py $(find $OUT/r06_cfg3twin_trace -name "*kernel_trace.csv" | head -1) 70 > $OUT/r06_cfg3twin_kernel_trace_summary.txt
  rm -rf $OUT/r06_cfg3twin_trace
  head -12 $OUT/r06_cfg3twin_kernel_trace_summary.txt | cut -c1-150 ;;
svdtrace)
  python bench.py --svd --warmup 1 --steps 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_svd_trace -o svd -- python bench.py --svd --warmup 1 --steps 3 --no-cpu-baseline > $OUT/r06_bench_svd.json 2> $OUT/r06_svd_traced.err
  python scripts/prof_summary.py $(find $OUT/r06_svd_trace -name "*kernel_trace.csv" | head -1) 30 > $OUT/r06_svd_kernel_trace_summary.txt
  rm -rf $OUT/r06_svd_trace
  LORA_AMD_SVD_ITERS=4 python bench.py --svd --warmup 2 --steps 5 --no-cpu-baseline > $OUT/r06_bench_svd_iters4.json 2> /dev/null
  head -8 $OUT/r06_svd_kernel_trace_summary.txt | cut -c1-150; cut -c1-250 $OUT/r06_bench_svd_iters4.json ;;
kbench)
  timeout 200 python scripts/kbench.py --what ${KBENCH_WHAT:-ws} > $OUT/r06_kbench_${KBENCH_WHAT:-ws}.log 2>&1
  tail -40 $OUT/r06_kbench_${KBENCH_WHAT:-ws}.log ;;
esac; done
