export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/tune_cfg3.json
OUT=gpurun_out
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline --conv-find 0"
timeout 150 python bench.py $CFG3 --channels-last 1 --steps 3 --warmup 2 > /dev/null 2>&1   # fills the attention tune cache
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n5_trace -o bench -- python bench.py $CFG3 --channels-last 1 --steps 5 --warmup 3 > $OUT/n5_traced.json 2> $OUT/n5_traced.err
python scripts/prof_summary.py $(find $OUT/n5_trace -name "*kernel_trace.csv" | head -1) 90 > $OUT/n5_trace_summary.txt
rm -rf $OUT/n5_trace
head -3 $OUT/n5_trace_summary.txt
