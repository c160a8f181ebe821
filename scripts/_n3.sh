export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/tune_cfg3.json
OUT=gpurun_out
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline"
timeout 200 python bench.py $CFG3 --channels-last 1 > $OUT/n3_cfg3_cl.json 2> $OUT/n3_cfg3_cl.err
timeout 200 python bench.py $CFG3 --channels-last 0 > $OUT/n3_cfg3_nchw.json 2> $OUT/n3_cfg3_nchw.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n3_trace -o bench -- python bench.py $CFG3 --channels-last 1 --steps 5 --warmup 3 > $OUT/n3_traced.json 2> $OUT/n3_traced.err
python scripts/prof_summary.py $(find $OUT/n3_trace -name "*kernel_trace.csv" | head -1) 70 > $OUT/n3_trace_summary.txt
rm -rf $OUT/n3_trace
for f in n3_cfg3_cl n3_cfg3_nchw; do tail -1 $OUT/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])" || tail -5 $OUT/$f.err; done
