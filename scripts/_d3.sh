export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/tune_cfg3.json
OUT=gpurun_out
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline --channels-last 1"
LORA_AMD_WS_DROPOUT=3 timeout 300 python bench.py $CFG3 > $OUT/d3_a.json 2> $OUT/d3_a.err
LORA_AMD_WS_DROPOUT=2 timeout 300 python bench.py $CFG3 > $OUT/d3_wide.json 2> $OUT/d3_wide.err
for f in d3_a d3_wide; do tail -1 $OUT/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])" || tail -3 $OUT/$f.err; done
