export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/tune_cfg3.json
OUT=gpurun_out
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline"
timeout 300 python bench.py $CFG3 > $OUT/d2_cfg3_fused.json 2> $OUT/d2_cfg3_fused.err
LORA_AMD_WS_DROPOUT=0 timeout 300 python bench.py $CFG3 > $OUT/d2_cfg3_unfused.json 2> /dev/null
timeout 300 python bench.py $CFG3 --channels-last 1 > $OUT/d2_cfg3_fused_nhwc.json 2> /dev/null
for f in d2_cfg3_fused d2_cfg3_unfused d2_cfg3_fused_nhwc; do tail -1 $OUT/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])" || tail -3 $OUT/$f.err; done
