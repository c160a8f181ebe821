export TMPDIR=/tmp
export LORA_AMD_TUNE_CACHE=/tmp/tune_cfg3.json
OUT=gpurun_out
CFG3="--extended 1 --rank 16 --res 768 --batch 1 --no-cpu-baseline --no-roofline --conv-find 0"
S=$(date +%s); timeout 150 python bench.py $CFG3 --channels-last 1 > $OUT/n4_cfg3_cl.json 2> $OUT/n4_cfg3_cl.err; echo "cfg3 cl wall $(( $(date +%s) - S )) s"
S=$(date +%s); timeout 150 python bench.py $CFG3 --channels-last 0 > $OUT/n4_cfg3_nchw.json 2> $OUT/n4_cfg3_nchw.err; echo "cfg3 nchw wall $(( $(date +%s) - S )) s"
S=$(date +%s); timeout 200 python bench.py --no-cpu-baseline > $OUT/n4_default_find.json 2> $OUT/n4_default_find.err; echo "default (find) wall $(( $(date +%s) - S )) s"
for f in n4_cfg3_cl n4_cfg3_nchw n4_default_find; do tail -1 $OUT/$f.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])" || tail -3 $OUT/$f.err; done
