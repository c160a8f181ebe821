#!/bin/bash
# round 6, call 26: the factor pass with pieces fetched four lanes per row (load layout) and the MFMA operands taken from the LDS
# tile: parity, stage stamps, kbench, the step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r4.py tests/test_gpu_parity_r6.py -q -x -k "factors_mfma or factor_pass or consecutive_optimizer_steps or extended_rank16" > $O/c26_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c26_tests.log | head -12
LORA_AMD_LIB=scripts/fm_trace/liblora_amd_trace.so timeout 600 python scripts/kbench.py --what fmtrace > $O/c26_fmtrace.log 2> $O/c26_fmtrace.err; echo "trace rc=$?"
python - <<PY
import json
for ln in open("$O/c26_fmtrace.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["group"], d["launch_us_events"], "life", d["wave_life_us_mean"], d["stage_us_mean"])
PY
LORA_AMD_FM_RINGS=0 timeout 900 python scripts/kbench.py --what fm > $O/c26_kbench_fm.log 2> $O/c26_kbench_fm.err; echo "kbench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/c26_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("part_") or k.startswith("mfma_") or k.startswith("class1_") or k=="max_rel_diff_valu_vs_matrix_core_last_run": print(k, v)
PY
timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $O/c26_bench.json 2> $O/c26_bench.err
python - <<PY
import json
d=json.loads(open("$O/c26_bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["roofline_in_step"]["factor_pass"])
PY
