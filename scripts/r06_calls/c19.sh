#!/bin/bash
# round 6, call 19: finer stage stamps of the factor pass (11 per wave)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
LORA_AMD_LIB=scripts/fm_trace/liblora_amd_trace.so timeout 600 python scripts/kbench.py --what fmtrace > $O/c19_fmtrace.log 2> $O/c19_fmtrace.err; echo "trace rc=$?"; tail -2 $O/c19_fmtrace.err
python - <<PY
import json
for ln in open("$O/c19_fmtrace.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["group"], d["launch_us_events"], "life", d["wave_life_us_mean"], d["stage_us_mean"])
PY
