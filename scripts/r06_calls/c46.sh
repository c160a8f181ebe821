#!/bin/bash
# round 6, call 46: stage stamps of the factor pass on the final kernel (block map is not used by the trace leg; compact fragments,
# non-temporal slabs)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
LORA_AMD_LIB=scripts/fm_trace/liblora_amd_trace.so timeout 600 python scripts/kbench.py --what fmtrace > $O/c46_fm_stage_stamps.log 2> $O/c46.err; echo rc=$?; tail -2 $O/c46.err
python - <<PY
import json
for l in open("$O/c46_fm_stage_stamps.log"):
    l=l.strip()
    if not l.startswith("{"): continue
    d=json.loads(l); print(d["group"], d["blocks"], "launch", d["launch_us_events"], "life", d["workgroup_life_us_mean"], "conc", d["concurrent_workgroups_mean"], d["stage_us_mean"])
PY
