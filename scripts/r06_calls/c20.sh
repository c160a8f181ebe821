#!/bin/bash
# round 6, call 20: table order — longest row blocks first — for the two factor-pass launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 900 python scripts/kbench.py --what fm > $O/c20_kbench_fm.log 2> $O/c20_kbench_fm.err; echo "kbench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/c20_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("part_") or k.startswith("mfma_"): print(k, v)
PY
