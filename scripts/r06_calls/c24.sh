#!/bin/bash
# round 6, call 24: ring depth per SITE TYPE (class 1 on its kernels, class 2 on ring variants of the 10-pair kernel)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1200 python scripts/kbench.py --what fm > $O/c24_kbench_fm.log 2> $O/c24_kbench_fm.err; echo "kbench rc=$?"; tail -3 $O/c24_kbench_fm.err
python - <<PY
import json
d=json.loads(open("$O/c24_kbench_fm.log").read().strip().splitlines()[-1])
for k,v in d.items():
    if k.startswith("part_") or k.startswith("mfma_") or k=="max_rel_diff_valu_vs_matrix_core_last_run": print(k, v)
PY
