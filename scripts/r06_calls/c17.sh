#!/bin/bash
# round 6, call 17: per-stage wall-clock stamps of the factor pass (scripts/fm_trace: the same source under -DFM_TRACE)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
LORA_AMD_LIB=scripts/fm_trace/liblora_amd_trace.so timeout 600 python scripts/kbench.py --what fmtrace > $O/c17_fmtrace.log 2> $O/c17_fmtrace.err; echo "rc=$?"
tail -3 $O/c17_fmtrace.err; cat $O/c17_fmtrace.log
