#!/bin/bash
# rocprofv3 kernel trace of the kernel microbenchmarks: per-kernel average durations without host effects.
#   bash scripts/prof_kbench.sh <tag> <what>
set -u
TAG=${1:-r01}; WHAT=${2:-conv}
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_kb_trace -o kb -- python scripts/kbench.py --what $WHAT --iters 3 > gpurun_out/${TAG}_kbench_${WHAT}.log 2> gpurun_out/${TAG}_kbench_${WHAT}.err
python scripts/prof_summary.py $(find gpurun_out/${TAG}_kb_trace -name "*kernel_trace.csv" | head -1) 80 by-grid > gpurun_out/${TAG}_kbench_${WHAT}_kernels.txt
rm -rf gpurun_out/${TAG}_kb_trace
cut -c1-200 gpurun_out/${TAG}_kbench_${WHAT}_kernels.txt | head -70
