#!/bin/bash
# round 6, call 25: what a CU's load path sustains per load SHAPE when the data sit in the L2 / the MALL (ld_shape_probe, wrap)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w scripts/ld_shape_probe.hip -o /tmp/ldp 2> $O/c25_build.err || { tail -5 $O/c25_build.err; exit 1; }
for wrap in 0 512 64; do for C in 320 640; do timeout 120 /tmp/ldp 1572864 $C $wrap; done; done > $O/c25_ld_shape_probe.log 2>&1
cat $O/c25_ld_shape_probe.log
