#!/bin/bash
# round 6, call 35: does the 3-3-2-2 split of a 320-wide operand over four waves cost the factor pass? widths 256 / 320 / 384
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python scripts/fm_width_probe.py > $O/c35_fm_width_probe.log 2> $O/c35_fm_width_probe.err; echo rc=$?; cat $O/c35_fm_width_probe.log; tail -3 $O/c35_fm_width_probe.err
