#!/bin/bash
# round 6, call 8: noisy latents formed in f32 -> the batch-4 bracket tests (r3 fixture x 3 configurations, r5, the r6 bisect);
# the svd tests on the rebuilt library; SQ counters of the factor pass kernels (stall breakdown)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_parity_r5.py tests/test_gpu_parity_r6.py -q -s -k "sd15_size_step or bracket" > $O/c08_bracket.log 2>&1; echo "bracket tests rc=$?"
grep -E "^\[bracket|passed|failed|^FAILED|^E   " $O/c08_bracket.log | cut -c1-230 | tail -30
timeout 600 python -m pytest tests/test_gpu_svd_small.py -q > $O/c08_svd_tests.log 2>&1; echo "svd tests rc=$?"; grep -E "passed|failed|^FAILED|^E   " $O/c08_svd_tests.log | cut -c1-200 | head
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -c4-12 | tr -d ' ')
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/c08_fm_pmc_$tag -o p -- python scripts/kbench.py --what fm --iters 2 > /dev/null 2> $O/c08_fm_pmc_$tag.err
done
python scripts/fm_pmc.py $O/c08_fm_pmc_* > $O/c08_fm_pmc.jsonl 2> $O/c08_fm_pmc_reduce.err; rm -rf $O/c08_fm_pmc_*/
grep factors_reg $O/c08_fm_pmc.jsonl | cut -c1-1200
