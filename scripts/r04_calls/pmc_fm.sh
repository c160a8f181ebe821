#!/bin/bash
# Counter passes over the factor-gradient kernels (kbench --what fm): where do the waves' cycles go?
set -u
OUT=gpurun_out
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
RUN="python scripts/kbench.py --what fm --iters 2"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/fm_pmc_a -o p -- $RUN > /dev/null 2> $OUT/r04e_fm_pmc_a.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/fm_pmc_b -o p -- $RUN > /dev/null 2> $OUT/r04e_fm_pmc_b.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fm_pmc_f -o p -- $RUN > /dev/null 2> $OUT/r04e_fm_pmc_f.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/fm_pmc_c -o p -- $RUN > /dev/null 2> $OUT/r04e_fm_pmc_c.err
python scripts/fm_pmc.py $OUT/fm_pmc_a $OUT/fm_pmc_b $OUT/fm_pmc_f $OUT/fm_pmc_c > $OUT/r04e_fm_pmc.jsonl
cat $OUT/r04e_fm_pmc.jsonl
tail -3 $OUT/r04e_fm_pmc_a.err $OUT/r04e_fm_pmc_c.err
rm -rf $OUT/fm_pmc_a $OUT/fm_pmc_b $OUT/fm_pmc_f $OUT/fm_pmc_c
