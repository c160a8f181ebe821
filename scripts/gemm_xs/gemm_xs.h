/* Entry points of the input-stationary K1/K2 experiment (scripts/gemm_xs/gemm_xs.hip).  NOT part of the product C-ABI
 * (include/lora_amd.h): round 5 measured the kernel (profiles/r05_kbench_xs.log: it beats the library GEMM on the plain
 * K = 320 product, loses fused and at K = 640), the step never routed to it, and round 6 moved it out of liblora_amd.so. */
#pragma once
#include "lora_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* K1/K2 input-stationary form (gemm_xs.hip): the same contract as lora_amd_linear_ws for ONE site — same packed weight
 * (lora_amd_ws_pack), same site struct (flayout 0: forward; 3: input gradient; no accumulate form), same dropout mask
 * indexing — with the roles turned round: a wave keeps its 32 input rows in registers (a lane's 16-byte piece IS the MFMA
 * operand), the weight panel goes through LDS.  For the short-contraction / many-row sites: K = 320 or 640
 * (lora_amd_xs_config returns 0 otherwise).  site->down == NULL: plain Y = X B^T + bias (a merged-weight site).
 * site->reserved = 1: site->wp is the row-major weight [N][K] itself (no packed copy).
 * replaces: lora_diffusion/lora.py:53-58 and its input gradient at those sites. */
int lora_amd_xs_config(int32_t K, int32_t *panel_cols, int32_t *block_rows);
/* measurement override (scripts/kbench.py): 16-row slabs per wave (1, 2, 4) and panels per workgroup; 0 = choose */
void lora_amd_xs_set_tuning(int32_t slabs, int32_t panels_per_group);
int lora_amd_linear_xs(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t act_dtype,
                       const lora_amd_ws_site *site, void *stream);
#ifdef __cplusplus
}
#endif
