// Host-side entry points of csrc/rank16_mfma.hip for the C-ABI functions of linear.hip / linear_fused.hip: each returns
// true when it took the launch (16-bit activations, f32 factor, rank 9..16, aligned rows, hook enabled), false when the
// caller's VALU kernel has to run.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lora_amd {

extern int g_r16_mfma;

bool r16_rowdot(const void *x, int64_t ldx, const void *f, int fdt, int layout, float *t_out, int64_t M, int C, int r,
                int act_dtype, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st);
bool r16_rank_update(void *y, int64_t ldy, const float *t, int nparts, int64_t part_stride, const void *f, int fdt, int layout,
                     int64_t M, int N, int r, int act_dtype, float scale, float p, uint64_t seed, uint64_t offset,
                     const uint64_t *offset_dev, hipStream_t st);
bool r16_bwd_g(const void *g, int64_t ldg, const float *t, const void *up, int fdt, float *gt_part, float *up_part, int64_t M,
               int N, int r, int log_ct8, int nct, int rows_per_block, int64_t nrb, int act_dtype, float scale, float p,
               uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st, float *gt_out = nullptr,
               unsigned *counters = nullptr);

}  // namespace lora_amd
