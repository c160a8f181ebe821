/*
 * lora_amd.h — C-ABI of the MI355X (gfx950) LoRA hot-path kernels.
 *
 * The reference (cloneofsimo/lora) has no FFI: its hot path is a sequence of
 * ATen calls issued from Python.  Each entry point below replaces one such
 * sequence; the `replaces:` tag names the reference lines (paths relative to
 * /root/reference).  A reference maintainer binds these with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is a DEVICE pointer unless the
 *     parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and
 *     the call never synchronises, allocates or frees (hipGraph-capturable);
 *   - row-major, contiguous unless a leading dimension `ld*` is given;
 *   - return value: LORA_AMD_OK or a negative LORA_AMD_E* code; the message of
 *     the last failure on the calling thread is lora_amd_last_error().
 */
#ifndef LORA_AMD_H
#define LORA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LORA_AMD_ABI_VERSION 7

/* status codes */
#define LORA_AMD_OK 0
#define LORA_AMD_EINVAL (-1)   /* bad argument (shape, dtype, alignment, null) */
#define LORA_AMD_ERANK (-2)    /* rank r outside [1, LORA_AMD_MAX_RANK] */
#define LORA_AMD_ELAUNCH (-3)  /* hipLaunchKernel reported an error */
#define LORA_AMD_EWORKSPACE (-4) /* workspace too small */
#define LORA_AMD_EUNSUPPORTED (-5) /* this entry has no kernel for the (dtype, rank, shape): the caller takes the documented other path */

#define LORA_AMD_MAX_RANK 64

/* element types of activations / frozen weights */
#define LORA_AMD_F32 0
#define LORA_AMD_F16 1
#define LORA_AMD_BF16 2

/* layout of a small (low-rank) factor */
#define LORA_AMD_FACTOR_RK 0 /* [r, K]  (lora_down.weight, lora.py:44) */
#define LORA_AMD_FACTOR_KR 1 /* [K, r]  (lora_up.weight,   lora.py:46) */

/* rounding of the merge kernel */
#define LORA_AMD_ROUND_REFERENCE 0 /* round exactly where collapse_lora does */
#define LORA_AMD_ROUND_ONCE 1      /* fp32 throughout, one final rounding   */
#define LORA_AMD_ROUND_DITHER 2    /* lora_amd_merge_step only: one final rounding with a fixed per-element dither */

int lora_amd_abi_version(void);
const char *lora_amd_last_error(void);
/* gfx arch string the library was compiled for ("gfx950"). */
const char *lora_amd_target_arch(void);

/* ------------------------------------------------------------------------
 * K3  fused merge  W' = W + alpha * (up @ down)
 * replaces: lora_diffusion/lora.py:646-655 (Linear) and :659-669 (Conv2d,
 *           both factors flattened from dim 1) — mm + mul + add + new Parameter.
 *
 * One descriptor per adapter site; ALL sites are merged by ONE launch.
 * `rows_per_tile`/`cols_per_tile` are chosen by lora_amd_merge_plan().
 * ---------------------------------------------------------------------- */
typedef struct lora_amd_merge_site {
  const void *w_in;  /* [N, K] frozen weight, w_dtype                        */
  void *w_out;       /* [N, K] result, w_dtype; may alias w_in (in place)    */
  const void *up;    /* [N, r] ab_dtype  (lora_up.weight,   flattened)       */
  const void *down;  /* [r, K] ab_dtype  (lora_down.weight, flattened)       */
  int32_t N, K, r;
  int32_t rows_per_tile; /* filled by lora_amd_merge_plan */
  int32_t cols_per_tile; /* filled by lora_amd_merge_plan; multiple of 8     */
  int32_t tiles_k;       /* filled by lora_amd_merge_plan                    */
  int64_t tile_begin;    /* filled by lora_amd_merge_plan (exclusive scan)   */
  int32_t flags;         /* filled by lora_amd_merge_plan; bit 0: 16-byte lanes
                            (K % 8 == 0 and w_in/w_out 16-byte aligned);
                            bit 1: column-owner kernel (rank <= 16, K/8 has a
                            power-of-two factor >= 4)                        */
  int32_t out_heads;     /* caller: 0 = w_out rows are dense [K]; d | (D << 16): w_out is the head-padded layout of a
                            weight whose INPUT is head-padded — row stride (K/d)*D, logical column k at
                            (k/d)*D + k%d (d, D multiples of 8, K % d == 0; pad columns are never written: zero them
                            once).  Column-owner sites only (lora_amd_merge_plan refuses it elsewhere)             */
  int32_t transposed;    /* caller: 0 = as described above.  1 = the site describes the TRANSPOSED product
                            W'^T = W^T + alpha (up down)^T: w_in / w_out are [N, K] with N = the ORIGINAL K and K = the
                            ORIGINAL N, `up` points at the original lora_down.weight [r, N] (row n of this site, rank j:
                            up[j*N + n]) and `down` at the original lora_up.weight [K, r] (column k, rank j: down[k*r + j]).
                            Same values as the untransposed site, element for element (same rank order of the fma chain);
                            what the input-gradient GEMM G W' wants as its [out, in] operand.  Column-owner sites only */
  int32_t reserved;
} lora_amd_merge_site;

typedef struct lora_amd_merge_summary {
  int64_t total_tiles;    /* grid size of the launch */
  int32_t n_fast_sites;   /* sites on the column-owner kernel (flags bit 1) */
  int32_t rank_tile_fast; /* 4 / 8 / 16: largest rank tile among them */
} lora_amd_merge_summary;

/* Host-side planner: fills rows_per_tile/cols_per_tile/tiles_k/tile_begin/flags of
 * `sites_host[0..n_sites)` and *summary.  Pure CPU arithmetic: callable without a GPU. */
int lora_amd_merge_plan(lora_amd_merge_site *sites_host, int32_t n_sites,
                        int32_t w_dtype, lora_amd_merge_summary *summary);

/* `sites_dev`: the planned descriptor array copied to device memory; `summary_host`: what the
 * planner returned.  alpha: collapse_lora's alpha (lora.py:635).  A negative alpha un-merges. */
int lora_amd_merge_batched(const lora_amd_merge_site *sites_dev, int32_t n_sites,
                           const lora_amd_merge_summary *summary_host,
                           int32_t w_dtype, int32_t ab_dtype, float alpha,
                           int32_t rounding, void *stream);

/* The merge INSIDE the training step (ops.MergedWeights; csrc/merge_step.hip): per site, from ONE read of the frozen
 * 16-bit W [N, K], the scratch weight W_eff = W + alpha up down (the operand of the forward GEMM, lora.py:53-58 without
 * dropout) and, when out_t is given, its transpose W_eff^T (the [out, in] operand of the input-gradient GEMM) — the same
 * values, written in the layouts the GEMMs consume:
 *   out   : logical (n, k) at out   + rowmap(n) * ld_out   + colmap(k)      rowmap(n) = (n / row_d) * row_D + n % row_d
 *   out_t : logical (k, n) at out_t + colmap(k) * ld_out_t + rowmap(n)      colmap(k) = (k / col_d) * col_D + k % col_d
 * (row_d / col_d = 0: dense; pad rows / columns are never written: zero them once).  ld_out / ld_out_t in elements: several
 * sites may share one wider buffer (q, k, v of an attention block: one GEMM forward).  f32 factors up [N, r], down [r, K].
 * rounding: LORA_AMD_ROUND_ONCE (nearest even) or LORA_AMD_ROUND_DITHER: nearest after adding a fixed per-element dither
 * hash(dither_key, n, k) in [0, 1) ulp — P(round away from zero) = frac((W + delta) / ulp), exactly W where delta = 0,
 * identical from step to step: a delta below half an ulp of the frozen weight survives in the sum over a row instead of
 * vanishing element by element.  tiles_k / tile_begin are filled by the plan (host, no GPU needed). */
typedef struct lora_amd_mstep_site {
  const void *w;
  const float *up, *down;
  void *out, *out_t;
  int64_t ld_out, ld_out_t;
  int32_t N, K, r;
  int32_t row_d, row_D, col_d, col_D;
  int32_t dither_key;
  int32_t tiles_k, reserved;
  int64_t tile_begin;
} lora_amd_mstep_site;
int lora_amd_merge_step_plan(lora_amd_mstep_site *sites_host, int32_t n, int32_t w_dtype, int64_t *plan_value);
/* plan_value: what the plan returned (tile count and tile geometry, opaque) */
int lora_amd_merge_step(const lora_amd_mstep_site *sites_dev, int32_t n, int64_t plan_value, int32_t rank_max,
                        int32_t w_dtype, float alpha, int32_t rounding, void *stream);
/* Tuning hook (scripts/kbench.py): tile geometry 0..3 = 128x64 | 64x128 | 128x128 | 256x64 (rows x columns; applies to
 * tables planned after the call), dither form 1 = one hash per element, 2 = one hash chain per 16-byte chunk.  < 0 keeps. */
int lora_amd_merge_step_set_tuning(int32_t tile, int32_t dither);
/* Ranks 9..16 on 16-bit activations with f32 factors: rowdot / rowdot_masked / rank_update / linear_fwd / linear_bwd_g run
 * their matrix-core forms (csrc/rank16_mfma.hip; same arguments, outputs and partial-buffer geometry).  enable = 0 routes
 * them back to the VALU kernels (parity tests compare the two), 1 = default, < 0 only reads.  Returns the previous value. */
int lora_amd_rank16_mfma(int32_t enable);

/* Tuning knobs of the planner/launcher (<= 0 keeps the current value):
 * target elements per tile and resident workgroups per CU (the LDS-slab kernel's grid cap).  blocks_per_cu >= 100
 * means "blocks_per_cu - 100, with non-temporal W loads/stores" (the default), < 100 turns them off (A/B runs). */
int lora_amd_merge_set_tuning(int64_t tile_elems, int64_t blocks_per_cu);

/* ------------------------------------------------------------------------
 * K1/K2 primitives.  LoraInjectedLinear.forward (lora.py:53-58) is
 *     T = rowdot(X, down)            lora.py:56  self.lora_down(input) [+ selector]
 *     Y = rank_update(Y0, T, up, s)  lora.py:56-57 lora_up, dropout, *scale, +
 * and its autograd (implicit in the reference) is
 *     Gt = rowdot(G, up, scale=s)          dT
 *     dUp   += colreduce(G, T,  s)         [N, r]
 *     dDown += colreduce(X, Gt, 1)         [r, K]
 *     dX  = rank_update(G @ W, Gt, down, 1)
 * ---------------------------------------------------------------------- */

/* T[M, r] (f32) = scale * X[M, K] @ F^T, F given as [r,K] or [K,r];
 * optional selector S [r, r] (f32): T <- T @ S^T (sel_transposed=0, forward,
 * lora.py:56,63-70) or T <- T @ S (sel_transposed=1, backward).  sel may be NULL. */
int lora_amd_rowdot(const void *x, int64_t ldx, const void *factor, void *t_out,
                    int64_t M, int32_t K, int32_t r, int32_t x_dtype,
                    int32_t factor_dtype, int32_t factor_layout, float scale,
                    const float *sel, int32_t sel_transposed, void *stream);

/* Y[M, N] (y_dtype, in place) += scale * mask * T[M, r] (f32) @ F, F given as
 * [r,N] or [N,r].  Dropout (lora.py:45,56): keep-probability 1-p with inverted
 * scaling, generated in-kernel from Philox(seed, offset, offset_dev, element index); p = 0
 * disables it.  The same (seed, offset, offset_dev) reproduces the same mask in rowdot_masked. */
int lora_amd_rank_update(void *y, int64_t ldy, const float *t, const void *factor,
                         int64_t M, int32_t N, int32_t r, int32_t y_dtype,
                         int32_t factor_dtype, int32_t factor_layout,
                         float scale, float dropout_p, uint64_t seed,
                         uint64_t offset, const uint64_t *offset_dev, void *stream);

/* As lora_amd_rowdot but X is first multiplied elementwise by the dropout mask
 * of (seed, offset, offset_dev, p): the backward of a dropped-out rank_update. */
int lora_amd_rowdot_masked(const void *x, int64_t ldx, const void *factor,
                           void *t_out, int64_t M, int32_t K, int32_t r,
                           int32_t x_dtype, int32_t factor_dtype,
                           int32_t factor_layout, float scale, const float *sel,
                           int32_t sel_transposed, float dropout_p,
                           uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream);

/* D (f32) = beta * D + scale * sum_m T[m, j] * mask * X[m, k]; D is [r,K] or
 * [K,r] per out_layout.  Two-stage reduction through `workspace` (bytes given
 * by lora_amd_colreduce_workspace).  dropout args as above (p = 0: no mask). */
size_t lora_amd_colreduce_workspace(int64_t M, int32_t K, int32_t r);
int lora_amd_colreduce(const void *x, int64_t ldx, const float *t, float *d_out,
                       int64_t M, int32_t K, int32_t r, int32_t x_dtype,
                       int32_t out_layout, float scale, float beta,
                       float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                       void *workspace, size_t workspace_bytes, void *stream);



/* Batched forms (grid.y = matrix index; strides in elements between consecutive matrices of a stack) and the small
 * dense step of CholeskyQR, for the SVD distillation of cli_svd.py:24-92 (every same-shape site of a model in one
 * launch instead of one torch.linalg.svd per site).
 *   rowdot_batched:    T_b[M, r] (f32) = scale * X_b[M, K] @ F_b^T           (F_b [r, K] or [K, r])
 *   colreduce_batched: D_b (f32, [r, K] or [K, r]) = scale * T_b^T @ X_b     (workspace: batch * colreduce_workspace)
 *   chol_inverse_batched: out_b = L_b^{-1} with G_b + shift_rel * tr(G_b)/l * I = L_b L_b^T  (l <= 32), so that
 *                      Q_b = Y_b L_b^{-T} = rowdot_batched(Y_b, out_b) is orthonormal when G_b = Y_b^T Y_b. */
int lora_amd_rowdot_batched(const void *x, int64_t ldx, int64_t stride_x, const void *factor, int64_t stride_factor,
                            float *t_out, int64_t stride_t, int32_t batch, int64_t M, int32_t K, int32_t r,
                            int32_t x_dtype, int32_t factor_dtype, int32_t factor_layout, float scale, void *stream);
int lora_amd_colreduce_batched(const void *x, int64_t ldx, int64_t stride_x, const float *t, int64_t stride_t,
                               float *d_out, int64_t stride_d, int32_t batch, int64_t M, int32_t K, int32_t r,
                               int32_t x_dtype, int32_t out_layout, float scale, void *workspace,
                               size_t workspace_bytes, void *stream);
int lora_amd_chol_inverse_batched(const float *gram, float *out, int32_t l, int32_t batch, float shift_rel,
                                  void *stream);

/* Ragged forms: the same passes over a TABLE of stacks of different shapes in one launch (cli_svd.py:24-92 walks 224
 * sites of 31 shapes; one table entry per shape group).  f32 matrices and factors; the rank r is common to the table.
 * The caller fills the first block of fields, lora_amd_ragged_plan fills the second (host side) and returns the grid
 * sizes; the table is then copied to the device once and reused by every launch on the same buffers.
 *   rowdot_ragged:    out_g[b] [M, r] = scale * x_g[b] [M, K] @ f_g[b]^T          (f [r, K] or [K, r] per `layout`)
 *   colreduce_ragged: out_g[b] ([r, K] or [K, r]) = scale * f_g[b]^T @ x_g[b]    (f = T [M, r]; partial: workspace of
 *                     batch * lora_amd_colreduce_workspace(M, K, r) bytes per group)
 *   sub_ragged:       out = (f32) a - (f32) b over a table of flat arrays (the residuals W_tuned - W_base, ref :30-32) */
typedef struct lora_amd_ragged_desc {
  const void *x;          /* stack of `batch` [M, K] matrices: row stride ldx, matrix stride stride_x (elements) */
  const void *f;          /* rowdot: factor stack; colreduce: T stack [M, r]; matrix stride stride_f (elements) */
  float *out;             /* matrix stride stride_out (elements) */
  float *partial;         /* colreduce only */
  int64_t ldx, stride_x, stride_f, stride_out, M;
  int32_t K, batch;
  /* filled by lora_amd_ragged_plan */
  int64_t stride_partial, begin1, begin2;
  int32_t blocks1, blocks2, col_tiles, kt_cols, logL, rows_per_block;
} lora_amd_ragged_desc;
enum { LORA_AMD_RAGGED_ROWDOT = 0, LORA_AMD_RAGGED_COLREDUCE = 1 };
int lora_amd_ragged_plan(int32_t op, lora_amd_ragged_desc *descs, int32_t n, int32_t r, int64_t *grid1, int64_t *grid2);
int lora_amd_rowdot_ragged(const lora_amd_ragged_desc *descs_dev, int32_t n, int64_t grid1, int32_t r,
                           int32_t factor_layout, float scale, void *stream);
int lora_amd_colreduce_ragged(const lora_amd_ragged_desc *descs_dev, int32_t n, int64_t grid1, int64_t grid2, int32_t r,
                              int32_t out_layout, float scale, void *stream);
/* The same skinny products on the matrix cores for stacks held as two 16-bit planes (X = hi + lo, the bytes of f32):
 * out[b] [M, r] = X[b] [M, C] @ F[b] [C, r]  (r <= 16, C % 32 == 0).  lora_amd_split16_ragged makes the planes of flat f32
 * arrays (n % 8 == 0 each; `begin` = running count of 4096-element blocks, filled by the caller).
 * replaces: the colreduce passes over dW / dW^T of the subspace iteration (cli_svd.py:24-92 as restated in cli_svd.py). */
typedef struct lora_amd_planes_desc {
  const void *hi, *lo;    /* planes of a dense stack [batch][M][C] */
  const float *f;         /* [batch][C][r] */
  float *out;             /* [batch][M][r] */
  int64_t M;
  int32_t C, batch;
  /* filled by lora_amd_rowdot16_planes_plan */
  int32_t wps, slabs_per_wg;
  int64_t wg_begin;
} lora_amd_planes_desc;
int lora_amd_rowdot16_planes_plan(lora_amd_planes_desc *descs_host, int32_t n, int64_t *grid);
int lora_amd_rowdot16_planes(const lora_amd_planes_desc *descs_dev, int32_t n, int64_t grid, int32_t r, int32_t plane_dtype,
                             void *stream);
/* The same with the factor given as PACKED fragments (lora_amd_thin_pack): desc.f points at 16-bit elements
 * [batch][C / 32][hi 512 | lo 512] (the bytes of the f32 [C][16] factor), 16 output columns.  hi_only != 0 (ABI 6): the hi
 * plane alone (X to 8 mantissa bits, half the bytes) — the power iterations of cli_svd's subspace iteration, which only steer a
 * subspace; the pass that forms the returned factors reads both planes. */
int lora_amd_rowdot16_planes_packed(const lora_amd_planes_desc *descs_dev, int32_t n, int64_t grid, int32_t plane_dtype,
                                    int32_t hi_only, void *stream);
typedef struct lora_amd_split_desc {
  const float *src;
  void *hi, *lo;
  int64_t n, begin;
} lora_amd_split_desc;
int lora_amd_split16_ragged(const lora_amd_split_desc *descs_dev, int32_t n, int64_t blocks, int32_t plane_dtype, void *stream);
/* The planes of a stack [batch][N][K] AND of its transpose [batch][K][N] from one read (N % 8 == 0, K % 8 == 0; 64 x 64
 * tiles; `tile_begin` = running count of batch * ceil(N / 64) * ceil(K / 64), filled by the caller). */
typedef struct lora_amd_splitt_desc {
  const float *src;
  void *hi, *lo, *thi, *tlo;
  int32_t batch, N, K, reserved;
  int64_t tile_begin;
} lora_amd_splitt_desc;
int lora_amd_split16_transpose(const lora_amd_splitt_desc *descs_dev, int32_t n, int64_t tiles, int32_t plane_dtype,
                               void *stream);
typedef struct lora_amd_sub_desc {
  const void *a, *b;
  float *out;
  int64_t n, begin;       /* elements; begin: running count of 4096-element blocks (filled by the caller) */
} lora_amd_sub_desc;
int lora_amd_sub_ragged(const lora_amd_sub_desc *descs_dev, int32_t n, int64_t blocks, int32_t in_dtype, void *stream);
/* The small dense steps of the same subspace iteration, fused (csrc/svd_small.hip): every site of a model in ONE launch, the
 * reduction behind a launch finished inside it by the last-arriving workgroup of each site.  A THIN matrix is [rows][16] f32,
 * row-major, at element offset `off` of a flat buffer (`off` % 16 == 0); the same site table serves every buffer of that
 * layout.  `blockmap` [total_blocks] int32 = the site of every 256-row block; counters [nsites] uint32, zero before the first
 * launch (the last arriver of a site resets its word).
 * replaces (cli_svd.py:24-92 as restated in lora_amd/cli_svd.py): colreduce_ragged (Gram) + chol_inverse_batched +
 * rowdot_ragged per CholeskyQR pass, torch.linalg.svd of the 16 x 16 cores (rocSOLVER, one call per site), the per-group sign
 * fixes, torch.quantile's sort and the clamp. */
typedef struct lora_amd_thin_site {
  int64_t off, rows;
  int64_t block_begin;    /* running count of ceil(rows / 256) */
  int32_t blocks, reserved;
} lora_amd_thin_site;
typedef struct lora_amd_thin_finish {
  float *part;            /* [total_blocks][256] f32 workspace */
  uint32_t *counters;     /* [nsites] */
  int32_t mode;           /* 1: out = L^{-1} of (sum + shift_rel tr(sum)/16 I) = L L^T; 2: SVD of the sum */
  float shift_rel;
  float *linv_out;        /* mode 1: [nsites][16][16] */
  float *ritz_out;        /* mode 1, or NULL: [nsites][2] = (sum of the `rank` largest eigenvalues of the un-shifted sum,
                           * sum of the other 16 - rank) */
  float *ubt, *vb;        /* mode 2: [nsites][rank][16] = U[:, :rank]^T and V[:, :rank]^T of sum = U S V^T */
  float *s_out;           /* mode 2: [nsites][16], descending */
  int32_t rank, reserved;
} lora_amd_thin_finish;
/* sum = A^T B over the rows of every site (b = NULL: B = A, the Gram matrix), then `fin`. */
int lora_amd_thin_gram(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                       const float *a, const float *b, const lora_amd_thin_finish *fin, void *stream);
/* dst = src M^T per site (mats [nsites][16][16]: the L^{-1} of a finish); fin != NULL: the Gram matrix of dst, then `fin`. */
int lora_amd_thin_apply(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                        const float *src, const float *mats, float *dst, const lora_amd_thin_finish *fin, void *stream);
/* dst [rows][rank] (at element offset off / 16 * rank) = (src M^T)[:, :rank] * scale_a * scale_b, mats [nsites][rank][16],
 * scales [nsites][16] or NULL.  sign_out != NULL: also sign_out [nsites][16] = the sign of the largest-magnitude entry of
 * every output column BEFORE scaling (first row on ties; cli_svd's deterministic sign rule), through sign_part
 * [total_blocks][32] f32 and sign_rows [total_blocks][16] int32. */
int lora_amd_thin_rotate(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                         const float *src, const float *mats, int32_t rank, const float *scale_a, const float *scale_b,
                         float *dst, float *sign_part, int32_t *sign_rows, uint32_t *counters, float *sign_out, void *stream);
/* src [rows][16] f32 of every site -> the (hi, lo) MFMA fragments lora_amd_rowdot16_planes_packed reads, at 16-bit element
 * offset 2 * off of dst: [rows / 32][hi 512 | lo 512], lane l = (j = l & 15, kq = l >> 4) holds src[32 ks + 8 kq + e][j]. */
int lora_amd_thin_pack(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                       const float *src, void *dst, int32_t plane_dtype, void *stream);
/* Order statistics of a site's joint values {u [n_u], v [n_v] * sign[j % rank]} (the distribution cli_svd.py:39-47 takes its
 * quantile of): three launches, pass = 0, 1, 2 (radix 11 + 11 + 10 bits of the order-preserving key); state [nsites][8]
 * uint32 = {0, k, 0, 0xffffffff, 0, 0, 0, 0} before pass 0 with k = the 0-based ascending index wanted; after pass 2
 * out2 [nsites][2] = (statistic k, statistic min(k + 1, n - 1)), both NaN if the site holds a NaN (torch.quantile's rule).
 * hist [nsites][2048] uint32, zero before pass 0.  Blocks of 8192 values. */
typedef struct lora_amd_thin_qsite {
  int64_t off_u, off_v, n_u, n_v;
  int64_t block_begin;    /* running count of ceil((n_u + n_v) / 8192) */
  int32_t blocks, reserved;
} lora_amd_thin_qsite;
int lora_amd_thin_select(const lora_amd_thin_qsite *qsites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                         const float *u, const float *v, const float *sign, int32_t rank, int32_t pass, uint32_t *hist,
                         uint32_t *counters, void *state, float *out2, void *stream);
/* u <- clamp(u, -hi, hi) in place ([N][rank] = `up`); down [rank][K] (at off_v) = clamp(v [K][rank] * sign, -hi, hi);
 * hi [nsites]; NaN in a value or in hi propagates (torch.minimum / maximum). */
int lora_amd_thin_clamp(const lora_amd_thin_qsite *qsites_dev, const int32_t *blockmap_dev, int64_t total_blocks, float *u,
                        const float *v, const float *sign, const float *hi, float *down, int32_t rank, void *stream);
/* dW = (f32) tuned - (f32) base of ONE site [N][K] (cli_svd.py:30-32; N % 8 == 0, K % 8 == 0) -> the (hi, lo) planes of dW
 * ([N][K]) and of dW^T ([K][N]) from one read of the two weights, and norm_part[tile] = the sum of squares of every 64 x 64
 * tile (tile_begin = running count of ceil(N / 64) * ceil(K / 64), filled by the caller; the tiles of a site are consecutive).
 * replaces: lora_amd_sub_ragged + lora_amd_split16_transpose (the f32 residual is never written). */
typedef struct lora_amd_resid_desc {
  const void *tuned, *base;
  void *hi, *lo, *thi, *tlo;
  int32_t N, K;
  int64_t tile_begin;
} lora_amd_resid_desc;
int lora_amd_split16_residual(const lora_amd_resid_desc *descs_dev, int32_t n, int64_t tiles, int32_t in_dtype,
                              int32_t plane_dtype, float *norm_part, void *stream);
/* ------------------------------------------------------------------------
 * K1/K2 fused: one launch forward, two backward, one batched reduction per step.
 * These are what LoraInjectedLinear runs for 16-byte-friendly shapes (K%8==0, N%8==0,
 * rank <= 16); lora_amd_linear_plan says whether a shape qualifies and sizes the
 * workspaces.  replaces: lora.py:53-58 (+ its autograd) — 5 + ~10 ATen launches per site.
 * ---------------------------------------------------------------------- */
typedef struct lora_amd_linear_plan_t {
  int32_t fused;      /* 1: the fused entry points accept this shape */
  int32_t rank_tile;  /* RT: r rounded up to 4 / 8 / 16 */
  int32_t nct_g;      /* column tiles of the G pass (= number of Gt partials) */
  int32_t nparts_up, nparts_down; /* row-block partials of dUp / dDown */
  int32_t reserved;
  int64_t gt_part_floats;   /* nct_g * M * r          */
  int64_t up_part_floats;   /* nparts_up * RT * N     */
  int64_t down_part_floats; /* nparts_down * RT * K   */
} lora_amd_linear_plan_t;

int lora_amd_linear_plan(int64_t M, int32_t K, int32_t N, int32_t r, lora_amd_linear_plan_t *out);

/* Y[M,N] (in place, holds X W^T + b) += scale * mask * T @ up^T with T[M,r] = (X @ down^T) @ S^T,
 * T also written to t_out (f32) for the backward.  lora.py:53-58. */
int lora_amd_linear_fwd(const void *x, int64_t ldx, void *y, int64_t ldy, const void *down,
                        const void *up, float *t_out, int64_t M, int32_t K, int32_t N, int32_t r,
                        int32_t act_dtype, int32_t factor_dtype, float scale, const float *sel,
                        float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream);

/* One pass over G[M,N]: gt_part[nct_g][M][r] = scale * (mask*G) @ up (per column tile) and
 * up_part[nparts_up][RT][N] = scale * (mask*G)^T @ T (per row block).  gt_part may be NULL (dUp partials only). */
int lora_amd_linear_bwd_g(const void *g, int64_t ldg, const float *t, const void *up, float *gt_part,
                          float *up_part, int64_t M, int32_t N, int32_t r, int32_t act_dtype,
                          int32_t factor_dtype, float scale, float dropout_p, uint64_t seed,
                          uint64_t offset, const uint64_t *offset_dev, void *stream);

/* The same with the fold of the nct_g column-tile partials INSIDE the launch: gt_out [M, r] is complete when the launch ends
 * (the last-arriving column workgroup of a row block sums the tiles in order and resets the block's counter), no
 * lora_amd_sum_parts behind it.  counters: lora_amd_linear_bwd_g_blocks(M, N, r) uint32, zeroed ONCE.  Matrix-core form
 * only: bf16 rows, f32 factors, rank 9..16 — anything else returns LORA_AMD_EUNSUPPORTED. */
int lora_amd_linear_bwd_g_blocks(int64_t M, int32_t N, int32_t r, int64_t *row_blocks);
int lora_amd_linear_bwd_g_folded(const void *g, int64_t ldg, const float *t, const void *up, float *gt_part, float *gt_out,
                                 uint32_t *counters, float *up_part, int64_t M, int32_t N, int32_t r, int32_t act_dtype,
                                 int32_t factor_dtype, float scale, float dropout_p, uint64_t seed, uint64_t offset,
                                 const uint64_t *offset_dev, void *stream);

/* One pass over X[M,K] (and dX, in place, holding G @ W; may be NULL): with Gt' = (sum gt_part) @ S,
 * down_part[nparts_down][RT][K] = Gt'^T @ X (per row block) and dX += Gt' @ down. */
int lora_amd_linear_bwd_x(const void *x, int64_t ldx, void *dx, int64_t lddx, const float *gt_part,
                          int32_t nct_g, const void *down, const float *sel, float *down_part, int64_t M,
                          int32_t K, int32_t r, int32_t act_dtype, int32_t factor_dtype, void *stream);

/* Both factor-gradient partials in ONE launch, for sites whose input gradient and Gt [M, r] came out of
 * lora_amd_linear_gemm_fwd (factor_layout 3): up_part as in lora_amd_linear_bwd_g (with T scaled by `scale`),
 * down_part as in lora_amd_linear_bwd_x with dx = NULL and one Gt part (`sel` [r, r] or NULL).  No dropout. */
int lora_amd_linear_bwd_factors(const void *g, int64_t ldg, const float *t, float *up_part, const void *x,
                                int64_t ldx, const float *gt, const float *sel, float *down_part, int64_t M,
                                int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale, void *stream);
/* The same for a site whose branch had nn.Dropout (lora.py:45, 56): the G pass sees mask * G, the mask regenerated
 * from (seed, offset [+ *offset_dev]) as lora_amd_linear_fwd indexed it.  Pairs with the dropout fields of
 * lora_amd_ws_site (forward and input-gradient calls). */
int lora_amd_linear_bwd_factors_drop(const void *g, int64_t ldg, const float *t, float *up_part, const void *x,
                                     int64_t ldx, const float *gt, const float *sel, float *down_part, int64_t M,
                                     int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale, float dropout_p,
                                     uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream);
/* The same for head-padded G and / or X rows (see lora_amd_linear_gemm_fwd_heads): g_head_* describe G [M, heads*D],
 * x_head_* describe X; the partials stay dense. */
int lora_amd_linear_bwd_factors_heads(const void *g, int64_t ldg, const float *t, float *up_part, const void *x,
                                      int64_t ldx, const float *gt, const float *sel, float *down_part, int64_t M,
                                      int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale,
                                      int32_t g_head_dim, int32_t g_head_pad, int32_t x_head_dim, int32_t x_head_pad,
                                      void *stream);

/* K1 fully fused on the matrix cores: Y[M,N] = X[M,K] W[N,K]^T + bias + scale * (X down^T) up^T, and
 * t_out[M,r] (f32) = t_scale * X down^T for the backward.  ONE launch replaces the frozen addmm AND the low-rank branch
 * of lora.py:53-58 (no dropout, no selector: those keep lora_amd_linear_fwd).  bf16/f16 X, W, bias, Y; f32 factors;
 * K % 64 == 0, N % 8 == 0, r <= 16, 16-byte-aligned rows.
 * factor_layout: bit 0 set = `down` is stored [K, r] instead of [r, K]; bit 1 set = `up` is stored [r, N] instead of
 * [N, r].  With both set the SAME entry point computes the input gradient of the site,
 *     dX[M,K] = G[M,N] W[N,K] + scale * (G up) down,   Gt = scale * G up,
 * when called as (x = G, w = W^T [K,N], y = dX, down = up [N,r], up = down [r,K], K <-> N swapped, t_scale = scale).
 * tile = 10 * stages + shape (shape 1: 64x320, 2: 64x160, 3: 32x160, 4: 128x160 outputs per workgroup; stages 2|3 LDS
 * ring slots); 0 = pick by grid size. */
int lora_amd_linear_gemm_supported(int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype);
int lora_amd_linear_gemm_fwd(const void *x, int64_t ldx, const void *w, int64_t ldw, const void *bias, void *y,
                             int64_t ldy, const float *down, const float *up, float *t_out, int64_t M, int32_t K,
                             int32_t N, int32_t r, int32_t act_dtype, float scale, float t_scale,
                             int32_t factor_layout, int32_t tile, void *stream);

/* K1/K2 weight-stationary form (csrc/gemm_ws.hip): one launch for up to LORA_AMD_WS_MAX_SITES sites that read the SAME
 * input X[M,K] (lora.py:53-58 called on to_q / to_k / to_v with one tensor): per site
 *     Y = X B^T + bias + scale * (X down^T) up^T,     t_out[M,r] (f32) = t_scale * X down^T   (t_out may be NULL)
 * with B [N,K] given PRE-PACKED in MFMA fragment order (lora_amd_ws_pack; frozen weights are packed once and stay
 * resident), so that a workgroup keeps its [64*CS columns][K] panel in registers and only X streams (LDS-DMA ring).
 * Supported contraction lengths K: 320, 640, 768, 1280 (lora_amd_ws_config returns 0 otherwise), bf16/f16, rank <= 16,
 * N % 4 == 0, ldy % 4 == 0.  flayout as in lora_amd_linear_gemm_fwd (bit 0: down is [K,r]; bit 1: up is [r,N]);
 * bit 2: accumulate, Y += ... (the input gradients of sites that shared an input meet in one dX).
 * The input gradient of a site is the same call on the weight packed in the other orientation:
 *     dX[M,K'] = G[M,N'] W[N',K'] + scale (G up) down,  Gt = scale G up:
 *     x = G, K = N', site = {wp = pack(W viewed as B[k'][n'] : stride_n = 1, stride_k = ldw), N = K', down = up [N',r]
 *     (flayout bit 0), up = down [r,K'] (bit 1), t_scale = scale, bias = NULL}.
 * row_groups: 0 = choose (about one workgroup per CU; a multiple of 8 so that the panels of a row group share an XCD). */
#define LORA_AMD_WS_MAX_SITES 4
typedef struct lora_amd_ws_site {
  const void *wp;    /* packed B, lora_amd_ws_packed_elems(N, K) elements */
  const void *bias;  /* [N] activation dtype, or NULL */
  void *y;           /* [M, N] activation dtype, row stride ldy */
  const float *down; /* [r, K] (or [K, r]) f32 */
  const float *up;   /* [N, r] (or [r, N]) f32 */
  float *t_out;      /* [M, r] f32 or NULL */
  int64_t ldy;
  int32_t N, r, panel_begin /* filled by the launcher */, flayout;
  float scale, t_scale;
  /* nn.Dropout on the low-rank branch (lora.py:45, 56); 0 = off.  Mask indexing as lora_amd_linear_fwd /
   * lora_amd_linear_bwd_g: element (m, n) of the site's [M, N] output (forward) resp. of G (input-gradient call, where
   * the launch's contraction length K is that N).  Needs N % 8 == 0. */
  float dropout_p;
  int32_t y_heads;   /* head-padded output: d | D << 16 (heads of d columns stored in slots of D, pad written as zeros; ldy >=
                      * N / d * D); 0 = dense rows.  Needs N a multiple of the panel width (lora_amd_ws_config), no
                      * accumulation, D - d <= d <= 2 (D - d) (40 in 64, 80 in 128, 160 in 256) */
  uint64_t seed, offset;
  const uint64_t *offset_dev; /* device int64 added to `offset` (graph replay / checkpoint recompute), or NULL */
} lora_amd_ws_site;

int lora_amd_ws_config(int32_t K, int32_t *panel_cols, int32_t *tile_rows);
int64_t lora_amd_ws_packed_elems(int32_t N, int32_t K);
/* Pack B (element (n, k) at w[n * stride_n + k * stride_k], n < N, k < K) into fragment order (zero-padded panels). */
int lora_amd_ws_pack(const void *w, int64_t stride_n, int64_t stride_k, int32_t N, int32_t K, int32_t dtype, void *out,
                     void *stream);
int lora_amd_linear_ws(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t act_dtype,
                       const lora_amd_ws_site *sites /* host array */, int32_t nsites, int32_t row_groups, void *stream);
/* lora_amd_linear_ws on a head-padded INPUT (ABI 7; the dropout sites around the attention core, lora.py:53-58 called by
 * CrossAttention.to_q / to_k / to_v / to_out): x rows hold K / x_head_dim heads of x_head_dim columns in slots of x_head_pad
 * (multiples of 8; pad never read); 0, 0 = dense.  A site's y_heads does the same for its output.  With both, the pad /
 * slice copies around the kernel disappear in either direction (forward: to_out reads the attention output as it is, q / k / v
 * leave padded; input-gradient call: G of q / k / v arrives padded, dX of to_out leaves padded). */
int lora_amd_linear_ws_heads(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t x_head_dim, int32_t x_head_pad,
                             int32_t act_dtype, const lora_amd_ws_site *sites /* host array */, int32_t nsites,
                             int32_t row_groups, void *stream);

/* lora_amd_linear_gemm_fwd with head-padded activations: a row of `heads` runs of d elements stored with every run padded to D
 * (d, D multiples of 8; the layout attention kernels want for head sizes 40 / 80).  x_head_dim/x_head_pad describe
 * the X operand (logical K = heads * d, physical row length heads * D, pad never read), y_head_dim/y_head_pad the
 * output (pad written as zeros; needs 160 % d == 0 so that an output tile owns whole heads).  0 = dense.  Removes
 * the pad / slice copies around the attention core: q, k, v projections write the padded layout, the output
 * projection reads it, and the input-gradient call (factor_layout 3) does the same in the other direction. */
int lora_amd_linear_gemm_fwd_heads(const void *x, int64_t ldx, const void *w, int64_t ldw, const void *bias, void *y,
                                   int64_t ldy, const float *down, const float *up, float *t_out, int64_t M,
                                   int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale, float t_scale,
                                   int32_t factor_layout, int32_t tile, int32_t x_head_dim, int32_t x_head_pad,
                                   int32_t y_head_dim, int32_t y_head_pad, void *stream);

/* out (f32, [r,C] or [C,r]) = beta*out + scale * sum_p part[p][j][c], part laid out [nparts][RT][C].
 * ONE launch covers every descriptor: the trainer reduces all sites' partials into its flat gradient
 * buffer once per step.  `begin` = exclusive prefix sum of r*C over the table; total = its end. */
typedef struct lora_amd_reduce_desc {
  const float *part;
  float *out;
  int64_t begin;
  int32_t nparts, RT, C, r, layout, reserved;
  float scale, beta;
} lora_amd_reduce_desc;

/* The parameter gradients of a site on the merged-weight path (no T, no Gt saved or produced by another launch):
 * up_part[rb][RT][N] = sum_m (s X down^T)[m, j] G[m, n],  down_part[rb][RT][K] = sum_m (s G up)[m, j] X[m, k] per row
 * block rb, in ONE launch (each row block of G and X is read twice: row-dot phase, column-sum phase; replaces the autograd of lora.py:53-58 for dA, dB when the
 * forward ran on W + s up down).  f32 factors; G / X rows may be head-padded (d, D as in linear_bwd_factors_heads).
 * The plan sizes the partial slabs: `nparts` row blocks, folded by lora_amd_reduce_batched. */
typedef struct lora_amd_factors_self_plan_t {
  int32_t supported, rank_tile, nparts, reserved;
  int64_t up_part_floats, down_part_floats;
} lora_amd_factors_self_plan_t;
int lora_amd_linear_factors_self_plan(int64_t M, int32_t K, int32_t N, int32_t r, lora_amd_factors_self_plan_t *out);
/* the same with the rows of a block chosen by the caller (0: as above); the one-launch pass takes `rows_per_block` of
 * lora_amd_self_site as that choice, so a site's slabs must be sized with the same value */
int lora_amd_linear_factors_self_plan_rows(int64_t M, int32_t K, int32_t N, int32_t r, int32_t rows,
                                           lora_amd_factors_self_plan_t *out);
int lora_amd_linear_bwd_factors_self(const void *g, int64_t ldg, const void *x, int64_t ldx, const float *down,
                                     const float *up, float *up_part, float *down_part, int64_t M, int32_t K,
                                     int32_t N, int32_t r, int32_t act_dtype, float scale, int32_t g_head_dim,
                                     int32_t g_head_pad, int32_t x_head_dim, int32_t x_head_pad, void *stream);
/* ... and for every site of a model in ONE launch (issued once per step after the backward, when every G and X is at
 * hand): one lora_amd_self_site per adapter; same activation dtype and rank tile (4 / 8 / 16) throughout a table.  The
 * caller fills the first block, lora_amd_linear_factors_self_ragged_plan (host) the second and returns the grid; the
 * partial slabs are sized by lora_amd_linear_factors_self_plan per site and folded by lora_amd_reduce_batched. */
typedef struct lora_amd_self_site {
  const void *g, *x;            /* [M, N] output gradient, [M, K] input (act_dtype; rows may be head-padded) */
  const float *down, *up;       /* f32 [r, K], [N, r] */
  float *up_part, *down_part;   /* [nparts][RT][N], [nparts][RT][K] */
  int64_t ldg, ldx, M;
  int32_t N, K, r;
  float scale;
  int32_t g_head_dim, g_head_pad, x_head_dim, x_head_pad;
  /* rows_per_block: caller's choice (0: the per-site default); everything else is filled by the plan */
  int32_t rows_per_block, nsplit, kt_g, logL_g, kt_x, logL_x, tile_g, nct_g, tile_x, nct_x;
  int32_t reserved0, reserved;  /* 0 */
  int64_t block_begin;
} lora_amd_self_site;
int lora_amd_linear_factors_self_ragged_plan(lora_amd_self_site *sites, int32_t n, int32_t act_dtype, int64_t *grid);
int lora_amd_linear_bwd_factors_self_ragged(const lora_amd_self_site *sites_dev, int32_t n, int64_t grid, int32_t rank,
                                            int32_t act_dtype, void *stream);
int lora_amd_reduce_batched(const lora_amd_reduce_desc *descs_dev, int32_t n, int64_t total, void *stream);
/* The same pass on the matrix cores, every row of G and X read from HBM ONCE (16-bit activations; csrc/factor_mfma.hip):
 * a workgroup keeps the R rows of the narrower of (X, G) resident in LDS, streams the wider one through LDS in column
 * chunks, and runs both contractions of lora.py:53-58's factor autograd as v_mfma_f32_16x16x32 tiles
 *     T = s X down^T,  Gt = s G up    (rank padded to 16, factors and T / Gt split hi + lo: f32-grade results)
 *     up_part[rb] = T^T G,  down_part[rb] = Gt^T X    (the row-contraction operand = ds_read_b64_tr_b16 transpose reads)
 * Same partial-slab layout as lora_amd_linear_bwd_factors_self_ragged ([nparts][RT][C], folded by lora_amd_reduce_batched).
 * The factors arrive packed in MFMA fragment order (lora_amd_factor_pack, one launch per step for all sites).
 * A workgroup walks blocks_per_wg consecutive row blocks (the next block's loads in flight while the current one is
 * consumed) and leaves ONE partial slab for them.
 * lds_class (a REGISTER class since the register-resident kernel of round 4; the name is the ABI's): 1 = the block's rows of
 * the narrower operand fit 6 resident (row step, column group) pairs per wave (rows x columns <= 64 x 320 or 32 x 640) — the
 * kernel that runs three workgroups per CU; 2 = up to 10 pairs (64 x 640, 32 x 1280), two per CU.  One launch per class; a
 * table planned for class 2 may hold class-1 sites (they then run the two-per-CU kernel).
 * Shapes: N, K multiples of 32, rank <= 16; anything else: supported = 0, use the _self_ragged pass. */
typedef struct lora_amd_factors_mfma_plan_t {
  int32_t supported, lds_class, rank_tile, rows_per_block, nparts, lds_bytes;
  int32_t blocks_per_wg, reserved;          /* row blocks one workgroup walks: nparts = ceil(ceil(M / rows) / blocks_per_wg) */
  int64_t up_part_floats, down_part_floats;
  int64_t pack_up_elems, pack_down_elems;   /* elements (activation dtype) of the two fragment packs of a site */
} lora_amd_factors_mfma_plan_t;
/* rows: 0 = the planner's choice (64, else 32), else 32 / 64 tried first.  flags: reserved (0). */
int lora_amd_factors_mfma_plan(int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype, int32_t rows, int32_t flags,
                               lora_amd_factors_mfma_plan_t *out);
/* f32 masters -> fragment packs: pk[split][c/8][RT][8] in the activation dtype (RT = the rank tile 4 / 8 / 16 of r; ABI 7 — it was
 * [16] whatever the rank: a rank-4 pack is a quarter of the bytes now), split 0 = rounded value, split 1 = the remainder, ranks
 * r .. RT - 1 zero; pk_down from down [r, K], pk_up from up [N, r]; sizes from lora_amd_factors_mfma_plan (pack_*_elems =
 * 2 C RT + 8).  `begin` is filled by the plan (host). */
typedef struct lora_amd_pack_site {
  const float *down, *up;
  void *pk_down, *pk_up;
  int32_t N, K, r, reserved;
  int64_t begin;
} lora_amd_pack_site;
int lora_amd_factor_pack_plan(lora_amd_pack_site *sites, int32_t n, int64_t *total);
int lora_amd_factor_pack(const lora_amd_pack_site *sites_dev, int32_t n, int64_t total, int32_t act_dtype, void *stream);
typedef struct lora_amd_fm_site {
  const void *g, *x;              /* [M, N] output gradient, [M, K] input (act_dtype; rows may be head-padded) */
  const void *pk_up, *pk_down;    /* fragment packs of this step's factors (lora_amd_factor_pack) */
  float *up_part, *down_part;     /* [nparts][RT][N], [nparts][RT][K] */
  int64_t ldg, ldx, M;
  int32_t N, K, r;
  float scale;
  int32_t g_head_dim, g_head_pad, x_head_dim, x_head_pad;
  int32_t rows_per_block;         /* caller: lora_amd_factors_mfma_plan's rows_per_block and ... */
  int32_t blocks_per_wg;          /* ... blocks_per_wg for this site */
  /* filled by lora_amd_factors_mfma_ragged_plan */
  int32_t resident_is_x, cw, nchunk;
  int32_t x_head_magic, g_head_magic; /* ceil(2^32 / (head_dim / 8)) as a bit pattern, 0 = dense rows (the kernel's division-free
                                       * column -> padded-column map) */
  int32_t lds_bytes;
  int64_t block_begin;
  /* nn.Dropout(p) on the branch (lora.py:45, 57): p > 0 makes G enter as mask (.) G with the forward's mask regenerated from
   * (seed, offset + *offset_dev) — the caller folds 1 / (1 - p) into `scale`; T = X down^T is unaffected */
  float dropout_p;
  int32_t reserved;
  uint64_t seed, offset;
  const uint64_t *offset_dev;
} lora_amd_fm_site;
int lora_amd_factors_mfma_ragged_plan(lora_amd_fm_site *sites, int32_t n, int32_t act_dtype, int32_t lds_class,
                                      int64_t *grid);
/* masked: 0 = no site of the table has dropout_p > 0; 1 = the kernel that regenerates the dropout mask on G (sites with
 * dropout_p = 0 in such a table run unmasked) */
/* rows_per_block (ABI 6): the block height (32 / 64) every site of the table was planned with — a compile-time constant of
 * the kernel — or 0 for a class-2 table that holds sites of both heights (one launch, the workgroup enters its site's
 * instantiation); a class-1 table has one height (lora_amd_factors_mfma_ragged_plan refuses a mixed one). */
int lora_amd_linear_bwd_factors_mfma_ragged(const lora_amd_fm_site *sites_dev, int32_t n, int64_t grid, int32_t lds_class,
                                            int32_t rows_per_block, int32_t act_dtype, int32_t masked, void *stream);
/* ABI 7: the same launch with the plan's block -> site map on the device (grid int32 entries, filled on the host by
 * lora_amd_factors_mfma_block_map from a planned table): a workgroup finds its site with ONE scalar load instead of copying the
 * table's block prefix to LDS and searching it (1.2 us of a 12.8 us block of the 320-wide sites).  block_map_dev = NULL is the
 * entry above. */
int lora_amd_factors_mfma_block_map(const lora_amd_fm_site *sites /* host, planned */, int32_t n, int64_t grid, int32_t *map);
int lora_amd_linear_bwd_factors_mfma_ragged_mapped(const lora_amd_fm_site *sites_dev, int32_t n, int64_t grid,
                                                   const int32_t *block_map_dev, int32_t lds_class, int32_t rows_per_block,
                                                   int32_t act_dtype, int32_t masked, void *stream);
/* Tuning / test hook: which kernel a table of class 1 (64-row blocks) runs — 0 = the 10-pair kernel every class can take (two
 * workgroups per CU, two column groups in flight per wave), 1 / 2 = the 6-pair kernel at three workgroups per CU with 1 / 2
 * groups in flight, 3 / 4 / 5 = the 6-pair kernel at two per CU with 2 / 3 / 4 groups in flight; < 0 only reads.  Returns the
 * previous value. */
int lora_amd_factors_mfma_set_tuning(int32_t narrow);


/* ------------------------------------------------------------------------
 * K4  LoraInjectedConv2d low-rank branch (lora.py:94-135) and its autograd, NCHW.
 *     T  = conv_kxk(X; down)        [B, r, H, W]   lora.py:131 self.lora_down(input) [+ selector]
 *     Y += scale * mask * conv_1x1(T; up)          lora.py:131-134 lora_up, dropout, *scale, +
 *     Gt = scale * up^T (mask*G)    [B, r, H, W] ; dUp = scale * sum_{b,p} (mask*G) T
 *     dDown = sum_{b,p} Gt' (x) patches(X) ;  dX += conv_transpose(Gt'; down)   (Gt' = S^T Gt)
 * Native geometry (lora_amd_conv_plan says whether a site qualifies): stride 1, dilation 1, groups 1,
 * kernel 1x1 (padding 0) or 3x3 (padding 1), contiguous NCHW, H*W % 8 == 0 (3x3: W % 8 == 0, W <= 512),
 * rank <= 16.  Every ResnetBlock2D conv of SD1.5 at 512^2 / 768^2 qualifies except the 12x12 maps.
 * Pixels are processed in 16-byte chunks of 8; a wave owns `cpw` consecutive chunks (whole image rows),
 * the 4 waves of a workgroup split the channel loop, `split_*` workgroups split it further.
 * ---------------------------------------------------------------------- */
typedef struct lora_amd_conv_plan_t {
  int32_t native;      /* 1: the entry points below accept this geometry */
  int32_t cpw_in;      /* chunks per wave of the passes over X / dX (whole rows for 3x3) */
  int32_t ngroups_in;  /* wave-groups of those passes  = number of dDown partials */
  int32_t ngroups_out; /* wave-groups of the passes over Y / G (64 chunks each) = number of dUp partials */
  int32_t split_in;    /* channel splits of conv_down_fwd (partial T sums) */
  int32_t split_out;   /* channel splits of conv_bwd_g    (partial Gt sums) */
  int32_t rank_pad;    /* r rounded up to a multiple of 4: row stride of the dUp/dDown partials */
  int32_t reserved;
  int64_t t_part_floats;    /* split_in  * B * r * H*W   workspace of conv_down_fwd */
  int64_t gt_part_floats;   /* split_out * B * r * H*W   workspace of conv_bwd_g    */
  int64_t up_part_floats;   /* ngroups_out * rank_pad * C_out            */
  int64_t down_part_floats; /* ngroups_in  * rank_pad * C_in * ks * ks   */
} lora_amd_conv_plan_t;

int lora_amd_conv_plan(int32_t B, int32_t C_in, int32_t C_out, int32_t H, int32_t W, int32_t ks, int32_t r,
                       lora_amd_conv_plan_t *out);

/* t_out[B, r, H, W] (f32) = (conv_kxk(X; down)) with the optional selector S [r,r] applied across the
 * rank dimension (lora.py:131, 140-156).  down: [r, C_in, ks, ks].  t_part: plan.t_part_floats floats. */
int lora_amd_conv_down_fwd(const void *x, const void *down, const float *sel, float *t_part, float *t_out,
                           int32_t B, int32_t C_in, int32_t H, int32_t W, int32_t ks, int32_t r,
                           int32_t act_dtype, int32_t factor_dtype, void *stream);

/* Y[B, C_out, H, W] (in place; holds the frozen conv's output) += scale * mask * conv_1x1(T; up),
 * up: [C_out, r].  Dropout as lora_amd_rank_update, element index = NCHW offset of Y. */
int lora_amd_conv_up_fwd(void *y, const float *t, const void *up, int32_t B, int32_t C_out, int32_t H, int32_t W,
                         int32_t r, int32_t act_dtype, int32_t factor_dtype, float scale, float dropout_p,
                         uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream);

/* One pass over G[B, C_out, H, W]: gt_out[B, r, H, W] (f32) = S^T (scale * up^T (mask*G)) and
 * up_part[ngroups_out][rank_pad][C_out] = scale * sum over the group's pixels of (mask*G) T. */
int lora_amd_conv_bwd_g(const void *g, const float *t, const void *up, const float *sel, float *gt_part,
                        float *gt_out, float *up_part, int32_t B, int32_t C_out, int32_t H, int32_t W, int32_t r,
                        int32_t act_dtype, int32_t factor_dtype, float scale, float dropout_p, uint64_t seed,
                        uint64_t offset, const uint64_t *offset_dev, void *stream);

/* One pass over X[B, C_in, H, W] (and dX in place, holding the frozen conv's input gradient; may be NULL):
 * down_part[ngroups_in][rank_pad][C_in*ks*ks] = sum over the group's pixels of Gt' (x) patches(X), and
 * dX += conv_transpose(Gt'; down). */
int lora_amd_conv_bwd_x(const void *x, void *dx, const float *gt, const void *down, float *down_part, int32_t B,
                        int32_t C_in, int32_t H, int32_t W, int32_t ks, int32_t r, int32_t act_dtype,
                        int32_t factor_dtype, void *stream);

/* ------------------------------------------------------------------------
 * K4, channels-last form (csrc/conv_nhwc.hip): the 3x3 down-projection of LoraInjectedConv2d (lora.py:131
 * `self.lora_down(input)`, Conv2d in -> r, kernel 3, padding 1, stride 1) and its two gradients on the matrix cores,
 * for activations stored [B, H, W, C] (torch.channels_last).  In that layout everything 1x1 in the adapter — the
 * up-projection (lora.py:131-134), its gradient pass over G, a 1x1 down-projection — is the Linear adapter on the
 * [B*H*W, C] matrix and uses lora_amd_linear_fwd / lora_amd_rank_update / lora_amd_linear_bwd_g unchanged, so
 * T, Gt are [B*H*W, r] f32 row-major here (not [B, r, H, W]).
 *     T[p, j]         = sum_{tap, c} down[j, c, tap] X[p + tap, c]
 *     dX[p, c]       += sum_{tap, j} down[j, c, tap] Gt[p - tap, j]
 *     dDown[j, c, tap] = sum_p Gt[p, j] X[p + tap, c]         (per-split partials, folded by lora_amd_reduce_batched)
 * Native geometry: C_in % 64 == 0, rank in {4, 8, 12, 16}, bf16 / f16 activations, B*H*W*C_in < 2^31, W <= 128;
 * any H (pixel tiles are 16 columns x pt rows with masked edges, so the 12x12 maps qualify too).
 * `down` is [r, C_in, 3, 3] f32; lora_amd_conv3_nhwc_pack rewrites it per call into MFMA fragment order in the
 * activation dtype: pf for the forward (at rank <= 8 the spare fragment rows carry the low 16-bit parts of the f32
 * values, T is then as precise as with f32 factors; at rank 12 / 16 the factor is rounded to the activation dtype,
 * which is what the reference's autocast does to lora_down.weight), pd for the input gradient.
 * ---------------------------------------------------------------------- */
typedef struct lora_amd_conv3_nhwc_plan_t {
  int32_t native;   /* 1: the entry points below accept this geometry */
  int32_t pt;       /* image rows of a forward pixel tile (tile = 16 columns x pt rows) */
  int32_t ksplit;   /* forward: channel shares per pixel tile (small maps); > 1 = that many T partials in t_part */
  int32_t csplit;   /* input gradient: shares of the 64-channel blocks per pixel tile */
  int32_t ks;       /* 32-slot steps of the (tap, rank) contraction = ceil(9 r / 32) */
  int32_t pr;       /* factor gradient: image rows per LDS-staged strip */
  int32_t nsplit;   /* factor gradient: workgroups per 64-channel chunk = number of dDown partials */
  int32_t rank_pad; /* rows of one dDown partial (r rounded up to 4 / 8 / 16) */
  int32_t fwd_tiles; /* forward pixel tiles = arrival counters (uint32, zeroed once) of lora_amd_conv3_nhwc_fwd_fused */
  int32_t reserved;
  int64_t pf_elems, pd_elems; /* packed factor sizes in activation-dtype elements */
  int64_t t_part_floats;      /* ksplit > 1: ksplit * B*H*W * r, else 0 */
  int64_t down_part_floats;   /* nsplit * rank_pad * C_in * 9 */
} lora_amd_conv3_nhwc_plan_t;

int lora_amd_conv3_nhwc_plan(int32_t B, int32_t C_in, int32_t H, int32_t W, int32_t r,
                             lora_amd_conv3_nhwc_plan_t *out);
int lora_amd_conv3_nhwc_pack(const float *down, int32_t r, int32_t C_in, int32_t act_dtype, void *pf, void *pd,
                             void *stream);
/* t_out [B*H*W, r] f32 = conv3x3(X; down), X [B, H, W, C_in] contiguous.  t_part: plan.t_part_floats floats
 * (may be NULL when plan.ksplit == 1). */
int lora_amd_conv3_nhwc_down_fwd(const void *x, const void *pf, float *t_part, float *t_out, int32_t B, int32_t C_in,
                                 int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream);
/* dX [B, H, W, C_in] (in place; holds the frozen conv's input gradient) += conv_transpose3x3(Gt; down). */
int lora_amd_conv3_nhwc_bwd_dx(void *dx, const float *gt, const void *pd, int32_t B, int32_t C_in, int32_t H,
                               int32_t W, int32_t r, int32_t act_dtype, void *stream);
/* down_part [nsplit][rank_pad][C_in * 9] f32: per-workgroup partial sums of dDown (rows >= r are not written). */
int lora_amd_conv3_nhwc_bwd_down(const void *x, const float *gt, float *down_part, int32_t B, int32_t C_in,
                                 int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream);
/* Round 6 — launch fusion of the 3x3 site.
 * (1) The fragment packs of EVERY conv site in ONE launch per optimiser step (the factors change once per step, not per
 *     forward): pf / pd as lora_amd_conv3_nhwc_pack writes them plus pu, `up` [C_out, r] as the matrix-core operand of the
 *     up-projection (C_out * 32 elements: per 32-column group two interleaved tiles of (hi | lo) fragments).  The caller
 *     fills the first block of every site, lora_amd_conv3_nhwc_pack_plan (host) KS and `begin`, and returns the piece total. */
typedef struct lora_amd_conv3_pack_site {
  const float *down, *up;   /* f32 [r, C_in, 3, 3], [C_out, r] */
  void *pf, *pd, *pu;       /* plan.pf_elems, plan.pd_elems, C_out * 32 elements of the activation dtype */
  int32_t r, C_in, C_out, KS;
  int64_t begin;
} lora_amd_conv3_pack_site;
int lora_amd_conv3_nhwc_pack_plan(lora_amd_conv3_pack_site *sites, int32_t n, int64_t *total);
int lora_amd_conv3_nhwc_pack_batched(const lora_amd_conv3_pack_site *sites_dev, int32_t n, int64_t total, int32_t act_dtype,
                                     void *stream);
/* (2) lora.py:130-135's low-rank branch in ONE launch: T = conv3x3(X; down) -> t_out [B*H*W, r] f32 (saved for the backward)
 *     and Y += scale * mask o (T up^T) in place on the frozen convolution's output Y [B, H, W, C_out]; dropout as
 *     lora_amd_rank_update draws it (chunk = 8 consecutive columns of a row of [B*H*W, C_out]).  Geometries with
 *     plan.ksplit > 1 (small maps: the channel loop is split over several workgroups per pixel tile) need t_part
 *     (plan.t_part_floats) and `counters` (plan.fwd_tiles uint32, zeroed ONCE: the last-arriving workgroup of a tile folds
 *     the shares, runs the up-projection and resets its counter).  bf16 activations, C_out % 32 == 0. */
int lora_amd_conv3_nhwc_fwd_fused(const void *x, const void *pf, const void *pu, void *y, float *t_out, float *t_part,
                                  uint32_t *counters, int32_t B, int32_t C_in, int32_t C_out, int32_t H, int32_t W,
                                  int32_t r, int32_t act_dtype, float scale, float dropout_p, uint64_t seed, uint64_t offset,
                                  const uint64_t *offset_dev, void *stream);
/* out[n] = sum_p part[p * stride + i]: folds the column-tile partials of lora_amd_linear_bwd_g into one Gt. */
int lora_amd_sum_parts(const float *part, int32_t nparts, int64_t stride, float *out, int64_t n, void *stream);

/* ------------------------------------------------------------------------
 * C2/K6  flat-buffer gradient clipping + AdamW.
 * replaces: train_lora_dreambooth.py:878-888 (clip_grad_norm_ over every UNet
 *           parameter, AdamW.step, zero_grad) for the LoRA parameters, which
 *           the trainer keeps in ONE flat f32 buffer (also the RCCL all-reduce
 *           payload).
 * ---------------------------------------------------------------------- */

/* out_sumsq[0] (f32) = sum(g[i]^2).  workspace >= lora_amd_sumsq_workspace(n). */
size_t lora_amd_sumsq_workspace(int64_t n);
int lora_amd_sumsq(const float *g, int64_t n, float *out_sumsq, void *workspace,
                   size_t workspace_bytes, void *stream);

typedef struct lora_amd_adamw_group {
  int64_t begin, end; /* element range [begin, end) of the flat buffer */
  float lr;
  float weight_decay;
} lora_amd_adamw_group;

/* One fused pass over the flat buffers.  clip coefficient is computed on the
 * device from sumsq[0]: c = min(1, max_norm / (sqrt(sumsq) + 1e-6))
 * (torch.nn.utils.clip_grad_norm_); max_norm <= 0 disables clipping.
 * grad_scale multiplies g before clipping (1/world_size after a SUM all-reduce).
 * `step` is the 1-based optimiser step (bias correction).  zero_grad != 0
 * writes zeros back to g (optimizer.zero_grad, train_lora_dreambooth.py:888). */
int lora_amd_clip_adamw(float *p, float *g, float *exp_avg, float *exp_avg_sq,
                        int64_t n, const lora_amd_adamw_group *groups_dev,
                        int32_t n_groups, const float *sumsq, float grad_scale,
                        float max_norm, float beta1, float beta2, float eps,
                        int64_t step, int32_t zero_grad, void *stream);

/* hipGraph-replayable form: the 1-based step is read from device memory, so a captured
 * launch is identical from step to step; lora_amd_step_advance does step_dev[0] += 1.
 * `scaler` (NULL = off): the 4-float loss-scaling state lora_amd_loss_scale_update maintains; the gradient is
 * additionally multiplied by scaler[2] and the whole update is skipped (grads still zeroed) when scaler[3] == 0. */
int lora_amd_clip_adamw_dev(float *p, float *g, float *exp_avg, float *exp_avg_sq,
                            int64_t n, const lora_amd_adamw_group *groups_dev,
                            int32_t n_groups, const float *sumsq, float grad_scale,
                            float max_norm, float beta1, float beta2, float eps,
                            const int64_t *step_dev, const float *scaler, int32_t zero_grad, void *stream);
int lora_amd_step_advance(int64_t *step_dev, void *stream);

/* Dynamic fp16 loss scaling on the device (the GradScaler accelerate wraps around the reference's step when
 * mixed_precision="fp16", train_lora_dreambooth.py:489-494).  state = {scale for the next backward, finite-step
 * count, 1/scale of the step being applied, finite flag}.  Call after lora_amd_sumsq of the scaled (all-reduced)
 * gradient and before lora_amd_clip_adamw_dev: inf/nan -> scale *= backoff_factor and the update is skipped;
 * growth_interval finite steps in a row -> scale *= growth_factor.  step_dev (may be NULL) advances only on
 * applied steps. */
int lora_amd_loss_scale_update(float *state, const float *sumsq, int64_t *step_dev, float growth_factor,
                               float backoff_factor, int32_t growth_interval, void *stream);

/* Textual-inversion step of pivotal tuning over the placeholder rows ONLY (replaces cli_lora_pti.py:433-479: AdamW
 * over the whole [vocab, hidden] embedding table, norm decay, restoring every other row).  One launch:
 * for token t = ids_dev[i]: g = grad_scale * table_grad[t]; AdamW(lr, betas, eps, decoupled weight_decay, 1-based
 * step) on the f32 master row rows[i] with moments exp_avg[i] / exp_avg_sq[i]; if decay_lambda >= 0 the row is rescaled
 * to norm + decay_lambda * (target_norm - norm) (the reference: lambda = min(1, 100 lr), target 0.4); table[t] = row. */
int lora_amd_ti_rows_step(void *table, const void *table_grad, const int64_t *ids_dev, int32_t n_tokens,
                          int32_t hidden, int32_t table_dtype, float *rows, float *exp_avg, float *exp_avg_sq,
                          float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                          int64_t step, float decay_lambda, float target_norm, void *stream);

/* ------------------------------------------------------------------------
 * Frozen host-model fusions of the training step (train_lora_dreambooth.py:838-892 calls unet(...)): the
 * normalisation / activation passes between the adapted sites.  Not part of the reference's LoRA interface —
 * they replace ATen sequences of the UNet the reference trains through (ResnetBlock2D norm+SiLU, GEGLU gate).
 * Affine parameters are frozen: backward produces the input gradient only.
 * ---------------------------------------------------------------------- */

/* GroupNorm over NCHW-contiguous x [B, C, HW] (HW % 8 == 0, C % groups == 0), y = act(gn(x) * gamma + beta) with
 * act = SiLU when `act` != 0.  gamma / beta have the activation dtype.  stats [B*groups][2] f32 receives
 * (mean, rstd) for the backward.  Two launches; workspace >= lora_amd_groupnorm_workspace bytes (0 = unsupported). */
size_t lora_amd_groupnorm_workspace(int32_t B, int32_t C, int32_t HW, int32_t groups);
int lora_amd_groupnorm_supported(int32_t B, int32_t C, int32_t HW, int32_t groups);
int lora_amd_groupnorm_fwd(const void *x, const void *gamma, const void *beta, void *y, float *stats,
                           void *workspace, size_t workspace_bytes, int32_t B, int32_t C, int32_t HW,
                           int32_t groups, float eps, int32_t act, int32_t dtype, void *stream);
/* dx = d loss / d x given gout = d loss / d y; x, stats as in the forward. */
int lora_amd_groupnorm_bwd(const void *x, const void *gout, const void *gamma, const void *beta,
                           const float *stats, void *dx, void *workspace, size_t workspace_bytes, int32_t B,
                           int32_t C, int32_t HW, int32_t groups, int32_t act, int32_t dtype, void *stream);

/* The same for channels_last activations (memory [B][HW][C], C % 8 == 0): aff [B][4][C] f32 receives the per-channel
 * (gamma*rstd, beta - mean*gamma*rstd, mean, rstd) the backward needs.  Three launches each way.
 * `addend` [B][C] f32 (or NULL) is added to x before the normalisation (time-embedding projection + the bias of the
 * producing convolution in ResnetBlock2D) at no streaming cost; it is folded into aff, the backward needs no change
 * (the gradient w.r.t. the addend is the per-(sample, channel) sum of dx). */
size_t lora_amd_groupnorm_nhwc_workspace(int32_t B, int32_t C, int32_t HW, int32_t groups);
int lora_amd_groupnorm_nhwc_fwd(const void *x, const void *gamma, const void *beta, const float *addend, void *y,
                                float *aff, void *workspace, size_t workspace_bytes, int32_t B, int32_t C, int32_t HW,
                                int32_t groups, float eps, int32_t act, int32_t dtype, void *stream);
int lora_amd_groupnorm_nhwc_bwd(const void *x, const void *gout, const void *gamma, const float *aff, void *dx,
                                void *workspace, size_t workspace_bytes, int32_t B, int32_t C, int32_t HW,
                                int32_t groups, int32_t act, int32_t dtype, void *stream);

/* LayerNorm over the last dimension of row-contiguous x [M, K] (K % 8 == 0, K <= 2560; gamma / beta in the activation
 * dtype, frozen).  stats [M][2] f32 = (mean, rstd).  One launch each way. */
int lora_amd_layernorm_supported(int32_t K);
int lora_amd_layernorm_fwd(const void *x, const void *gamma, const void *beta, void *y, float *stats, int64_t M,
                           int32_t K, float eps, int32_t dtype, void *stream);
int lora_amd_layernorm_bwd(const void *x, const void *gout, const void *gamma, const float *stats, void *dx,
                           int64_t M, int32_t K, int32_t dtype, void *stream);
/* The residual add in front of the norm in the same pass: sum_out = x + res (rounded to the activation dtype, what the
 * residual stream carries on), y = layernorm(sum_out).  Backward: dx = d layernorm / d sum (given gout, evaluated at
 * x = sum_out) + gsum, the gradient arriving through the residual stream; it is the gradient of both addends. */
int lora_amd_add_layernorm_fwd(const void *x, const void *res, const void *gamma, const void *beta, void *sum_out,
                               void *y, float *stats, int64_t M, int32_t K, float eps, int32_t dtype, void *stream);
int lora_amd_add_layernorm_bwd(const void *x, const void *gout, const void *gsum, const void *gamma,
                               const float *stats, void *dx, int64_t M, int32_t K, int32_t dtype, void *stream);

/* GEGLU gate behind the adapted projection: y [M, 2*inner] = [h | gate]; out [M, inner] = h * gelu(gate) (erf form).
 * Backward writes gy [M, 2*inner] = [gout * gelu(gate) | gout * h * gelu'(gate)] in one pass (no cat). */
int lora_amd_geglu_fwd(const void *y, int64_t ldy, void *out, int64_t ldo, int64_t M, int32_t inner, int32_t dtype,
                       void *stream);
int lora_amd_geglu_bwd(const void *y, int64_t ldy, const void *gout, int64_t ldg, void *gy, int64_t ldgy, int64_t M,
                       int32_t inner, int32_t dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LORA_AMD_H */
