// K1/K2 — the low-rank branch of LoraInjectedLinear (lora_diffusion/lora.py:53-58)
// and of its autograd, as three HBM-streaming primitives:
//
//   rowdot       T[M,r]  = scale * X[M,K] @ F^T          (lora_down; dT in backward)
//   rank_update  Y[M,N] += scale * mask * T[M,r] @ F      (lora_up + dropout + *scale + add)
//   colreduce    D[r,K]  = beta*D + scale * T^T @ (mask*X) (dUp, dDown)
//
// The reference issues mm, mm, dropout, mul, add (5 launches, 3 extra [M,N]
// round trips) forward and ~10 launches backward.  Here every activation byte is
// touched once per primitive: each lane moves 16-byte chunks of 8 elements, the
// small factor is staged in LDS as f32 (bank-conflict-free two-plane layout), the
// per-row T vector sits in LDS, reductions over K are wave shuffles, reductions
// over M are per-lane accumulators + an LDS tree + a second tiny kernel.
// All of it is HBM-bound (AI ~ r flop/B): no MFMA here by design.
#include <algorithm>

#include "common.hpp"
#include "rank16_mfma.hpp"

namespace lora_amd {

constexpr int kThreads = 256;
constexpr int kFactorLdsFloats = 8192;  // 32 KiB  [RT][cols]
constexpr int kTLdsFloats = 2048;       // 8 KiB   [rows][RT]

// Batched launches (grid.y = matrix index): element strides between consecutive matrices of a stack.  The SVD
// distillation (cli_svd.py) runs the same pass over every same-shape site of a model in one launch.
struct BatchStride {
  int64_t x, f_bytes, t, d, partial;
};

__device__ inline float ld_factor(const void *p, int dt, int64_t i) {
  if (dt == LORA_AMD_F32) return gl(reinterpret_cast<const float *>(p))[i];
  if (dt == LORA_AMD_F16) return (float)gl(reinterpret_cast<const _Float16 *>(p))[i];
  return (float)gl(reinterpret_cast<const __bf16 *>(p))[i];
}

// Stage factor[:, c0:c0+ncols] (logical [r, C]; stored [r,C] or [C,r]) into LDS as
// [RT][2][ncols/8][4] (vec) so that a lane's 8 columns are two 16-byte slots whose
// addresses advance 16 B per lane (conflict-free ds_read_b128).  Ranks >= r are zero.
template <int RT>
__device__ __forceinline__ void stage_factor_vec(float *s_f, const void *f, int fdt, int layout, int r,
                                        int64_t C, int c0, int ncols) {
  const int c8 = ncols >> 3;
  for (int i = threadIdx.x; i < RT * ncols; i += kThreads) {
    int j, c;
    if (layout == LORA_AMD_FACTOR_RK) { j = i / ncols; c = i - j * ncols; }
    else { c = i / RT; j = i - c * RT; }  // coalesced along r for [C, r]
    float v = 0.f;
    if (j < r) v = ld_factor(f, fdt, layout == LORA_AMD_FACTOR_RK ? (int64_t)j * C + c0 + c
                                                                   : (int64_t)(c0 + c) * r + j);
    s_f[((j * 2 + ((c >> 2) & 1)) * c8 + (c >> 3)) * 4 + (c & 3)] = v;
  }
}

template <int RT>
__device__ __forceinline__ void fma_chunk(const float *s_f, int c8, int cc, const float (&x)[8], float (&acc)[RT]) {
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + cc) * 4]);
    const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + cc) * 4]);
    float a = acc[j];
    a = fmaf(x[0], d0.x, a); a = fmaf(x[1], d0.y, a); a = fmaf(x[2], d0.z, a); a = fmaf(x[3], d0.w, a);
    a = fmaf(x[4], d1.x, a); a = fmaf(x[5], d1.y, a); a = fmaf(x[6], d1.z, a); a = fmaf(x[7], d1.w, a);
    acc[j] = a;
  }
}

// ---------------------------------------------------------------------------
// rowdot: L lanes cooperate on one row (L = 2^logL chosen on the host so that
// K/8 chunks divide evenly), a wave covers 64/L rows at a time, the K-reduction
// finishes with logL xor-shuffles.  Lane 0 of each group applies scale/selector
// and writes the r outputs.
// ---------------------------------------------------------------------------
// `bx` = row block of the matrix; the pointers already address the matrix (batched and ragged launches offset them).
template <class EX, int RT, bool MASKED>
__device__ __forceinline__ void rowdot_body(
    const typename EX::storage *__restrict__ x, int64_t ldx, const void *__restrict__ f, int fdt,
    int layout, float *__restrict__ t_out, int64_t M, int K, int r, int kt_cols, int logL,
    int rows_per_block, float scale, const float *__restrict__ sel, int sel_transposed, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev, int64_t bx) {
  __shared__ __attribute__((aligned(16))) float s_f[kFactorLdsFloats];
  __shared__ float s_sel[RT * RT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = 1 << logL, G = 64 >> logL;  // lanes per row, rows per wave-iteration
  const int l = lane & (L - 1), g = lane >> logL;
  const int64_t m0 = bx * rows_per_block;
  const int64_t m1 = min(M, m0 + rows_per_block);
  const int rows_iter = G * (kThreads / 64);
  const int niter = (int)((m1 - m0 + rows_iter - 1) / rows_iter);

  if (sel != nullptr) {
    for (int i = tid; i < RT * RT; i += kThreads) {
      int a = i / RT, b = i - a * RT;
      s_sel[i] = (a < r && b < r) ? sel[a * r + b] : 0.f;
    }
  }

  const bool single_tile = kt_cols >= K;
  if (single_tile) {
    stage_factor_vec<RT>(s_f, f, fdt, layout, r, K, 0, K);
    __syncthreads();
  }

  constexpr int U = 4;
  for (int it = 0; it < niter; ++it) {
    const int64_t row = m0 + (int64_t)it * rows_iter + wave * G + g;
    const bool live = row < m1;
    float acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.f;

    for (int k0 = 0; k0 < K; k0 += kt_cols) {
      const int ncols = min(kt_cols, K - k0);
      const int c8 = ncols >> 3;
      if (!single_tile) {
        __syncthreads();
        stage_factor_vec<RT>(s_f, f, fdt, layout, r, K, k0, ncols);
        __syncthreads();
      }
      if (live) {
        const typename EX::storage *xr = x + row * ldx + k0;
        for (int cb = l; cb < c8; cb += L * U) {
          float xv[U][8];
#pragma unroll
          for (int u = 0; u < U; ++u) {  // clamped address + select: all U loads in flight (common.hpp)
            const int cc = cb + u * L;
            load8_sel<EX>(xr + (cc < c8 ? cc : cb) * 8, cc < c8, xv[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int cc = cb + u * L;
            if (cc >= c8) continue;
            if (MASKED) {
              float mk[8];
              dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)((row * K + k0) >> 3) + cc, p, mk);
#pragma unroll
              for (int i = 0; i < 8; ++i) xv[u][i] *= mk[i];
            }
            fma_chunk<RT>(s_f, c8, cc, xv[u], acc);
          }
        }
      }
    }
    // reduce the L partial sums of each row
    for (int off = L >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int j = 0; j < RT; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    }
    if (live && l == 0) {
      float o[RT];
      if (sel != nullptr) {
#pragma unroll
        for (int a = 0; a < RT; ++a) {
          float v = 0.f;
#pragma unroll
          for (int b = 0; b < RT; ++b)
            v = fmaf(acc[b], sel_transposed ? s_sel[b * RT + a] : s_sel[a * RT + b], v);
          o[a] = v * scale;
        }
      } else {
#pragma unroll
        for (int j = 0; j < RT; ++j) o[j] = acc[j] * scale;
      }
      float LORA_AMD_AS_GLOBAL *tr = gl(t_out) + row * r;
#pragma unroll
      for (int j = 0; j < RT; ++j)
        if (j < r) tr[j] = o[j];
    }
  }
}

template <class EX, int RT, bool MASKED>
__global__ __launch_bounds__(kThreads) void rowdot_kernel(
    const typename EX::storage *__restrict__ x, int64_t ldx, const void *__restrict__ f, int fdt,
    int layout, float *__restrict__ t_out, int64_t M, int K, int r, int kt_cols, int logL,
    int rows_per_block, float scale, const float *__restrict__ sel, int sel_transposed, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev, BatchStride bs) {
  rowdot_body<EX, RT, MASKED>(x + blockIdx.y * bs.x, ldx, reinterpret_cast<const char *>(f) + blockIdx.y * bs.f_bytes, fdt,
                              layout, t_out + blockIdx.y * bs.t, M, K, r, kt_cols, logL, rows_per_block, scale, sel,
                              sel_transposed, p, seed, offset, offset_dev, blockIdx.x);
}

// ---- ragged launches (cli_svd.py: every shape group of a model in ONE launch) -----------------------------------
// A table of jobs, one per stack of same-shape matrices; blocks are numbered through the table (begin1 / begin2 are the
// running block counts of the two launch kinds, filled by lora_amd_ragged_plan).  Uniform per workgroup: the lookup is
// a binary search on scalar loads.
__device__ __forceinline__ int ragged_find(const lora_amd_ragged_desc *__restrict__ d, int n, int64_t b, bool second) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((second ? d[mid].begin2 : d[mid].begin1) <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <int RT>
__global__ __launch_bounds__(kThreads) void rowdot_ragged_kernel(const lora_amd_ragged_desc *__restrict__ descs, int n,
                                                                 int r, int layout, float scale) {
  const int g = ragged_find(descs, n, blockIdx.x, false);
  const lora_amd_ragged_desc d = descs[g];
  const int64_t local = (int64_t)blockIdx.x - d.begin1;
  const int64_t by = local / d.blocks1, bx = local - by * d.blocks1;
  rowdot_body<f32_t, RT, false>(reinterpret_cast<const float *>(d.x) + by * d.stride_x, d.ldx,
                                reinterpret_cast<const float *>(d.f) + by * d.stride_f, LORA_AMD_F32, layout,
                                d.out + by * d.stride_out, d.M, d.K, r, d.kt_cols, d.logL, d.rows_per_block, scale,
                                nullptr, 0, 0.f, 0, 0, nullptr, bx);
}

// Any K / any alignment: one wave per row, scalar lanes.
template <class EX, bool MASKED>
__global__ __launch_bounds__(kThreads) void rowdot_generic_kernel(
    const typename EX::storage *__restrict__ x, int64_t ldx, const void *__restrict__ f, int fdt,
    int layout, float *__restrict__ t_out, int64_t M, int K, int r, float scale,
    const float *__restrict__ sel, int sel_transposed, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  __shared__ float s_o[kThreads / 64][LORA_AMD_MAX_RANK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * (kThreads / 64) + wave;
  const bool live = row < M;
  for (int j = 0; j < r; ++j) {
    float a = 0.f;
    for (int k = lane; live && k < K; k += 64) {
      float xv = EX::to_f(x[row * ldx + k]);
      if (MASKED) {
        uint64_t e = (uint64_t)row * K + k;
        float mk[8];
        dropout_mult8(seed, dropout_offset(offset, offset_dev), e >> 3, p, mk);
        xv *= mk[e & 7];
      }
      float fv = ld_factor(f, fdt, layout == LORA_AMD_FACTOR_RK ? (int64_t)j * K + k : (int64_t)k * r + j);
      a = fmaf(xv, fv, a);
    }
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) s_o[wave][j] = a;
  }
  __syncthreads();
  if (live && lane < r) {
    float v;
    if (sel != nullptr) {
      v = 0.f;
      for (int b = 0; b < r; ++b) v = fmaf(s_o[wave][b], sel_transposed ? sel[b * r + lane] : sel[lane * r + b], v);
    } else {
      v = s_o[wave][lane];
    }
    t_out[row * r + lane] = v * scale;
  }
}

// ---------------------------------------------------------------------------
// rank_update: tile = rows_per_tile x cols_per_tile of Y; the factor slab and the
// tile's T rows live in LDS; lanes walk the tile's 16-byte chunks in flat order
// (perfectly coalesced when ldy == N), 4 chunks in flight per lane.
// ---------------------------------------------------------------------------
template <class EY, int RT, bool DROP>
__global__ __launch_bounds__(kThreads) void rank_update_kernel(
    typename EY::storage *__restrict__ y, int64_t ldy, const float *__restrict__ t,
    const void *__restrict__ f, int fdt, int layout, int64_t M, int N, int r, int rows_per_tile,
    int cols_per_tile, int tiles_n, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  __shared__ __attribute__((aligned(16))) float s_f[kFactorLdsFloats];
  __shared__ __attribute__((aligned(16))) float s_t[kTLdsFloats];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  const int64_t tr = tile / tiles_n;
  const int tc = (int)(tile - tr * tiles_n);
  const int64_t row0 = tr * rows_per_tile;
  const int col0 = tc * cols_per_tile;
  const int nrows = (int)min((int64_t)rows_per_tile, M - row0);
  const int ncols = min(cols_per_tile, N - col0);
  const int c8 = ncols >> 3;

  stage_factor_vec<RT>(s_f, f, fdt, layout, r, N, col0, ncols);
  for (int i = tid; i < nrows * RT; i += kThreads) {
    int rl = i / RT, j = i - rl * RT;
    s_t[i] = j < r ? t[(row0 + rl) * r + j] : 0.f;
  }
  __syncthreads();

  constexpr int U = 4;
  const int nchunk = nrows * c8;
  const int dq = kThreads / c8, dr = kThreads % c8;
  int rl = tid / c8, cc = tid % c8;
  for (int c = tid; c < nchunk; c += kThreads * U) {
    float v[U][8];
    int rls[U], ccs[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = (c + u * kThreads) < nchunk;
      rls[u] = rl; ccs[u] = cc;
      load8_sel<EY>(y + (row0 + (ok[u] ? rl : rls[0])) * ldy + col0 + (ok[u] ? cc : ccs[0]) * 8, ok[u], v[u]);
      rl += dq; cc += dr;
      if (cc >= c8) { cc -= c8; ++rl; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float pr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const float *tr_ = s_t + rls[u] * RT;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float tj = tr_[j];
        const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + ccs[u]) * 4]);
        const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + ccs[u]) * 4]);
        pr[0] = fmaf(tj, d0.x, pr[0]); pr[1] = fmaf(tj, d0.y, pr[1]);
        pr[2] = fmaf(tj, d0.z, pr[2]); pr[3] = fmaf(tj, d0.w, pr[3]);
        pr[4] = fmaf(tj, d1.x, pr[4]); pr[5] = fmaf(tj, d1.y, pr[5]);
        pr[6] = fmaf(tj, d1.z, pr[6]); pr[7] = fmaf(tj, d1.w, pr[7]);
      }
      if (DROP) {
        float mk[8];
        const int64_t e = (row0 + rls[u]) * (int64_t)N + col0 + ccs[u] * 8;
        dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
        for (int i = 0; i < 8; ++i) pr[i] *= mk[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[u][i] = fmaf(scale, pr[i], v[u][i]);
      store8<EY>(y + (row0 + rls[u]) * ldy + col0 + ccs[u] * 8, v[u]);
    }
  }
}

template <class EY, bool DROP>
__global__ __launch_bounds__(kThreads) void rank_update_generic_kernel(
    typename EY::storage *__restrict__ y, int64_t ldy, const float *__restrict__ t,
    const void *__restrict__ f, int fdt, int layout, int64_t M, int N, int r, float scale, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  const int64_t total = M * (int64_t)N;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int64_t row = e / N;
    const int col = (int)(e - row * N);
    float a = 0.f;
    for (int j = 0; j < r; ++j)
      a = fmaf(t[row * r + j], ld_factor(f, fdt, layout == LORA_AMD_FACTOR_RK ? (int64_t)j * N + col : (int64_t)col * r + j), a);
    if (DROP) {
      float mk[8];
      dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)e >> 3, p, mk);
      a *= mk[e & 7];
    }
    y[row * ldy + col] = EY::from_f(fmaf(scale, a, EY::to_f(y[row * ldy + col])));
  }
}

// ---------------------------------------------------------------------------
// colreduce, stage 1: block = (row block, column tile of <= 64 chunks: one wave
// reads one contiguous 1-2 KB run per row).  Thread t owns chunk column t % c8 and
// row slot t / c8 (>= 4 slots), walks its rows accumulating acc[RT][8]; the row
// slots are then summed through LDS and the block's partial [RT][ncols] goes to
// the workspace.  Stage 2 sums partials over row blocks.
// ---------------------------------------------------------------------------
constexpr int kColRowsPerBlock = 64;
constexpr int kColMaxChunks = 64;

// RB = rows per block: 64 for the single / batched launches (many of them are short stacks of activations), 256 for the
// ragged SVD passes (tall residual stacks: a quarter of the partial slabs and of the slot reductions per byte streamed).
constexpr int kColRowsRagged = 256;

template <class EX, int RT, bool MASKED, int RB = kColRowsPerBlock>
__device__ __forceinline__ void colreduce_stage1_body(
    const typename EX::storage *__restrict__ x, int64_t ldx, const float *__restrict__ t,
    float *__restrict__ partial, int64_t M, int K, int r, int rank0, int col_tiles, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev, int64_t bx) {
  // s_red doubles as the slot-reduction buffer: [slot][c8*8][4 ranks]
  __shared__ __attribute__((aligned(16))) float s_red[kThreads * 8 * 4];
  __shared__ float s_t[RB * RT];
  const int tid = threadIdx.x;
  const int64_t rb = bx / col_tiles;
  const int ct = (int)(bx - rb * col_tiles);
  const int col0 = ct * kColMaxChunks * 8;
  const int ncols = min(kColMaxChunks * 8, K - col0);
  const int c8 = ncols >> 3;
  const int64_t m0 = rb * RB;
  const int nrows = (int)min((int64_t)RB, M - m0);
  const int nslots = kThreads / c8;  // >= 1
  const int slot = tid / c8, cc = tid - slot * c8;

  for (int i = tid; i < nrows * RT; i += kThreads) {
    int rl = i / RT, j = i - rl * RT;
    s_t[i] = (rank0 + j) < r ? t[(m0 + rl) * r + rank0 + j] : 0.f;
  }
  __syncthreads();

  float acc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;

  if (slot < nslots) {
    constexpr int U = 4;
    for (int rb0 = slot; rb0 < nrows; rb0 += nslots * U) {
      float xv[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rl = rb0 + u * nslots;
        load8_sel<EX>(x + (m0 + (rl < nrows ? rl : rb0)) * ldx + col0 + cc * 8, rl < nrows, xv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rl = rb0 + u * nslots;
        if (rl >= nrows) continue;
        if (MASKED) {
          float mk[8];
          const int64_t e = (m0 + rl) * (int64_t)K + col0 + cc * 8;
          dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
          for (int i = 0; i < 8; ++i) xv[u][i] *= mk[i];
        }
        const float *tr_ = s_t + rl * RT;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const float tj = tr_[j];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(tj, xv[u][i], acc[j][i]);
        }
      }
    }
  }

  // slot reduction, 4 ranks at a time: s_red[(slot*ncols + col)*4 + jj]
  float *pout = partial + ((int64_t)rb * RT) * K;  // [rb][RT][K] (ranks rank0..rank0+RT of this pass)
#pragma unroll
  for (int jb = 0; jb < RT; jb += 4) {
    __syncthreads();
    if (slot < nslots) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 v;
        v.x = acc[jb][i];
        v.y = (jb + 1 < RT) ? acc[(jb + 1) % RT][i] : 0.f;
        v.z = (jb + 2 < RT) ? acc[(jb + 2) % RT][i] : 0.f;
        v.w = (jb + 3 < RT) ? acc[(jb + 3) % RT][i] : 0.f;
        *reinterpret_cast<float4 *>(&s_red[((slot * ncols) + cc * 8 + i) * 4]) = v;
      }
    }
    __syncthreads();
    for (int i = tid; i < ncols * 4; i += kThreads) {
      const int col = i >> 2, jj = i & 3;
      if (jb + jj < RT) {
        float sum = 0.f;
        for (int s = 0; s < nslots; ++s) sum += s_red[(s * ncols + col) * 4 + jj];
        gl(pout)[(int64_t)(jb + jj) * K + col0 + col] = sum;
      }
    }
  }
}

template <class EX, int RT, bool MASKED>
__global__ __launch_bounds__(kThreads) void colreduce_stage1_kernel(
    const typename EX::storage *__restrict__ x, int64_t ldx, const float *__restrict__ t,
    float *__restrict__ partial, int64_t M, int K, int r, int rank0, int col_tiles, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev, BatchStride bs) {
  colreduce_stage1_body<EX, RT, MASKED>(x + blockIdx.y * bs.x, ldx, t + blockIdx.y * bs.t, partial + blockIdx.y * bs.partial,
                                        M, K, r, rank0, col_tiles, p, seed, offset, offset_dev, blockIdx.x);
}

template <int RT>
__global__ __launch_bounds__(kThreads) void colreduce_stage1_ragged_kernel(const lora_amd_ragged_desc *__restrict__ descs,
                                                                           int n, int r, int rank0) {
  const int g = ragged_find(descs, n, blockIdx.x, false);
  const lora_amd_ragged_desc d = descs[g];
  const int64_t local = (int64_t)blockIdx.x - d.begin1;
  const int64_t by = local / d.blocks1, bx = local - by * d.blocks1;
  colreduce_stage1_body<f32_t, RT, false, kColRowsRagged>(reinterpret_cast<const float *>(d.x) + by * d.stride_x, d.ldx,
                                          reinterpret_cast<const float *>(d.f) + by * d.stride_f,
                                          d.partial + by * d.stride_partial, d.M, d.K, r, rank0, d.col_tiles, 0.f, 0, 0,
                                          nullptr, bx);
}

// stage 2: D[j,k] = beta*D + scale * sum_b partial[b][j][k]  (j in [rank0, rank0+RT)).
// Block = 64 consecutive (j,k) elements x 4 waves; wave w sums row blocks w, w+4, ... (8 loads in
// flight per lane), the four wave sums meet in LDS.
__device__ __forceinline__ void colreduce_stage2_body(
    const float *__restrict__ partial, float *__restrict__ d, int64_t nblocks, int K, int r,
    int RT, int rank0, int out_layout, float scale, float beta, int64_t bx) {
  __shared__ float s_sum[kThreads];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)RT * K;
  const int64_t i = bx * 64 + lane;
  const int64_t ic = i < total ? i : total - 1;
  const int64_t stride = total;  // floats between consecutive row blocks
  constexpr int U = 8;
  float sum = 0.f;
  for (int64_t b = wave; b < nblocks; b += 4 * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t bb = b + 4 * u;
      const float x = gl(partial)[(bb < nblocks ? bb : b) * stride + ic];
      v[u] = bb < nblocks ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sum += v[u];
  }
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  if (wave == 0 && i < total) {
    sum = (s_sum[lane] + s_sum[64 + lane]) + (s_sum[128 + lane] + s_sum[192 + lane]);
    const int j = (int)(i / K);
    const int k = (int)(i - (int64_t)j * K);
    if (rank0 + j < r) {
      const int64_t o = out_layout == LORA_AMD_FACTOR_RK ? (int64_t)(rank0 + j) * K + k : (int64_t)k * r + rank0 + j;
      d[o] = (beta == 0.f ? 0.f : beta * d[o]) + scale * sum;
    }
  }
}

__global__ __launch_bounds__(kThreads) void colreduce_stage2_kernel(
    const float *__restrict__ partial, float *__restrict__ d, int64_t nblocks, int K, int r,
    int RT, int rank0, int out_layout, float scale, float beta, BatchStride bs) {
  colreduce_stage2_body(partial + blockIdx.y * bs.partial, d + blockIdx.y * bs.d, nblocks, K, r, RT, rank0, out_layout,
                        scale, beta, blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void colreduce_stage2_ragged_kernel(const lora_amd_ragged_desc *__restrict__ descs,
                                                                           int n, int r, int RT, int rank0, int out_layout,
                                                                           float scale) {
  const int g = ragged_find(descs, n, blockIdx.x, true);
  const lora_amd_ragged_desc d = descs[g];
  const int64_t local = (int64_t)blockIdx.x - d.begin2;
  const int64_t by = local / d.blocks2, bx = local - by * d.blocks2;
  const int64_t nrb = (d.M + kColRowsRagged - 1) / kColRowsRagged;
  colreduce_stage2_body(d.partial + by * d.stride_partial, d.out + by * d.stride_out, nrb, d.K, r, RT, rank0, out_layout,
                        scale, 0.f, bx);
}

template <class EX, bool MASKED>
__global__ __launch_bounds__(kThreads) void colreduce_generic_kernel(
    const typename EX::storage *__restrict__ x, int64_t ldx, const float *__restrict__ t,
    float *__restrict__ d, int64_t M, int K, int r, int out_layout, float scale, float beta, float p,
    uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  // one thread per (j, k); serial over M — correctness path for odd shapes only
  const int64_t total = (int64_t)r * K;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int j = (int)(i / K);
    const int k = (int)(i - (int64_t)j * K);
    float sum = 0.f;
    for (int64_t m = 0; m < M; ++m) {
      float xv = EX::to_f(x[m * ldx + k]);
      if (MASKED) {
        uint64_t e = (uint64_t)m * K + k;
        float mk[8];
        dropout_mult8(seed, dropout_offset(offset, offset_dev), e >> 3, p, mk);
        xv *= mk[e & 7];
      }
      sum = fmaf(t[m * r + j], xv, sum);
    }
    const int64_t o = out_layout == LORA_AMD_FACTOR_RK ? (int64_t)j * K + k : (int64_t)k * r + j;
    d[o] = (beta == 0.f ? 0.f : beta * d[o]) + scale * sum;
  }
}

// ---- host-side helpers ------------------------------------------------------
static inline int rank_tile(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : r <= 16 ? 16 : r <= 32 ? 32 : 64; }

static inline bool vec_ok(const void *p, int64_t ld, int cols, int dt) {
  const uintptr_t align = dt == LORA_AMD_F32 ? 32 : 16;
  return cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)p % align) == 0;
}

// lanes per row for rowdot: maximise lane efficiency c8 / (L * ceil(c8 / L)); ties -> larger L
static inline int pick_logL(int c8) {
  int best = 0;
  double best_eff = -1.0;
  for (int lg = 0; lg <= 6; ++lg) {
    int L = 1 << lg;
    double eff = (double)c8 / ((double)L * ((c8 + L - 1) / L));
    if (eff >= best_eff - 1e-9) { best_eff = eff > best_eff ? eff : best_eff; best = lg; }
  }
  return best;
}

template <class EX, bool MASKED>
static int launch_rowdot(const void *x, int64_t ldx, const void *f, void *t_out, int64_t M, int K, int r,
                         int fdt, int layout, float scale, const float *sel, int selT, float p,
                         uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st, int batch = 1,
                         BatchStride bs = BatchStride{0, 0, 0, 0, 0}) {
  using S = typename EX::storage;
  const S *xp = reinterpret_cast<const S *>(x);
  float *tp = reinterpret_cast<float *>(t_out);
  LORA_AMD_CHECK(batch == 1 || (vec_ok(x, ldx, K, EX::kCode) && bs.x % 8 == 0), LORA_AMD_EINVAL,
                 "rowdot: batched launches need 16-byte-friendly rows");
  if (!vec_ok(x, ldx, K, EX::kCode)) {
    int grid = (int)((M + 3) / 4);
    hipLaunchKernelGGL((rowdot_generic_kernel<EX, MASKED>), dim3(grid), dim3(kThreads), 0, st, xp, ldx, f, fdt,
                       layout, tp, M, K, r, scale, sel, selT, p, seed, offset, offset_dev);
    return check_launch("lora_amd_rowdot(generic)");
  }
  const int RT = rank_tile(r);
  int kt_cols = (kFactorLdsFloats / RT) & ~7;
  if (kt_cols > K) kt_cols = K;
  const int logL = pick_logL(kt_cols >> 3);
  const int rows_iter = (64 >> logL) * (kThreads / 64);
  // enough blocks to fill 256 CUs, enough rows per block to amortise the LDS staging
  int64_t rows_per_block = (M + 1023) / 1024;
  rows_per_block = ((rows_per_block + rows_iter - 1) / rows_iter) * rows_iter;
  const int grid = (int)((M + rows_per_block - 1) / rows_per_block);
#define RD(RTV)                                                                                         \
  hipLaunchKernelGGL((rowdot_kernel<EX, RTV, MASKED>), dim3(grid, batch), dim3(kThreads), 0, st, xp, ldx, f, fdt, \
                     layout, tp, M, K, r, kt_cols, logL, (int)rows_per_block, scale, sel, selT, p, seed, offset, offset_dev, bs)
  switch (RT) {
    case 4: RD(4); break;
    case 8: RD(8); break;
    case 16: RD(16); break;
    case 32: RD(32); break;
    default: RD(64); break;
  }
#undef RD
  return check_launch("lora_amd_rowdot");
}

template <class EY, bool DROP>
static int launch_rank_update(void *y, int64_t ldy, const float *t, const void *f, int64_t M, int N, int r,
                              int fdt, int layout, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                              hipStream_t st) {
  using S = typename EY::storage;
  S *yp = reinterpret_cast<S *>(y);
  if (!vec_ok(y, ldy, N, EY::kCode)) {
    int64_t total = M * (int64_t)N;
    int grid = (int)std::min<int64_t>((total + kThreads - 1) / kThreads, 4096);
    hipLaunchKernelGGL((rank_update_generic_kernel<EY, DROP>), dim3(grid), dim3(kThreads), 0, st, yp, ldy, t, f,
                       fdt, layout, M, N, r, scale, p, seed, offset, offset_dev);
    return check_launch("lora_amd_rank_update(generic)");
  }
  const int RT = rank_tile(r);
  int cols = (kFactorLdsFloats / RT) & ~7;
  if (cols > N) cols = N;
  int64_t rows = 32768 / cols;
  const int max_rows = kTLdsFloats / RT;
  if (rows > max_rows) rows = max_rows;
  if (rows > 128) rows = 128;
  if (rows < 8) rows = 8;
  if (rows > M) rows = M;
  const int tiles_n = (N + cols - 1) / cols;
  const int64_t tiles = tiles_n * ((M + rows - 1) / rows);
#define RU(RTV)                                                                                            \
  hipLaunchKernelGGL((rank_update_kernel<EY, RTV, DROP>), dim3((unsigned)tiles), dim3(kThreads), 0, st, yp, ldy, t, \
                     f, fdt, layout, M, N, r, (int)rows, cols, tiles_n, scale, p, seed, offset, offset_dev)
  switch (RT) {
    case 4: RU(4); break;
    case 8: RU(8); break;
    case 16: RU(16); break;
    case 32: RU(32); break;
    default: RU(64); break;
  }
#undef RU
  return check_launch("lora_amd_rank_update");
}

static inline int col_rank_tile(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : 16; }

template <class EX, bool MASKED>
static int launch_colreduce(const void *x, int64_t ldx, const float *t, float *d, int64_t M, int K, int r,
                            int out_layout, float scale, float beta, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                            void *ws, size_t ws_bytes, hipStream_t st, int batch = 1,
                            BatchStride bs = BatchStride{0, 0, 0, 0, 0}) {
  using S = typename EX::storage;
  const S *xp = reinterpret_cast<const S *>(x);
  LORA_AMD_CHECK(batch == 1 || (vec_ok(x, ldx, K, EX::kCode) && bs.x % 8 == 0), LORA_AMD_EINVAL,
                 "colreduce: batched launches need 16-byte-friendly rows");
  if (!vec_ok(x, ldx, K, EX::kCode)) {
    int grid = (int)std::min<int64_t>(((int64_t)r * K + kThreads - 1) / kThreads, 4096);
    hipLaunchKernelGGL((colreduce_generic_kernel<EX, MASKED>), dim3(grid), dim3(kThreads), 0, st, xp, ldx, t, d, M,
                       K, r, out_layout, scale, beta, p, seed, offset, offset_dev);
    return check_launch("lora_amd_colreduce(generic)");
  }
  LORA_AMD_CHECK(ws_bytes >= lora_amd_colreduce_workspace(M, K, r) * (size_t)batch, LORA_AMD_EWORKSPACE,
                 "colreduce: workspace %zu < %zu bytes", ws_bytes, lora_amd_colreduce_workspace(M, K, r) * (size_t)batch);
  bs.partial = (int64_t)(lora_amd_colreduce_workspace(M, K, r) / sizeof(float));
  const int RT = col_rank_tile(r);
  const int64_t nrb = (M + kColRowsPerBlock - 1) / kColRowsPerBlock;
  const int col_tiles = (K + kColMaxChunks * 8 - 1) / (kColMaxChunks * 8);
  float *partial = reinterpret_cast<float *>(ws);
  for (int rank0 = 0; rank0 < r; rank0 += RT) {
    if (nrb > 0) {  // ranks beyond 16 take extra passes over X
#define CR(RTV)                                                                                              \
  hipLaunchKernelGGL((colreduce_stage1_kernel<EX, RTV, MASKED>), dim3((unsigned)(nrb * col_tiles), batch), dim3(kThreads), \
                     0, st, xp, ldx, t, partial, M, K, r, rank0, col_tiles, p, seed, offset, offset_dev, bs)
    switch (RT) {
      case 4: CR(4); break;
      case 8: CR(8); break;
      default: CR(16); break;
    }
    }
#undef CR
    const int grid2 = (int)(((int64_t)RT * K + 63) / 64);
    hipLaunchKernelGGL(colreduce_stage2_kernel, dim3(grid2, batch), dim3(kThreads), 0, st, partial, d, nrb, K, r, RT,
                       rank0, out_layout, scale, beta, bs);
  }
  return check_launch("lora_amd_colreduce");
}

}  // namespace lora_amd

using namespace lora_amd;

#define COMMON_CHECKS(name, M, K, r, dt)                                                         \
  LORA_AMD_CHECK((M) >= 0 && (K) > 0, LORA_AMD_EINVAL, name ": bad shape M=%lld K=%d", (long long)(M), (int)(K)); \
  LORA_AMD_CHECK((r) >= 1 && (r) <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, name ": rank %d outside [1,%d]", (int)(r), LORA_AMD_MAX_RANK); \
  LORA_AMD_CHECK(dtype_ok(dt), LORA_AMD_EINVAL, name ": bad dtype %d", (int)(dt));               \
  if ((M) == 0) return LORA_AMD_OK;

extern "C" int lora_amd_rowdot_masked(const void *x, int64_t ldx, const void *factor, void *t_out, int64_t M,
                                      int32_t K, int32_t r, int32_t x_dtype, int32_t factor_dtype,
                                      int32_t factor_layout, float scale, const float *sel,
                                      int32_t sel_transposed, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                      void *stream) {
  COMMON_CHECKS("rowdot", M, K, r, x_dtype);
  LORA_AMD_CHECK(x && factor && t_out, LORA_AMD_EINVAL, "rowdot: null pointer");
  LORA_AMD_CHECK(dtype_ok(factor_dtype), LORA_AMD_EINVAL, "rowdot: bad factor dtype %d", factor_dtype);
  LORA_AMD_CHECK(ldx >= K, LORA_AMD_EINVAL, "rowdot: ldx %lld < K %d", (long long)ldx, K);
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "rowdot: dropout p=%f", dropout_p);
  hipStream_t st = (hipStream_t)stream;
  const bool masked = dropout_p > 0.f;
  // ranks 9..16 on 16-bit rows: the matrix-core form (csrc/rank16_mfma.hip)
  if (sel == nullptr && r16_rowdot(x, ldx, factor, factor_dtype, factor_layout, reinterpret_cast<float *>(t_out), M, K, r,
                                   x_dtype, scale, dropout_p, seed, offset, offset_dev, st))
    return check_launch("lora_amd_rowdot(mfma)");
#define GO(E)                                                                                          \
  return masked ? launch_rowdot<E, true>(x, ldx, factor, t_out, M, K, r, factor_dtype, factor_layout, scale, sel, \
                                         sel_transposed, dropout_p, seed, offset, offset_dev, st)                  \
                : launch_rowdot<E, false>(x, ldx, factor, t_out, M, K, r, factor_dtype, factor_layout, scale, sel, \
                                          sel_transposed, 0.f, 0, 0, nullptr, st)
  switch (x_dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
}

extern "C" int lora_amd_rowdot(const void *x, int64_t ldx, const void *factor, void *t_out, int64_t M, int32_t K,
                               int32_t r, int32_t x_dtype, int32_t factor_dtype, int32_t factor_layout,
                               float scale, const float *sel, int32_t sel_transposed, void *stream) {
  return lora_amd_rowdot_masked(x, ldx, factor, t_out, M, K, r, x_dtype, factor_dtype, factor_layout, scale, sel,
                                sel_transposed, 0.f, 0, 0, nullptr, stream);
}

extern "C" int lora_amd_rank_update(void *y, int64_t ldy, const float *t, const void *factor, int64_t M, int32_t N,
                                    int32_t r, int32_t y_dtype, int32_t factor_dtype, int32_t factor_layout,
                                    float scale, float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream) {
  COMMON_CHECKS("rank_update", M, N, r, y_dtype);
  LORA_AMD_CHECK(y && t && factor, LORA_AMD_EINVAL, "rank_update: null pointer");
  LORA_AMD_CHECK(dtype_ok(factor_dtype), LORA_AMD_EINVAL, "rank_update: bad factor dtype %d", factor_dtype);
  LORA_AMD_CHECK(ldy >= N, LORA_AMD_EINVAL, "rank_update: ldy %lld < N %d", (long long)ldy, N);
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "rank_update: dropout p=%f", dropout_p);
  hipStream_t st = (hipStream_t)stream;
  const bool drop = dropout_p > 0.f;
  if (r16_rank_update(y, ldy, t, 1, 0, factor, factor_dtype, factor_layout, M, N, r, y_dtype, scale, dropout_p, seed, offset,
                      offset_dev, st))
    return check_launch("lora_amd_rank_update(mfma)");
#define GO(E)                                                                                              \
  return drop ? launch_rank_update<E, true>(y, ldy, t, factor, M, N, r, factor_dtype, factor_layout, scale, \
                                            dropout_p, seed, offset, offset_dev, st)                                   \
              : launch_rank_update<E, false>(y, ldy, t, factor, M, N, r, factor_dtype, factor_layout, scale, 0.f, 0, 0, nullptr, st)
  switch (y_dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
}

extern "C" size_t lora_amd_colreduce_workspace(int64_t M, int32_t K, int32_t r) {
  if (M <= 0 || K <= 0 || r <= 0) return 0;
  const int RT = col_rank_tile(r);
  const int64_t nrb = (M + kColRowsPerBlock - 1) / kColRowsPerBlock;
  return (size_t)nrb * RT * K * sizeof(float);
}

extern "C" int lora_amd_colreduce(const void *x, int64_t ldx, const float *t, float *d_out, int64_t M, int32_t K,
                                  int32_t r, int32_t x_dtype, int32_t out_layout, float scale, float beta,
                                  float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  LORA_AMD_CHECK(M >= 0 && K > 0, LORA_AMD_EINVAL, "colreduce: bad shape M=%lld K=%d", (long long)M, K);
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "colreduce: rank %d outside [1,%d]", r, LORA_AMD_MAX_RANK);
  LORA_AMD_CHECK(dtype_ok(x_dtype), LORA_AMD_EINVAL, "colreduce: bad dtype %d", x_dtype);
  LORA_AMD_CHECK(x && t && d_out, LORA_AMD_EINVAL, "colreduce: null pointer");
  LORA_AMD_CHECK(ldx >= K, LORA_AMD_EINVAL, "colreduce: ldx %lld < K %d", (long long)ldx, K);
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "colreduce: dropout p=%f", dropout_p);
  hipStream_t st = (hipStream_t)stream;
  const bool masked = dropout_p > 0.f;
#define GO(E)                                                                                               \
  return masked ? launch_colreduce<E, true>(x, ldx, t, d_out, M, K, r, out_layout, scale, beta, dropout_p, seed, \
                                            offset, offset_dev, workspace, workspace_bytes, st)                         \
                : launch_colreduce<E, false>(x, ldx, t, d_out, M, K, r, out_layout, scale, beta, 0.f, 0, 0, nullptr, \
                                             workspace, workspace_bytes, st)
  switch (x_dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
}

// ---- batched forms (cli_svd.py: every same-shape site of a model in one launch) ---------------------------------
extern "C" int lora_amd_rowdot_batched(const void *x, int64_t ldx, int64_t stride_x, const void *factor,
                                       int64_t stride_factor, float *t_out, int64_t stride_t, int32_t batch, int64_t M,
                                       int32_t K, int32_t r, int32_t x_dtype, int32_t factor_dtype,
                                       int32_t factor_layout, float scale, void *stream) {
  COMMON_CHECKS("rowdot_batched", M, K, r, x_dtype);
  LORA_AMD_CHECK(x && factor && t_out && batch >= 1 && batch <= 65535, LORA_AMD_EINVAL, "rowdot_batched: bad argument");
  LORA_AMD_CHECK(dtype_ok(factor_dtype) && ldx >= K, LORA_AMD_EINVAL, "rowdot_batched: bad factor dtype / ldx");
  BatchStride bs{stride_x, stride_factor * dtype_size(factor_dtype), stride_t, 0, 0};
  hipStream_t st = (hipStream_t)stream;
#define GO(E)                                                                                                     \
  return launch_rowdot<E, false>(x, ldx, factor, t_out, M, K, r, factor_dtype, factor_layout, scale, nullptr, 0, 0.f, 0, \
                                 0, nullptr, st, batch, bs)
  switch (x_dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
}

extern "C" int lora_amd_colreduce_batched(const void *x, int64_t ldx, int64_t stride_x, const float *t, int64_t stride_t,
                                          float *d_out, int64_t stride_d, int32_t batch, int64_t M, int32_t K, int32_t r,
                                          int32_t x_dtype, int32_t out_layout, float scale, void *workspace,
                                          size_t workspace_bytes, void *stream) {
  LORA_AMD_CHECK(M >= 0 && K > 0 && r >= 1 && r <= LORA_AMD_MAX_RANK && dtype_ok(x_dtype), LORA_AMD_EINVAL,
                 "colreduce_batched: bad shape / dtype");
  LORA_AMD_CHECK(x && t && d_out && batch >= 1 && batch <= 65535 && ldx >= K, LORA_AMD_EINVAL,
                 "colreduce_batched: bad argument");
  BatchStride bs{stride_x, 0, stride_t, stride_d, 0};
  hipStream_t st = (hipStream_t)stream;
#define GO(E)                                                                                                       \
  return launch_colreduce<E, false>(x, ldx, t, d_out, M, K, r, out_layout, scale, 0.f, 0.f, 0, 0, nullptr, workspace, \
                                    workspace_bytes, st, batch, bs)
  switch (x_dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
}

// ---- ragged forms (cli_svd.py: every shape group of a model in one launch) ---------------------------------------
extern "C" int lora_amd_ragged_plan(int32_t op, lora_amd_ragged_desc *descs, int32_t n, int32_t r, int64_t *grid1,
                                    int64_t *grid2) {
  LORA_AMD_CHECK(descs && n >= 1 && grid1 && grid2, LORA_AMD_EINVAL, "ragged_plan: bad argument");
  LORA_AMD_CHECK(op == LORA_AMD_RAGGED_ROWDOT || op == LORA_AMD_RAGGED_COLREDUCE, LORA_AMD_EINVAL, "ragged_plan: op %d", op);
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "ragged_plan: rank %d outside [1,%d]", r, LORA_AMD_MAX_RANK);
  int64_t b1 = 0, b2 = 0;
  for (int i = 0; i < n; ++i) {
    lora_amd_ragged_desc &d = descs[i];
    LORA_AMD_CHECK(d.x && d.f && d.out && d.M > 0 && d.K > 0 && d.batch >= 1 && d.ldx >= d.K, LORA_AMD_EINVAL,
                   "ragged_plan: group %d: bad shape or null pointer", i);
    LORA_AMD_CHECK(vec_ok(d.x, d.ldx, d.K, LORA_AMD_F32) && d.stride_x % 8 == 0, LORA_AMD_EINVAL,
                   "ragged_plan: group %d: rows must be 32-byte aligned with K %% 8 == 0", i);
    d.begin1 = b1; d.begin2 = b2;
    d.blocks2 = 0; d.col_tiles = 0; d.kt_cols = 0; d.logL = 0; d.rows_per_block = 0; d.stride_partial = 0;
    if (op == LORA_AMD_RAGGED_ROWDOT) {
      const int RT = rank_tile(r);
      int kt_cols = (kFactorLdsFloats / RT) & ~7;
      if (kt_cols > d.K) kt_cols = d.K;
      const int logL = pick_logL(kt_cols >> 3);
      const int rows_iter = (64 >> logL) * (kThreads / 64);
      int64_t rows_per_block = (d.M + 1023) / 1024;
      rows_per_block = ((rows_per_block + rows_iter - 1) / rows_iter) * rows_iter;
      d.kt_cols = kt_cols; d.logL = logL; d.rows_per_block = (int32_t)rows_per_block;
      d.blocks1 = (int32_t)((d.M + rows_per_block - 1) / rows_per_block);
    } else {
      LORA_AMD_CHECK(d.partial, LORA_AMD_EWORKSPACE, "ragged_plan: group %d: colreduce needs a workspace", i);
      const int64_t nrb = (d.M + kColRowsRagged - 1) / kColRowsRagged;
      d.col_tiles = (d.K + kColMaxChunks * 8 - 1) / (kColMaxChunks * 8);
      d.blocks1 = (int32_t)(nrb * d.col_tiles);
      d.blocks2 = (int32_t)(((int64_t)col_rank_tile(r) * d.K + 63) / 64);
      d.stride_partial = (int64_t)(lora_amd_colreduce_workspace(d.M, d.K, r) / sizeof(float));
    }
    b1 += (int64_t)d.blocks1 * d.batch;
    b2 += (int64_t)d.blocks2 * d.batch;
  }
  LORA_AMD_CHECK(b1 < (1ll << 31) && b2 < (1ll << 31), LORA_AMD_EINVAL, "ragged_plan: too many blocks");
  *grid1 = b1; *grid2 = b2;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_rowdot_ragged(const lora_amd_ragged_desc *descs_dev, int32_t n, int64_t grid1, int32_t r,
                                      int32_t factor_layout, float scale, void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && grid1 >= 1 && grid1 < (1ll << 31), LORA_AMD_EINVAL, "rowdot_ragged: bad argument");
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "rowdot_ragged: rank %d outside [1,%d]", r, LORA_AMD_MAX_RANK);
  hipStream_t st = (hipStream_t)stream;
#define RD(RTV) hipLaunchKernelGGL((rowdot_ragged_kernel<RTV>), dim3((unsigned)grid1), dim3(kThreads), 0, st, descs_dev, n, r, factor_layout, scale)
  switch (rank_tile(r)) {
    case 4: RD(4); break;
    case 8: RD(8); break;
    case 16: RD(16); break;
    case 32: RD(32); break;
    default: RD(64); break;
  }
#undef RD
  return check_launch("lora_amd_rowdot_ragged");
}

extern "C" int lora_amd_colreduce_ragged(const lora_amd_ragged_desc *descs_dev, int32_t n, int64_t grid1, int64_t grid2,
                                         int32_t r, int32_t out_layout, float scale, void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && grid1 >= 1 && grid2 >= 1 && grid1 < (1ll << 31) && grid2 < (1ll << 31),
                 LORA_AMD_EINVAL, "colreduce_ragged: bad argument");
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "colreduce_ragged: rank %d outside [1,%d]", r, LORA_AMD_MAX_RANK);
  hipStream_t st = (hipStream_t)stream;
  const int RT = col_rank_tile(r);
  for (int rank0 = 0; rank0 < r; rank0 += RT) {  // ranks beyond 16 take extra passes over X
#define CR(RTV) hipLaunchKernelGGL((colreduce_stage1_ragged_kernel<RTV>), dim3((unsigned)grid1), dim3(kThreads), 0, st, descs_dev, n, r, rank0)
    switch (RT) {
      case 4: CR(4); break;
      case 8: CR(8); break;
      default: CR(16); break;
    }
#undef CR
    hipLaunchKernelGGL(colreduce_stage2_ragged_kernel, dim3((unsigned)grid2), dim3(kThreads), 0, st, descs_dev, n, r, RT,
                       rank0, out_layout, scale);
  }
  return check_launch("lora_amd_colreduce_ragged");
}

namespace lora_amd {
// out = (f32) a - (f32) b over a table of flat arrays; one workgroup per 4096 elements (16 per thread, 16-byte accesses
// when the three pointers allow it).
template <class E>
__global__ __launch_bounds__(kThreads) void sub_ragged_kernel(const lora_amd_sub_desc *__restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_sub_desc d = descs[lo];
  using S = typename E::storage;
  const S *a = reinterpret_cast<const S *>(d.a), *b = reinterpret_cast<const S *>(d.b);
  const int64_t e0 = ((int64_t)blockIdx.x - d.begin) * 4096;
  const bool vec = ((((uintptr_t)d.a | (uintptr_t)d.b) % (8 * sizeof(S))) == 0) && (((uintptr_t)d.out % 32) == 0);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t e = e0 + (u * kThreads + threadIdx.x) * 8;
    if (e >= d.n) continue;
    if (vec && e + 8 <= d.n) {
      float av[8], bv[8];
      load8<E>(a + e, av);
      load8<E>(b + e, bv);
#pragma unroll
      for (int i = 0; i < 8; ++i) av[i] -= bv[i];
      store8<f32_t>(d.out + e, av);
    } else {
      for (int i = 0; i < 8 && e + i < d.n; ++i) gl(d.out)[e + i] = E::to_f(gl(a)[e + i]) - E::to_f(gl(b)[e + i]);
    }
  }
}
}  // namespace lora_amd

extern "C" int lora_amd_sub_ragged(const lora_amd_sub_desc *descs_dev, int32_t n, int64_t blocks, int32_t in_dtype,
                                   void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && blocks >= 1 && blocks < (1ll << 31) && dtype_ok(in_dtype), LORA_AMD_EINVAL,
                 "sub_ragged: bad argument");
  hipStream_t st = (hipStream_t)stream;
  switch (in_dtype) {
    case LORA_AMD_F32: hipLaunchKernelGGL((sub_ragged_kernel<f32_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, descs_dev, n); break;
    case LORA_AMD_F16: hipLaunchKernelGGL((sub_ragged_kernel<f16_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, descs_dev, n); break;
    default: hipLaunchKernelGGL((sub_ragged_kernel<bf16_t>), dim3((unsigned)blocks), dim3(kThreads), 0, st, descs_dev, n); break;
  }
  return check_launch("lora_amd_sub_ragged");
}

// Inverse Cholesky factor of a stack of small Gram matrices, for CholeskyQR on the device (cli_svd.py):
//   G' = G + shift_rel * trace(G) / l * I,   G' = L L^T,   out = L^{-1}  ([l][l] row-major, lower triangular)
// so that Q = Y L^{-T} — one lora_amd_rowdot_batched(Y, factor = out) — has orthonormal columns when G = Y^T Y.
// One wave per matrix (l <= 32), f64 inside (the matrices are tiny; the input Gram is what limits accuracy: callers run
// the shifted pass first and an unshifted one after it, "shifted CholeskyQR3").  Rank-deficient blocks: see `floor_`.
__global__ __launch_bounds__(64) void chol_inverse_kernel(const float *__restrict__ gram, float *__restrict__ out, int l,
                                                           float shift_rel) {
  __shared__ double A[32][33];
  __shared__ double X[32][33];
  const int lane = threadIdx.x;
  const float *g = gram + (int64_t)blockIdx.x * l * l;
  float *o = out + (int64_t)blockIdx.x * l * l;
  double tr = 0.0;
  for (int i = 0; i < l; ++i) tr += (double)g[i * l + i];
  const double shift = (double)shift_rel * tr / (double)l;
  for (int idx = lane; idx < l * l; idx += 64) {
    const int i = idx / l, j = idx - i * l;
    A[i][j] = 0.5 * ((double)g[i * l + j] + (double)g[j * l + i]) + (i == j ? shift : 0.0);
  }
  __syncthreads();
  // a pivot below the noise floor of an f32-accumulated Gram matrix is a direction the block does not have (an exactly
  // low-rank residual sketched wider than its rank): it is dropped — L[j][j] = inf makes row and column j of L^{-1}
  // zero, so the corresponding column of Q = Y L^{-T} is zero instead of amplified rounding noise
  double dmax = 0.0;
  for (int i = 0; i < l; ++i) dmax = fmax(dmax, A[i][i]);
  const double floor_ = 1e-30 + 1e-6 * dmax;
  for (int j = 0; j < l; ++j) {
    double d = A[j][j];
    d = d > floor_ ? sqrt(d) : (double)INFINITY;
    __syncthreads();
    if (lane == j) A[j][j] = d;
    if (lane > j && lane < l) A[lane][j] /= d;
    __syncthreads();
    if (lane > j && lane < l) {
      const double lij = A[lane][j];
      for (int k = j + 1; k <= lane; ++k) A[lane][k] -= lij * A[k][j];
    }
    __syncthreads();
  }
  // forward substitution, lane = column c of the inverse: L x = e_c
  if (lane < l) {
    for (int i = 0; i < l; ++i) {
      double s = i == lane ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= A[i][k] * X[k][lane];
      X[i][lane] = i < lane ? 0.0 : s / A[i][i];
    }
  }
  __syncthreads();
  for (int idx = lane; idx < l * l; idx += 64) {
    const int i = idx / l, j = idx - i * l;
    o[idx] = (float)X[i][j];
  }
}

extern "C" int lora_amd_chol_inverse_batched(const float *gram, float *out, int32_t l, int32_t batch, float shift_rel,
                                             void *stream) {
  LORA_AMD_CHECK(gram && out && l >= 1 && l <= 32 && batch >= 1, LORA_AMD_EINVAL,
                 "chol_inverse_batched: l in [1,32], batch >= 1");
  hipLaunchKernelGGL(chol_inverse_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, gram, out, l, shift_rel);
  return check_launch("lora_amd_chol_inverse_batched");
}
