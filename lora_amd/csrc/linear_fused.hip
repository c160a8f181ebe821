// K1/K2 fused — one launch for the adapter forward, two for its backward.
//
//   linear_fwd    Y += scale * mask * ((X A^T) S^T) B^T,  T = (X A^T) S^T saved     [after the frozen GEMM]
//   linear_bwd_g  one pass over G:  Gt = scale*(mask*G) B   and   dUp partials = (mask*G)^T T
//   linear_bwd_x  one pass over X and dX:  dDown partials = Gt'^T X  and  dX += Gt' A   (Gt' = Gt S)
//   reduce_batched  ONE launch per optimiser step sums every site's partials into the flat grad buffer
//
// replaces: lora_diffusion/lora.py:53-58 (5 ATen launches forward, ~10 in its autograd) per site.
//
// Mapping ("column owner"): a thread owns 8 consecutive columns (one 16-byte chunk) of a column tile of
// ct8 chunks (ct8 = power of two dividing the row length, <= 64 so a row segment is one wave-slice) and
// walks the rows of its row slot.  The rank-r factor columns it needs live in registers (r*8 floats,
// loaded once per block with 16-byte loads that are all in flight together), per-row r-vectors (T, Gt)
// live in LDS, K-reductions are xor-shuffles over the ct8 lanes, M-reductions are register accumulators
// + an LDS slot reduction + per-block partials.  Everything is HBM-streaming: 16 B/lane, 4 loads in flight.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "rank16_mfma.hpp"

namespace lora_amd {

constexpr int kFT = 256;             // threads per block
constexpr int kFLdsFactor = 8192;    // floats, staged factor slab (forward)
constexpr int kFLdsT = 2048;         // floats, per-row r-vectors

__device__ inline float ldf(const void *p, int dt, int64_t i) {
  if (dt == LORA_AMD_F32) return gl(reinterpret_cast<const float *>(p))[i];
  if (dt == LORA_AMD_F16) return (float)gl(reinterpret_cast<const _Float16 *>(p))[i];
  return (float)gl(reinterpret_cast<const __bf16 *>(p))[i];
}

// Factor slab -> LDS layout [RT][2][ncols/8][4] (two conflict-free 16-byte planes per lane).
// f32 [r,C] slabs move as float4 with 4 loads in flight per thread; anything else element-wise.
template <int RT>
__device__ __forceinline__ void stage_factor(float *s_f, const void *f, int fdt, int layout, int r, int64_t C, int c0,
                                    int ncols) {
  const int c8 = ncols >> 3;
  const bool fast = fdt == LORA_AMD_F32 && layout == LORA_AMD_FACTOR_RK && (C & 3) == 0 && (c0 & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(f) & 15u) == 0);
  if (fast) {
    const int n4 = ncols >> 2, total = RT * n4;
    const float *fp = reinterpret_cast<const float *>(f);
    for (int i0 = threadIdx.x; i0 < total; i0 += kFT * 4) {
      float4 v[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * kFT;
        dst[u] = -1;
        if (i < total) {
          const int j = i / n4, c4 = i - j * n4;
          dst[u] = ((j * 2 + (c4 & 1)) * c8 + (c4 >> 1)) * 4;
          v[u] = j < r ? gl_ld4(fp + (int64_t)j * C + c0 + c4 * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (dst[u] >= 0) *reinterpret_cast<float4 *>(&s_f[dst[u]]) = v[u];
    }
    return;
  }
  if (fdt == LORA_AMD_F32 && layout == LORA_AMD_FACTOR_KR && r == 4 && RT == 4 &&
      ((reinterpret_cast<uintptr_t>(f) & 15u) == 0)) {
    // up [C, 4] f32: one 16-byte load per column brings all 4 ranks; 4 loads in flight per thread
    const float *fp = reinterpret_cast<const float *>(f);
    for (int c0b = threadIdx.x; c0b < ncols; c0b += kFT * 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0b + u * kFT;
        if (c < ncols) v[u] = gl_ld4(fp + (int64_t)(c0 + c) * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0b + u * kFT;
        if (c >= ncols) continue;
        const int base = ((c >> 2) & 1) * c8 * 4 + (c >> 3) * 4 + (c & 3);
        s_f[base + 0 * 2 * c8 * 4] = v[u].x; s_f[base + 1 * 2 * c8 * 4] = v[u].y;
        s_f[base + 2 * 2 * c8 * 4] = v[u].z; s_f[base + 3 * 2 * c8 * 4] = v[u].w;
      }
    }
    return;
  }
  const int total = RT * ncols;
#pragma unroll 4
  for (int i = threadIdx.x; i < total; i += kFT) {
    int j, c;
    if (layout == LORA_AMD_FACTOR_RK) { j = i / ncols; c = i - j * ncols; }
    else { c = i / RT; j = i - c * RT; }
    float v = 0.f;
    if (j < r) v = ldf(f, fdt, layout == LORA_AMD_FACTOR_RK ? (int64_t)j * C + c0 + c : (int64_t)(c0 + c) * r + j);
    s_f[((j * 2 + ((c >> 2) & 1)) * c8 + (c >> 3)) * 4 + (c & 3)] = v;
  }
}

// The r x 8 factor block a column-owner thread needs, straight into registers.
template <int RT>
__device__ __forceinline__ void load_factor_cols(float (&fc)[RT][8], const void *f, int fdt, int layout, int r, int64_t C,
                                        int col) {
  if (fdt == LORA_AMD_F32 && layout == LORA_AMD_FACTOR_RK && (C & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(f) & 15u) == 0)) {
    const float *fp = reinterpret_cast<const float *>(f);
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (j < r) {
        a = *reinterpret_cast<const float4 *>(fp + (int64_t)j * C + col);
        b = *reinterpret_cast<const float4 *>(fp + (int64_t)j * C + col + 4);
      }
      fc[j][0] = a.x; fc[j][1] = a.y; fc[j][2] = a.z; fc[j][3] = a.w;
      fc[j][4] = b.x; fc[j][5] = b.y; fc[j][6] = b.z; fc[j][7] = b.w;
    }
  } else if (fdt == LORA_AMD_F32 && layout == LORA_AMD_FACTOR_KR && r == 4 && RT == 4 &&
             ((reinterpret_cast<uintptr_t>(f) & 15u) == 0)) {
    const float *fp = reinterpret_cast<const float *>(f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 a = *reinterpret_cast<const float4 *>(fp + (int64_t)(col + i) * 4);
      fc[0][i] = a.x; fc[1 % RT][i] = a.y; fc[2 % RT][i] = a.z; fc[3 % RT][i] = a.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        fc[j][i] = j < r ? ldf(f, fdt, layout == LORA_AMD_FACTOR_RK ? (int64_t)j * C + col + i
                                                                     : (int64_t)(col + i) * r + j)
                         : 0.f;
  }
}

// s_t[row][RT] = (sum over nparts of part[p][m0+row][0..r)) (@ S | @ S^T), zero padded to RT.
template <int RT>
__device__ __forceinline__ void stage_rowvecs(float *s_t, const float *part, int nparts, int64_t part_stride, int64_t m0,
                                     int nrows, int r, const float *sel, int sel_transposed, float mult) {
  for (int i = threadIdx.x; i < nrows * RT; i += kFT) {
    const int rl = i / RT, j = i - rl * RT;
    float v = 0.f;
    if (j < r) {
      if (sel == nullptr) {
        for (int p = 0; p < nparts; ++p) v += part[p * part_stride + (m0 + rl) * r + j];
      } else {
        for (int b = 0; b < r; ++b) {
          float tb = 0.f;
          for (int p = 0; p < nparts; ++p) tb += part[p * part_stride + (m0 + rl) * r + b];
          v = fmaf(tb, sel_transposed ? sel[b * r + j] : sel[j * r + b], v);
        }
      }
    }
    s_t[i] = v * mult;
  }
}

// Slot reduction of per-thread accumulators acc[RT][8] -> out[j*ld + col0 + c] for the block's column tile.
// LDS image [4 ranks][slot][col]: every thread drops its 8 columns as two 16-byte writes per rank, the summing pass
// reads consecutive columns with consecutive lanes (conflict-free) and stores 256-byte runs of one output row.
template <int RT>
__device__ __forceinline__ void slot_reduce_store(float *s_red, const float (&acc)[RT][8], int slot, int nslots, int cl,
                                         int ct8, float *out, int64_t ld, int col0) {  // ld = row length: the last tile may overhang
  const int ncols = ct8 * 8;
#pragma unroll
  for (int jb = 0; jb < RT; jb += 4) {
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float *dst = &s_red[((jj * nslots + slot) * ncols) + cl * 8];
      *reinterpret_cast<float4 *>(dst) = make_float4(acc[(jb + jj) % RT][0], acc[(jb + jj) % RT][1],
                                                     acc[(jb + jj) % RT][2], acc[(jb + jj) % RT][3]);
      *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[(jb + jj) % RT][4], acc[(jb + jj) % RT][5],
                                                         acc[(jb + jj) % RT][6], acc[(jb + jj) % RT][7]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncols * 4; i += kFT) {
      const int jj = i / ncols, col = i - jj * ncols;
      const float *src = &s_red[(jj * nslots) * ncols + col];
      float sum = 0.f;
      for (int s = 0; s < nslots; ++s) sum += src[s * ncols];
      if (col0 + col < ld) out[(int64_t)(jb + jj) * ld + col0 + col] = sum;
    }
  }
}

// ============================================================================ forward
template <class E, int RT, bool DROP>
__global__ __launch_bounds__(kFT) void linear_fwd_kernel(
    const typename E::storage *__restrict__ x, int64_t ldx, typename E::storage *__restrict__ y, int64_t ldy,
    const void *__restrict__ down, const void *__restrict__ up, int fdt, float *__restrict__ t_out, int64_t M, int K,
    int N, int r, int kt_cols, int nt_cols, int cols_per_y, int logL, int rows_per_block, float scale,
    const float *__restrict__ sel, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  __shared__ __attribute__((aligned(16))) float s_f[kFLdsFactor];
  __shared__ __attribute__((aligned(16))) float s_t[kFLdsT];
  __shared__ float s_sel[RT * RT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = 1 << logL, G = 64 >> logL;
  const int l = lane & (L - 1), g = lane >> logL;
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int rows_iter = G * (kFT / 64);
  const int niter = (nrows + rows_iter - 1) / rows_iter;

  if (sel != nullptr)
    for (int i = tid; i < RT * RT; i += kFT) {
      const int a = i / RT, b = i - a * RT;
      s_sel[i] = (a < r && b < r) ? sel[a * r + b] : 0.f;
    }
  // ---- phase 1: T rows of this block (lora_down + selector), L lanes per row
  const bool single = kt_cols >= K;
  if (single) {
    stage_factor<RT>(s_f, down, fdt, LORA_AMD_FACTOR_RK, r, K, 0, K);
    __syncthreads();
  }
  constexpr int U = 4;
  for (int it = 0; it < niter; ++it) {
    const int rl = it * rows_iter + wave * G + g;
    const bool live = rl < nrows;
    float acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += kt_cols) {
      const int ncols = min(kt_cols, K - k0), c8 = ncols >> 3;
      if (!single) {
        __syncthreads();
        stage_factor<RT>(s_f, down, fdt, LORA_AMD_FACTOR_RK, r, K, k0, ncols);
        __syncthreads();
      }
      if (live) {
        const typename E::storage *xr = x + (m0 + rl) * ldx + k0;
        for (int cb = l; cb < c8; cb += L * U) {
          float xv[U][8];
#pragma unroll
          for (int u = 0; u < U; ++u) {  // clamped address + select: the U loads issue back to back (no branch/wait each)
            const int cc = cb + u * L;
            load8_sel<E>(xr + (cc < c8 ? cc : cb) * 8, cc < c8, xv[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int cc = cb + u * L;
            if (cc >= c8) continue;
#pragma unroll
            for (int j = 0; j < RT; ++j) {
              const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + cc) * 4]);
              const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + cc) * 4]);
              float a = acc[j];
              a = fmaf(xv[u][0], d0.x, a); a = fmaf(xv[u][1], d0.y, a); a = fmaf(xv[u][2], d0.z, a);
              a = fmaf(xv[u][3], d0.w, a); a = fmaf(xv[u][4], d1.x, a); a = fmaf(xv[u][5], d1.y, a);
              a = fmaf(xv[u][6], d1.z, a); a = fmaf(xv[u][7], d1.w, a);
              acc[j] = a;
            }
          }
        }
      }
    }
    if (sel == nullptr && logL >= 2) {
      // butterfly: lanes l < 4 of the row's lane group end up with the totals of ranks jb + idx4(l)
#pragma unroll
      for (int jb = 0; jb < RT; jb += 4) {
        const float tot = group_sum4(acc[jb], acc[(jb + 1) % RT], acc[(jb + 2) % RT], acc[(jb + 3) % RT], lane, logL);
        const int j = jb + idx4(l);
        if (live && l < 4) {
          s_t[rl * RT + j] = tot;
          if (j < r && blockIdx.y == 0) t_out[(m0 + rl) * r + j] = tot;
        }
      }
      continue;
    }
    for (int off = L >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < RT; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    if (live && l == 0) {
      float o[RT];
      if (sel != nullptr) {
#pragma unroll
        for (int a = 0; a < RT; ++a) {
          float v = 0.f;
#pragma unroll
          for (int b = 0; b < RT; ++b) v = fmaf(acc[b], s_sel[a * RT + b], v);
          o[a] = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < RT; ++j) o[j] = acc[j];
      }
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        s_t[rl * RT + j] = o[j];
        if (j < r && blockIdx.y == 0) t_out[(m0 + rl) * r + j] = o[j];
      }
    }
  }
  // ---- phase 2: Y[rows of this block, :] += scale * mask * T @ up^T, column tile by column tile
  // wide outputs (GEGLU) put column groups on grid.y: each group recomputes its rows of T (X is K/N of the traffic)
  const int n_begin = blockIdx.y * cols_per_y, n_end = min(N, n_begin + cols_per_y);
  for (int n0 = n_begin; n0 < n_end; n0 += nt_cols) {
    const int ncols = min(nt_cols, n_end - n0), c8 = ncols >> 3;
    __syncthreads();  // s_t complete / previous tile's readers done
    stage_factor<RT>(s_f, up, fdt, LORA_AMD_FACTOR_KR, r, N, n0, ncols);
    __syncthreads();
    const int nchunk = nrows * c8;
    const int dq = kFT / c8, dr = kFT % c8;
    int rl = tid / c8, cc = tid % c8;
    for (int c = tid; c < nchunk; c += kFT * U) {
      float v[U][8];
      int rls[U], ccs[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[u] = (c + u * kFT) < nchunk;
        rls[u] = rl; ccs[u] = cc;
        load8_sel<E>(y + (m0 + (ok[u] ? rl : rls[0])) * ldy + n0 + (ok[u] ? cc : ccs[0]) * 8, ok[u], v[u]);
        rl += dq; cc += dr;
        if (cc >= c8) { cc -= c8; ++rl; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float pr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float *tr = s_t + rls[u] * RT;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const float tj = tr[j];
          const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + ccs[u]) * 4]);
          const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + ccs[u]) * 4]);
          pr[0] = fmaf(tj, d0.x, pr[0]); pr[1] = fmaf(tj, d0.y, pr[1]); pr[2] = fmaf(tj, d0.z, pr[2]);
          pr[3] = fmaf(tj, d0.w, pr[3]); pr[4] = fmaf(tj, d1.x, pr[4]); pr[5] = fmaf(tj, d1.y, pr[5]);
          pr[6] = fmaf(tj, d1.z, pr[6]); pr[7] = fmaf(tj, d1.w, pr[7]);
        }
        if (DROP) {
          float mk[8];
          const int64_t e = (m0 + rls[u]) * (int64_t)N + n0 + ccs[u] * 8;
          dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
          for (int i = 0; i < 8; ++i) pr[i] *= mk[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] = fmaf(scale, pr[i], v[u][i]);
        store8<E>(y + (m0 + rls[u]) * ldy + n0 + ccs[u] * 8, v[u]);
      }
    }
  }
}

// ============================================================================ backward, pass over G
// grid = row blocks x column tiles (column tile fastest).  Outputs:
//   gt_part[ct][M][r]  = scale * sum over this tile's columns of (mask*G)[m, n] * up[n, j]
//   up_part[rb][RT][N] = scale * sum over this block's rows of (mask*G)[m, n] * T[m, j]
// FCL: the thread's RT x 8 block of `up` lives in LDS instead of registers (rank tile 16: 128 registers less — two waves
// per SIMD instead of one; the block is the same for every row slot, so the slot-0 threads stage it once, in the area
// the slot reduction uses at the end; rows of RT*8 + 4 floats keep the 16-byte reads of different columns off each other's banks)
constexpr int kFclStride16 = 16 * 8 + 4;

template <class E, int RT, bool DROP, bool FCL = false>
__global__ __launch_bounds__(kFT) void linear_bwd_g_kernel(
    const typename E::storage *__restrict__ g, int64_t ldg, const float *__restrict__ t,
    const void *__restrict__ up, int fdt, float *__restrict__ gt_part, float *__restrict__ up_part, int64_t M,
    int N, int r, int log_ct8, int nct, int rows_per_block, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  __shared__ __attribute__((aligned(16))) float s_red[FCL ? 64 * kFclStride16 : kFT * 8 * 4];
  static_assert(64 * kFclStride16 >= kFT * 8 * 4, "slot-reduction area");
  __shared__ __attribute__((aligned(16))) float s_t[kFLdsT];
  __shared__ __attribute__((aligned(16))) float s_gt[kFLdsT];  // this block's Gt rows, stored once at the end
  const int tid = threadIdx.x;
  const int ct8 = 1 << log_ct8, nslots = kFT >> log_ct8;
  const int slot = tid >> log_ct8, cl = tid & (ct8 - 1);
  const int64_t rb = blockIdx.x / nct;
  const int ct = (int)(blockIdx.x - rb * nct);
  const int64_t m0 = rb * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int col = (ct * ct8 + cl) * 8;
  const bool colok = col < N;  // the last tile of a row whose chunk count is not a multiple of ct8 overhangs: idle lanes

  float fc[FCL ? 1 : RT][8];
  constexpr int FS = RT * 8 + 4;
  const float *s_fc = s_red + cl * FS;
  if constexpr (FCL) {
    if (slot == 0) {
      float tmp[RT][8];
      load_factor_cols<RT>(tmp, up, fdt, LORA_AMD_FACTOR_KR, r, N, colok ? col : 0);
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        *reinterpret_cast<float4 *>(s_red + cl * FS + j * 8) = make_float4(tmp[j][0], tmp[j][1], tmp[j][2], tmp[j][3]);
        *reinterpret_cast<float4 *>(s_red + cl * FS + j * 8 + 4) = make_float4(tmp[j][4], tmp[j][5], tmp[j][6], tmp[j][7]);
      }
    }
  } else {
    load_factor_cols<RT>(fc, up, fdt, LORA_AMD_FACTOR_KR, r, N, colok ? col : 0);
  }
  stage_rowvecs<RT>(s_t, t, 1, 0, m0, nrows, r, nullptr, 0, scale);  // dUp partials carry `scale`
  __syncthreads();

  float acc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;

  constexpr int U = 4;
  const bool want_gt = gt_part != nullptr;  // nullptr: dUp partials only (Gt came out of the fused MFMA dX kernel)
  float *gtp = gt_part + (int64_t)ct * M * r;
  // every lane of a wave runs the same trip count (shuffles below): bound by the first slot's rows
  for (int rb0 = 0; rb0 < nrows; rb0 += nslots * U) {
    float gv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots + slot;
      load8_sel<E>(g + (m0 + (rl < nrows ? rl : nrows - 1)) * ldg + (colok ? col : 0), rl < nrows && colok, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots + slot;
      const bool live = rl < nrows;
      float dot[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) dot[j] = 0.f;
      if (live) {
        if (DROP) {
          float mk[8];
          const int64_t e = (m0 + rl) * (int64_t)N + col;
          dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
          for (int i = 0; i < 8; ++i) gv[u][i] *= mk[i];
        }
        const float *tr = s_t + rl * RT;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const float tj = tr[j];
          float d = 0.f;
          float fj[8];
          if constexpr (FCL) {
            const float4 f0 = *reinterpret_cast<const float4 *>(s_fc + j * 8), f1 = *reinterpret_cast<const float4 *>(s_fc + j * 8 + 4);
            fj[0] = f0.x; fj[1] = f0.y; fj[2] = f0.z; fj[3] = f0.w; fj[4] = f1.x; fj[5] = f1.y; fj[6] = f1.z; fj[7] = f1.w;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) fj[i] = fc[j][i];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            d = fmaf(gv[u][i], fj[i], d);
            acc[j][i] = fmaf(tj, gv[u][i], acc[j][i]);
          }
          dot[j] = d;
        }
      }
      if (want_gt) {
#pragma unroll
        for (int jb = 0; jb < RT; jb += 4) {  // ct8 >= 4 lanes per row segment (plan): butterfly, 5..7 cross-lane ops
          const float tot =
              group_sum4(dot[jb], dot[(jb + 1) % RT], dot[(jb + 2) % RT], dot[(jb + 3) % RT], tid, log_ct8);
          const int j = jb + idx4(cl);
          if (live && cl < 4) s_gt[rl * RT + j] = scale * tot;
        }
      }
    }
  }
  __syncthreads();
  if (want_gt)
    for (int i = tid; i < nrows * r; i += kFT) {  // contiguous [nrows][r] run of gt_part
      const int rl = i / r, j = i - rl * r;
      gtp[m0 * r + i] = s_gt[rl * RT + j];
    }
  slot_reduce_store<RT>(s_red, acc, slot, nslots, cl, ct8, up_part + (int64_t)rb * RT * N, N, ct * ct8 * 8);
}

// ============================================================================ backward, pass over X (and dX)
//   Gt' = (sum_ct gt_part[ct]) @ S ;  down_part[rb][RT][K] = Gt'^T X (block rows) ;  dX += Gt' @ down
template <class E, int RT, bool HAS_DX>
__global__ __launch_bounds__(kFT) void linear_bwd_x_kernel(
    const typename E::storage *__restrict__ x, int64_t ldx, typename E::storage *__restrict__ dx, int64_t lddx,
    const float *__restrict__ gt_part, int nct_g, const void *__restrict__ down, int fdt,
    const float *__restrict__ sel, float *__restrict__ down_part, int64_t M, int K, int r, int log_ct8, int nct,
    int rows_per_block) {
  __shared__ __attribute__((aligned(16))) float s_red[kFT * 8 * 4];
  __shared__ __attribute__((aligned(16))) float s_t[kFLdsT];
  const int tid = threadIdx.x;
  const int ct8 = 1 << log_ct8, nslots = kFT >> log_ct8;
  const int slot = tid >> log_ct8, cl = tid & (ct8 - 1);
  const int64_t rb = blockIdx.x / nct;
  const int ct = (int)(blockIdx.x - rb * nct);
  const int64_t m0 = rb * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int colv = (ct * ct8 + cl) * 8;
  const bool colok = colv < K;  // overhanging lanes of the last tile idle
  const int col = colok ? colv : 0;

  float fc[RT][8];
  if (HAS_DX) load_factor_cols<RT>(fc, down, fdt, LORA_AMD_FACTOR_RK, r, K, col);
  stage_rowvecs<RT>(s_t, gt_part, nct_g, M * r, m0, nrows, r, sel, 1, 1.0f);
  __syncthreads();

  float acc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;

  constexpr int U = 4;
  for (int rb0 = slot; rb0 < nrows; rb0 += nslots * U) {
    float xv[U][8], dv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      const int rc = rl < nrows ? rl : nrows - 1;
      load8_sel<E>(x + (m0 + rc) * ldx + col, rl < nrows && colok, xv[u]);
      if (HAS_DX) load8_sel<E>(dx + (m0 + rc) * lddx + col, rl < nrows && colok, dv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      if (rl >= nrows) continue;
      const float *tr = s_t + rl * RT;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float tj = tr[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[j][i] = fmaf(tj, xv[u][i], acc[j][i]);
          if (HAS_DX) dv[u][i] = fmaf(tj, fc[j][i], dv[u][i]);
        }
      }
      if (HAS_DX && colok) store8<E>(dx + (m0 + rl) * lddx + col, dv[u]);
    }
  }
  slot_reduce_store<RT>(s_red, acc, slot, nslots, cl, ct8, down_part + (int64_t)rb * RT * K, K, ct * ct8 * 8);
}

// ============================================================================ backward, both factor gradients
// When the input gradient and Gt came out of the fused MFMA launch (gemm_fused.hip), what is left of the backward are
// two outer-product column sums of the same shape:
//   up_part[rb][RT][N]   = sum over the block's rows of (scale * T)[m, j]  * G[m, n]
//   down_part[rb][RT][K] = sum over the block's rows of (Gt @ S)[m, j]     * X[m, k]
// ONE launch runs both: workgroups [0, a.nblocks) stream G, the rest stream X (no dropout here: the fused MFMA
// launch does not take it).
struct FactorJob {
  const void *data;      // [M, C] activations
  int64_t ld;
  const float *rowvec;   // [M, r] f32
  const float *sel;      // [r, r] applied to the row vectors (X job), or null
  float *part;           // [nrb][RT][C]
  float mult;
  int C, log_ct8, nct, rows_per_block, nblocks;
  int hc, hp;            // head-padded rows of `data`: logical chunk c lives at (c / hc) * hp + c % hc; hc == 0: dense
  float p;               // dropout on `data` (the G job of a site whose branch had nn.Dropout): mask chunk = (row*C + col) / 8
  uint64_t seed, offset;
  const uint64_t *offset_dev;
};

template <class E, int RT>
__global__ __launch_bounds__(kFT) void linear_bwd_factors_kernel(FactorJob a, FactorJob b, int64_t M, int r) {
  __shared__ __attribute__((aligned(16))) float s_red[kFT * 8 * 4];
  __shared__ __attribute__((aligned(16))) float s_t[kFLdsT];
  const bool first = (int)blockIdx.x < a.nblocks;
  const int bid = first ? (int)blockIdx.x : (int)blockIdx.x - a.nblocks;
  const typename E::storage *data = reinterpret_cast<const typename E::storage *>(first ? a.data : b.data);
  const int64_t ld = first ? a.ld : b.ld;
  const float *rowvec = first ? a.rowvec : b.rowvec;
  const float *sel = first ? a.sel : b.sel;
  float *part = first ? a.part : b.part;
  const float mult = first ? a.mult : b.mult;
  const int C = first ? a.C : b.C, log_ct8 = first ? a.log_ct8 : b.log_ct8, nct = first ? a.nct : b.nct;
  const int rows_per_block = first ? a.rows_per_block : b.rows_per_block;
  const int hc = first ? a.hc : b.hc, hp = first ? a.hp : b.hp;
  const float dp = first ? a.p : b.p;  // block-uniform
  const uint64_t dseed = first ? a.seed : b.seed;
  const uint64_t doff = dp > 0.f ? dropout_offset(first ? a.offset : b.offset, first ? a.offset_dev : b.offset_dev) : 0;

  const int tid = threadIdx.x;
  const int ct8 = 1 << log_ct8, nslots = kFT >> log_ct8;
  const int slot = tid >> log_ct8, cl = tid & (ct8 - 1);
  const int64_t rb = bid / nct;
  const int ct = (int)(bid - rb * nct);
  const int64_t m0 = rb * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int colv = (ct * ct8 + cl) * 8;                                             // logical column (partials)
  const bool colok = colv < C;                                                      // last tile may overhang the row
  const int col = colok ? colv : 0;
  const int pcol = hc ? (((col >> 3) / hc) * hp + ((col >> 3) % hc)) * 8 : col;     // where it lives in `data`

  stage_rowvecs<RT>(s_t, rowvec, 1, 0, m0, nrows, r, sel, 1, mult);
  __syncthreads();

  float acc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;

  constexpr int U = 4;
  for (int rb0 = slot; rb0 < nrows; rb0 += nslots * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      load8_sel<E>(data + (m0 + (rl < nrows ? rl : nrows - 1)) * ld + pcol, rl < nrows && colok, v[u]);
    }
    if (dp > 0.f) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float mk[8];
        const int64_t e = (m0 + rb0 + u * nslots) * (int64_t)C + col;
        dropout_mult8(dseed, doff, (uint64_t)(e >> 3), dp, mk);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[u][i] *= mk[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      const float *tr = s_t + (rl < nrows ? rl : 0) * RT;  // rows past the end contribute v = 0
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float tj = tr[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(tj, v[u][i], acc[j][i]);
      }
    }
  }
  slot_reduce_store<RT>(s_red, acc, slot, nslots, cl, ct8, part + (int64_t)rb * RT * C, C, ct * ct8 * 8);
}

// ============================================================================ backward, both factor gradients, self-sufficient
// The merged-weight training path (ops.LoraLinearMergedFunction: Y = X W_eff^T with W_eff = W + s up down refreshed once
// per step by the K3 merge launch, dX = G W_eff) leaves the backward with ONLY the parameter gradients, and with neither
// T = X down^T nor Gt = s G up at hand.  This kernel recomputes both per row block and consumes them in place:
//   phase A   s_t[m][j]  = s * sum_k X[m, k] down[j, k]          s_gt[m][j] = s * sum_n G[m, n] up[n, j]      (LDS)
//   phase B   up_part[rb][j][n] = sum_m s_t[m][j] G[m, n]        down_part[rb][j][k] = sum_m s_gt[m][j] X[m, k]
// (phase B reads the block's rows a second time: rows * (N + K) * 2 bytes per block; in the one-launch pass that second
// read is served past the L2 — FETCH_SIZE 2.0x algorithmic).  grid = row blocks x `nsplit`; the splits of a row block share out the column tiles of
// phase B (and each redo phase A: only taken when M is too small to fill the chip with row blocks alone).
constexpr int kSelfRowsCap = 128;  // rows of a block: their r-vectors live in LDS next to the 32 KiB work buffer

struct SelfArgs {
  const void *g, *x;
  int64_t ldg, ldx, M;
  const float *down, *up;          // f32 masters: [r, K] and [N, r]
  float *up_part, *down_part;      // [nrb][RT][N], [nrb][RT][K]
  float scale;
  int N, K, r, rows_per_block, nsplit;
  int ghc, ghp, xhc, xhp;          // head-padded rows (16-byte chunks per head: logical, physical); 0 = dense
  int kt_g, logL_g, kt_x, logL_x;  // phase A: factor columns per LDS stage, lanes per row (log2)
  int tile_g, nct_g, tile_x, nct_x;  // phase B: 16-byte chunks per column tile (<= 256), number of tiles
};

__device__ __forceinline__ int hchunk(int c, int hc, int hp) { return hc ? (c / hc) * hp + (c % hc) : c; }

// s_out[rl * RT + j] = mult * sum_c data[m0 + rl, c] * factor(j, c) for the block's rows; L lanes per row.  The factor
// passes through LDS in slabs of kt_cols columns (outer loop: a slab is staged once per block, not once per row group).
template <class E, int RT, int U>
__device__ __forceinline__ void block_rowdots(float *s_f, float *s_out, const typename E::storage *data, int64_t ld, int64_t m0,
                                     int nrows, int C, const float *factor, int layout, int r, int kt_cols, int logL,
                                     float mult, int hc, int hp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = 1 << logL, G = 64 >> logL;
  const int l = lane & (L - 1), g = lane >> logL;
  const int rows_iter = G * (kFT / 64);
  const int niter = (nrows + rows_iter - 1) / rows_iter;
  for (int k0 = 0; k0 < C; k0 += kt_cols) {  // U = loads in flight per lane
    const int ncols = min(kt_cols, C - k0), c8 = ncols >> 3;
    __syncthreads();
    stage_factor<RT>(s_f, factor, LORA_AMD_F32, layout, r, C, k0, ncols);
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < niter; ++it) {
      const int rl = it * rows_iter + wave * G + g;
      const bool live = rl < nrows;
      float acc[RT];
#pragma unroll
      for (int j = 0; j < RT; ++j) acc[j] = 0.f;
      if (live) {
        const typename E::storage *xr = data + (m0 + rl) * ld;
        const int cbase = k0 >> 3;
#pragma unroll 1
        for (int cb = l; cb < c8; cb += L * U) {
          float xv[U][8];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int cc = cb + u * L;
            load8_sel<E>(xr + hchunk(cbase + (cc < c8 ? cc : cb), hc, hp) * 8, cc < c8, xv[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int cc = cb + u * L;
            if (cc >= c8) continue;
#pragma unroll
            for (int j = 0; j < RT; ++j) {
              const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + cc) * 4]);
              const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + cc) * 4]);
              float a = acc[j];
              a = fmaf(xv[u][0], d0.x, a); a = fmaf(xv[u][1], d0.y, a); a = fmaf(xv[u][2], d0.z, a);
              a = fmaf(xv[u][3], d0.w, a); a = fmaf(xv[u][4], d1.x, a); a = fmaf(xv[u][5], d1.y, a);
              a = fmaf(xv[u][6], d1.z, a); a = fmaf(xv[u][7], d1.w, a);
              acc[j] = a;
            }
          }
        }
      }
      for (int off = L >> 1; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
      if (live && l == 0) {  // the row belongs to this lane group in every slab: no race on s_out
#pragma unroll
        for (int j = 0; j < RT; ++j) s_out[rl * RT + j] = (k0 == 0 ? 0.f : s_out[rl * RT + j]) + acc[j] * mult;
      }
    }
  }
}

// Column sums of one tile [c0, c0 + tc8) of 16-byte chunks (tc8 <= 256, any value: slot = tid / tc8, so a 320-wide row is
// ONE tile of 40 chunks x 6 row slots), in two steps so that a caller can accumulate over several row groups first:
//   colsum_acc:    acc[j][i] += s_vec[row][j] * data[row, chunk] over the rows of this thread's slot
//   colsum_finish: the slots' accumulators meet in LDS ([4 ranks][slot][col]) -> part[j][c] (this block's [RT][C] slab)
template <class E, int RT, int U>
__device__ __forceinline__ void colsum_acc(float (&acc)[RT][8], const float *s_vec, const typename E::storage *data,
                                           int64_t ld, int64_t m0, int nrows, int c0, int tc8, int hc, int hp) {
  const int tid = threadIdx.x;
  const int nslots = kFT / tc8;
  const int slot = tid / tc8, cl = tid - slot * tc8;
  if (slot >= nslots) return;  // kFT % tc8 threads idle in the streaming part
  const int pcol = hchunk(c0 + cl, hc, hp) * 8;
#pragma unroll 1
  for (int rb0 = slot; rb0 < nrows; rb0 += nslots * U) {
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      load8_sel<E>(data + (m0 + (rl < nrows ? rl : nrows - 1)) * ld + pcol, rl < nrows, v[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb0 + u * nslots;
      const float *tr = s_vec + (rl < nrows ? rl : 0) * RT;  // rows past the end contribute v = 0
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float tj = tr[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(tj, v[u][i], acc[j][i]);
      }
    }
  }
}

template <int RT>
__device__ __forceinline__ void colsum_finish(float *s_red, const float (&acc)[RT][8], int C, int c0, int tc8, float *part) {
  const int tid = threadIdx.x;
  const int nslots = kFT / tc8;
  const int slot = tid / tc8, cl = tid - slot * tc8;
  const bool owner = slot < nslots;
  const int col = (c0 + cl) * 8;
  if (nslots == 1) {  // a tile of more than 128 chunks: one row slot, every owner holds final column sums
#pragma unroll
    for (int j = 0; owner && j < RT; ++j) {
      gl_st4(part + (int64_t)j * C + col, acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      gl_st4(part + (int64_t)j * C + col + 4, acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
    }
    return;
  }
  const int ncols = tc8 * 8;
#pragma unroll
  for (int jb = 0; jb < RT; jb += 4) {
    __syncthreads();
    if (owner) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float *dst = &s_red[((jj * nslots + slot) * ncols) + cl * 8];
        *reinterpret_cast<float4 *>(dst) = make_float4(acc[(jb + jj) % RT][0], acc[(jb + jj) % RT][1],
                                                       acc[(jb + jj) % RT][2], acc[(jb + jj) % RT][3]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[(jb + jj) % RT][4], acc[(jb + jj) % RT][5],
                                                           acc[(jb + jj) % RT][6], acc[(jb + jj) % RT][7]);
      }
    }
    __syncthreads();
    for (int i = tid; i < ncols * 4; i += kFT) {
      const int jj = i / ncols, cc = i - jj * ncols;
      const float *src = &s_red[(jj * nslots) * ncols + cc];
      float sum = 0.f;
      for (int q = 0; q < nslots; ++q) sum += src[q * ncols];
      gl(part)[(int64_t)(jb + jj) * C + c0 * 8 + cc] = sum;
    }
  }
}

template <class E, int RT, int U>
__device__ __forceinline__ void block_colsums(float *s_red, const float *s_vec, const typename E::storage *data, int64_t ld,
                                     int64_t m0, int nrows, int C, int c0, int tc8, float *part, int hc, int hp) {
  float acc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
  colsum_acc<E, RT, U>(acc, s_vec, data, ld, m0, nrows, c0, tc8, hc, hp);
  colsum_finish<RT>(s_red, acc, C, c0, tc8, part);
}

template <class E, int RT, int U = 4>
__device__ __forceinline__ void factors_self_body(const SelfArgs &a, int64_t bid) {
  // phase A stages the factor slab, phase B reduces row slots: never live together
  __shared__ __attribute__((aligned(16))) float s_buf[kFT * 8 * 4];
  __shared__ __attribute__((aligned(16))) float s_t[kSelfRowsCap * RT];   // rows_per_block <= kSelfRowsCap (factors_self_geom)
  __shared__ __attribute__((aligned(16))) float s_gt[kSelfRowsCap * RT];
  static_assert(kFT * 8 * 4 >= kFLdsFactor, "shared buffer");
  using S = typename E::storage;
  const S *g = reinterpret_cast<const S *>(a.g), *x = reinterpret_cast<const S *>(a.x);
  const int64_t rb = bid / a.nsplit;
  const int sp = (int)(bid - rb * a.nsplit);
  const int64_t m0 = rb * a.rows_per_block;
  const int nrows = (int)min((int64_t)a.rows_per_block, a.M - m0);
  block_rowdots<E, RT, U>(s_buf, s_t, x, a.ldx, m0, nrows, a.K, a.down, LORA_AMD_FACTOR_RK, a.r, a.kt_x, a.logL_x, a.scale,
                       a.xhc, a.xhp);
  block_rowdots<E, RT, U>(s_buf, s_gt, g, a.ldg, m0, nrows, a.N, a.up, LORA_AMD_FACTOR_KR, a.r, a.kt_g, a.logL_g, a.scale,
                       a.ghc, a.ghp);
  __syncthreads();
  // phase B: column tiles of <= 256 chunks, G's first, shared out among the splits of the row block
  const int c8g = a.N >> 3, c8x = a.K >> 3;
  for (int t = sp; t < a.nct_g + a.nct_x; t += a.nsplit) {
    if (t < a.nct_g) {
      const int c0 = t * a.tile_g;
      block_colsums<E, RT, U>(s_buf, s_t, g, a.ldg, m0, nrows, a.N, c0, min(a.tile_g, c8g - c0),
                           a.up_part + rb * RT * (int64_t)a.N, a.ghc, a.ghp);
    } else {
      const int c0 = (t - a.nct_g) * a.tile_x;
      block_colsums<E, RT, U>(s_buf, s_gt, x, a.ldx, m0, nrows, a.K, c0, min(a.tile_x, c8x - c0),
                           a.down_part + rb * RT * (int64_t)a.K, a.xhc, a.xhp);
    }
  }
}

template <class E, int RT>
__global__ __launch_bounds__(kFT) void linear_bwd_factors_self_kernel(const SelfArgs a) {
  factors_self_body<E, RT>(a, blockIdx.x);
}

// Every site of a model in ONE launch: the table holds one lora_amd_self_site per adapter (same activation dtype and rank
// tile), blocks are numbered through it.  The per-site launches of a training step are latency-bound (a site is 1-90 MB,
// one or two rounds of workgroups, four dependent trips to memory each); deferred to the end of the backward and issued
// together, the sites' phases overlap across ~50 000 workgroups and the pass runs at memory throughput.
// Occupancy: the pass is bound by how many loads the chip keeps in flight.  The helpers are __forceinline__: left as
// plain `inline`, hipcc emitted real calls to them (s_swappc) and the call ABI cost the kernel 232 registers (2 waves per
// SIMD); inlined, rank tile 4 takes 117 (4 waves per SIMD, 36 KiB of LDS per workgroup).
template <class E, int RT, int U>
__global__ __launch_bounds__(kFT) void linear_bwd_factors_self_ragged_kernel(const lora_amd_self_site *__restrict__ sites,
                                                                             int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (sites[mid].block_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_self_site q = sites[lo];
  SelfArgs a;
  a.g = q.g; a.x = q.x; a.ldg = q.ldg; a.ldx = q.ldx; a.M = q.M; a.down = q.down; a.up = q.up;
  a.up_part = q.up_part; a.down_part = q.down_part; a.scale = q.scale; a.N = q.N; a.K = q.K; a.r = q.r;
  a.rows_per_block = q.rows_per_block; a.nsplit = q.nsplit;
  a.ghc = q.g_head_dim >> 3; a.ghp = q.g_head_pad >> 3; a.xhc = q.x_head_dim >> 3; a.xhp = q.x_head_pad >> 3;
  a.kt_g = q.kt_g; a.logL_g = q.logL_g; a.kt_x = q.kt_x; a.logL_x = q.logL_x;
  a.tile_g = q.tile_g; a.nct_g = q.nct_g; a.tile_x = q.tile_x; a.nct_x = q.nct_x;
  factors_self_body<E, RT, U>(a, (int64_t)blockIdx.x - q.block_begin);
}

// ---- the same, wave-specialised: the two halves of the workgroup work on the two tensors at the same time, so a block's
// critical path is ONE trip to HBM (phase A: waves 0-1 take X rows against `down`, waves 2-3 the G rows against `up`)
// and ONE to L2 (phase B: waves 0-1 sum G columns weighted by T, waves 2-3 X columns weighted by Gt) instead of four
// trips in sequence.  Needs both factor slabs in LDS together: RT * (N + K) <= kSelfLdsFloats (every attention site and
// the smallest GEGLU projection at rank 4; the sequential kernel above takes the rest).
constexpr int kSelfLdsFloats = 16384;  // 64 KiB: [down slab | up slab] in phase A, the two slot-reduction areas in phase B
constexpr int kDualThreads = 512;      // 8 waves: 4 per tensor
constexpr int kHalf = kDualThreads / 2;
constexpr int kSelfRowsMax = kSelfRowsCap;

template <class E, int RT>
__device__ __forceinline__ void half_rowdots(const float *s_f, float *s_out, const typename E::storage *data, int64_t ld, int64_t m0,
                                    int nrows, int C, int logL, float mult, int hc, int hp, int hw) {
  const int lane = threadIdx.x & 63;
  const int L = 1 << logL, G = 64 >> logL;
  const int l = lane & (L - 1), g = lane >> logL;
  const int rows_iter = G * (kHalf / 64), c8 = C >> 3;
  constexpr int U = 8;  // loads in flight per lane: a 320-wide row block is one trip to memory
  for (int rl = hw * G + g; rl < nrows; rl += rows_iter) {
    float acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.f;
    const typename E::storage *xr = data + (m0 + rl) * ld;
    for (int cb = l; cb < c8; cb += L * U) {
      float xv[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = cb + u * L;
        load8_sel<E>(xr + hchunk(cc < c8 ? cc : cb, hc, hp) * 8, cc < c8, xv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = cb + u * L;
        if (cc >= c8) continue;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const float4 d0 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 0) * c8 + cc) * 4]);
          const float4 d1 = *reinterpret_cast<const float4 *>(&s_f[((j * 2 + 1) * c8 + cc) * 4]);
          float a = acc[j];
          a = fmaf(xv[u][0], d0.x, a); a = fmaf(xv[u][1], d0.y, a); a = fmaf(xv[u][2], d0.z, a);
          a = fmaf(xv[u][3], d0.w, a); a = fmaf(xv[u][4], d1.x, a); a = fmaf(xv[u][5], d1.y, a);
          a = fmaf(xv[u][6], d1.z, a); a = fmaf(xv[u][7], d1.w, a);
          acc[j] = a;
        }
      }
    }
    for (int off = L >> 1; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < RT; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    if (l == 0) {
#pragma unroll
      for (int j = 0; j < RT; ++j) s_out[rl * RT + j] = acc[j] * mult;
    }
  }
}

// staging with all kDualThreads threads (stage_factor strides by kFT: its callers are 256-thread kernels)
template <int RT>
__device__ __forceinline__ void stage_factor_dual(float *s_f, const float *f, int layout, int r, int C) {
  const int c8 = C >> 3;
  for (int i = threadIdx.x; i < RT * C; i += kDualThreads) {
    int j, c;
    if (layout == LORA_AMD_FACTOR_RK) { j = i / C; c = i - j * C; }
    else { c = i / RT; j = i - c * RT; }  // [C, r]: consecutive threads read consecutive ranks of a column
    const float v = j < r ? f[layout == LORA_AMD_FACTOR_RK ? (int64_t)j * C + c : (int64_t)c * r + j] : 0.f;
    s_f[((j * 2 + ((c >> 2) & 1)) * c8 + (c >> 3)) * 4 + (c & 3)] = v;
  }
}

template <class E, int RT>
__global__ __launch_bounds__(kDualThreads) void linear_bwd_factors_self_dual_kernel(const SelfArgs a) {
  __shared__ __attribute__((aligned(16))) float s_buf[kSelfLdsFloats];
  __shared__ __attribute__((aligned(16))) float s_t[kSelfRowsMax * RT];
  __shared__ __attribute__((aligned(16))) float s_gt[kSelfRowsMax * RT];
  using S = typename E::storage;
  const S *g = reinterpret_cast<const S *>(a.g), *x = reinterpret_cast<const S *>(a.x);
  const int64_t rb = blockIdx.x;
  const int64_t m0 = rb * a.rows_per_block;
  const int nrows = (int)min((int64_t)a.rows_per_block, a.M - m0);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool first = wave < kHalf / 64;  // wave-uniform
  float *s_fx = s_buf, *s_fg = s_buf + RT * a.K;
  stage_factor_dual<RT>(s_fx, a.down, LORA_AMD_FACTOR_RK, a.r, a.K);
  stage_factor_dual<RT>(s_fg, a.up, LORA_AMD_FACTOR_KR, a.r, a.N);
  __syncthreads();
  // ---- phase A
  if (first) half_rowdots<E, RT>(s_fx, s_t, x, a.ldx, m0, nrows, a.K, a.logL_x, a.scale, a.xhc, a.xhp, wave);
  else half_rowdots<E, RT>(s_fg, s_gt, g, a.ldg, m0, nrows, a.N, a.logL_g, a.scale, a.ghc, a.ghp, wave - kHalf / 64);
  __syncthreads();
  // ---- phase B: this half's tensor, its row vectors, its partial slab and its half of the slot-reduction area
  const S *data = first ? g : x;
  const int64_t ld = first ? a.ldg : a.ldx;
  const int C = first ? a.N : a.K, hc = first ? a.ghc : a.xhc, hp = first ? a.ghp : a.xhp;
  const float *s_vec = first ? s_t : s_gt;
  float *part = first ? a.up_part + rb * RT * (int64_t)a.N : a.down_part + rb * RT * (int64_t)a.K;
  float *s_red = s_buf + (first ? 0 : kSelfLdsFloats / 2);
  const int c8 = C >> 3;
  const int ntile = (c8 + kHalf - 1) / kHalf, tile = (c8 + ntile - 1) / ntile;
  const int ntile_max = max(((a.N >> 3) + kHalf - 1) / kHalf, ((a.K >> 3) + kHalf - 1) / kHalf);
  const int ht = tid & (kHalf - 1);
  for (int t = 0; t < ntile_max; ++t) {  // both halves pass the same barriers
    const bool have = t < ntile;
    const int c0 = t * tile, tc8 = have ? min(tile, c8 - c0) : 1;
    const int nslots = kHalf / tc8;
    const int slot = ht / tc8, cl = ht - slot * tc8;
    const bool owner = have && slot < nslots;
    const int col = (c0 + cl) * 8;
    float acc[RT][8];
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    constexpr int U = RT <= 8 ? 8 : 4;
    if (owner) {
      const int pcol = hchunk(c0 + cl, hc, hp) * 8;
      for (int rb0 = slot; rb0 < nrows; rb0 += nslots * U) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rl = rb0 + u * nslots;
          load8_sel<E>(data + (m0 + (rl < nrows ? rl : nrows - 1)) * ld + pcol, rl < nrows, v[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rl = rb0 + u * nslots;
          const float *tr = s_vec + (rl < nrows ? rl : 0) * RT;
#pragma unroll
          for (int j = 0; j < RT; ++j) {
            const float tj = tr[j];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(tj, v[u][i], acc[j][i]);
          }
        }
      }
    }
    const int ncols = tc8 * 8;
#pragma unroll
    for (int jb = 0; jb < RT; jb += 4) {
      __syncthreads();  // phase A's factor slabs / the previous round's sums have been read
      if (owner) {
        if (nslots == 1) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float *dst = part + (int64_t)(jb + jj) * C + col;
            *reinterpret_cast<float4 *>(dst) = make_float4(acc[(jb + jj) % RT][0], acc[(jb + jj) % RT][1],
                                                           acc[(jb + jj) % RT][2], acc[(jb + jj) % RT][3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[(jb + jj) % RT][4], acc[(jb + jj) % RT][5],
                                                               acc[(jb + jj) % RT][6], acc[(jb + jj) % RT][7]);
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            float *dst = &s_red[((jj * nslots + slot) * ncols) + cl * 8];
            *reinterpret_cast<float4 *>(dst) = make_float4(acc[(jb + jj) % RT][0], acc[(jb + jj) % RT][1],
                                                           acc[(jb + jj) % RT][2], acc[(jb + jj) % RT][3]);
            *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[(jb + jj) % RT][4], acc[(jb + jj) % RT][5],
                                                               acc[(jb + jj) % RT][6], acc[(jb + jj) % RT][7]);
          }
        }
      }
      __syncthreads();
      if (have && nslots > 1) {
        for (int i = ht; i < ncols * 4; i += kHalf) {
          const int jj = i / ncols, cc = i - jj * ncols;
          const float *src = &s_red[(jj * nslots) * ncols + cc];
          float sum = 0.f;
          for (int q = 0; q < nslots; ++q) sum += src[q * ncols];
          gl(part)[(int64_t)(jb + jj) * C + c0 * 8 + cc] = sum;
        }
      }
    }
  }
}

// ============================================================================ batched partial reduction
// out (f32; [r,C] or [C,r]) = beta*out + scale * sum_p part[p][j][c].  One launch for every descriptor.
// Latency, not bytes, bounded round 4's form (87-90 us for 0.29 GB in the step): a search of the table in memory (eight
// dependent loads) in front of a sum with 4 loads in flight over up to 256 parts.  Now the table's prefix is searched in LDS and
// 16 independent loads are in flight per thread.
constexpr int kReduceLds = 1024;
__global__ __launch_bounds__(kFT) void reduce_batched_kernel(const lora_amd_reduce_desc *__restrict__ descs, int n,
                                                             int64_t total) {
  __shared__ int64_t s_begin[kReduceLds];
  const bool cached = n <= kReduceLds;
  if (cached) {
    for (int i = threadIdx.x; i < n; i += kFT) s_begin[i] = gl(descs)[i].begin;
    __syncthreads();
  }
  for (int64_t i = (int64_t)blockIdx.x * kFT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kFT) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((cached ? s_begin[mid] : descs[mid].begin) <= i) lo = mid; else hi = mid - 1;
    }
    const lora_amd_reduce_desc d = descs[lo];
    const int64_t e = i - d.begin;
    const int j = (int)(e / d.C), c = (int)(e - (int64_t)j * d.C);
    const float LORA_AMD_AS_GLOBAL *pp = gl(d.part) + (int64_t)j * d.C + c;
    const int64_t stride = (int64_t)d.RT * d.C;
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
    int p = 0;
    for (; p + 16 <= d.nparts; p += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = pp[(int64_t)(p + u) * stride];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] += v[u];
    }
    for (; p < d.nparts; ++p) acc[p & 15] += pp[(int64_t)p * stride];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    const float sum = acc[0];
    const int64_t o = d.layout == LORA_AMD_FACTOR_RK ? (int64_t)j * d.C + c : (int64_t)c * d.r + j;
    gl(d.out)[o] = (d.beta == 0.f ? 0.f : d.beta * gl(d.out)[o]) + d.scale * sum;
  }
}

// ---------------------------------------------------------------------------- host helpers
static inline int frank_tile(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : 16; }
static inline int pow2_divisor(int c8, int cap) {
  int v = 1;
  while (v < cap && (c8 % (v * 2)) == 0) v *= 2;
  return v;
}
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline bool aligned_ok(const void *p, int64_t ld, int cols, int dt) {
  const uintptr_t a = dt == LORA_AMD_F32 ? 32 : 16;
  return cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)p % a) == 0;
}
static inline int pick_logL(int c8) {
  int best = 0;
  double best_eff = -1.0;
  for (int lg = 0; lg <= 6; ++lg) {
    const int L = 1 << lg;
    const double eff = (double)c8 / ((double)L * ((c8 + L - 1) / L));
    if (eff >= best_eff - 1e-9) { best_eff = std::max(eff, best_eff); best = lg; }
  }
  return best;
}

struct BwdGeom { int log_ct8, nct, rows_per_block; int64_t nrb; };
static BwdGeom bwd_geom(int64_t M, int cols, int RT, int cap) {
  BwdGeom q;
  const int c8 = cols / 8;
  // column tile = a power of two of 16-byte chunks (lane groups reduce by xor-shuffle): the largest one dividing the row
  // (320 columns = 40 chunks -> five 8-chunk tiles; the next power of two with the last tile overhanging measured 1 %
  // slower on configs[3], round 3)
  const int ct8 = pow2_divisor(c8, cap);
  q.log_ct8 = ilog2(ct8);
  q.nct = (c8 + ct8 - 1) / ct8;
  // rows per block: as many workgroups as stay co-resident (3 per CU by LDS -> 768) and no more, so that no CU is
  // left with an extra workgroup while the rest idle; >= 32 rows (partials are RT*4/(rows*e) of the stream) and
  // <= kFLdsT/RT rows (their r-vectors live in LDS)
  const int64_t max_rows = kFLdsT / RT;
  const int64_t resident = 3 * 256;
  int64_t rows = (M * q.nct + resident - 1) / resident;
  if (rows < 32) rows = 32;
  while (rows < max_rows && ((M + rows - 1) / rows) * q.nct > resident) ++rows;
  if (rows > max_rows) rows = max_rows;
  if (rows > M) rows = M;
  q.rows_per_block = (int)rows;
  q.nrb = (M + rows - 1) / rows;
  return q;
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_linear_plan(int64_t M, int32_t K, int32_t N, int32_t r, lora_amd_linear_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr, LORA_AMD_EINVAL, "linear_plan: null output");
  LORA_AMD_CHECK(M >= 0 && K > 0 && N > 0, LORA_AMD_EINVAL, "linear_plan: bad shape");
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "linear_plan: rank %d outside [1,%d]", r,
                 LORA_AMD_MAX_RANK);
  memset(out, 0, sizeof(*out));
  out->rank_tile = frank_tile(r);
  // fused kernels: 16-byte lanes, rank <= 16, row segments of >= 4 chunks per wave-slice
  if (M == 0 || r > 16 || K % 8 || N % 8 || pow2_divisor(N / 8, 64) < 4 || pow2_divisor(K / 8, 256) < 4) return LORA_AMD_OK;
  const BwdGeom gg = bwd_geom(M, N, out->rank_tile, 64), gx = bwd_geom(M, K, out->rank_tile, 256);
  out->fused = 1;
  out->nct_g = gg.nct;
  out->gt_part_floats = (int64_t)gg.nct * M * r;
  out->nparts_up = (int32_t)gg.nrb;
  out->up_part_floats = gg.nrb * out->rank_tile * (int64_t)N;
  out->nparts_down = (int32_t)gx.nrb;
  out->down_part_floats = gx.nrb * out->rank_tile * (int64_t)K;
  return LORA_AMD_OK;
}

#define FUSED_COMMON(name, dt, fdt)                                                                \
  LORA_AMD_CHECK(dtype_ok(dt) && dtype_ok(fdt), LORA_AMD_EINVAL, name ": bad dtype");              \
  LORA_AMD_CHECK(r >= 1 && r <= 16, LORA_AMD_ERANK, name ": fused path needs rank in [1,16], got %d", r); \
  if (M == 0) return LORA_AMD_OK;

extern "C" int lora_amd_linear_fwd(const void *x, int64_t ldx, void *y, int64_t ldy, const void *down,
                                   const void *up, float *t_out, int64_t M, int32_t K, int32_t N, int32_t r,
                                   int32_t act_dtype, int32_t factor_dtype, float scale, const float *sel,
                                   float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream) {
  FUSED_COMMON("linear_fwd", act_dtype, factor_dtype);
  LORA_AMD_CHECK(x && y && down && up && t_out, LORA_AMD_EINVAL, "linear_fwd: null pointer");
  LORA_AMD_CHECK(aligned_ok(x, ldx, K, act_dtype) && aligned_ok(y, ldy, N, act_dtype), LORA_AMD_EINVAL,
                 "linear_fwd: needs K%%8==0, N%%8==0, ld%%8==0 and 16-byte aligned rows (use the primitives otherwise)");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "linear_fwd: dropout p=%f", dropout_p);
  const int RT = frank_tile(r);
  // rank tile 16 on 16-bit rows: T = X down^T, then Y += s mask . (T up^T), both on the matrix cores (csrc/rank16_mfma.hip)
  if (RT == 16 && sel == nullptr && K % 32 == 0 && g_r16_mfma &&
      r16_rowdot(x, ldx, down, factor_dtype, LORA_AMD_FACTOR_RK, t_out, M, K, r, act_dtype, 1.0f, 0.f, 0, 0, nullptr,
                 (hipStream_t)stream)) {
    if (r16_rank_update(y, ldy, t_out, 1, 0, up, factor_dtype, LORA_AMD_FACTOR_KR, M, N, r, act_dtype, scale, dropout_p, seed,
                        offset, offset_dev, (hipStream_t)stream))
      return check_launch("lora_amd_linear_fwd(mfma)");
    return lora_amd_rank_update(y, ldy, t_out, up, M, N, r, act_dtype, factor_dtype, LORA_AMD_FACTOR_KR, scale, dropout_p, seed,
                                offset, offset_dev, stream);
  }
  int kt = (kFLdsFactor / RT) & ~7, nt = kt;
  if (kt > K) kt = K;
  if (nt > N) nt = N;
  const int logL = pick_logL(kt >> 3);
  const int rows_iter = (64 >> logL) * (kFT / 64);
  int64_t rpb = (M + 1023) / 1024;
  rpb = ((rpb + rows_iter - 1) / rows_iter) * rows_iter;
  const int64_t max_rows = ((kFLdsT / RT) / rows_iter) * rows_iter;
  if (rpb > max_rows) rpb = max_rows;
  if (rpb < rows_iter) rpb = rows_iter;
  const unsigned gx = (unsigned)((M + rpb - 1) / rpb);
  // column groups only where the output is much wider than the input (N >= 4K: the GEGLU projections) and the row
  // blocks alone are at most one per CU (measured: splitting 512 row blocks of the 16384-row site costs 15 %);
  // ~1024 workgroups in total
  int ny = 1;
  if (N >= 4 * K && gx <= 256) ny = (int)std::min<int64_t>(std::max<int64_t>(1, 1024 / gx), (N + 1023) / 1024);
  int cols_per_y = (((N + ny - 1) / ny) + 7) & ~7;
  ny = (N + cols_per_y - 1) / cols_per_y;
  if (nt > cols_per_y) nt = cols_per_y;
  const dim3 grid(gx, (unsigned)ny);
  hipStream_t st = (hipStream_t)stream;
  const bool drop = dropout_p > 0.f;
#define FW(E, RTV, D)                                                                                          \
  hipLaunchKernelGGL((linear_fwd_kernel<E, RTV, D>), grid, dim3(kFT), 0, st,                                   \
                     reinterpret_cast<const typename E::storage *>(x), ldx, reinterpret_cast<typename E::storage *>(y), \
                     ldy, down, up, factor_dtype, t_out, M, K, N, r, kt, nt, cols_per_y, logL, (int)rpb, scale, sel, \
                     dropout_p, seed, offset, offset_dev)
#define FW_RT(E, D) do { if (RT == 4) FW(E, 4, D); else if (RT == 8) FW(E, 8, D); else FW(E, 16, D); } while (0)
#define FW_E(E) do { if (drop) FW_RT(E, true); else FW_RT(E, false); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: FW_E(f32_t); break;
    case LORA_AMD_F16: FW_E(f16_t); break;
    default: FW_E(bf16_t); break;
  }
#undef FW_E
#undef FW_RT
#undef FW
  return check_launch("lora_amd_linear_fwd");
}

extern "C" int lora_amd_linear_bwd_g(const void *g, int64_t ldg, const float *t, const void *up, float *gt_part,
                                     float *up_part, int64_t M, int32_t N, int32_t r, int32_t act_dtype,
                                     int32_t factor_dtype, float scale, float dropout_p, uint64_t seed,
                                     uint64_t offset, const uint64_t *offset_dev, void *stream) {
  FUSED_COMMON("linear_bwd_g", act_dtype, factor_dtype);
  LORA_AMD_CHECK(g && t && up && up_part, LORA_AMD_EINVAL, "linear_bwd_g: null pointer");
  LORA_AMD_CHECK(aligned_ok(g, ldg, N, act_dtype) && pow2_divisor(N / 8, 64) >= 4, LORA_AMD_EINVAL,
                 "linear_bwd_g: shape/alignment not supported by the fused path (see lora_amd_linear_plan)");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "linear_bwd_g: dropout p=%f", dropout_p);
  const int RT = frank_tile(r);
  const BwdGeom q = bwd_geom(M, N, RT, 64);
  const unsigned grid = (unsigned)(q.nrb * q.nct);
  hipStream_t st = (hipStream_t)stream;
  const bool drop = dropout_p > 0.f;
  // rank tile 16 on 16-bit rows: the matrix-core form with the same outputs and geometry (csrc/rank16_mfma.hip)
  if (RT == 16 && r16_bwd_g(g, ldg, t, up, factor_dtype, gt_part, up_part, M, N, r, q.log_ct8, q.nct, q.rows_per_block, q.nrb,
                            act_dtype, scale, dropout_p, seed, offset, offset_dev, st))
    return check_launch("lora_amd_linear_bwd_g(mfma)");
#define BG(E, RTV, D)                                                                                       \
  hipLaunchKernelGGL((linear_bwd_g_kernel<E, RTV, D>), dim3(grid), dim3(kFT), 0, st,                        \
                     reinterpret_cast<const typename E::storage *>(g), ldg, t, up, factor_dtype, gt_part, up_part, M, \
                     N, r, q.log_ct8, q.nct, q.rows_per_block, scale, dropout_p, seed, offset, offset_dev)
#define BG16(E, D)                                                                                          \
  hipLaunchKernelGGL((linear_bwd_g_kernel<E, 16, D, true>), dim3(grid), dim3(kFT), 0, st,                   \
                     reinterpret_cast<const typename E::storage *>(g), ldg, t, up, factor_dtype, gt_part, up_part, M, \
                     N, r, q.log_ct8, q.nct, q.rows_per_block, scale, dropout_p, seed, offset, offset_dev)
#define BG_RT(E, D) do { if (RT == 4) BG(E, 4, D); else if (RT == 8) BG(E, 8, D); else BG16(E, D); } while (0)
#define BG_E(E) do { if (drop) BG_RT(E, true); else BG_RT(E, false); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: BG_E(f32_t); break;
    case LORA_AMD_F16: BG_E(f16_t); break;
    default: BG_E(bf16_t); break;
  }
#undef BG_E
#undef BG_RT
#undef BG16
#undef BG
  return check_launch("lora_amd_linear_bwd_g");
}

// lora_amd_linear_bwd_g with the fold of its Gt column-tile partials inside the launch (round 6): gt_out [M, r] complete when
// the launch ends, no lora_amd_sum_parts behind it.  Only the matrix-core form has the in-launch hand-off: bf16 rows, f32
// factors, ranks 9..16; anything else returns LORA_AMD_EUNSUPPORTED and the caller runs lora_amd_linear_bwd_g +
// lora_amd_sum_parts.  `counters`: one uint32 per row block (lora_amd_linear_bwd_g_blocks), zeroed ONCE.
extern "C" int lora_amd_linear_bwd_g_blocks(int64_t M, int32_t N, int32_t r, int64_t *row_blocks) {
  LORA_AMD_CHECK(row_blocks && M >= 0 && N > 0 && r >= 1 && r <= 16, LORA_AMD_EINVAL, "linear_bwd_g_blocks: bad argument");
  *row_blocks = bwd_geom(M, N, frank_tile(r), 64).nrb;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_linear_bwd_g_folded(const void *g, int64_t ldg, const float *t, const void *up, float *gt_part,
                                            float *gt_out, uint32_t *counters, float *up_part, int64_t M, int32_t N,
                                            int32_t r, int32_t act_dtype, int32_t factor_dtype, float scale,
                                            float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                            void *stream) {
  FUSED_COMMON("linear_bwd_g_folded", act_dtype, factor_dtype);
  LORA_AMD_CHECK(g && t && up && up_part && gt_part && gt_out && counters, LORA_AMD_EINVAL, "linear_bwd_g_folded: null pointer");
  LORA_AMD_CHECK(aligned_ok(g, ldg, N, act_dtype) && pow2_divisor(N / 8, 64) >= 4, LORA_AMD_EINVAL,
                 "linear_bwd_g_folded: shape/alignment not supported by the fused path (see lora_amd_linear_plan)");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "linear_bwd_g_folded: dropout p=%f", dropout_p);
  const int RT = frank_tile(r);
  const BwdGeom q = bwd_geom(M, N, RT, 64);
  if (RT == 16 && r16_bwd_g(g, ldg, t, up, factor_dtype, gt_part, up_part, M, N, r, q.log_ct8, q.nct, q.rows_per_block, q.nrb,
                            act_dtype, scale, dropout_p, seed, offset, offset_dev, (hipStream_t)stream, gt_out, counters))
    return check_launch("lora_amd_linear_bwd_g_folded");
  LORA_AMD_CHECK(false, LORA_AMD_EUNSUPPORTED, "linear_bwd_g_folded: the in-launch fold needs bf16 rows, f32 factors and a rank in 9..16");
}

extern "C" int lora_amd_linear_bwd_x(const void *x, int64_t ldx, void *dx, int64_t lddx, const float *gt_part,
                                     int32_t nct_g, const void *down, const float *sel, float *down_part,
                                     int64_t M, int32_t K, int32_t r, int32_t act_dtype, int32_t factor_dtype,
                                     void *stream) {
  FUSED_COMMON("linear_bwd_x", act_dtype, factor_dtype);
  LORA_AMD_CHECK(x && gt_part && down && down_part && nct_g >= 1, LORA_AMD_EINVAL, "linear_bwd_x: null pointer");
  LORA_AMD_CHECK(aligned_ok(x, ldx, K, act_dtype) && pow2_divisor(K / 8, 256) >= 4, LORA_AMD_EINVAL,
                 "linear_bwd_x: shape/alignment not supported by the fused path (see lora_amd_linear_plan)");
  LORA_AMD_CHECK(dx == nullptr || aligned_ok(dx, lddx, K, act_dtype), LORA_AMD_EINVAL, "linear_bwd_x: dX alignment");
  const int RT = frank_tile(r);
  const BwdGeom q = bwd_geom(M, K, RT, 256);
  const unsigned grid = (unsigned)(q.nrb * q.nct);
  hipStream_t st = (hipStream_t)stream;
  const bool has_dx = dx != nullptr;
#define BX(E, RTV, D)                                                                                        \
  hipLaunchKernelGGL((linear_bwd_x_kernel<E, RTV, D>), dim3(grid), dim3(kFT), 0, st,                         \
                     reinterpret_cast<const typename E::storage *>(x), ldx, reinterpret_cast<typename E::storage *>(dx), \
                     lddx, gt_part, nct_g, down, factor_dtype, sel, down_part, M, K, r, q.log_ct8, q.nct,     \
                     q.rows_per_block)
#define BX_RT(E, D) do { if (RT == 4) BX(E, 4, D); else if (RT == 8) BX(E, 8, D); else BX(E, 16, D); } while (0)
#define BX_E(E) do { if (has_dx) BX_RT(E, true); else BX_RT(E, false); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: BX_E(f32_t); break;
    case LORA_AMD_F16: BX_E(f16_t); break;
    default: BX_E(bf16_t); break;
  }
#undef BX_E
#undef BX_RT
#undef BX
  return check_launch("lora_amd_linear_bwd_x");
}

static int bwd_factors_impl(const void *g, int64_t ldg, const float *t, float *up_part, const void *x, int64_t ldx,
                            const float *gt, const float *sel, float *down_part, int64_t M, int32_t K, int32_t N,
                            int32_t r, int32_t act_dtype, float scale, int32_t g_head_dim, int32_t g_head_pad,
                            int32_t x_head_dim, int32_t x_head_pad, float dropout_p, uint64_t seed, uint64_t offset,
                            const uint64_t *offset_dev, void *stream);

extern "C" int lora_amd_linear_bwd_factors(const void *g, int64_t ldg, const float *t, float *up_part, const void *x,
                                           int64_t ldx, const float *gt, const float *sel, float *down_part,
                                           int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale,
                                           void *stream) {
  return bwd_factors_impl(g, ldg, t, up_part, x, ldx, gt, sel, down_part, M, K, N, r, act_dtype, scale, 0, 0, 0, 0,
                          0.f, 0, 0, nullptr, stream);
}

extern "C" int lora_amd_linear_bwd_factors_drop(const void *g, int64_t ldg, const float *t, float *up_part,
                                                const void *x, int64_t ldx, const float *gt, const float *sel,
                                                float *down_part, int64_t M, int32_t K, int32_t N, int32_t r,
                                                int32_t act_dtype, float scale, float dropout_p, uint64_t seed,
                                                uint64_t offset, const uint64_t *offset_dev, void *stream) {
  return bwd_factors_impl(g, ldg, t, up_part, x, ldx, gt, sel, down_part, M, K, N, r, act_dtype, scale, 0, 0, 0, 0,
                          dropout_p, seed, offset, offset_dev, stream);
}

extern "C" int lora_amd_linear_bwd_factors_heads(const void *g, int64_t ldg, const float *t, float *up_part,
                                                 const void *x, int64_t ldx, const float *gt, const float *sel,
                                                 float *down_part, int64_t M, int32_t K, int32_t N, int32_t r,
                                                 int32_t act_dtype, float scale, int32_t g_head_dim,
                                                 int32_t g_head_pad, int32_t x_head_dim, int32_t x_head_pad,
                                                 void *stream) {
  return bwd_factors_impl(g, ldg, t, up_part, x, ldx, gt, sel, down_part, M, K, N, r, act_dtype, scale, g_head_dim,
                          g_head_pad, x_head_dim, x_head_pad, 0.f, 0, 0, nullptr, stream);
}

static int bwd_factors_impl(const void *g, int64_t ldg, const float *t, float *up_part, const void *x, int64_t ldx,
                            const float *gt, const float *sel, float *down_part, int64_t M, int32_t K, int32_t N,
                            int32_t r, int32_t act_dtype, float scale, int32_t g_head_dim, int32_t g_head_pad,
                            int32_t x_head_dim, int32_t x_head_pad, float dropout_p, uint64_t seed, uint64_t offset,
                            const uint64_t *offset_dev, void *stream) {
  FUSED_COMMON("linear_bwd_factors", act_dtype, LORA_AMD_F32);
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || g_head_dim == 0), LORA_AMD_EINVAL,
                 "linear_bwd_factors: dropout p=%f (dense G rows only)", dropout_p);
  auto heads_ok = [](int d, int D, int cols, int64_t ld) {
    return d == 0 || (d > 0 && D >= d && d % 8 == 0 && D % 8 == 0 && cols % d == 0 && ld >= (int64_t)(cols / d) * D);
  };
  LORA_AMD_CHECK(heads_ok(g_head_dim, g_head_pad, N, ldg) && heads_ok(x_head_dim, x_head_pad, K, ldx), LORA_AMD_EINVAL,
                 "linear_bwd_factors: head layout not supported");
  LORA_AMD_CHECK(g && t && up_part && x && gt && down_part, LORA_AMD_EINVAL, "linear_bwd_factors: null pointer");
  LORA_AMD_CHECK(aligned_ok(g, ldg, N, act_dtype) && pow2_divisor(N / 8, 64) >= 4 && aligned_ok(x, ldx, K, act_dtype) &&
                     pow2_divisor(K / 8, 256) >= 4,
                 LORA_AMD_EINVAL, "linear_bwd_factors: shape/alignment not supported (see lora_amd_linear_plan)");
  const int RT = frank_tile(r);
  const BwdGeom qg = bwd_geom(M, N, RT, 64), qx = bwd_geom(M, K, RT, 256);
  FactorJob a{g, ldg, t, nullptr, up_part, scale, N, qg.log_ct8, qg.nct, qg.rows_per_block, (int)(qg.nrb * qg.nct),
              g_head_dim / 8, g_head_pad / 8, dropout_p, seed, offset, offset_dev};
  FactorJob b{x, ldx, gt, sel, down_part, 1.0f, K, qx.log_ct8, qx.nct, qx.rows_per_block, (int)(qx.nrb * qx.nct),
              x_head_dim / 8, x_head_pad / 8, 0.f, 0, 0, nullptr};
  const unsigned grid = (unsigned)(a.nblocks + b.nblocks);
  hipStream_t st = (hipStream_t)stream;
#define BF(E, RTV) hipLaunchKernelGGL((linear_bwd_factors_kernel<E, RTV>), dim3(grid), dim3(kFT), 0, st, a, b, M, r)
#define BF_E(E) do { if (RT == 4) BF(E, 4); else if (RT == 8) BF(E, 8); else BF(E, 16); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: BF_E(f32_t); break;
    case LORA_AMD_F16: BF_E(f16_t); break;
    default: BF_E(bf16_t); break;
  }
#undef BF_E
#undef BF
  return check_launch("lora_amd_linear_bwd_factors");
}

// Geometry of the self-sufficient factor-gradient launch (also the size of its partial workspaces).
static bool factors_self_geom(int64_t M, int K, int N, int r, SelfArgs *a, int64_t *nrb_out, int rows_hint = 0) {
  if (M <= 0 || r < 1 || r > 16 || K % 8 || N % 8 || K < 32 || N < 32) return false;
  const int RT = frank_tile(r);
  // rows per block: one workgroup streams its rows twice (phase A, phase B); enough blocks to give every CU two, at
  // least 16 rows (the partial slabs are RT*4 / (rows*2) of the stream) and at most what the LDS row vectors hold
  int64_t rows = (M + 511) / 512;
  rows = std::max<int64_t>(rows, 16);
  if (rows_hint > 0) rows = rows_hint;  // the one-launch pass is throughput-bound: its caller asks for tall blocks
  rows = std::min<int64_t>(rows, std::min<int64_t>(kSelfRowsMax, kFLdsT / RT));
  rows = std::min<int64_t>(rows, M);
  const int64_t nrb = (M + rows - 1) / rows;
  // phase B tiles: the whole row when it has <= 256 chunks, else equal tiles of <= 256 chunks
  auto tiles = [](int c8, int *tile, int *nct) { *nct = (c8 + 255) / 256; *tile = (c8 + *nct - 1) / *nct; };
  tiles(N / 8, &a->tile_g, &a->nct_g);
  tiles(K / 8, &a->tile_x, &a->nct_x);
  int nsplit = (int)std::min<int64_t>((512 + nrb - 1) / nrb, 4);
  nsplit = std::max(1, std::min(nsplit, a->nct_g + a->nct_x));
  a->nsplit = nsplit;
  a->rows_per_block = (int)rows;
  const int kt = (kFLdsFactor / RT) & ~7;
  a->kt_g = std::min(kt, N); a->logL_g = pick_logL(a->kt_g >> 3);
  a->kt_x = std::min(kt, K); a->logL_x = pick_logL(a->kt_x >> 3);
  *nrb_out = nrb;
  return true;
}

extern "C" int lora_amd_linear_factors_self_plan(int64_t M, int32_t K, int32_t N, int32_t r,
                                                 lora_amd_factors_self_plan_t *out) {
  return lora_amd_linear_factors_self_plan_rows(M, K, N, r, 0, out);
}

extern "C" int lora_amd_linear_factors_self_plan_rows(int64_t M, int32_t K, int32_t N, int32_t r, int32_t rows,
                                                      lora_amd_factors_self_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr && rows >= 0, LORA_AMD_EINVAL, "factors_self_plan: bad argument");
  memset(out, 0, sizeof(*out));
  SelfArgs a;
  int64_t nrb = 0;
  if (!factors_self_geom(M, K, N, r, &a, &nrb, rows)) return LORA_AMD_OK;
  out->supported = 1;
  out->rank_tile = frank_tile(r);
  out->nparts = (int32_t)nrb;
  out->up_part_floats = nrb * out->rank_tile * (int64_t)N;
  out->down_part_floats = nrb * out->rank_tile * (int64_t)K;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_linear_bwd_factors_self(const void *g, int64_t ldg, const void *x, int64_t ldx,
                                                const float *down, const float *up, float *up_part, float *down_part,
                                                int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype,
                                                float scale, int32_t g_head_dim, int32_t g_head_pad,
                                                int32_t x_head_dim, int32_t x_head_pad, void *stream) {
  FUSED_COMMON("linear_bwd_factors_self", act_dtype, LORA_AMD_F32);
  auto heads_ok = [](int d, int D, int cols, int64_t ld) {
    return d == 0 || (d > 0 && D >= d && d % 8 == 0 && D % 8 == 0 && cols % d == 0 && ld >= (int64_t)(cols / d) * D);
  };
  LORA_AMD_CHECK(heads_ok(g_head_dim, g_head_pad, N, ldg) && heads_ok(x_head_dim, x_head_pad, K, ldx), LORA_AMD_EINVAL,
                 "linear_bwd_factors_self: head layout not supported");
  LORA_AMD_CHECK(g && x && down && up && up_part && down_part, LORA_AMD_EINVAL, "linear_bwd_factors_self: null pointer");
  SelfArgs a;
  int64_t nrb = 0;
  LORA_AMD_CHECK(factors_self_geom(M, K, N, r, &a, &nrb) && aligned_ok(g, ldg, N, act_dtype) &&
                     aligned_ok(x, ldx, K, act_dtype) && ((uintptr_t)down % 16) == 0 && ((uintptr_t)up % 16) == 0,
                 LORA_AMD_EINVAL, "linear_bwd_factors_self: shape/alignment not supported (see lora_amd_linear_factors_self_plan)");
  a.g = g; a.x = x; a.ldg = ldg; a.ldx = ldx; a.M = M; a.down = down; a.up = up; a.up_part = up_part;
  a.down_part = down_part; a.scale = scale; a.N = N; a.K = K; a.r = r;
  a.ghc = g_head_dim / 8; a.ghp = g_head_pad / 8; a.xhc = x_head_dim / 8; a.xhp = x_head_pad / 8;
  const unsigned grid = (unsigned)(nrb * a.nsplit);
  const int RT = frank_tile(r);
  hipStream_t st = (hipStream_t)stream;
  // both factor slabs in LDS at once -> the wave-specialised kernel (one row block per workgroup, no column splits)
  const bool dual = (int64_t)RT * (N + K) <= kSelfLdsFloats && a.logL_x >= 0;
#define FS(E, RTV)                                                                                                    \
  do {                                                                                                                \
    if (dual) hipLaunchKernelGGL((linear_bwd_factors_self_dual_kernel<E, RTV>), dim3((unsigned)nrb), dim3(kDualThreads), 0, st, a); \
    else hipLaunchKernelGGL((linear_bwd_factors_self_kernel<E, RTV>), dim3(grid), dim3(kFT), 0, st, a);                \
  } while (0)
#define FS_E(E) do { if (RT == 4) FS(E, 4); else if (RT == 8) FS(E, 8); else FS(E, 16); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: FS_E(f32_t); break;
    case LORA_AMD_F16: FS_E(f16_t); break;
    default: FS_E(bf16_t); break;
  }
#undef FS_E
#undef FS
  return check_launch("lora_amd_linear_bwd_factors_self");
}

extern "C" int lora_amd_linear_factors_self_ragged_plan(lora_amd_self_site *sites, int32_t n, int32_t act_dtype,
                                                        int64_t *grid) {
  LORA_AMD_CHECK(sites && n >= 1 && grid && dtype_ok(act_dtype), LORA_AMD_EINVAL, "factors_self_ragged_plan: bad argument");
  auto heads_ok = [](int d, int D, int cols, int64_t ld) {
    return d == 0 || (d > 0 && D >= d && d % 8 == 0 && D % 8 == 0 && cols % d == 0 && ld >= (int64_t)(cols / d) * D);
  };
  int64_t begin = 0;
  const int rt0 = frank_tile(sites[0].r);
  for (int i = 0; i < n; ++i) {
    lora_amd_self_site &q = sites[i];
    SelfArgs a;
    int64_t nrb = 0;
    LORA_AMD_CHECK(q.r >= 1 && q.r <= 16 && frank_tile(q.r) == rt0, LORA_AMD_ERANK,
                   "factors_self_ragged_plan: site %d: rank %d (one rank tile per table)", i, q.r);
    LORA_AMD_CHECK(q.g && q.x && q.down && q.up && q.up_part && q.down_part, LORA_AMD_EINVAL,
                   "factors_self_ragged_plan: site %d: null pointer", i);
    LORA_AMD_CHECK(factors_self_geom(q.M, q.K, q.N, q.r, &a, &nrb, q.rows_per_block) && aligned_ok(q.g, q.ldg, q.N, act_dtype) &&
                       aligned_ok(q.x, q.ldx, q.K, act_dtype) && ((uintptr_t)q.down % 16) == 0 && ((uintptr_t)q.up % 16) == 0 &&
                       heads_ok(q.g_head_dim, q.g_head_pad, q.N, q.ldg) && heads_ok(q.x_head_dim, q.x_head_pad, q.K, q.ldx),
                   LORA_AMD_EINVAL, "factors_self_ragged_plan: site %d: shape / alignment / head layout not supported", i);
    // throughput-bound launch: no column splits (they redo phase A), the row blocks of all sites fill the chip
    q.rows_per_block = a.rows_per_block; q.nsplit = 1;
    q.kt_g = a.kt_g; q.logL_g = a.logL_g; q.kt_x = a.kt_x; q.logL_x = a.logL_x;
    q.tile_g = a.tile_g; q.nct_g = a.nct_g; q.tile_x = a.tile_x; q.nct_x = a.nct_x;
    q.block_begin = begin;
    begin += nrb;
  }
  LORA_AMD_CHECK(begin < (1ll << 31), LORA_AMD_EINVAL, "factors_self_ragged_plan: too many blocks");
  *grid = begin;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_linear_bwd_factors_self_ragged(const lora_amd_self_site *sites_dev, int32_t n, int64_t grid,
                                                       int32_t rank, int32_t act_dtype, void *stream) {
  LORA_AMD_CHECK(sites_dev && n >= 1 && grid >= 1 && grid < (1ll << 31) && dtype_ok(act_dtype), LORA_AMD_EINVAL,
                 "linear_bwd_factors_self_ragged: bad argument");
  LORA_AMD_CHECK(rank >= 1 && rank <= 16, LORA_AMD_ERANK, "linear_bwd_factors_self_ragged: rank %d outside [1,16]", rank);
  const int RT = frank_tile(rank);
  hipStream_t st = (hipStream_t)stream;
  // loads in flight per lane: 4 measured best in the one-launch pass (8 costs occupancy: 1.9 vs ~1.1 ms on configs[1])
#define FR(E, RTV) hipLaunchKernelGGL((linear_bwd_factors_self_ragged_kernel<E, RTV, 4>), dim3((unsigned)grid), dim3(kFT), 0, st, sites_dev, n)
#define FR_E(E) do { if (RT == 4) FR(E, 4); else if (RT == 8) FR(E, 8); else FR(E, 16); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: FR_E(f32_t); break;
    case LORA_AMD_F16: FR_E(f16_t); break;
    default: FR_E(bf16_t); break;
  }
#undef FR_E
#undef FR
  return check_launch("lora_amd_linear_bwd_factors_self_ragged");
}

extern "C" int lora_amd_reduce_batched(const lora_amd_reduce_desc *descs_dev, int32_t n, int64_t total,
                                       void *stream) {
  LORA_AMD_CHECK(n >= 0 && total >= 0, LORA_AMD_EINVAL, "reduce_batched: bad sizes");
  if (n == 0 || total == 0) return LORA_AMD_OK;
  LORA_AMD_CHECK(descs_dev != nullptr, LORA_AMD_EINVAL, "reduce_batched: null table");
  const int grid = (int)std::min<int64_t>((total + kFT - 1) / kFT, 8192);
  hipLaunchKernelGGL(reduce_batched_kernel, dim3(grid), dim3(kFT), 0, (hipStream_t)stream, descs_dev, n, total);
  return check_launch("lora_amd_reduce_batched");
}
