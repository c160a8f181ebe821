// Rank-tile-16 forms of the adapter's streaming kernels on the matrix cores (ranks 9..16, bf16 activations, f32 factors).
//
// replaces: csrc/linear.hip's rowdot / rank_update kernels and csrc/linear_fused.hip's linear_bwd_g_kernel<E, 16> for
//           lora_diffusion/lora.py:53-58 (Linear) and lora.py:130-135 (Conv2d, channels-last rows) at rank 16 — BASELINE
//           configs[3].  At rank 16 the VALU forms spend 16-32 FMAs per element and sit at 0.1-0.2 of the byte roof
//           (profiles/r04_cfg3_by_grid_before_rank16.txt: 25-110 us per launch whatever the size); the contractions are dense enough for
//           v_mfma_f32_16x16x32 (rank padded to the 16 of the tile), which leaves the dropout mask (one Philox call per
//           16-byte chunk) as the only per-element VALU work.
//
// All three kernels are WAVE-AUTONOMOUS: a wave owns a 16-row (rowdot, rank_update) or 32-row x 32-column (bwd_g) unit,
// takes the 16-byte piece a lane loads from a row-major row straight as the MFMA A operand (lane = row (l & 15), columns
// 8 (l >> 4) ..), and never meets another wave inside its loop — no workgroup barrier between the prologue and the epilogue.
// Operands that are not data (factors, T, Gt) are split hi + lo into two fragments of the same accumulator (f32-grade).
//
//   rowdot16:       T[M, r]  = s (mask . X)[M, C] F^T         wave = 16 rows x all C; fragments of F built per k-step
//   rank_update16:  Y[M, N] += s mask . (T[M, r] F)           workgroup = 64 rows; wave = its 32-column groups; the product
//                   is taken transposed, D[n, m] = sum_j U[n, j] T[m, j], with the rows of the two 16 x 16 tiles interleaved
//                   (tile t row i <-> n0 + 8 (i / 4) + 4 t + i % 4) so that a lane ends up with 8 consecutive columns of one
//                   row: one 16-byte read-modify-write per lane.  k = 32 holds (hi | lo) of the 16 ranks of U against T's hi
//                   twice, then T's lo twice: two MFMAs per tile.
//   bwd_g16:        gt_part[ct][M, r] = s (mask . G)[., column tile] up ;  up_part[rb][16][N] = s (mask . G)^T T (block rows)
//                   same outputs and launch geometry as linear_bwd_g_kernel.  Per unit: phase 1 straight from the registers,
//                   then the two pieces go through the wave's own 32 x 32 LDS tile and come back column-major through
//                   ds_read_b64_tr_b16 for phase 2 (A = G^T, B = T); the LDS queue of a wave is in order, so write -> transpose
//                   read -> next write needs no wait beyond the data dependence.
//
// The same row product also serves cli_svd (lora_diffusion/cli_svd.py:24-92 as restated in lora_amd/cli_svd.py): the residuals
// dW = W_tuned - W_base are f32, which has no full-rate matrix-core path on gfx950, so they are held as TWO bf16 planes
// (dW = hi + lo to ~16 mantissa bits: the bytes of f32) written by split16_transpose_kernel (one read of dW -> the planes of
// dW and of dW^T), and rowdot16_planes_kernel computes out = X F as hi F_hi + hi F_lo + lo F_hi: 3.6 TB/s per pass.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "mfma16.hpp"
#include "rank16_mfma.hpp"

namespace lora_amd {

int g_r16_mfma = 1;

constexpr int kR16Threads = 256;
constexpr int kR16Pitch = 96;  // bytes per row of a wave's 32 x 32 16-bit tile: ds_write_b128 and the transpose reads conflict-free

__device__ __forceinline__ mu32x4 r16_zero() { return mu32x4{0u, 0u, 0u, 0u}; }

// B-operand fragment pair (hi, lo) of the factor for the 32 columns c0 .. c0 + 31: lane (j = l & 15, kq = l >> 4) holds
// mult * F[j, c0 + 8 kq + e].  layout RK: F = [r, C] (8 consecutive floats); KR: F = [C, r] (stride r).  Two steps, so that
// a loop can have the next k-step's values in flight: r16_factor_load (global -> 8 floats), r16_factor_split.
struct R16Raw { float v[8]; };
__device__ __forceinline__ R16Raw r16_factor_load(const float *__restrict__ f, int layout, int r, int C, int c0) {
  const int lane = threadIdx.x & 63, jj = lane & 15, kq = lane >> 4;
  const int c = c0 + 8 * kq;
  R16Raw w;
#pragma unroll
  for (int e = 0; e < 8; ++e) w.v[e] = 0.f;
  if (jj < r && c < C) {  // C % 8 == 0: the chunk is inside or outside as a whole
    if (layout == LORA_AMD_FACTOR_RK) {
      const float4 a = gl_ld4(f + (int64_t)jj * C + c), b = gl_ld4(f + (int64_t)jj * C + c + 4);
      w.v[0] = a.x; w.v[1] = a.y; w.v[2] = a.z; w.v[3] = a.w; w.v[4] = b.x; w.v[5] = b.y; w.v[6] = b.z; w.v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) w.v[e] = gl(f)[(int64_t)(c + e) * r + jj];
    }
  }
  return w;
}
template <class E>
__device__ __forceinline__ void r16_factor_split(const R16Raw &w, float mult, mu32x4 &hi, mu32x4 &lo) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = w.v[e] * mult;
  split_hi_lo<E>(v, hi, lo);
}
template <class E>
__device__ __forceinline__ void r16_factor_frag(const float *__restrict__ f, int layout, int r, int C, int c0, float mult,
                                                mu32x4 &hi, mu32x4 &lo) {
  r16_factor_split<E>(r16_factor_load(f, layout, r, C, c0), mult, hi, lo);
}

// ============================================================================ rowdot16
// A workgroup = `slabs` 16-row slabs x `wps` waves per slab (slabs * wps = blockDim / 64); the waves of a slab take the
// k-steps ks = cw, cw + wps, ... and meet once in LDS.  Two data pieces and one factor fragment ahead of the MFMAs.
template <class E, bool DROP>
__global__ __launch_bounds__(1024) void rowdot16_mfma_kernel(
    const typename E::storage *__restrict__ x, int64_t ldx, const float *__restrict__ f, int layout, float *__restrict__ t_out,
    int64_t M, int C, int r, int wps, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  using S = typename E::storage;
  __shared__ __attribute__((aligned(16))) float s_red[16 * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, jj = lane & 15, q = lane >> 4;
  const int nwaves = blockDim.x >> 6, slabs = nwaves / wps;
  const int sl = wave / wps, cw = wave - sl * wps;
  const int64_t m0 = ((int64_t)blockIdx.x * slabs + sl) * 16;
  float sc = scale;
  uint32_t thr = 0;
  uint64_t off = 0;
  if constexpr (DROP) {
    sc = scale * (1.0f / (1.0f - p));
    thr = (uint32_t)(p * 65536.0f + 0.5f);
    off = dropout_offset(offset, offset_dev);
  }
  const int64_t row = m0 + jj;
  const bool rok = row < M;
  const S *xr = x + (rok ? row : 0) * ldx + 8 * q;
  const int nks = C >> 5;  // C % 32 == 0 (host check)
  mf32x4 d = {0.f, 0.f, 0.f, 0.f};
  if (m0 < M) {
    auto piece = [&](int ks) -> mu32x4 {
      return ks < nks ? *gl(reinterpret_cast<const mu32x4 *>(xr + (int64_t)ks * 32)) : r16_zero();
    };
    mu32x4 c0 = piece(cw), c1 = piece(cw + wps);
    R16Raw fr = r16_factor_load(f, layout, r, C, (cw < nks ? cw : 0) * 32);
#pragma unroll 1
    for (int ks = cw; ks < nks; ks += wps) {
      const mu32x4 c2 = piece(ks + 2 * wps);
      mu32x4 fh, fl;
      r16_factor_split<E>(fr, sc, fh, fl);
      if (ks + wps < nks) fr = r16_factor_load(f, layout, r, C, (ks + wps) * 32);
      mu32x4 cur = c0;
      if constexpr (DROP) cur &= dropout_and8(seed, off, (uint64_t)(row * (int64_t)(C >> 3) + ks * 4 + q), thr);
      if (!rok) cur = r16_zero();
      d = FmMfma<E>::mma(fm_frag<E>(cur), fm_frag<E>(fh), d);
      d = FmMfma<E>::mma(fm_frag<E>(cur), fm_frag<E>(fl), d);
      c0 = c1;
      c1 = c2;
    }
  }
  if (wps > 1) {  // block-uniform
    if (cw > 0) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) s_red[(wave * 4 + reg) * 64 + lane] = d[reg];
    }
    __syncthreads();
    if (cw > 0) return;
    for (int k = 1; k < wps; ++k)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) d[reg] += s_red[((wave + k) * 4 + reg) * 64 + lane];
  }
  // D: lane (column j = jj, rows 4 q + reg)
  if (jj < r && m0 < M) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t rw = m0 + 4 * q + reg;
      if (rw < M) t_out[rw * r + jj] = d[reg];
    }
  }
}

// ============================================================================ rank_update16
// grid (row blocks of `rows_per_block` = 64 | 128 | 256, column splits of 128 columns); a wave owns ONE 32-column group:
// its two A fragments are built once and meet every 16-row slab of the block, whose T fragments all four waves share
// through LDS.
template <class E, bool DROP>
__global__ __launch_bounds__(kR16Threads) void rank_update16_mfma_kernel(
    typename E::storage *__restrict__ y, int64_t ldy, const float *__restrict__ t, int nparts, int64_t part_stride,
    const float *__restrict__ f, int layout, int64_t M, int N, int r, int rows_per_block, float scale, float p, uint64_t seed,
    uint64_t offset, const uint64_t *offset_dev) {
  __shared__ __attribute__((aligned(16))) mu32x4 s_t[16 * 2 * 64];  // [slab][hi, lo][lane]: B operands (32 KB)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mm = lane & 15, q = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int nslabs = (nrows + 15) >> 4;
  uint64_t off = 0;
  if constexpr (DROP) off = dropout_offset(offset, offset_dev);
  // B operands: T of the block's 16-row slabs, (hi | hi) and (lo | lo) along k
  for (int s = wave; s < nslabs; s += 4) {
    const int64_t row = m0 + s * 16 + mm;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (row < M) {
      const int j0 = 8 * (q & 1);
      for (int pi = 0; pi < nparts; ++pi) {
        const float *tp = t + pi * part_stride + row * r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (j0 + e < r) v[e] += gl(tp)[j0 + e];
      }
    }
    mu32x4 hi, lo;
    split_hi_lo<E>(v, hi, lo);
    s_t[(s * 2 + 0) * 64 + lane] = hi;
    s_t[(s * 2 + 1) * 64 + lane] = lo;
  }
  const int n0 = (blockIdx.y * 4 + wave) * 32;
  // A operands of the two interleaved tiles: lane (i = mm, kq = q) holds (hi for kq < 2, lo for kq >= 2) of
  // U[n(i, tile), 8 (kq & 1) + e], n(i, tile) = n0 + 8 (i / 4) + 4 tile + i % 4
  mu32x4 ua[2];
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    const int n = n0 + 8 * (mm >> 2) + 4 * tile + (mm & 3);
    const int j0 = 8 * (q & 1);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (n < N) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (j0 + e < r) v[e] = layout == LORA_AMD_FACTOR_KR ? gl(f)[(int64_t)n * r + j0 + e] : gl(f)[(int64_t)(j0 + e) * N + n];
    }
    mu32x4 hi, lo;
    split_hi_lo<E>(v, hi, lo);
    ua[tile] = q < 2 ? hi : lo;
  }
  __syncthreads();
  const int col = n0 + 8 * q;  // this lane's 16-byte chunk of each row
  const bool cok = col < N;
  if (n0 >= N) return;         // wave-uniform (no barrier below)
  auto yload = [&](int s) -> mu32x4 {
    const int64_t row = m0 + s * 16 + mm;
    const bool ok = cok && row < M && s < nslabs;
    return *gl(reinterpret_cast<const mu32x4 *>(y + (ok ? row : m0) * ldy + (ok ? col : 0)));
  };
  mu32x4 y0 = yload(0), y1 = yload(1);
#pragma unroll 1
  for (int s = 0; s < nslabs; ++s) {
    const mu32x4 y2 = yload(s + 2);
    const mu32x4 th = s_t[(s * 2 + 0) * 64 + lane], tl = s_t[(s * 2 + 1) * 64 + lane];
    const int64_t row = m0 + s * 16 + mm;
    mf32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    d0 = FmMfma<E>::mma(fm_frag<E>(ua[0]), fm_frag<E>(th), d0);
    d0 = FmMfma<E>::mma(fm_frag<E>(ua[0]), fm_frag<E>(tl), d0);
    d1 = FmMfma<E>::mma(fm_frag<E>(ua[1]), fm_frag<E>(th), d1);
    d1 = FmMfma<E>::mma(fm_frag<E>(ua[1]), fm_frag<E>(tl), d1);
    // lane (row m = mm, q): d0[reg] -> column col + reg, d1[reg] -> column col + 4 + reg
    float pr[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
    if constexpr (DROP) {
      float mk[8];
      dropout_mult8(seed, off, (uint64_t)((row * (int64_t)N + col) >> 3), p, mk);
#pragma unroll
      for (int e = 0; e < 8; ++e) pr[e] *= mk[e];
    }
    if (cok && row < M) {
      union { mu32x4 u; Chunk8<E> c; } in, out;
      in.u = y0;
#pragma unroll
      for (int e = 0; e < 8; ++e) out.c.v[e] = E::from_f(fmaf(scale, pr[e], E::to_f(in.c.v[e])));
      *gl(reinterpret_cast<mu32x4 *>(y + row * ldy + col)) = out.u;
    }
    y0 = y1;
    y1 = y2;
  }
}

// ============================================================================ bwd_g16
template <class E, bool DROP>
__global__ __launch_bounds__(kR16Threads) void bwd_g16_mfma_kernel(
    const typename E::storage *__restrict__ g, int64_t ldg, const float *__restrict__ t, const float *__restrict__ up,
    float *__restrict__ gt_part, float *__restrict__ up_part, int64_t M, int N, int r, int log_ct8, int nct,
    int rows_per_block, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
    float *__restrict__ gt_out, unsigned *__restrict__ counters) {
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[4 * 32 * kR16Pitch];  // a 32 x 32 tile per wave (12 KB)
  __shared__ int s_last;
  __shared__ __attribute__((aligned(16))) mu32x4 s_tf[4 * 2 * 64];                     // T fragments [row step][hi, lo][lane]
  __shared__ __attribute__((aligned(16))) float s_gtw[4 * 128 * 16];                   // Gt partial of each column wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 15, q = lane >> 4;
  const int CW = 8 << log_ct8, ngroups = CW >> 5;     // 32-column groups of the tile: 1, 2, 4, 8, 16
  const int GW = ngroups < 4 ? ngroups : 4, RW = 4 / GW;  // waves across the groups x waves across the 32-row steps
  const int cgw = wave % GW, rw = wave / GW;
  const int ncg = ngroups / GW;                       // groups per wave: 1, 2, 4
  const int64_t rb = blockIdx.x / nct;
  const int ct = (int)(blockIdx.x - rb * nct);
  const int64_t m0 = rb * rows_per_block;
  const int nrows = (int)min((int64_t)rows_per_block, M - m0);
  const int nsteps = (nrows + 31) >> 5;
  const int col0 = ct * CW;
  float sc = scale;
  uint32_t thr = 0;
  uint64_t off = 0;
  if constexpr (DROP) {
    sc = scale * (1.0f / (1.0f - p));
    thr = (uint32_t)(p * 65536.0f + 0.5f);
    off = dropout_offset(offset, offset_dev);
  }
  // T fragments of the block's row steps (B operand of phase 2): k slot 8 q + e <-> row 4 q + e (e < 4) / 16 + 4 q + e - 4
  // of the step — the order the two transpose reads deliver G's rows in
  if (wave < nsteps) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = wave * 32 + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
      v[e] = (row < nrows && jj < r) ? sc * gl(t)[(m0 + row) * r + jj] : 0.f;
    }
    mu32x4 hi, lo;
    split_hi_lo<E>(v, hi, lo);
    s_tf[(wave * 2 + 0) * 64 + lane] = hi;
    s_tf[(wave * 2 + 1) * 64 + lane] = lo;
  }
  // `up` fragments of this wave's column groups (B operand of phase 1), scaled
  mu32x4 uh[4], ul[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uh[c] = ul[c] = r16_zero();
    if (c < ncg) r16_factor_frag<E>(up, LORA_AMD_FACTOR_KR, r, N, col0 + (cgw + GW * c) * 32, sc, uh[c], ul[c]);
  }
  __syncthreads();

  auto piece = [&](int rs, int c, int rg) -> mu32x4 {
    const int row = rs * 32 + rg * 16 + jj, col = col0 + (cgw + GW * c) * 32 + 8 * q;
    const bool ok = row < nrows && col < N;
    return *gl(reinterpret_cast<const mu32x4 *>(g + (m0 + (ok ? row : 0)) * ldg + (ok ? col : 0)));
  };
  auto finish = [&](mu32x4 v, int rs, int c, int rg) -> mu32x4 {  // out-of-range pieces -> 0, the forward's mask
    const int row = rs * 32 + rg * 16 + jj, col = col0 + (cgw + GW * c) * 32 + 8 * q;
    if (!(row < nrows && col < N)) return r16_zero();
    if constexpr (DROP) v &= dropout_and8(seed, off, (uint64_t)((m0 + row) * (int64_t)(N >> 3) + (col >> 3)), thr);
    return v;
  };
  mf32x4 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c][0] = acc[c][1] = mf32x4{0.f, 0.f, 0.f, 0.f};
  unsigned char *stage = s_stage + wave * 32 * kR16Pitch;
  mu32x4 cur0 = r16_zero(), cur1 = r16_zero();
  if (rw < nsteps) { cur0 = piece(rw, 0, 0); cur1 = piece(rw, 0, 1); }
#pragma unroll 1
  for (int rs = rw; rs < nsteps; rs += RW) {
    const mu32x4 tfh = s_tf[(rs * 2 + 0) * 64 + lane], tfl = s_tf[(rs * 2 + 1) * 64 + lane];
    mf32x4 d1[2] = {mf32x4{0.f, 0.f, 0.f, 0.f}, mf32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < ncg) {
        const bool last = c + 1 >= ncg;
        const int prs = last ? rs + RW : rs, pc = last ? 0 : c + 1;
        mu32x4 nx0 = r16_zero(), nx1 = r16_zero();
        if (prs < nsteps) { nx0 = piece(prs, pc, 0); nx1 = piece(prs, pc, 1); }
        cur0 = finish(cur0, rs, c, 0);
        cur1 = finish(cur1, rs, c, 1);
        d1[0] = FmMfma<E>::mma(fm_frag<E>(cur0), fm_frag<E>(uh[c]), d1[0]);
        d1[0] = FmMfma<E>::mma(fm_frag<E>(cur0), fm_frag<E>(ul[c]), d1[0]);
        d1[1] = FmMfma<E>::mma(fm_frag<E>(cur1), fm_frag<E>(uh[c]), d1[1]);
        d1[1] = FmMfma<E>::mma(fm_frag<E>(cur1), fm_frag<E>(ul[c]), d1[1]);
        *reinterpret_cast<mu32x4 *>(stage + jj * kR16Pitch + q * 16) = cur0;
        *reinterpret_cast<mu32x4 *>(stage + (16 + jj) * kR16Pitch + q * 16) = cur1;
        asm volatile("" ::: "memory");  // the LDS queue of a wave is in order: only the compiler must keep the order
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const unsigned char *pp = stage + (4 * q + (jj >> 2)) * kR16Pitch + (16 * nt + 4 * (jj & 3)) * 2;
          union { ms16x4 h[2]; mu32x4 u; } a;
          a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(pp));
          a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(pp + 16 * kR16Pitch));
          acc[c][nt] = FmMfma<E>::mma(fm_frag<E>(a.u), fm_frag<E>(tfh), acc[c][nt]);
          acc[c][nt] = FmMfma<E>::mma(fm_frag<E>(a.u), fm_frag<E>(tfl), acc[c][nt]);
        }
        asm volatile("" ::: "memory");
        cur0 = nx0;
        cur1 = nx1;
      }
    }
    // this wave's share of Gt for the step's 32 rows: D1 lane (j = jj, rows 4 q + reg)
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = rs * 32 + rg * 16 + 4 * q + reg;
        if (row < nrows) s_gtw[(cgw * 128 + row) * 16 + jj] = d1[rg][reg];
      }
  }
  __syncthreads();
  if (gt_part != nullptr) {
    // gt_out (round 6): the column-tile partials of a row block are folded INSIDE the launch by the block's last-arriving
    // column workgroup (csrc/svd_small.hip's hand-off: write-through partials, vmcnt(0), relaxed agent-scope arrival, one
    // acquire fence in the last arriver, which also resets the counter) — the separate lora_amd_sum_parts launch of every
    // conv / Linear dropout site's backward is gone
    const bool fold = gt_out != nullptr;
    float *gtp = (fold && nct == 1 ? gt_out : gt_part + (int64_t)ct * M * r) + m0 * r;
    for (int i = tid; i < nrows * r; i += kR16Threads) {
      const int row = i / r, j = i - row * r;
      float sum = 0.f;
      for (int w = 0; w < GW; ++w) sum += s_gtw[(w * 128 + row) * 16 + j];
      if (fold && nct > 1) __hip_atomic_store(gtp + i, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else gtp[i] = sum;
    }
    if (fold && nct > 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(counters + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (unsigned)(nct - 1);
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(counters + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
      }
      __syncthreads();
      if (s_last) {   // the tiles in order: the same sum whichever workgroup arrives last
        const float *src = gt_part + m0 * r;
        float *dst = gt_out + m0 * r;
        for (int i = tid; i < nrows * r; i += kR16Threads) {
          float sum = 0.f;
          for (int c = 0; c < nct; ++c) sum += gl(src)[(int64_t)c * M * r + i];
          dst[i] = sum;
        }
      }
    }
  }
  if (RW > 1) {  // ncg == 1: the row waves of a column group meet in LDS (the staging tiles are free now)
    float *red = reinterpret_cast<float *>(s_stage);
    if (rw > 0) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[((wave - GW) * 8 + nt * 4 + reg) * 64 + lane] = acc[0][nt][reg];
    }
    __syncthreads();
    if (rw > 0) return;
    for (int k = 1; k < RW; ++k)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[0][nt][reg] += red[((cgw + GW * k - GW) * 8 + nt * 4 + reg) * 64 + lane];
  }
  // D2: lane (j = jj, n = group base + 16 nt + 4 q + reg): 16 bytes of row j of the block's slab
  float *slab = up_part + (int64_t)rb * 16 * N + (int64_t)jj * N;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c < ncg) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int n = col0 + (cgw + GW * c) * 32 + 16 * nt + 4 * q;
        if (n < N) *gl(reinterpret_cast<mf32x4 *>(slab + n)) = acc[c][nt];
      }
    }
  }
}

// ============================================================================ rowdot16 over hi + lo planes (cli_svd)
// out[b][M, r] = X[b][M, C] F[b][C, r] for stacks of f32 matrices held as TWO 16-bit planes (X = hi + lo to ~16 mantissa
// bits; same bytes as f32): the skinny products of the randomized subspace iteration (cli_svd.py:24-92 restated in
// lora_amd/cli_svd.py) on the matrix cores — hi F_hi + hi F_lo + lo F_hi per k-step.  One launch for every shape group of
// a model (descriptor table); a workgroup = 16 waves = 16 / wps slabs of 16 rows, wps waves per slab dealing the k-steps.
// PK: the factor arrives as PACKED hi / lo fragments (lora_amd_thin_pack: [k step][hi 1 KB | lo 1 KB], lane l's 16 bytes at
// 16 l) instead of f32 [C][r]: two coalesced 16-byte loads per k-step where the f32 form needs eight strided scalar loads and
// ~50 VALU instructions of hi / lo splitting — per 16-row slab, i.e. once per three MFMAs (round 5: the pass was issue-bound,
// not byte-bound, at 3.6 TB/s).
// LO = false (round 6): the hi plane alone (X ~ hi to 8 mantissa bits) at half the bytes — the power iterations of the
// subspace iteration only steer a subspace (an O(2^-9) perturbation of it costs the rank-r Frobenius error to second order);
// the pass that forms the returned factors (b = Q^T dW) reads both planes.
// Slabs per wave (third session of round 6): a k-step's packed factor fragments are 2 KB (hi | lo) against 1 KB of hi-plane data
// per 16-row slab — two thirds of what a wave pulls through the CU's load path were fragments.  A wave now multiplies kPlNS
// consecutive slabs against each fragment pair (the plan sizes the workgroups: slabs_per_wg = kPlNS * 16 / wps).
constexpr int kPlNS = 2;
template <class E, bool PK, bool LO = true, bool ROWLD = false>
__global__ __launch_bounds__(1024) void rowdot16_planes_kernel(const lora_amd_planes_desc *__restrict__ descs, int n, int r) {
  using S = typename E::storage;
  constexpr int NS = kPlNS;
  __shared__ __attribute__((aligned(16))) float s_red[16 * NS * 4 * 64];
  __shared__ int64_t s_begin[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jj = lane & 15, q = lane >> 4;
  int lo = 0, hi = n - 1;
  if (n <= 64) {
    if (tid < n) s_begin[tid] = descs[tid].wg_begin;
    __syncthreads();
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_begin[mid] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (descs[mid].wg_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  }
  const lora_amd_planes_desc d = descs[lo];
  const int wps = d.wps, spw = NS * (16 / wps);   // == d.slabs_per_wg (lora_amd_rowdot16_planes_plan)
  const int64_t nslabs = (d.M + 15) >> 4, wgs_per = (nslabs + spw - 1) / spw;
  const int64_t wg = (int64_t)blockIdx.x - d.wg_begin;
  const int64_t b = wg / wgs_per;
  const int sl = wave / wps, cw = wave - sl * wps;
  const int64_t slab0 = (wg - b * wgs_per) * spw + (int64_t)sl * NS;   // this wave's NS consecutive slabs
  const int C = d.C, nks = C >> 5;
  const float *f = d.f + b * (int64_t)C * r;
  const S *pk = reinterpret_cast<const S *>(d.f) + b * (int64_t)C * 32 + lane * 8;   // PK: [C / 32][hi 512 | lo 512] elements
  mf32x4 accs[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) accs[s] = mf32x4{0.f, 0.f, 0.f, 0.f};
  const bool active = slab0 < nslabs;
  if constexpr (!PK) {
    // the f32-factor form (tests, the non-thin path): slab by slab
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int64_t m0 = (slab0 + s) * 16;
      const int64_t row = m0 + jj;
      const bool rok = slab0 + s < nslabs && row < d.M;
      const int64_t base = (b * d.M + (rok ? row : 0)) * (int64_t)C + 8 * q;
      const S *xh = reinterpret_cast<const S *>(d.hi) + base, *xl = reinterpret_cast<const S *>(d.lo) + base;
      auto piece = [&](const S *p, int ks) -> mu32x4 {
        return ks < nks ? *gl(reinterpret_cast<const mu32x4 *>(p + (int64_t)ks * 32)) : r16_zero();
      };
      mf32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (slab0 + s < nslabs) {
        mu32x4 h0 = piece(xh, cw), l0 = piece(xl, cw), h1 = piece(xh, cw + wps), l1 = piece(xl, cw + wps);
        R16Raw fr = r16_factor_load(f, LORA_AMD_FACTOR_KR, r, C, (cw < nks ? cw : 0) * 32);
#pragma unroll 1
        for (int ks = cw; ks < nks; ks += wps) {
          const mu32x4 h2 = piece(xh, ks + 2 * wps), l2 = piece(xl, ks + 2 * wps);
          mu32x4 fh, fl;
          r16_factor_split<E>(fr, 1.0f, fh, fl);
          if (ks + wps < nks) fr = r16_factor_load(f, LORA_AMD_FACTOR_KR, r, C, (ks + wps) * 32);
          mu32x4 ch = h0, cl = l0;
          if (!rok) { ch = r16_zero(); cl = r16_zero(); }
          acc = FmMfma<E>::mma(fm_frag<E>(ch), fm_frag<E>(fh), acc);
          acc = FmMfma<E>::mma(fm_frag<E>(ch), fm_frag<E>(fl), acc);
          acc = FmMfma<E>::mma(fm_frag<E>(cl), fm_frag<E>(fh), acc);
          h0 = h1; l0 = l1; h1 = h2; l1 = l2;
        }
      }
      accs[s] = acc;
    }
  }
  if (PK && active) {
    auto piece = [&](const S *p, int ks) -> mu32x4 {
      return ks < nks ? *gl(reinterpret_cast<const mu32x4 *>(p + (int64_t)ks * 32)) : r16_zero();
    };
    {
      // Round 6: a piece (16 rows x 64 bytes) is FETCHED four lanes per row (row l >> 2, chunk l & 3) and brought into the MFMA
      // operand order (row l & 15, chunk l >> 4) by four ds_bpermute_b32: the operand order's 64 scattered 16-byte accesses go
      // through the L1 tag pipeline one lane per cycle (scripts/ld_shape_probe.hip: 9.7 TB/s from cache against 23-28 for
      // this order).  Same box: 557 -> 503 us for the hi-plane pass, 911 -> 806 us for both planes (call c28).
      // (The factor pass's other finding — static slots instead of the rotating h0 = h1 = h2 — does NOT carry over: a wave
      // here has two or three k-steps, an unrolled three-slot loop runs up to two dead ones; it measured 25 % slower.)
      const int lr = ROWLD ? lane >> 2 : jj, lc = ROWLD ? lane & 3 : q;
      const S *ph[NS], *pl[NS];
      bool lrok[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int64_t lrow = (slab0 + s) * 16 + lr;
        lrok[s] = slab0 + s < nslabs && lrow < d.M;
        const int64_t lbase = (b * d.M + (lrok[s] ? lrow : 0)) * (int64_t)C + 8 * lc;
        ph[s] = reinterpret_cast<const S *>(d.hi) + lbase;
        pl[s] = reinterpret_cast<const S *>(d.lo) + lbase;
      }
      const int src = ((jj << 2) | q) << 2;   // byte address of the source lane for ds_bpermute: lane (row jj, chunk q) = 4 jj + q
      auto to_operand = [&](mu32x4 v) -> mu32x4 {
        if constexpr (!ROWLD) return v;
        mu32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v[i]);
        return o;
      };
      auto lpiece_lo = [&](int s, int ks) -> mu32x4 {
        if constexpr (LO) return piece(pl[s], ks);
        else return r16_zero();
      };
      mu32x4 h0[NS], l0[NS], h1[NS], l1[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        h0[s] = piece(ph[s], cw); l0[s] = lpiece_lo(s, cw);
        h1[s] = piece(ph[s], cw + wps); l1[s] = lpiece_lo(s, cw + wps);
      }
      auto frag = [&](int ks, int part) -> mu32x4 {
        return *gl(reinterpret_cast<const mu32x4 *>(pk + (int64_t)(ks < nks ? ks : 0) * 1024 + part * 512));
      };
      mu32x4 fh = frag(cw, 0), fl = frag(cw, 1);
#pragma unroll 1
      for (int ks = cw; ks < nks; ks += wps) {
        mu32x4 h2[NS], l2[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { h2[s] = piece(ph[s], ks + 2 * wps); l2[s] = lpiece_lo(s, ks + 2 * wps); }
        const mu32x4 nfh = frag(ks + wps, 0), nfl = frag(ks + wps, 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          mu32x4 ch = h0[s], cl = l0[s];
          if (!lrok[s]) { ch = r16_zero(); cl = r16_zero(); }
          ch = to_operand(ch);
          accs[s] = FmMfma<E>::mma(fm_frag<E>(ch), fm_frag<E>(fh), accs[s]);
          accs[s] = FmMfma<E>::mma(fm_frag<E>(ch), fm_frag<E>(fl), accs[s]);
          if constexpr (LO) {
            cl = to_operand(cl);
            accs[s] = FmMfma<E>::mma(fm_frag<E>(cl), fm_frag<E>(fh), accs[s]);
          }
          h0[s] = h1[s]; l0[s] = l1[s]; h1[s] = h2[s]; l1[s] = l2[s];
        }
        fh = nfh; fl = nfl;
      }
    }
  }
  if (wps > 1) {
    if (cw > 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) s_red[((wave * NS + s) * 4 + reg) * 64 + lane] = accs[s][reg];
    }
    __syncthreads();
    if (cw > 0) return;
    for (int k = 1; k < wps; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) accs[s][reg] += s_red[(((wave + k) * NS + s) * 4 + reg) * 64 + lane];
  }
  if (jj < r) {
    float *o = d.out + b * d.M * r;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (slab0 + s >= nslabs) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t rw = (slab0 + s) * 16 + 4 * q + reg;
        if (rw < d.M) o[rw * r + jj] = accs[s][reg];
      }
    }
  }
}

// f32 -> (hi, lo) 16-bit planes of flat arrays, one launch over a table (8 elements per thread, 4096 per workgroup)
template <class E>
__global__ __launch_bounds__(256) void split16_ragged_kernel(const lora_amd_split_desc *__restrict__ descs, int n) {
  using S = typename E::storage;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_split_desc d = descs[lo];
  const int64_t i0 = ((int64_t)blockIdx.x - d.begin) * 4096;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int64_t i = i0 + ((int64_t)u * 256 + threadIdx.x) * 8;
    if (i >= d.n) continue;   // n % 8 == 0 (host check)
    const float4 a = gl_ld4(d.src + i), bq = gl_ld4(d.src + i + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
    mu32x4 ph, pl;
    split_hi_lo<E>(v, ph, pl);
    *gl(reinterpret_cast<mu32x4 *>(reinterpret_cast<S *>(d.hi) + i)) = ph;
    *gl(reinterpret_cast<mu32x4 *>(reinterpret_cast<S *>(d.lo) + i)) = pl;
  }
}

// f32 stack [B][N][K] -> the (hi, lo) planes of every matrix AND of its transpose ([B][K][N]) from ONE read: 64 x 64 tiles,
// the transposed planes through an LDS image of the tile (2-byte column gathers, 16-byte stores along n — the in-step merge's
// way of writing W_eff^T).  Replaces a torch transpose copy per shape group plus a separate split.
constexpr int kSpT = 64, kSpPitch = kSpT * 2 + 4;   // bytes per image row
template <class E>
__global__ __launch_bounds__(256) void split16_transpose_kernel(const lora_amd_splitt_desc *__restrict__ descs, int n) {
  using S = typename E::storage;
  __shared__ __attribute__((aligned(16))) unsigned char s_hi[kSpT * kSpPitch], s_lo[kSpT * kSpPitch];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_splitt_desc d = descs[lo];
  const int tiles_k = (d.K + kSpT - 1) / kSpT, tiles_n = (d.N + kSpT - 1) / kSpT;
  int64_t t = (int64_t)blockIdx.x - d.tile_begin;
  const int64_t b = t / ((int64_t)tiles_k * tiles_n);
  t -= b * (int64_t)tiles_k * tiles_n;
  const int tn = (int)(t / tiles_k), tk = (int)(t - (int64_t)tn * tiles_k);
  const int n0 = tn * kSpT, k0 = tk * kSpT;
  const int tid = threadIdx.x;
  const float *src = d.src + b * (int64_t)d.N * d.K;
  S *ph = reinterpret_cast<S *>(d.hi) + b * (int64_t)d.N * d.K, *pl = reinterpret_cast<S *>(d.lo) + b * (int64_t)d.N * d.K;
  S *th = reinterpret_cast<S *>(d.thi) + b * (int64_t)d.N * d.K, *tl = reinterpret_cast<S *>(d.tlo) + b * (int64_t)d.N * d.K;
  // rows of the tile: 8 chunks of 8 columns per row, 32 rows per sweep
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rl = u * 32 + (tid >> 3), c8 = tid & 7;
    const int row = n0 + rl, col = k0 + c8 * 8;
    mu32x4 vh = r16_zero(), vl = r16_zero();
    if (row < d.N && col < d.K) {   // K % 8 == 0
      const float4 a = gl_ld4(src + (int64_t)row * d.K + col), bq = gl_ld4(src + (int64_t)row * d.K + col + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
      split_hi_lo<E>(v, vh, vl);
      *gl(reinterpret_cast<mu32x4 *>(ph + (int64_t)row * d.K + col)) = vh;
      *gl(reinterpret_cast<mu32x4 *>(pl + (int64_t)row * d.K + col)) = vl;
    }
    uint32_t *ih = reinterpret_cast<uint32_t *>(s_hi + rl * kSpPitch + c8 * 16), *il = reinterpret_cast<uint32_t *>(s_lo + rl * kSpPitch + c8 * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ih[i] = vh[i]; il[i] = vl[i]; }
  }
  __syncthreads();
  // columns of the tile: task = (column k, chunk of 8 rows): 64 x 8 tasks, k fastest in groups of 4
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int task = u * 256 + tid;
    const int kq = task >> 5, rem = task & 31;
    const int k = kq * 4 + (rem & 3), ch = rem >> 2;
    const int col = k0 + k, row0 = n0 + ch * 8;
    if (col >= d.K || row0 >= d.N) continue;   // N % 8 == 0
    const unsigned char *sh = s_hi + (ch * 8) * kSpPitch + k * 2, *sl = s_lo + (ch * 8) * kSpPitch + k * 2;
    mu32x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      oh[i] = (uint32_t)*reinterpret_cast<const unsigned short *>(sh + (2 * i) * kSpPitch) |
              ((uint32_t)*reinterpret_cast<const unsigned short *>(sh + (2 * i + 1) * kSpPitch) << 16);
      ol[i] = (uint32_t)*reinterpret_cast<const unsigned short *>(sl + (2 * i) * kSpPitch) |
              ((uint32_t)*reinterpret_cast<const unsigned short *>(sl + (2 * i + 1) * kSpPitch) << 16);
    }
    *gl(reinterpret_cast<mu32x4 *>(th + (int64_t)col * d.N + row0)) = oh;
    *gl(reinterpret_cast<mu32x4 *>(tl + (int64_t)col * d.N + row0)) = ol;
  }
}

// bf16 activations only (round 6): the (hi, lo) split of `up` ~ 1e-4 (lora.py:50-51 starts it at 0) and of Gt = s G up
// ~ 1e-8 .. 1e-3 needs f32's exponent range, which bf16 has and f16 has not — f16 activations at ranks 9..16 take the VALU
// kernels (exact f32 arithmetic on the converted data); the one-launch factor pass (factor_mfma.hip) pre-scales instead.
static bool r16_common_ok(int act_dtype, int fdt, int r) {
  return g_r16_mfma && act_dtype == LORA_AMD_BF16 && fdt == LORA_AMD_F32 && r > 8 && r <= 16;
}

bool r16_rowdot(const void *x, int64_t ldx, const void *f, int fdt, int layout, float *t_out, int64_t M, int C, int r,
                int act_dtype, float scale, float p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st) {
  if (!r16_common_ok(act_dtype, fdt, r) || C % 32 || ldx % 8 || ((uintptr_t)x & 15u) || M <= 0) return false;
  // waves per 16-row slab: >= 4 k-steps per wave, ~4096 waves in flight at most (a wave's chain of dependent loads is what
  // bounds a long row: 576 rows x 10240 columns on one wave per slab took 110-154 us, profiles/r04_kbench_r16_first.log)
  const int nks = C >> 5;
  const int64_t nslabs = (M + 15) / 16;
  int wps = 1;
  while (wps < 16 && nks / (2 * wps) >= 4 && nslabs * wps < 4096) wps *= 2;
  const int slabs = wps <= 4 ? 4 / wps : 1;
  const unsigned block = (unsigned)(64 * wps * slabs);
  const unsigned grid = (unsigned)((nslabs + slabs - 1) / slabs);
  const float *ff = reinterpret_cast<const float *>(f);
#define RD(E, D)                                                                                                      \
  hipLaunchKernelGGL((rowdot16_mfma_kernel<E, D>), dim3(grid), dim3(block), 0, st,                                    \
                     reinterpret_cast<const typename E::storage *>(x), ldx, ff, layout, t_out, M, C, r, wps, scale, p, seed, offset, offset_dev)
  if (p > 0.f) RD(bf16_t, true); else RD(bf16_t, false);
#undef RD
  return true;
}

bool r16_rank_update(void *y, int64_t ldy, const float *t, int nparts, int64_t part_stride, const void *f, int fdt, int layout,
                     int64_t M, int N, int r, int act_dtype, float scale, float p, uint64_t seed, uint64_t offset,
                     const uint64_t *offset_dev, hipStream_t st) {
  if (!r16_common_ok(act_dtype, fdt, r) || N % 8 || ldy % 8 || ((uintptr_t)y & 15u) || M <= 0 || nparts < 1) return false;
  const int64_t ny = (N + 127) / 128;  // a wave = one 32-column group
  // rows per workgroup: as many as keep >= 768 workgroups (the A fragments are rebuilt per workgroup)
  int rows = 256;
  while (rows > 64 && ((M + rows - 1) / rows) * ny < 768) rows >>= 1;
  const int64_t gx = (M + rows - 1) / rows;
  if (gx > 0x7fffffff || ny > 65535) return false;
  const float *ff = reinterpret_cast<const float *>(f);
#define RU(E, D)                                                                                                      \
  hipLaunchKernelGGL((rank_update16_mfma_kernel<E, D>), dim3((unsigned)gx, (unsigned)ny), dim3(kR16Threads), 0, st,   \
                     reinterpret_cast<typename E::storage *>(y), ldy, t, nparts, part_stride, ff, layout, M, N, r,    \
                     rows, scale, p, seed, offset, offset_dev)
  if (p > 0.f) RU(bf16_t, true); else RU(bf16_t, false);
#undef RU
  return true;
}

bool r16_bwd_g(const void *g, int64_t ldg, const float *t, const void *up, int fdt, float *gt_part, float *up_part, int64_t M,
               int N, int r, int log_ct8, int nct, int rows_per_block, int64_t nrb, int act_dtype, float scale, float p,
               uint64_t seed, uint64_t offset, const uint64_t *offset_dev, hipStream_t st, float *gt_out, unsigned *counters) {
  const int cw = 8 << log_ct8;
  if (!r16_common_ok(act_dtype, fdt, r) || cw < 32 || cw > 512 || rows_per_block > 128 || N % 8 || ldg % 8 ||
      ((uintptr_t)g & 15u) || M <= 0 || !up_part || ((uintptr_t)up_part & 15u) || nrb * nct > 0x7fffffff)
    return false;
  const unsigned grid = (unsigned)(nrb * nct);
  const float *uf = reinterpret_cast<const float *>(up);
#define BG(E, D)                                                                                                      \
  hipLaunchKernelGGL((bwd_g16_mfma_kernel<E, D>), dim3(grid), dim3(kR16Threads), 0, st,                               \
                     reinterpret_cast<const typename E::storage *>(g), ldg, t, uf, gt_part, up_part, M, N, r, log_ct8, nct, \
                     rows_per_block, scale, p, seed, offset, offset_dev, gt_out, counters)
  if (p > 0.f) BG(bf16_t, true); else BG(bf16_t, false);
#undef BG
  return true;
}

}  // namespace lora_amd

// Tuning / test hook: 0 routes ranks 9..16 back to the VALU kernels (parity tests compare both), 1 (default) the
// matrix-core forms.  Returns the previous value.
extern "C" int lora_amd_rank16_mfma(int32_t enable) {
  const int prev = lora_amd::g_r16_mfma;
  if (enable >= 0) lora_amd::g_r16_mfma = enable ? 1 : 0;
  return prev;
}

using namespace lora_amd;

extern "C" int lora_amd_rowdot16_planes_plan(lora_amd_planes_desc *descs, int32_t n, int64_t *grid) {
  LORA_AMD_CHECK(descs && n >= 1 && grid, LORA_AMD_EINVAL, "rowdot16_planes_plan: bad argument");
  int64_t begin = 0;
  for (int i = 0; i < n; ++i) {
    lora_amd_planes_desc &d = descs[i];
    LORA_AMD_CHECK(d.hi && d.lo && d.f && d.out && d.M >= 1 && d.batch >= 1, LORA_AMD_EINVAL, "rowdot16_planes_plan: group %d: null pointer / empty", i);
    LORA_AMD_CHECK(d.C >= 32 && d.C % 32 == 0 && (((uintptr_t)d.hi | (uintptr_t)d.lo) & 15u) == 0, LORA_AMD_EINVAL,
                   "rowdot16_planes_plan: group %d: C = %d must be a multiple of 32, planes 16-byte aligned", i, d.C);
    // waves per slab: >= 4 k-steps each (a wave's chain of dependent loads is what bounds a long row)
    const int nks = d.C >> 5;
    int wps = 1;
    while (wps < 16 && nks / (2 * wps) >= 4) wps *= 2;
    d.wps = wps;
    d.slabs_per_wg = kPlNS * (16 / wps);   // a wave multiplies kPlNS consecutive slabs against each fragment pair
    const int64_t nslabs = (d.M + 15) / 16;
    d.wg_begin = begin;
    begin += d.batch * ((nslabs + d.slabs_per_wg - 1) / d.slabs_per_wg);
  }
  LORA_AMD_CHECK(begin < (1ll << 31), LORA_AMD_EINVAL, "rowdot16_planes_plan: too many workgroups");
  *grid = begin;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_rowdot16_planes(const lora_amd_planes_desc *descs_dev, int32_t n, int64_t grid, int32_t r,
                                        int32_t plane_dtype, void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && grid >= 1 && grid < (1ll << 31), LORA_AMD_EINVAL, "rowdot16_planes: bad argument");
  LORA_AMD_CHECK(r >= 1 && r <= 16, LORA_AMD_ERANK, "rowdot16_planes: %d output columns outside [1,16]", r);
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16 || plane_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "rowdot16_planes: 16-bit planes only");
  hipStream_t st = (hipStream_t)stream;
  if (plane_dtype == LORA_AMD_F16) hipLaunchKernelGGL((rowdot16_planes_kernel<f16_t, false>), dim3((unsigned)grid), dim3(1024), 0, st, descs_dev, n, r);
  else hipLaunchKernelGGL((rowdot16_planes_kernel<bf16_t, false>), dim3((unsigned)grid), dim3(1024), 0, st, descs_dev, n, r);
  return check_launch("lora_amd_rowdot16_planes");
}

extern "C" int lora_amd_rowdot16_planes_packed(const lora_amd_planes_desc *descs_dev, int32_t n, int64_t grid,
                                               int32_t plane_dtype, int32_t hi_only, void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && grid >= 1 && grid < (1ll << 31), LORA_AMD_EINVAL, "rowdot16_planes_packed: bad argument");
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16 || plane_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "rowdot16_planes_packed: 16-bit planes only");
  hipStream_t st = (hipStream_t)stream;
#define RP(E)                                                                                                          \
  do {   /* ROWLD: pieces fetched four lanes per row (kernel comment) */                                               \
    if (hi_only) hipLaunchKernelGGL((rowdot16_planes_kernel<E, true, false, true>), dim3((unsigned)grid), dim3(1024), 0, st, descs_dev, n, 16); \
    else hipLaunchKernelGGL((rowdot16_planes_kernel<E, true, true, true>), dim3((unsigned)grid), dim3(1024), 0, st, descs_dev, n, 16);          \
  } while (0)
  if (plane_dtype == LORA_AMD_F16) RP(f16_t); else RP(bf16_t);
#undef RP
  return check_launch("lora_amd_rowdot16_planes_packed");
}

extern "C" int lora_amd_split16_ragged(const lora_amd_split_desc *descs_dev, int32_t n, int64_t blocks, int32_t plane_dtype,
                                       void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && blocks >= 1 && blocks < (1ll << 31), LORA_AMD_EINVAL, "split16_ragged: bad argument");
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16 || plane_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "split16_ragged: 16-bit planes only");
  hipStream_t st = (hipStream_t)stream;
  if (plane_dtype == LORA_AMD_F16) hipLaunchKernelGGL(split16_ragged_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, descs_dev, n);
  else hipLaunchKernelGGL(split16_ragged_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, descs_dev, n);
  return check_launch("lora_amd_split16_ragged");
}

extern "C" int lora_amd_split16_transpose(const lora_amd_splitt_desc *descs_dev, int32_t n, int64_t tiles, int32_t plane_dtype,
                                          void *stream) {
  LORA_AMD_CHECK(descs_dev && n >= 1 && tiles >= 1 && tiles < (1ll << 31), LORA_AMD_EINVAL, "split16_transpose: bad argument");
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16 || plane_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "split16_transpose: 16-bit planes only");
  hipStream_t st = (hipStream_t)stream;
  if (plane_dtype == LORA_AMD_F16) hipLaunchKernelGGL(split16_transpose_kernel<f16_t>, dim3((unsigned)tiles), dim3(256), 0, st, descs_dev, n);
  else hipLaunchKernelGGL(split16_transpose_kernel<bf16_t>, dim3((unsigned)tiles), dim3(256), 0, st, descs_dev, n);
  return check_launch("lora_amd_split16_transpose");
}
