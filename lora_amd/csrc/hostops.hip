// Pointwise / normalisation fusions of the FROZEN host model that the training step
// (training_scripts/train_lora_dreambooth.py:838-892 -> unet(...)) spends its launches on
// between the adapted Linear / Conv2d sites:
//
//   GroupNorm (+ SiLU)   ResnetBlock2D.norm1/norm2 + nonlinearity, Transformer2DModel.norm, conv_norm_out
//                        ATen: 4 launches forward (moments, fused params, normalise, silu), 5 backward, and it
//                        keeps the normalised tensor AND the SiLU input alive for backward.
//                        Here: 2 launches each way (slice statistics, apply), only x and [B,G] (mean, rstd) kept.
//   GEGLU gate           GEGLU.forward right behind the adapted `proj` (lora_diffusion/lora.py:14 targets GEGLU):
//                        h * gelu(gate): ATen = chunk, gelu, mul (5 tensor passes) forward and
//                        gelu_backward, 2 mul, cat (13 passes) backward.  Here: one launch each way (3 / 5 passes);
//                        the backward writes the [M, 2*inner] gradient in place of the cat, which is exactly the G
//                        operand the adapter backward kernels stream next.
//
// All of it is HBM-bound streaming with 16-byte lanes; reductions are per (sample, group) and split into slices so
// that ~1000 workgroups are in flight even when B*G = 128.  Affine parameters are frozen (the reference trains only
// the LoRA factors), so no dgamma / dbeta is produced; callers that train them must use the library path.
#include <algorithm>

#include "common.hpp"

namespace lora_amd {

constexpr int kHT = 256;       // threads per workgroup
constexpr int kHU = 4;         // 16-byte chunks per thread held in registers
constexpr int kHMaxSlices = 256;

__device__ inline float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ inline float sigmoidf(float z) { return fast_rcp(1.f + __expf(-z)); }
// d/dz [z * sigmoid(z)]
__device__ inline float silu_grad(float z) {
  const float s = sigmoidf(z);
  return s * (1.f + z * (1.f - s));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below f32 round-off of the products it feeds)
__device__ inline float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = fast_rcp(1.f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __expf(-ax * ax);
  return copysignf(e, x);
}
__device__ inline float gelu_cdf(float g) { return 0.5f * (1.f + erf_as(g * 0.70710678118654752f)); }
__device__ inline float gelu_pdf(float g) { return 0.39894228040143268f * __expf(-0.5f * g * g); }

// Sum of two per-thread values over the workgroup; every thread receives both totals.
__device__ inline void block_sum2(float &a, float &b, float (*s)[2]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();  // previous use of s is over
  if (lane == 0) { s[wave][0] = a; s[wave][1] = b; }
  __syncthreads();
  a = (s[0][0] + s[1][0]) + (s[2][0] + s[3][0]);
  b = (s[0][1] + s[1][1]) + (s[2][1] + s[3][1]);
}

// One (sample, group) = `chunks` consecutive 16-byte chunks of an NCHW tensor, cut into S slices of `per_block`.
struct GnGeo {
  int cpg, chunks, per_block, S;
  int64_t span;
  bool ok;
};
static GnGeo gn_geo(int B, int C, int HW, int G) {
  GnGeo q{};
  q.ok = B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && HW % 8 == 0;
  if (!q.ok) return q;
  q.cpg = C / G;
  q.span = (int64_t)q.cpg * HW;
  q.ok = q.span / 8 <= (int64_t)kHMaxSlices * kHT * kHU;
  if (!q.ok) return q;
  q.chunks = (int)(q.span / 8);
  int64_t pb = (int64_t)B * G * q.chunks / 1024;  // aim at ~1000 workgroups
  pb = std::max<int64_t>(64, std::min<int64_t>(pb, kHT * kHU));
  pb = std::min<int64_t>(pb, q.chunks);
  q.per_block = (int)pb;
  q.S = (q.chunks + q.per_block - 1) / q.per_block;
  return q;
}

struct GnBlock {
  int64_t bg;       // sample * G + group
  int slice, nloc;  // slice index, chunks in this slice
  int64_t base;     // first element of the slice in the tensor
  int c0;           // first channel of the group
};
__device__ inline GnBlock gn_block(int G, int cpg, int HW, int chunks, int per_block, int S) {
  GnBlock b;
  b.bg = blockIdx.x / S;
  b.slice = (int)(blockIdx.x - b.bg * S);
  b.nloc = min(per_block, chunks - b.slice * per_block);
  b.base = b.bg * (int64_t)cpg * HW + (int64_t)b.slice * per_block * 8;
  b.c0 = (int)(b.bg % G) * cpg;
  return b;
}

// gamma / beta of the channel each of the thread's chunks lies in (HW % 8 == 0: a chunk never straddles channels).
// Loaded up front, next to the activation loads, so that nothing later waits on a dependent scalar fetch.
template <class E>
__device__ inline void gn_load_affine(const typename E::storage *__restrict__ gamma,
                                      const typename E::storage *__restrict__ beta, const GnBlock &b, int per_block,
                                      int HW, float (&ga)[kHU], float (&be)[kHU]) {
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = threadIdx.x + u * kHT;
    const int c = b.c0 + ((b.slice * per_block + (i < b.nloc ? i : 0)) * 8) / HW;
    ga[u] = E::to_f(gamma[c]);
    be[u] = E::to_f(beta[c]);
  }
}
// Keep the compiler from sinking loads below (or hoisting their consumers above) this point: without it hipcc
// consumes the first chunk before issuing the second load, i.e. one HBM round trip per chunk.
#define LORA_AMD_LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)

// ---- GroupNorm forward ---------------------------------------------------------------------------------------------
// Stage 1: slice mean and M2 = sum (x - slice mean)^2 with the slice held in registers (no E[x^2] - mean^2
// cancellation).  part[bg][slice] = (mean, M2); the slice's element count follows from the geometry.
template <class E>
__global__ __launch_bounds__(kHT) void gn_stats_kernel(const typename E::storage *__restrict__ x,
                                                       float *__restrict__ part, int G, int cpg, int HW, int chunks,
                                                       int per_block, int S) {
  __shared__ float s_red[4][2];
  const GnBlock b = gn_block(G, cpg, HW, chunks, per_block, S);
  const int tid = threadIdx.x;
  float v[kHU][8];
  float sum = 0.f, zero = 0.f;
  Raw8<E> raw[kHU];
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    raw[u] = load8_raw<E>(x + b.base + (int64_t)(i < b.nloc ? i : 0) * 8);
  }
  LORA_AMD_LOADS_ISSUED();
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    unpack8_sel<E>(raw[u], tid + u * kHT < b.nloc, v[u]);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[u][e];
  }
  block_sum2(sum, zero, s_red);
  const float mean = sum / (float)(b.nloc * 8);
  float m2 = 0.f;
  zero = 0.f;
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    if (tid + u * kHT < b.nloc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[u][e] - mean;
        m2 = fmaf(d, d, m2);
      }
    }
  }
  block_sum2(m2, zero, s_red);
  if (tid == 0) {
    part[(b.bg * S + b.slice) * 2 + 0] = mean;
    part[(b.bg * S + b.slice) * 2 + 1] = m2;
  }
}

// Stage 2: merge the S slice statistics (Chan et al.), normalise, affine, optional SiLU.
template <class E, bool ACT>
__global__ __launch_bounds__(kHT) void gn_apply_kernel(const typename E::storage *__restrict__ x,
                                                       const typename E::storage *__restrict__ gamma,
                                                       const typename E::storage *__restrict__ beta,
                                                       typename E::storage *__restrict__ y,
                                                       const float *__restrict__ part, float *__restrict__ stats,
                                                       int G, int cpg, int HW, int chunks, int per_block, int S,
                                                       float eps) {
  __shared__ float s_red[4][2];
  const GnBlock b = gn_block(G, cpg, HW, chunks, per_block, S);
  const int tid = threadIdx.x;
  // issue the slice's loads first; the statistics merge below overlaps their latency
  Raw8<E> raw[kHU];
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    raw[u] = load8_raw<E>(x + b.base + (int64_t)(i < b.nloc ? i : 0) * 8);
  }
  float gam[kHU], bet[kHU];
  gn_load_affine<E>(gamma, beta, b, per_block, HW, gam, bet);
  const float n_all = (float)chunks * 8.f;
  const int ts = tid < S ? tid : 0;
  const float n_t = tid < S ? 8.f * (float)min(per_block, chunks - ts * per_block) : 0.f;
  const float mean_t = part[(b.bg * S + ts) * 2 + 0];
  const float m2_t = tid < S ? part[(b.bg * S + ts) * 2 + 1] : 0.f;
  LORA_AMD_LOADS_ISSUED();
  float a = n_t * mean_t, zero = 0.f;
  block_sum2(a, zero, s_red);
  const float mean = a / n_all;
  const float dm = mean_t - mean;
  float m2 = fmaf(n_t * dm, dm, m2_t);
  zero = 0.f;
  block_sum2(m2, zero, s_red);
  const float rstd = rsqrtf(m2 / n_all + eps);
  if (b.slice == 0 && tid == 0) {
    stats[b.bg * 2 + 0] = mean;
    stats[b.bg * 2 + 1] = rstd;
  }
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    if (i < b.nloc) {
      const float ga = gam[u] * rstd;
      const float be = fmaf(-mean, ga, bet[u]);
      float o[8], v[8];
      unpack8_sel<E>(raw[u], true, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = fmaf(v[e], ga, be);
        o[e] = ACT ? z * sigmoidf(z) : z;
      }
      store8<E>(y + b.base + (int64_t)i * 8, o);
    }
  }
}

// ---- GroupNorm backward (input gradient only: gamma / beta are frozen) --------------------------------------------
// t = gamma * dL/dz,  xh = (x - mean) * rstd;  dx = rstd * (t - mean_g(t) - xh * mean_g(t * xh)).
template <class E, bool ACT>
__device__ inline void gn_bwd_terms(const float (&xv)[8], const float (&gv)[8], float ga_r, float be, float gam,
                                    float mean, float rstd, float (&t)[8], float (&xh)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xh[e] = (xv[e] - mean) * rstd;
    float dz = gv[e];
    if (ACT) dz *= silu_grad(fmaf(xv[e], ga_r, be));
    t[e] = gam * dz;
  }
}

template <class E, bool ACT>
__global__ __launch_bounds__(kHT) void gn_bwd_stats_kernel(const typename E::storage *__restrict__ x,
                                                           const typename E::storage *__restrict__ gout,
                                                           const typename E::storage *__restrict__ gamma,
                                                           const typename E::storage *__restrict__ beta,
                                                           const float *__restrict__ stats, float *__restrict__ part,
                                                           int G, int cpg, int HW, int chunks, int per_block, int S) {
  __shared__ float s_red[4][2];
  const GnBlock b = gn_block(G, cpg, HW, chunks, per_block, S);
  const int tid = threadIdx.x;
  Raw8<E> xr[kHU], gr[kHU];
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    const int64_t off = b.base + (int64_t)(i < b.nloc ? i : 0) * 8;
    xr[u] = load8_raw<E>(x + off);
    gr[u] = load8_raw<E>(gout + off);
  }
  float gam[kHU], bet[kHU];
  gn_load_affine<E>(gamma, beta, b, per_block, HW, gam, bet);
  const float mean = stats[b.bg * 2 + 0], rstd = stats[b.bg * 2 + 1];
  LORA_AMD_LOADS_ISSUED();
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    if (i < b.nloc) {
      const float ga_r = gam[u] * rstd, be = fmaf(-mean, ga_r, bet[u]);
      float t[8], xh[8], xv[8], gv[8];
      unpack8_sel<E>(xr[u], true, xv);
      unpack8_sel<E>(gr[u], true, gv);
      gn_bwd_terms<E, ACT>(xv, gv, ga_r, be, gam[u], mean, rstd, t, xh);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1 += t[e];
        s2 = fmaf(t[e], xh[e], s2);
      }
    }
  }
  block_sum2(s1, s2, s_red);
  if (tid == 0) {
    part[(b.bg * S + b.slice) * 2 + 0] = s1;
    part[(b.bg * S + b.slice) * 2 + 1] = s2;
  }
}

template <class E, bool ACT>
__global__ __launch_bounds__(kHT) void gn_bwd_apply_kernel(const typename E::storage *__restrict__ x,
                                                           const typename E::storage *__restrict__ gout,
                                                           const typename E::storage *__restrict__ gamma,
                                                           const typename E::storage *__restrict__ beta,
                                                           const float *__restrict__ stats,
                                                           const float *__restrict__ part,
                                                           typename E::storage *__restrict__ dx, int G, int cpg,
                                                           int HW, int chunks, int per_block, int S) {
  __shared__ float s_red[4][2];
  const GnBlock b = gn_block(G, cpg, HW, chunks, per_block, S);
  const int tid = threadIdx.x;
  Raw8<E> xr[kHU], gr[kHU];
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    const int64_t off = b.base + (int64_t)(i < b.nloc ? i : 0) * 8;
    xr[u] = load8_raw<E>(x + off);
    gr[u] = load8_raw<E>(gout + off);
  }
  float gam[kHU], bet[kHU];
  gn_load_affine<E>(gamma, beta, b, per_block, HW, gam, bet);
  const int ts = tid < S ? tid : 0;
  float s1 = part[(b.bg * S + ts) * 2 + 0], s2 = part[(b.bg * S + ts) * 2 + 1];
  const float mean = stats[b.bg * 2 + 0], rstd = stats[b.bg * 2 + 1];
  LORA_AMD_LOADS_ISSUED();
  if (tid >= S) s1 = s2 = 0.f;
  block_sum2(s1, s2, s_red);
  const float inv_n = 1.f / ((float)chunks * 8.f);
  const float c1 = s1 * inv_n, c2 = s2 * inv_n;
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const int i = tid + u * kHT;
    if (i < b.nloc) {
      const float ga_r = gam[u] * rstd, be = fmaf(-mean, ga_r, bet[u]);
      float t[8], xh[8], o[8], xv[8], gv[8];
      unpack8_sel<E>(xr[u], true, xv);
      unpack8_sel<E>(gr[u], true, gv);
      gn_bwd_terms<E, ACT>(xv, gv, ga_r, be, gam[u], mean, rstd, t, xh);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (t[e] - c1 - xh[e] * c2);
      store8<E>(dx + b.base + (int64_t)i * 8, o);
    }
  }
}

// ---- GEGLU gate ----------------------------------------------------------------------------------------------------
// y = [h | gate] per row (2 * inner columns);  out = h * gelu(gate)   (exact erf form, F.gelu's default).
template <class E>
__global__ __launch_bounds__(kHT) void geglu_fwd_kernel(const typename E::storage *__restrict__ y, int64_t ldy,
                                                        typename E::storage *__restrict__ out, int64_t ldo,
                                                        int64_t M, int c8) {
  const uint32_t total = (uint32_t)(M * c8);  // host checks M * c8 < 2^31
  const uint32_t i0 = blockIdx.x * (uint32_t)(kHT * kHU) + threadIdx.x;
  Raw8<E> hr[kHU], gr[kHU];
  int64_t rows[kHU];
  int cols[kHU];
  const uint32_t dq = (uint32_t)kHT / (uint32_t)c8, dr = (uint32_t)kHT % (uint32_t)c8;
  const uint32_t row0 = min(i0, total - 1) / (uint32_t)c8, col0 = min(i0, total - 1) - row0 * (uint32_t)c8;
  uint32_t row = row0, col = col0;
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    const bool ok = i0 + u * kHT < total;  // chunks past the end re-read the block's first chunk
    rows[u] = ok ? row : row0;
    cols[u] = (int)(ok ? col : col0) * 8;
    row += dq; col += dr;
    if (col >= (uint32_t)c8) { col -= c8; ++row; }
    const typename E::storage *p = y + rows[u] * ldy + cols[u];
    hr[u] = load8_raw<E>(p);
    gr[u] = load8_raw<E>(p + (int64_t)c8 * 8);
  }
  LORA_AMD_LOADS_ISSUED();
#pragma unroll
  for (int u = 0; u < kHU; ++u) {
    if (i0 + u * kHT < total) {
      float o[8], h[8], g[8];
      unpack8_sel<E>(hr[u], true, h);
      unpack8_sel<E>(gr[u], true, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = h[e] * (g[e] * gelu_cdf(g[e]));
      store8<E>(out + rows[u] * ldo + cols[u], o);
    }
  }
}

// gy[:, :inner] = gout * gelu(gate);  gy[:, inner:] = gout * h * gelu'(gate)
template <class E>
__global__ __launch_bounds__(kHT) void geglu_bwd_kernel(const typename E::storage *__restrict__ y, int64_t ldy,
                                                        const typename E::storage *__restrict__ gout, int64_t ldg,
                                                        typename E::storage *__restrict__ gy, int64_t ldgy,
                                                        int64_t M, int c8) {
  constexpr int U = 2;
  const uint32_t total = (uint32_t)(M * c8);  // host checks M * c8 < 2^31
  const uint32_t i0 = blockIdx.x * (uint32_t)(kHT * U) + threadIdx.x;
  Raw8<E> hr[U], gr[U], gor[U];
  int64_t rows[U];
  int cols[U];
  const uint32_t dq = (uint32_t)kHT / (uint32_t)c8, dr = (uint32_t)kHT % (uint32_t)c8;
  const uint32_t row0 = min(i0, total - 1) / (uint32_t)c8, col0 = min(i0, total - 1) - row0 * (uint32_t)c8;
  uint32_t row = row0, col = col0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool ok = i0 + u * kHT < total;
    rows[u] = ok ? row : row0;
    cols[u] = (int)(ok ? col : col0) * 8;
    row += dq; col += dr;
    if (col >= (uint32_t)c8) { col -= c8; ++row; }
    const typename E::storage *p = y + rows[u] * ldy + cols[u];
    hr[u] = load8_raw<E>(p);
    gr[u] = load8_raw<E>(p + (int64_t)c8 * 8);
    gor[u] = load8_raw<E>(gout + rows[u] * ldg + cols[u]);
  }
  LORA_AMD_LOADS_ISSUED();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (i0 + u * kHT < total) {
      float dh[8], dg[8], h[8], g[8], go[8];
      unpack8_sel<E>(hr[u], true, h);
      unpack8_sel<E>(gr[u], true, g);
      unpack8_sel<E>(gor[u], true, go);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float cdf = gelu_cdf(g[e]);
        dh[e] = go[e] * (g[e] * cdf);
        dg[e] = go[e] * h[e] * fmaf(g[e], gelu_pdf(g[e]), cdf);
      }
      typename E::storage *q = gy + rows[u] * ldgy + cols[u];
      store8<E>(q, dh);
      store8<E>(q + (int64_t)c8 * 8, dg);
    }
  }
}

// ---- LayerNorm over the last dimension (BasicTransformerBlock.norm1/2/3) -----------------------------------------
// One row = K/8 chunks spread over L = 2^logL consecutive lanes (<= kLU chunks per lane, all in registers); a wave
// holds 64/L rows.  Mean, then M2 around it, by xor-shuffles inside the lane group: one launch forward, one backward
// (ATen: 24 us / 34 us per call at [16384, 320] bf16, i.e. < 1 TB/s).
constexpr int kLU = 5;

__device__ inline float group_allsum(float v, int logL) {
  for (int off = 1; off < (1 << logL); off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ADD: the row is x + res (the residual add in front of norm2 / norm3), rounded to the activation dtype exactly as the
// separate add would, written to sum_out for the residual stream, and normalised in the same pass.
template <class E, bool ADD>
__global__ __launch_bounds__(kHT) void ln_fwd_kernel(const typename E::storage *__restrict__ x,
                                                     const typename E::storage *__restrict__ res,
                                                     const typename E::storage *__restrict__ gamma,
                                                     const typename E::storage *__restrict__ beta,
                                                     typename E::storage *__restrict__ sum_out,
                                                     typename E::storage *__restrict__ y, float *__restrict__ stats,
                                                     int64_t M, int c8, int logL, float eps) {
  const int L = 1 << logL, l = threadIdx.x & (L - 1);
  const int64_t row = (int64_t)blockIdx.x * (kHT >> logL) + (threadIdx.x >> logL);
  const bool live = row < M;
  const int64_t rbase = (live ? row : 0) * (int64_t)c8 * 8;
  const typename E::storage *xr = x + rbase;
  Raw8<E> raw[kLU], rr[ADD ? kLU : 1], gr[kLU], br[kLU];
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const int cc = l + u * L, cs = cc < c8 ? cc : l;
    raw[u] = load8_raw<E>(xr + cs * 8);
    if (ADD) rr[u] = load8_raw<E>(res + rbase + cs * 8);
    gr[u] = load8_raw<E>(gamma + cs * 8);
    br[u] = load8_raw<E>(beta + cs * 8);
  }
  LORA_AMD_LOADS_ISSUED();
  float v[kLU][8], sum = 0.f;
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const bool ok = l + u * L < c8;
    unpack8_sel<E>(raw[u], ok, v[u]);
    if (ADD) {
      float rv[8];
      unpack8_sel<E>(rr[u], ok, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[u][e] = round_to<E>(v[u][e] + rv[e]);
      if (ok && live) store8<E>(sum_out + row * (int64_t)c8 * 8 + (l + u * L) * 8, v[u]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += v[u][e];
  }
  const float inv_n = 1.f / (float)(c8 * 8);
  const float mean = group_allsum(sum, logL) * inv_n;
  float m2 = 0.f;
#pragma unroll
  for (int u = 0; u < kLU; ++u)
    if (l + u * L < c8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[u][e] - mean;
        m2 = fmaf(d, d, m2);
      }
    }
  const float rstd = rsqrtf(group_allsum(m2, logL) * inv_n + eps);
  if (!live) return;
  if (l == 0) {
    stats[row * 2 + 0] = mean;
    stats[row * 2 + 1] = rstd;
  }
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const int cc = l + u * L;
    if (cc < c8) {
      float ga[8], be[8], o[8];
      unpack8_sel<E>(gr[u], true, ga);
      unpack8_sel<E>(br[u], true, be);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf((v[u][e] - mean) * rstd, ga[e], be[e]);
      store8<E>(y + row * (int64_t)c8 * 8 + cc * 8, o);
    }
  }
}

// dx = rstd * (t - mean(t) - xh * mean(t * xh)),  t = gamma * gout,  xh = (x - mean) * rstd   (gamma / beta frozen)
// ADDG: dx additionally receives gsum, the gradient that reaches the same row through the residual stream.
template <class E, bool ADDG>
__global__ __launch_bounds__(kHT) void ln_bwd_kernel(const typename E::storage *__restrict__ x,
                                                     const typename E::storage *__restrict__ gout,
                                                     const typename E::storage *__restrict__ gsum,
                                                     const typename E::storage *__restrict__ gamma,
                                                     const float *__restrict__ stats,
                                                     typename E::storage *__restrict__ dx, int64_t M, int c8,
                                                     int logL) {
  const int L = 1 << logL, l = threadIdx.x & (L - 1);
  const int64_t row = (int64_t)blockIdx.x * (kHT >> logL) + (threadIdx.x >> logL);
  const bool live = row < M;
  const int64_t rbase = (live ? row : 0) * (int64_t)c8 * 8;
  Raw8<E> xr[kLU], gor[kLU], gsr[ADDG ? kLU : 1], gr[kLU];
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const int cc = l + u * L, cs = cc < c8 ? cc : l;
    xr[u] = load8_raw<E>(x + rbase + cs * 8);
    gor[u] = load8_raw<E>(gout + rbase + cs * 8);
    if (ADDG) gsr[u] = load8_raw<E>(gsum + rbase + cs * 8);
    gr[u] = load8_raw<E>(gamma + cs * 8);
  }
  const float mean = stats[(live ? row : 0) * 2 + 0], rstd = stats[(live ? row : 0) * 2 + 1];
  LORA_AMD_LOADS_ISSUED();
  float t[kLU][8], xh[kLU][8], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const bool ok = l + u * L < c8;
    float xv[8], gv[8], ga[8];
    unpack8_sel<E>(xr[u], ok, xv);
    unpack8_sel<E>(gor[u], ok, gv);
    unpack8_sel<E>(gr[u], ok, ga);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[u][e] = ok ? (xv[e] - mean) * rstd : 0.f;
      t[u][e] = ga[e] * gv[e];
      s1 += t[u][e];
      s2 = fmaf(t[u][e], xh[u][e], s2);
    }
  }
  const float inv_n = 1.f / (float)(c8 * 8);
  const float c1 = group_allsum(s1, logL) * inv_n, c2 = group_allsum(s2, logL) * inv_n;
  if (!live) return;
#pragma unroll
  for (int u = 0; u < kLU; ++u) {
    const int cc = l + u * L;
    if (cc < c8) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (t[u][e] - c1 - xh[u][e] * c2);
      if (ADDG) {
        float gs[8];
        unpack8_sel<E>(gsr[u], true, gs);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += gs[e];
      }
      store8<E>(dx + row * (int64_t)c8 * 8 + cc * 8, o);
    }
  }
}

// lanes per row: the smallest power of two that leaves <= kLU chunks per lane (-1: row too long)
static inline int ln_logL(int c8) {
  for (int lg = 0; lg <= 6; ++lg)
    if (((c8 + (1 << lg) - 1) >> lg) <= kLU) return lg;
  return -1;
}

// ---- GroupNorm on channels_last (NHWC) activations -------------------------------------------------------------------
// Memory is [B][HW][C]: a group is C/G adjacent channels of every pixel, so the reduction runs over pixels.  Every
// streaming kernel uses a column-owner mapping: thread = (pixel slot, 16-byte channel chunk), the chunk is fixed for
// the thread's lifetime (its per-channel vectors are loaded once), consecutive lanes read consecutive chunks of one
// pixel (contiguous runs of cw * 16 bytes).  Statistics are per channel and per pixel slice (shifted sums -> mean, M2),
// a tiny finalize kernel folds slices and the group's channels together (Chan) and expands the result per channel:
//   aff[b][0][c] = gamma*rstd, aff[b][1][c] = beta - mean*gamma*rstd, aff[b][2][c] = mean, aff[b][3][c] = rstd
// so that the apply kernels are pure per-channel affine maps.  3 + 3 launches; MIOpen's NHWC kernels then need no
// NCHW<->NHWC transposes around them and the transformer blocks read the activations as tokens without a copy.
constexpr int kNU = 4;  // pixels in flight per thread

struct GnNhwcGeo {
  int c8, cw, tiles, nslots, px, S;
  bool ok;
};
static GnNhwcGeo gn_nhwc_geo(int B, int C, int HW, int G) {
  GnNhwcGeo q{};
  q.ok = B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0 && C % 8 == 0;
  if (!q.ok) return q;
  q.c8 = C / 8;
  int best = 1, best_act = 0;
  for (int d = 1; d <= std::min(q.c8, 128); ++d) {
    if (q.c8 % d) continue;
    const int act = (kHT / d) * d;
    if (act >= best_act) { best_act = act; best = d; }
  }
  q.cw = best;
  q.tiles = q.c8 / q.cw;
  q.nslots = kHT / q.cw;
  // pixels per workgroup: ~768 workgroups on the large maps, at least two pixels per slot on the small ones (the
  // per-channel partials are 8 bytes per channel and slice: keep them a small fraction of the stream)
  int64_t px = std::max<int64_t>(2 * q.nslots, (int64_t)HW * B * q.tiles / 768);
  px = std::min<int64_t>(px, HW);
  q.px = (int)px;
  q.S = (HW + q.px - 1) / q.px;
  return q;
}

struct NhwcBlock {
  int b, s, col, slot, np;
  bool active;
  int64_t base;  // element offset of (b, first pixel of the slice, this thread's chunk)
};
__device__ inline NhwcBlock nhwc_block(int C, int HW, int cw, int tiles, int nslots, int px, int S) {
  NhwcBlock k;
  const int bs = blockIdx.x / tiles, tile = blockIdx.x - bs * tiles;
  k.b = bs / S;
  k.s = bs - k.b * S;
  k.slot = threadIdx.x / cw;
  const int cl = threadIdx.x - k.slot * cw;
  k.active = k.slot < nslots;
  k.col = tile * cw + cl;
  const int p0 = k.s * px;
  k.np = min(px, HW - p0);
  k.base = ((int64_t)k.b * HW + p0) * C + (int64_t)k.col * 8;
  return k;
}

__device__ inline void ld8f(const float *p, float (&v)[8]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 a = *reinterpret_cast<const f4 *>(p), b = *reinterpret_cast<const f4 *>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
}
__device__ inline void st8f(float *p, const float (&v)[8]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[i + 4]; }
  *reinterpret_cast<f4 *>(p) = a;
  *reinterpret_cast<f4 *>(p + 4) = b;
}

// Sum acc0 / acc1 (8 channels each) over the pixel slots of the block; slot 0 receives the totals.
__device__ inline void nhwc_slot_reduce(float (&a0)[8], float (&a1)[8], const NhwcBlock &k, int cw, int nslots,
                                        float *s_red /* [2][kHT][8] */) {
  if (k.active) {
    st8f(s_red + (size_t)threadIdx.x * 8, a0);
    st8f(s_red + (size_t)(kHT + threadIdx.x) * 8, a1);
  }
  __syncthreads();
  if (k.active && k.slot == 0) {
    for (int sl = 1; sl < nslots; ++sl) {
      float t0[8], t1[8];
      ld8f(s_red + (size_t)(sl * cw + threadIdx.x) * 8, t0);
      ld8f(s_red + (size_t)(kHT + sl * cw + threadIdx.x) * 8, t1);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0[e] += t0[e]; a1[e] += t1[e]; }
    }
  }
}

// part[((b*S + s)*2 + 0)*C + c] = slice mean of channel c, [.. + 1] = slice M2
template <class E>
__global__ __launch_bounds__(kHT) void gn_nhwc_stats_kernel(const typename E::storage *__restrict__ x,
                                                            float *__restrict__ part, int C, int HW, int cw, int tiles,
                                                            int nslots, int px, int S) {
  __shared__ __attribute__((aligned(16))) float s_red[2 * kHT * 8];
  const NhwcBlock k = nhwc_block(C, HW, cw, tiles, nslots, px, S);
  float sh[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  if (k.active) {
    unpack8_sel<E>(load8_raw<E>(x + k.base), true, sh);  // shift = the slice's first pixel (same for every slot)
    for (int p = k.slot; p < k.np; p += nslots * kNU) {
      Raw8<E> raw[kNU];
#pragma unroll
      for (int u = 0; u < kNU; ++u) {
        const int pp = p + u * nslots;
        raw[u] = load8_raw<E>(x + k.base + (int64_t)(pp < k.np ? pp : p) * C);
      }
      LORA_AMD_LOADS_ISSUED();
#pragma unroll
      for (int u = 0; u < kNU; ++u) {
        float v[8];
        unpack8_sel<E>(raw[u], true, v);
        if (p + u * nslots < k.np) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[e] - sh[e];
            s1[e] += d;
            s2[e] = fmaf(d, d, s2[e]);
          }
        }
      }
    }
  }
  nhwc_slot_reduce(s1, s2, k, cw, nslots, s_red);
  if (k.active && k.slot == 0) {
    const float inv_n = 1.f / (float)k.np;
    float mean[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = sh[e] + s1[e] * inv_n;
      m2[e] = fmaxf(s2[e] - s1[e] * s1[e] * inv_n, 0.f);
    }
    float *o = part + ((int64_t)(k.b * S + k.s) * 2) * C + (int64_t)k.col * 8;
    st8f(o, mean);
    st8f(o + C, m2);
  }
}

// One workgroup per (sample, group): fold slices x channels, expand per channel.  `addend` [B][C] (f32, may be null)
// is a per-(sample, channel) term added to x BEFORE the normalisation (ResnetBlock2D: the time-embedding projection and
// the bias of the convolution that produced x).  A per-channel shift moves only the channel means, so it costs nothing
// in the streaming kernels: slice means are shifted here, and the expanded affine absorbs it,
//   z = ((x + add) - mean) * rstd * gamma + beta = x * a + (beta + (add - mean) * a),   xh = (x - (mean - add)) * rstd.
template <class E>
__global__ __launch_bounds__(kHT) void gn_nhwc_finalize_kernel(const float *__restrict__ part,
                                                               const typename E::storage *__restrict__ gamma,
                                                               const typename E::storage *__restrict__ beta,
                                                               const float *__restrict__ addend,
                                                               float *__restrict__ aff, int C, int HW, int G, int px,
                                                               int S, float eps) {
  __shared__ float s_red[4][2];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, cpg = C / G;
  const int items = S * cpg;
  float a = 0.f, zero = 0.f;
  for (int i = threadIdx.x; i < items; i += kHT) {
    const int s = i / cpg, c = g * cpg + (i - s * cpg);
    const float add = addend != nullptr ? addend[(int64_t)b * C + c] : 0.f;
    a += (float)min(px, HW - s * px) * (part[((int64_t)(b * S + s) * 2) * C + c] + add);
  }
  block_sum2(a, zero, s_red);
  const float n_all = (float)HW * (float)cpg;
  const float mean = a / n_all;
  float m2 = 0.f;
  zero = 0.f;
  for (int i = threadIdx.x; i < items; i += kHT) {
    const int s = i / cpg, c = g * cpg + (i - s * cpg);
    const float *pp = part + ((int64_t)(b * S + s) * 2) * C + c;
    const float add = addend != nullptr ? addend[(int64_t)b * C + c] : 0.f;
    const float dm = pp[0] + add - mean;
    m2 += pp[C] + (float)min(px, HW - s * px) * dm * dm;
  }
  block_sum2(m2, zero, s_red);
  const float rstd = rsqrtf(m2 / n_all + eps);
  for (int cc = threadIdx.x; cc < cpg; cc += kHT) {
    const int c = g * cpg + cc;
    const float add = addend != nullptr ? addend[(int64_t)b * C + c] : 0.f;
    const float ga = E::to_f(gamma[c]) * rstd;
    float *o = aff + (int64_t)b * 4 * C + c;
    o[0] = ga;
    o[C] = fmaf(add - mean, ga, E::to_f(beta[c]));
    o[2 * C] = mean - add;
    o[3 * C] = rstd;
  }
}

template <class E, bool ACT>
__global__ __launch_bounds__(kHT) void gn_nhwc_apply_kernel(const typename E::storage *__restrict__ x,
                                                            const float *__restrict__ aff,
                                                            typename E::storage *__restrict__ y, int C, int HW, int cw,
                                                            int tiles, int nslots, int px, int S) {
  const NhwcBlock k = nhwc_block(C, HW, cw, tiles, nslots, px, S);
  if (!k.active) return;
  float ga[8], be[8];
  ld8f(aff + (int64_t)k.b * 4 * C + (int64_t)k.col * 8, ga);
  ld8f(aff + (int64_t)k.b * 4 * C + C + (int64_t)k.col * 8, be);
  for (int p = k.slot; p < k.np; p += nslots * kNU) {
    Raw8<E> raw[kNU];
#pragma unroll
    for (int u = 0; u < kNU; ++u) {
      const int pp = p + u * nslots;
      raw[u] = load8_raw<E>(x + k.base + (int64_t)(pp < k.np ? pp : p) * C);
    }
    LORA_AMD_LOADS_ISSUED();
#pragma unroll
    for (int u = 0; u < kNU; ++u) {
      const int pp = p + u * nslots;
      if (pp < k.np) {
        float v[8], o[8];
        unpack8_sel<E>(raw[u], true, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float z = fmaf(v[e], ga[e], be[e]);
          o[e] = ACT ? z * sigmoidf(z) : z;
        }
        store8<E>(y + k.base + (int64_t)pp * C, o);
      }
    }
  }
}

// backward: t = gamma * dz, xh = (x - mean) * rstd;  per channel and pixel slice: sum t, sum t*xh
template <class E, bool ACT>
__device__ inline void gn_nhwc_terms(const float (&v)[8], const float (&go)[8], const float (&ga)[8],
                                     const float (&be)[8], const float (&mean)[8], const float (&rstd)[8],
                                     const float (&gam)[8], float (&t)[8], float (&xh)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    xh[e] = (v[e] - mean[e]) * rstd[e];
    float dz = go[e];
    if (ACT) dz *= silu_grad(fmaf(v[e], ga[e], be[e]));
    t[e] = gam[e] * dz;
  }
}

template <class E, bool ACT, bool APPLY>
__global__ __launch_bounds__(kHT) void gn_nhwc_bwd_kernel(const typename E::storage *__restrict__ x,
                                                          const typename E::storage *__restrict__ gout,
                                                          const typename E::storage *__restrict__ gamma,
                                                          const float *__restrict__ aff,
                                                          float *__restrict__ part,        // !APPLY: out [B][S][2][C]
                                                          const float *__restrict__ cvec,  // APPLY: [B][2][C] = c1, c2
                                                          typename E::storage *__restrict__ dx, int C, int HW, int cw,
                                                          int tiles, int nslots, int px, int S) {
  __shared__ __attribute__((aligned(16))) float s_red[APPLY ? 8 : 2 * kHT * 8];
  const NhwcBlock k = nhwc_block(C, HW, cw, tiles, nslots, px, S);
  float ga[8], be[8], mean[8], rstd[8], gam[8], c1[8], c2[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = c1[e] = c2[e] = 0.f;
  if (k.active) {
    const float *ab = aff + (int64_t)k.b * 4 * C + (int64_t)k.col * 8;
    ld8f(ab, ga);
    ld8f(ab + C, be);
    ld8f(ab + 2 * C, mean);
    ld8f(ab + 3 * C, rstd);
    unpack8_sel<E>(load8_raw<E>(gamma + (int64_t)k.col * 8), true, gam);
    if (APPLY) {
      ld8f(cvec + (int64_t)k.b * 2 * C + (int64_t)k.col * 8, c1);
      ld8f(cvec + (int64_t)k.b * 2 * C + C + (int64_t)k.col * 8, c2);
    }
    for (int p = k.slot; p < k.np; p += nslots * kNU) {
      Raw8<E> xr[kNU], gr[kNU];
#pragma unroll
      for (int u = 0; u < kNU; ++u) {
        const int pp = p + u * nslots;
        const int64_t off = k.base + (int64_t)(pp < k.np ? pp : p) * C;
        xr[u] = load8_raw<E>(x + off);
        gr[u] = load8_raw<E>(gout + off);
      }
      LORA_AMD_LOADS_ISSUED();
#pragma unroll
      for (int u = 0; u < kNU; ++u) {
        const int pp = p + u * nslots;
        if (pp < k.np) {
          float v[8], go[8], t[8], xh[8];
          unpack8_sel<E>(xr[u], true, v);
          unpack8_sel<E>(gr[u], true, go);
          gn_nhwc_terms<E, ACT>(v, go, ga, be, mean, rstd, gam, t, xh);
          if (APPLY) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rstd[e] * (t[e] - c1[e] - xh[e] * c2[e]);
            store8<E>(dx + k.base + (int64_t)pp * C, o);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              s1[e] += t[e];
              s2[e] = fmaf(t[e], xh[e], s2[e]);
            }
          }
        }
      }
    }
  }
  if (!APPLY) {
    nhwc_slot_reduce(s1, s2, k, cw, nslots, s_red);
    if (k.active && k.slot == 0) {
      float *o = part + ((int64_t)(k.b * S + k.s) * 2) * C + (int64_t)k.col * 8;
      st8f(o, s1);
      st8f(o + C, s2);
    }
  }
}

// cvec[b][0][c] = (sum over the group of s1) / n, cvec[b][1][c] = (sum of s2) / n, expanded per channel
__global__ __launch_bounds__(kHT) void gn_nhwc_bwd_finalize_kernel(const float *__restrict__ part,
                                                                   float *__restrict__ cvec, int C, int HW, int G,
                                                                   int S) {
  __shared__ float s_red[4][2];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, cpg = C / G;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < S * cpg; i += kHT) {
    const int s = i / cpg, c = g * cpg + (i - s * cpg);
    const float *pp = part + ((int64_t)(b * S + s) * 2) * C + c;
    s1 += pp[0];
    s2 += pp[C];
  }
  block_sum2(s1, s2, s_red);
  const float inv_n = 1.f / ((float)HW * (float)cpg);
  for (int cc = threadIdx.x; cc < cpg; cc += kHT) {
    cvec[(int64_t)b * 2 * C + g * cpg + cc] = s1 * inv_n;
    cvec[(int64_t)b * 2 * C + C + g * cpg + cc] = s2 * inv_n;
  }
}

static inline bool aligned_for(const void *p, int dt) { return ((uintptr_t)p % (dt == LORA_AMD_F32 ? 32 : 16)) == 0; }

}  // namespace lora_amd

using namespace lora_amd;

extern "C" size_t lora_amd_groupnorm_workspace(int32_t B, int32_t C, int32_t HW, int32_t groups) {
  const GnGeo q = gn_geo(B, C, HW, groups);
  return q.ok ? (size_t)B * groups * q.S * 2 * sizeof(float) : 0;
}

extern "C" int lora_amd_groupnorm_supported(int32_t B, int32_t C, int32_t HW, int32_t groups) {
  return gn_geo(B, C, HW, groups).ok ? 1 : 0;
}

#define GN_CHECKS(name)                                                                                              \
  const GnGeo q = gn_geo(B, C, HW, groups);                                                                          \
  LORA_AMD_CHECK(q.ok, LORA_AMD_EINVAL, name ": geometry B=%d C=%d HW=%d groups=%d not supported", B, C, HW, groups); \
  LORA_AMD_CHECK(dtype_ok(dtype), LORA_AMD_EINVAL, name ": bad dtype %d", dtype);                                    \
  LORA_AMD_CHECK(workspace_bytes >= lora_amd_groupnorm_workspace(B, C, HW, groups), LORA_AMD_EWORKSPACE,             \
                 name ": workspace %zu < %zu bytes", workspace_bytes, lora_amd_groupnorm_workspace(B, C, HW, groups)); \
  const dim3 grid((unsigned)((int64_t)B * groups * q.S)), block(kHT);                                                \
  hipStream_t st = (hipStream_t)stream;                                                                              \
  float *part = reinterpret_cast<float *>(workspace)

extern "C" int lora_amd_groupnorm_fwd(const void *x, const void *gamma, const void *beta, void *y, float *stats,
                                      void *workspace, size_t workspace_bytes, int32_t B, int32_t C, int32_t HW,
                                      int32_t groups, float eps, int32_t act, int32_t dtype, void *stream) {
  GN_CHECKS("groupnorm_fwd");
  LORA_AMD_CHECK(x && gamma && beta && y && stats && workspace, LORA_AMD_EINVAL, "groupnorm_fwd: null pointer");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(y, dtype), LORA_AMD_EINVAL, "groupnorm_fwd: unaligned tensor");
#define GO(E)                                                                                                        \
  {                                                                                                                  \
    using S = typename E::storage;                                                                                   \
    hipLaunchKernelGGL((gn_stats_kernel<E>), grid, block, 0, st, (const S *)x, part, groups, q.cpg, HW, q.chunks,    \
                       q.per_block, q.S);                                                                            \
    if (act)                                                                                                         \
      hipLaunchKernelGGL((gn_apply_kernel<E, true>), grid, block, 0, st, (const S *)x, (const S *)gamma,             \
                         (const S *)beta, (S *)y, part, stats, groups, q.cpg, HW, q.chunks, q.per_block, q.S, eps);  \
    else                                                                                                             \
      hipLaunchKernelGGL((gn_apply_kernel<E, false>), grid, block, 0, st, (const S *)x, (const S *)gamma,            \
                         (const S *)beta, (S *)y, part, stats, groups, q.cpg, HW, q.chunks, q.per_block, q.S, eps);  \
  }                                                                                                                  \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_groupnorm_fwd");
}

extern "C" int lora_amd_groupnorm_bwd(const void *x, const void *gout, const void *gamma, const void *beta,
                                      const float *stats, void *dx, void *workspace, size_t workspace_bytes,
                                      int32_t B, int32_t C, int32_t HW, int32_t groups, int32_t act, int32_t dtype,
                                      void *stream) {
  GN_CHECKS("groupnorm_bwd");
  LORA_AMD_CHECK(x && gout && gamma && beta && stats && dx && workspace, LORA_AMD_EINVAL,
                 "groupnorm_bwd: null pointer");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(gout, dtype) && aligned_for(dx, dtype), LORA_AMD_EINVAL,
                 "groupnorm_bwd: unaligned tensor");
#define GO2(E, A)                                                                                                    \
  {                                                                                                                  \
    using S = typename E::storage;                                                                                   \
    hipLaunchKernelGGL((gn_bwd_stats_kernel<E, A>), grid, block, 0, st, (const S *)x, (const S *)gout,               \
                       (const S *)gamma, (const S *)beta, stats, part, groups, q.cpg, HW, q.chunks, q.per_block,     \
                       q.S);                                                                                         \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<E, A>), grid, block, 0, st, (const S *)x, (const S *)gout,               \
                       (const S *)gamma, (const S *)beta, stats, part, (S *)dx, groups, q.cpg, HW, q.chunks,         \
                       q.per_block, q.S);                                                                            \
  }
#define GO(E)                  \
  if (act) GO2(E, true) else GO2(E, false) \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
#undef GO2
  return check_launch("lora_amd_groupnorm_bwd");
}

#define GEGLU_CHECKS(name)                                                                                      \
  LORA_AMD_CHECK(M >= 0 && inner > 0 && inner % 8 == 0, LORA_AMD_EINVAL, name ": bad shape M=%lld inner=%d",     \
                 (long long)M, inner);                                                                          \
  LORA_AMD_CHECK(dtype_ok(dtype), LORA_AMD_EINVAL, name ": bad dtype %d", dtype);                                \
  if (M == 0) return LORA_AMD_OK;                                                                               \
  hipStream_t st = (hipStream_t)stream;                                                                         \
  const int c8 = inner / 8;                                                                                     \
  const int64_t total = M * c8;                                                                                 \
  LORA_AMD_CHECK(total < (1ll << 31), LORA_AMD_EINVAL, name ": M * inner / 8 = %lld exceeds 2^31", (long long)total)

extern "C" int lora_amd_geglu_fwd(const void *y, int64_t ldy, void *out, int64_t ldo, int64_t M, int32_t inner,
                                  int32_t dtype, void *stream) {
  GEGLU_CHECKS("geglu_fwd");
  LORA_AMD_CHECK(y && out, LORA_AMD_EINVAL, "geglu_fwd: null pointer");
  LORA_AMD_CHECK(ldy >= 2 * (int64_t)inner && ldo >= inner && ldy % 8 == 0 && ldo % 8 == 0 && aligned_for(y, dtype) &&
                     aligned_for(out, dtype),
                 LORA_AMD_EINVAL, "geglu_fwd: rows must be 16-byte aligned (ldy=%lld ldo=%lld)", (long long)ldy,
                 (long long)ldo);
  const dim3 grid((unsigned)((total + kHT * kHU - 1) / (kHT * kHU))), block(kHT);
#define GO(E)                                                                                                    \
  hipLaunchKernelGGL((geglu_fwd_kernel<E>), grid, block, 0, st, (const typename E::storage *)y, ldy,             \
                     (typename E::storage *)out, ldo, M, c8);                                                    \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_geglu_fwd");
}

extern "C" int lora_amd_geglu_bwd(const void *y, int64_t ldy, const void *gout, int64_t ldg, void *gy, int64_t ldgy,
                                  int64_t M, int32_t inner, int32_t dtype, void *stream) {
  GEGLU_CHECKS("geglu_bwd");
  LORA_AMD_CHECK(y && gout && gy, LORA_AMD_EINVAL, "geglu_bwd: null pointer");
  LORA_AMD_CHECK(ldy >= 2 * (int64_t)inner && ldgy >= 2 * (int64_t)inner && ldg >= inner && ldy % 8 == 0 &&
                     ldgy % 8 == 0 && ldg % 8 == 0 && aligned_for(y, dtype) && aligned_for(gout, dtype) &&
                     aligned_for(gy, dtype),
                 LORA_AMD_EINVAL, "geglu_bwd: rows must be 16-byte aligned");
  const dim3 grid((unsigned)((total + kHT * 2 - 1) / (kHT * 2))), block(kHT);
#define GO(E)                                                                                                    \
  hipLaunchKernelGGL((geglu_bwd_kernel<E>), grid, block, 0, st, (const typename E::storage *)y, ldy,             \
                     (const typename E::storage *)gout, ldg, (typename E::storage *)gy, ldgy, M, c8);            \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_geglu_bwd");
}

extern "C" int lora_amd_layernorm_supported(int32_t K) { return K > 0 && K % 8 == 0 && ln_logL(K / 8) >= 0 ? 1 : 0; }

#define LN_CHECKS(name)                                                                                          \
  LORA_AMD_CHECK(M >= 0 && lora_amd_layernorm_supported(K), LORA_AMD_EINVAL, name ": bad shape M=%lld K=%d",      \
                 (long long)M, K);                                                                               \
  LORA_AMD_CHECK(dtype_ok(dtype), LORA_AMD_EINVAL, name ": bad dtype %d", dtype);                                 \
  if (M == 0) return LORA_AMD_OK;                                                                                \
  hipStream_t st = (hipStream_t)stream;                                                                          \
  const int c8 = K / 8, logL = ln_logL(c8);                                                                      \
  const int rows_per_block = kHT >> logL;                                                                        \
  const dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block)), block(kHT)

extern "C" int lora_amd_layernorm_fwd(const void *x, const void *gamma, const void *beta, void *y, float *stats,
                                      int64_t M, int32_t K, float eps, int32_t dtype, void *stream) {
  return lora_amd_add_layernorm_fwd(x, nullptr, gamma, beta, nullptr, y, stats, M, K, eps, dtype, stream);
}

extern "C" int lora_amd_add_layernorm_fwd(const void *x, const void *res, const void *gamma, const void *beta,
                                          void *sum_out, void *y, float *stats, int64_t M, int32_t K, float eps,
                                          int32_t dtype, void *stream) {
  LN_CHECKS("layernorm_fwd");
  LORA_AMD_CHECK(x && gamma && beta && y && stats, LORA_AMD_EINVAL, "layernorm_fwd: null pointer");
  LORA_AMD_CHECK((res == nullptr) == (sum_out == nullptr), LORA_AMD_EINVAL,
                 "layernorm_fwd: res and sum_out go together");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(y, dtype) && aligned_for(gamma, dtype) && aligned_for(beta, dtype) &&
                     aligned_for(res, dtype) && aligned_for(sum_out, dtype),
                 LORA_AMD_EINVAL, "layernorm_fwd: unaligned tensor");
  const bool add = res != nullptr;
#define GO(E)                                                                                                    \
  if (add)                                                                                                       \
    hipLaunchKernelGGL((ln_fwd_kernel<E, true>), grid, block, 0, st, (const typename E::storage *)x,             \
                       (const typename E::storage *)res, (const typename E::storage *)gamma,                     \
                       (const typename E::storage *)beta, (typename E::storage *)sum_out,                        \
                       (typename E::storage *)y, stats, M, c8, logL, eps);                                       \
  else                                                                                                           \
    hipLaunchKernelGGL((ln_fwd_kernel<E, false>), grid, block, 0, st, (const typename E::storage *)x,            \
                       (const typename E::storage *)nullptr, (const typename E::storage *)gamma,                 \
                       (const typename E::storage *)beta, (typename E::storage *)nullptr,                        \
                       (typename E::storage *)y, stats, M, c8, logL, eps);                                       \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_layernorm_fwd");
}

extern "C" int lora_amd_layernorm_bwd(const void *x, const void *gout, const void *gamma, const float *stats,
                                      void *dx, int64_t M, int32_t K, int32_t dtype, void *stream) {
  return lora_amd_add_layernorm_bwd(x, gout, nullptr, gamma, stats, dx, M, K, dtype, stream);
}

extern "C" int lora_amd_add_layernorm_bwd(const void *x, const void *gout, const void *gsum, const void *gamma,
                                          const float *stats, void *dx, int64_t M, int32_t K, int32_t dtype,
                                          void *stream) {
  LN_CHECKS("layernorm_bwd");
  LORA_AMD_CHECK(x && gout && gamma && stats && dx, LORA_AMD_EINVAL, "layernorm_bwd: null pointer");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(gout, dtype) && aligned_for(dx, dtype) && aligned_for(gamma, dtype) &&
                     aligned_for(gsum, dtype),
                 LORA_AMD_EINVAL, "layernorm_bwd: unaligned tensor");
#define GO(E)                                                                                                    \
  if (gsum != nullptr)                                                                                           \
    hipLaunchKernelGGL((ln_bwd_kernel<E, true>), grid, block, 0, st, (const typename E::storage *)x,             \
                       (const typename E::storage *)gout, (const typename E::storage *)gsum,                     \
                       (const typename E::storage *)gamma, stats, (typename E::storage *)dx, M, c8, logL);       \
  else                                                                                                           \
    hipLaunchKernelGGL((ln_bwd_kernel<E, false>), grid, block, 0, st, (const typename E::storage *)x,            \
                       (const typename E::storage *)gout, (const typename E::storage *)nullptr,                  \
                       (const typename E::storage *)gamma, stats, (typename E::storage *)dx, M, c8, logL);       \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_layernorm_bwd");
}

// ---- channels_last GroupNorm entry points ----------------------------------------------------------------------
static size_t gn_nhwc_part_floats(const GnNhwcGeo &q, int B, int C) { return (size_t)B * q.S * 2 * C; }

extern "C" size_t lora_amd_groupnorm_nhwc_workspace(int32_t B, int32_t C, int32_t HW, int32_t groups) {
  const GnNhwcGeo q = gn_nhwc_geo(B, C, HW, groups);
  return q.ok ? (gn_nhwc_part_floats(q, B, C) + (size_t)B * 2 * C) * sizeof(float) : 0;
}

#define GN_NHWC_CHECKS(name)                                                                                          \
  const GnNhwcGeo q = gn_nhwc_geo(B, C, HW, groups);                                                                  \
  LORA_AMD_CHECK(q.ok, LORA_AMD_EINVAL, name ": geometry B=%d C=%d HW=%d groups=%d not supported", B, C, HW, groups); \
  LORA_AMD_CHECK(dtype_ok(dtype), LORA_AMD_EINVAL, name ": bad dtype %d", dtype);                                     \
  LORA_AMD_CHECK(workspace_bytes >= lora_amd_groupnorm_nhwc_workspace(B, C, HW, groups), LORA_AMD_EWORKSPACE,         \
                 name ": workspace %zu bytes too small", workspace_bytes);                                            \
  const dim3 grid((unsigned)((int64_t)B * q.S * q.tiles)), gridg((unsigned)(B * groups)), block(kHT);                 \
  hipStream_t st = (hipStream_t)stream;                                                                               \
  float *part = reinterpret_cast<float *>(workspace);                                                                 \
  float *cvec = part + gn_nhwc_part_floats(q, B, C);                                                                  \
  (void)cvec

extern "C" int lora_amd_groupnorm_nhwc_fwd(const void *x, const void *gamma, const void *beta, const float *addend,
                                           void *y, float *aff, void *workspace, size_t workspace_bytes, int32_t B, int32_t C, int32_t HW,
                                           int32_t groups, float eps, int32_t act, int32_t dtype, void *stream) {
  GN_NHWC_CHECKS("groupnorm_nhwc_fwd");
  LORA_AMD_CHECK(x && gamma && beta && y && aff && workspace, LORA_AMD_EINVAL, "groupnorm_nhwc_fwd: null pointer");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(y, dtype) && ((uintptr_t)aff % 32) == 0, LORA_AMD_EINVAL,
                 "groupnorm_nhwc_fwd: unaligned tensor");
#define GO(E)                                                                                                         \
  {                                                                                                                   \
    using S_ = typename E::storage;                                                                                   \
    hipLaunchKernelGGL((gn_nhwc_stats_kernel<E>), grid, block, 0, st, (const S_ *)x, part, C, HW, q.cw, q.tiles,      \
                       q.nslots, q.px, q.S);                                                                          \
    hipLaunchKernelGGL((gn_nhwc_finalize_kernel<E>), gridg, block, 0, st, part, (const S_ *)gamma, (const S_ *)beta,  \
                       addend, aff, C, HW, groups, q.px, q.S, eps);                                                   \
    if (act)                                                                                                          \
      hipLaunchKernelGGL((gn_nhwc_apply_kernel<E, true>), grid, block, 0, st, (const S_ *)x, aff, (S_ *)y, C, HW,     \
                         q.cw, q.tiles, q.nslots, q.px, q.S);                                                         \
    else                                                                                                              \
      hipLaunchKernelGGL((gn_nhwc_apply_kernel<E, false>), grid, block, 0, st, (const S_ *)x, aff, (S_ *)y, C, HW,    \
                         q.cw, q.tiles, q.nslots, q.px, q.S);                                                         \
  }                                                                                                                   \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
  return check_launch("lora_amd_groupnorm_nhwc_fwd");
}

extern "C" int lora_amd_groupnorm_nhwc_bwd(const void *x, const void *gout, const void *gamma, const float *aff,
                                           void *dx, void *workspace, size_t workspace_bytes, int32_t B, int32_t C,
                                           int32_t HW, int32_t groups, int32_t act, int32_t dtype, void *stream) {
  GN_NHWC_CHECKS("groupnorm_nhwc_bwd");
  LORA_AMD_CHECK(x && gout && gamma && aff && dx && workspace, LORA_AMD_EINVAL, "groupnorm_nhwc_bwd: null pointer");
  LORA_AMD_CHECK(aligned_for(x, dtype) && aligned_for(gout, dtype) && aligned_for(dx, dtype) &&
                     aligned_for(gamma, dtype) && ((uintptr_t)aff % 32) == 0,
                 LORA_AMD_EINVAL, "groupnorm_nhwc_bwd: unaligned tensor");
#define GO2(E, A)                                                                                                     \
  {                                                                                                                   \
    using S_ = typename E::storage;                                                                                   \
    hipLaunchKernelGGL((gn_nhwc_bwd_kernel<E, A, false>), grid, block, 0, st, (const S_ *)x, (const S_ *)gout,        \
                       (const S_ *)gamma, aff, part, (const float *)nullptr, (S_ *)nullptr, C, HW, q.cw, q.tiles,     \
                       q.nslots, q.px, q.S);                                                                          \
    hipLaunchKernelGGL(gn_nhwc_bwd_finalize_kernel, gridg, block, 0, st, part, cvec, C, HW, groups, q.S);             \
    hipLaunchKernelGGL((gn_nhwc_bwd_kernel<E, A, true>), grid, block, 0, st, (const S_ *)x, (const S_ *)gout,         \
                       (const S_ *)gamma, aff, (float *)nullptr, cvec, (S_ *)dx, C, HW, q.cw, q.tiles, q.nslots,      \
                       q.px, q.S);                                                                                    \
  }
#define GO(E) \
  if (act) GO2(E, true) else GO2(E, false) \
  break
  switch (dtype) {
    case LORA_AMD_F32: GO(f32_t);
    case LORA_AMD_F16: GO(f16_t);
    default: GO(bf16_t);
  }
#undef GO
#undef GO2
  return check_launch("lora_amd_groupnorm_nhwc_bwd");
}
