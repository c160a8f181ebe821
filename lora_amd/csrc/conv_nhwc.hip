// K4, channels-last form — the 3x3 down-projection of LoraInjectedConv2d and its two gradients on the matrix cores.
//
// replaces: lora_diffusion/lora.py:131 `self.lora_down(input)` (Conv2d in -> r, 3x3, padding 1) and the autograd of
//           that call (input gradient + weight gradient), for activations stored channels-last ([B, H, W, C] in
//           memory, what MIOpen's NHWC convolutions of the frozen UNet produce and consume).
//
// Why a second form.  In NHWC a pixel's channels are contiguous, so the contraction over input channels is the K
// dimension of an MFMA with 16-byte operand loads straight from the tensor (no LDS transpose), and everything that is
// 1x1 in the adapter — the up-projection, its gradient pass over G, a 1x1 down-projection — IS the Linear adapter on
// the [B*H*W, C] matrix: those reuse csrc/linear_fused.hip unchanged.  What is left are the three 3x3 contractions:
//   T[p, j]        = sum_{tap, c} down[j, c, tap] X[p + tap, c]            conv3_down_nhwc_kernel
//   dX[p, c]      += sum_{tap, j} down[j, c, tap] Gt[p - tap, j]           conv3_dx_nhwc_kernel
//   dDown[j,c,tap] = sum_p        Gt[p, j] X[p + tap, c]                   conv3_ddown_nhwc_kernel
// (tap = (dy, dx) in {-1,0,1}^2, zero outside the image).  The NCHW kernels of conv.hip do these on the f32 VALU with
// channel-split partial sums (1.4-1.7x the algorithmic traffic, rank groups of 4 re-reading X); here
//   * every one is v_mfma_f32_16x16x32 with the operands swapped so that a lane's accumulator holds 4 CONSECUTIVE
//     ranks / channels of one pixel: T leaves as 16-byte stores, dX as 8-byte read-modify-writes;
//   * a workgroup owns a 16-column x PT-row pixel tile; its four waves split the CHANNEL range and meet once in LDS
//     (T) or own disjoint channels (dX): no partial buffers in HBM, X and dX are touched exactly once
//     (+ the halo rows/columns, which are L2 hits);
//   * `down` is re-packed per call into MFMA fragment order in the activation dtype (conv3_pack_kernel; 9*C*16
//     elements, 92 KB at C = 320) so that a wave fetches a fragment with one coalesced 1 KB load; at r <= 8 the spare
//     fragment rows carry the low 16-bit parts of the f32 masters, so T is as precise as f32 factors at no cost;
//   * dDown contracts over PIXELS, which are strided in NHWC: the X^T fragment is gathered with 2-byte loads (a lane =
//     one channel x 8 pixels), the shifted-Gt fragment with 4-byte loads (Gt is r/C of the stream); the pixel range is
//     split over `nsplit` workgroups x 4 waves, the waves meet in LDS, and the `nsplit` partials are folded by the
//     trainer's batched reduce (same [parts][rank_pad][C*9] layout as conv.hip's).
// Algorithmic bytes per site: forward B*H*W*C*e (X once); backward 2*B*H*W*C*e (dX read + write) + B*H*W*C*e (X once);
// the [B*H*W, r] f32 tensors are 2r/C of that.  All three are HBM/L2-stream bound (9*r*2 flop per 2-byte element is
// 144 flop/B at r = 16, under the 312 flop/B MFMA balance).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace lora_amd {

typedef float nf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int nu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int nu32x2 __attribute__((ext_vector_type(2)));

template <class E> struct NhMfma;
template <> struct NhMfma<bf16_t> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  __device__ static nf32x4 mma(frag a, frag b, nf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct NhMfma<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  __device__ static nf32x4 mma(frag a, frag b, nf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int kNhThreads = 256;  // 4 waves

struct NhGeom {
  int B, H, W, C, r;
  int ntc, nrg;  // 16-column tiles per image row, PT-row groups per image
};

template <class E>
__device__ inline typename NhMfma<E>::frag nh_frag(const float (&v)[8]) {
  union { typename NhMfma<E>::frag f; typename E::storage s[8]; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.s[i] = E::from_f(v[i]);
  return u.f;
}
template <class E>
__device__ inline typename NhMfma<E>::frag nh_frag_bits(nu32x4 v) {
  union { typename NhMfma<E>::frag f; nu32x4 u; } c;
  c.u = v;
  return c.f;
}
__device__ inline int nh_clamp(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// ============================================================================ down [r, C, 3, 3] f32 -> fragment order
// pf (forward, A operand of T^T = down X^T): fragment (tap, kc) = 64 lanes x 8 elements,
//     lane l, element e  <-  down[row(l & 15)][kc*32 + (l >> 4)*8 + e][tap]
//     rows 0..r-1 = the ranks; at r <= 8 rows 8..8+r-1 = the low parts (v - round(v)) of ranks 0..r-1, else zero.
// pd (input gradient, A operand of dX^T = down^T Gt^T): fragment (ct, ks),
//     lane l, element e  <-  down[j][ct*16 + (l & 15)][tap]  with slot s = ks*32 + (l >> 4)*8 + e = tap*r + j (0 beyond 9r).
template <class E>
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float *__restrict__ down, int r, int C, int KS,
                                                         typename E::storage *__restrict__ pf,
                                                         typename E::storage *__restrict__ pd) {
  const int KC = C >> 5;
  const int64_t npf = (int64_t)9 * KC * 64, npd = (int64_t)(C >> 4) * KS * 64;
  const bool lo_rows = r <= 8;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < npf + npd; id += (int64_t)gridDim.x * 256) {
    Chunk8<E> c;
    if (id < npf) {
      const int lane = (int)(id & 63);
      const int f = (int)(id >> 6);
      const int kc = f % KC, tap = f / KC;
      const int row = lane & 15, c0 = kc * 32 + (lane >> 4) * 8;
      const bool lo = lo_rows && row >= 8;
      const int j = lo ? row - 8 : row;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = j < r ? down[((int64_t)j * C + c0 + e) * 9 + tap] : 0.f;
        c.v[e] = E::from_f(lo ? v - E::to_f(E::from_f(v)) : v);
      }
      *reinterpret_cast<Chunk8<E> *>(pf + id * 8) = c;
    } else {
      const int64_t id2 = id - npf;
      const int lane = (int)(id2 & 63);
      const int f = (int)(id2 >> 6);
      const int ks = f % KS, ct = f / KS;
      const int ch = ct * 16 + (lane & 15), s0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int s = s0 + e;
        const int tap = s / r, j = s - tap * r;
        c.v[e] = E::from_f(s < 9 * r ? down[((int64_t)j * C + ch) * 9 + tap] : 0.f);
      }
      *reinterpret_cast<Chunk8<E> *>(pd + id2 * 8) = c;
    }
  }
}

// ============================================================================ T = conv3x3(X; down), [B*H*W, r] f32
// grid = B * nrg * ntc workgroups; workgroup = pixel tile (PT rows x 16 columns); wave w takes the 32-channel k-steps
// kc = w, w + 4, ...  Per k-step and column shift dx: 3 weight fragments (dy = -1, 0, 1) and the PT + 2 input rows the
// tile's taps touch, each row fragment feeding up to 3 MFMAs (one per dy).  Out-of-image taps: clamped address, value
// ANDed to zero (no branch, no load behind a branch).
template <class E, int PT>
__global__ __launch_bounds__(kNhThreads) void conv3_down_nhwc_kernel(const typename E::storage *__restrict__ x,
                                                                     const typename E::storage *__restrict__ pf,
                                                                     float *__restrict__ t_out, const NhGeom g) {
  using S = typename E::storage;
  using F = typename NhMfma<E>::frag;
  __shared__ __attribute__((aligned(16))) float red[4 * PT * 64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  int bid = blockIdx.x;
  const int tc = bid % g.ntc;
  bid /= g.ntc;
  const int rgi = bid % g.nrg, b = bid / g.nrg;
  const int H = g.H, W = g.W, C = g.C, KC = C >> 5;
  const int y0 = rgi * PT, xx = tc * 16 + l15;

  int rowoff[PT + 2], coloff[3];
  unsigned rowmask[PT + 2], colmask[3];
#pragma unroll
  for (int q = 0; q < PT + 2; ++q) {
    const int yy = y0 + q - 1;
    rowoff[q] = ((b * H + nh_clamp(yy, H - 1)) * W) * C;
    rowmask[q] = (yy >= 0 && yy < H) ? 0xFFFFFFFFu : 0u;
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int xs = xx + d - 1;
    coloff[d] = nh_clamp(xs, W - 1) * C + lg * 8;
    colmask[d] = (xs >= 0 && xs < W) ? 0xFFFFFFFFu : 0u;
  }

  nf32x4 acc[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) acc[t] = (nf32x4){0.f, 0.f, 0.f, 0.f};

  for (int kc = wave; kc < KC; kc += 4) {
    const S *xk = x + kc * 32;
    const S *pk = pf + ((int64_t)kc * 64 + lane) * 8;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      F pa[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) pa[dy] = *reinterpret_cast<const F *>(pk + (int64_t)((dy * 3 + d) * KC) * 512);
      nu32x4 xr[PT + 2];
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) xr[q] = *reinterpret_cast<const nu32x4 *>(xk + rowoff[q] + coloff[d]);
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) {
        const unsigned m = rowmask[q] & colmask[d];
        xr[q] = xr[q] & (nu32x4){m, m, m, m};
      }
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) acc[t] = NhMfma<E>::mma(pa[dy], nh_frag_bits<E>(xr[t + dy]), acc[t]);
    }
  }

  // the four channel shares meet in LDS; wave w finishes rows t = w, w + 4, ... of the tile.
  // acc[e] = T^T[row lg*4 + e][pixel l15]: rows 0..7 ranks (+ rows 8..15 their low-part products when r <= 8)
#pragma unroll
  for (int t = 0; t < PT; ++t) *reinterpret_cast<nf32x4 *>(red + ((wave * PT + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  const int r = g.r;
  for (int t = wave; t < PT; t += 4) {
    nf32x4 v = (nf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const nf32x4 *>(red + ((w * PT + t) * 64 + lane) * 4);
    if (r <= 8 && lg < 2) {
#pragma unroll
      for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const nf32x4 *>(red + ((w * PT + t) * 64 + lane + 32) * 4);
    }
    const int yy = y0 + t;
    if (yy < H && xx < W && lg * 4 < r) {
      float *dst = t_out + (((int64_t)b * H + yy) * W + xx) * r + lg * 4;
      if ((r & 3) == 0) {
        *reinterpret_cast<nf32x4 *>(dst) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (lg * 4 + e < r) dst[e] = v[e];
      }
    }
  }
}

// ============================================================================ dX += conv_transpose3x3(Gt; down)
// Same pixel tiles.  The (tap, rank) contraction has 9r slots = KS k-steps of 32; the tile's Gt fragments (shifted by
// the tap of each slot group, zero outside the image) are built once and stay in registers; the waves walk disjoint
// 64-channel blocks: per 16-channel subtile KS weight fragments, KS MFMAs per tile row, one 8-byte read-modify-write of
// dX per lane (a lane owns 4 consecutive channels of one pixel; the 4 subtiles of a block cover the pixel's 128-byte line).
template <class E, int PT, int KS>
__global__ __launch_bounds__(kNhThreads) void conv3_dx_nhwc_kernel(typename E::storage *__restrict__ dx,
                                                                   const float *__restrict__ gt,
                                                                   const typename E::storage *__restrict__ pd,
                                                                   const NhGeom g) {
  using S = typename E::storage;
  using F = typename NhMfma<E>::frag;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  int bid = blockIdx.x;
  const int tc = bid % g.ntc;
  bid /= g.ntc;
  const int rgi = bid % g.nrg, b = bid / g.nrg;
  const int H = g.H, W = g.W, C = g.C, r = g.r;
  const int y0 = rgi * PT, xx = tc * 16 + l15;

  // raw loads first (clamped addresses), masks applied afterwards: PT*KS*2 independent 16-byte loads in flight
  nu32x4 graw[PT][KS][2];
  unsigned gmask[PT][KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int s = ks * 32 + lg * 8 + 4 * h;
      const bool live = s < 9 * r;
      const int tap = live ? s / r : 0;
      const int j = live ? s - tap * r : 0;
      const int dy = tap / 3 - 1, dxx = tap % 3 - 1;
      const int xs = xx - dxx;
      const bool xok = live && xs >= 0 && xs < W;
      const int xc = nh_clamp(xs, W - 1);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const int ys = y0 + t - dy;
        const float *src = gt + (((int64_t)b * H + nh_clamp(ys, H - 1)) * W + xc) * r + j;
        graw[t][ks][h] = *reinterpret_cast<const nu32x4 *>(src);
        gmask[t][ks][h] = (xok && ys >= 0 && ys < H) ? 0xFFFFFFFFu : 0u;
      }
    }
  F bg[PT][KS];
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned m = gmask[t][ks][h];
        const nu32x4 q = graw[t][ks][h] & (nu32x4){m, m, m, m};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[h * 4 + e] = __builtin_bit_cast(float, q[e]);
      }
      bg[t][ks] = nh_frag<E>(v);
    }

  bool pok[PT];
  int poff[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int yy = y0 + t;
    pok[t] = yy < H && xx < W;
    poff[t] = ((b * H + nh_clamp(yy, H - 1)) * W + nh_clamp(xx, W - 1)) * C + lg * 4;
  }
  const int NCB = C >> 6;
  for (int cb = wave; cb < NCB; cb += 4) {
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int ct = cb * 4 + sub;
      F pa[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) pa[ks] = *reinterpret_cast<const F *>(pd + (((int64_t)ct * KS + ks) * 64 + lane) * 8);
      nu32x2 prev[PT];
#pragma unroll
      for (int t = 0; t < PT; ++t) prev[t] = *reinterpret_cast<const nu32x2 *>(dx + poff[t] + ct * 16);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        nf32x4 a = (nf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = NhMfma<E>::mma(pa[ks], bg[t][ks], a);
        union { nu32x2 v; S s[4]; } o;
        o.v = prev[t];
#pragma unroll
        for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(E::to_f(o.s[e]) + a[e]);
        if (pok[t]) *reinterpret_cast<nu32x2 *>(dx + poff[t] + ct * 16) = o.v;
      }
    }
  }
}

// ============================================================================ dDown partials
// grid (C / 32, nsplit).  Workgroup (cg, sp): channels cg*32 .. +31, pixel blocks (32 consecutive flat pixels = one
// k-step) sp*4 + wave, + 4*nsplit, ...  A = X^T (rows = channels; a lane gathers one channel of 8 consecutive pixels
// with 2-byte loads), B = Gt shifted by the slot's tap (columns = (tap, rank) slots, NT tiles of 16).  The waves meet in
// LDS; wave 0 writes part[sp][j][c*9 + tap].
template <class E, int NT>
__global__ __launch_bounds__(kNhThreads) void conv3_ddown_nhwc_kernel(const typename E::storage *__restrict__ x,
                                                                      const float *__restrict__ gt,
                                                                      float *__restrict__ part, const NhGeom g,
                                                                      int nblk, int nsplit, int rank_pad) {
  using F = typename NhMfma<E>::frag;
  constexpr int CT = 2;
  __shared__ __attribute__((aligned(16))) float red[3 * CT * NT * 64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int cg = blockIdx.x, sp = blockIdx.y;
  const int H = g.H, W = g.W, C = g.C, r = g.r, HW = H * W;
  const int M = g.B * HW;
  const unsigned short *xs16 = reinterpret_cast<const unsigned short *>(x);

  int s_dy[NT], s_dx[NT], s_off[NT], s_j[NT], s_tap[NT];
  bool s_live[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int s = nt * 16 + l15;
    s_live[nt] = s < 9 * r;
    const int tap = s_live[nt] ? s / r : 0;
    s_tap[nt] = tap;
    s_j[nt] = s_live[nt] ? s - tap * r : 0;
    s_dy[nt] = tap / 3 - 1;
    s_dx[nt] = tap % 3 - 1;
    s_off[nt] = -(s_dy[nt] * W + s_dx[nt]) * r + s_j[nt];  // Gt index of pixel p shifted by -tap: p*r + s_off
  }

  nf32x4 acc[CT][NT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[ct][nt] = (nf32x4){0.f, 0.f, 0.f, 0.f};

  for (int blk = sp * 4 + wave; blk < nblk; blk += nsplit * 4) {
    const int p0 = blk * 32 + lg * 8;
    int py[8], px[8];
    bool pv[8];
    {
      const int pc = p0 < M ? p0 : 0;
      const int rem = pc % HW;
      int y = rem / W, xq = rem - y * W;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = p0 + e < M;
        py[e] = y;
        px[e] = xq;
        if (++xq == W) { xq = 0; if (++y == H) y = 0; }
      }
    }
    // X^T fragments: raw 2-byte gathers (clamped rows), zeroed past M
    unsigned short xraw[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = pv[e] ? p0 + e : M - 1;
        xraw[ct][e] = xs16[(int64_t)p * C + cg * 32 + ct * 16 + l15];
      }
    F xa[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      union { F f; unsigned short s[8]; } u;
#pragma unroll
      for (int e = 0; e < 8; ++e) u.s[e] = pv[e] ? xraw[ct][e] : (unsigned short)0;
      xa[ct] = u.f;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float graw[8];
      bool ok[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ys = py[e] - s_dy[nt], xs = px[e] - s_dx[nt];
        ok[e] = s_live[nt] && pv[e] && ys >= 0 && ys < H && xs >= 0 && xs < W;
        const int64_t idx = ok[e] ? (int64_t)(p0 + e) * r + s_off[nt] : 0;
        graw[e] = gt[idx];
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float q = graw[e];
        asm volatile("" : "+v"(q));  // keep the load where it is: a select, not a load behind a branch
        v[e] = ok[e] ? q : 0.f;
      }
      const F bq = nh_frag<E>(v);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct][nt] = NhMfma<E>::mma(xa[ct], bq, acc[ct][nt]);
    }
  }

  if (wave > 0) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<nf32x4 *>(red + ((((wave - 1) * CT + ct) * NT + nt) * 64 + lane) * 4) = acc[ct][nt];
  }
  __syncthreads();
  if (wave == 0) {
    const int64_t row_len = (int64_t)C * 9;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        nf32x4 v = acc[ct][nt];
#pragma unroll
        for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const nf32x4 *>(red + (((w * CT + ct) * NT + nt) * 64 + lane) * 4);
        // v[e] = dDown^T[channel cg*32 + ct*16 + lg*4 + e][slot nt*16 + l15]
        if (s_live[nt]) {
          float *dst = part + ((int64_t)sp * rank_pad + s_j[nt]) * row_len + (int64_t)(cg * 32 + ct * 16 + lg * 4) * 9 + s_tap[nt];
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e * 9] = v[e];
        }
      }
  }
}

// ============================================================================ out = sum_p part[p]  (Gt column-tile partials)
__global__ __launch_bounds__(256) void nh_sum_parts_kernel(const float *__restrict__ part, int nparts, int64_t stride4,
                                                           float *__restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const nf32x4 *pp = reinterpret_cast<const nf32x4 *>(part) + i;
  nf32x4 a = pp[0];
  for (int p = 1; p < nparts; ++p) a += pp[(int64_t)p * stride4];
  reinterpret_cast<nf32x4 *>(out)[i] = a;
}

// ---------------------------------------------------------------------------- host
static inline int nh_rank_pad(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : 16; }
static inline int nh_env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static int nh_pick_pt(int B, int H, int W) {
  const int forced = nh_env_int("LORA_AMD_NHWC_PT", 0);
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  const int64_t ntc = (W + 15) / 16;
  const int64_t t4 = (int64_t)B * ((H + 3) / 4) * ntc, t2 = (int64_t)B * ((H + 1) / 2) * ntc;
  if (t4 >= 200) return 4;
  if (t2 >= 100) return 2;
  return 1;
}
static inline bool nh_native(int B, int C, int H, int W, int r) {
  return B >= 1 && H >= 1 && W >= 1 && C >= 64 && C % 64 == 0 && r >= 4 && r <= 16 && r % 4 == 0 &&
         (int64_t)B * H * W * C < ((int64_t)1 << 31);
}
static int nh_pick_split(int64_t M, int C, int r) {
  const int forced = nh_env_int("LORA_AMD_NHWC_SPLIT", 0);
  const int64_t nblk = (M + 31) / 32;
  int64_t s = forced > 0 ? forced : (256 + C / 32 - 1) / (C / 32);
  // partials are nsplit * 18 r / M of the X stream: keep them under ~15 %
  const int64_t cap_bytes = std::max<int64_t>(1, (int64_t)(0.15 * (double)M / (18.0 * r)));
  if (forced <= 0) s = std::min<int64_t>(s, std::min<int64_t>(16, cap_bytes));
  s = std::min<int64_t>(s, std::max<int64_t>(1, nblk / 4));
  return (int)std::max<int64_t>(1, s);
}
static inline NhGeom nh_geom(int B, int C, int H, int W, int r, int pt) {
  NhGeom g;
  g.B = B; g.H = H; g.W = W; g.C = C; g.r = r;
  g.ntc = (W + 15) / 16;
  g.nrg = (H + pt - 1) / pt;
  return g;
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_conv3_nhwc_plan(int32_t B, int32_t C_in, int32_t H, int32_t W, int32_t r,
                                        lora_amd_conv3_nhwc_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr, LORA_AMD_EINVAL, "conv3_nhwc_plan: null output");
  LORA_AMD_CHECK(B >= 0 && C_in > 0 && H > 0 && W > 0, LORA_AMD_EINVAL, "conv3_nhwc_plan: bad shape");
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "conv3_nhwc_plan: rank %d outside [1,%d]", r,
                 LORA_AMD_MAX_RANK);
  memset(out, 0, sizeof(*out));
  if (!nh_native(B, C_in, H, W, r)) return LORA_AMD_OK;
  out->native = 1;
  out->pt = nh_pick_pt(B, H, W);
  out->ks = (9 * r + 31) / 32;
  out->rank_pad = nh_rank_pad(r);
  out->nsplit = nh_pick_split((int64_t)B * H * W, C_in, r);
  out->pf_elems = (int64_t)9 * C_in * 16;
  out->pd_elems = (int64_t)C_in * out->ks * 32;
  out->down_part_floats = (int64_t)out->nsplit * out->rank_pad * C_in * 9;
  return LORA_AMD_OK;
}

#define NH_COMMON(name)                                                                                          \
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16, LORA_AMD_EINVAL,                       \
                 name ": activations must be bf16 or f16");                                                       \
  LORA_AMD_CHECK(r >= 1 && r <= 16, LORA_AMD_ERANK, name ": rank %d outside [1,16]", r);                          \
  LORA_AMD_CHECK(nh_native(B, C_in, H, W, r), LORA_AMD_EINVAL,                                                    \
                 name ": needs C_in %% 64 == 0, rank in {4, 8, 12, 16}, B*H*W*C_in < 2^31 (lora_amd_conv3_nhwc_plan)")

extern "C" int lora_amd_conv3_nhwc_pack(const float *down, int32_t r, int32_t C_in, int32_t act_dtype, void *pf,
                                        void *pd, void *stream) {
  const int B = 1, H = 1, W = 1;
  NH_COMMON("conv3_nhwc_pack");
  LORA_AMD_CHECK(down && pf && pd, LORA_AMD_EINVAL, "conv3_nhwc_pack: null pointer");
  const int KS = (9 * r + 31) / 32;
  const int64_t pieces = (int64_t)9 * (C_in / 32) * 64 + (int64_t)(C_in / 16) * KS * 64;
  const unsigned grid = (unsigned)std::min<int64_t>((pieces + 255) / 256, 1024);
  if (act_dtype == LORA_AMD_BF16)
    hipLaunchKernelGGL(conv3_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, down, r, C_in, KS,
                       (__bf16 *)pf, (__bf16 *)pd);
  else
    hipLaunchKernelGGL(conv3_pack_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, down, r, C_in, KS,
                       (_Float16 *)pf, (_Float16 *)pd);
  return check_launch("lora_amd_conv3_nhwc_pack");
}

#define NH_BY_DTYPE(LAUNCH)                        \
  if (act_dtype == LORA_AMD_BF16) { LAUNCH(bf16_t) } \
  else { LAUNCH(f16_t) }

extern "C" int lora_amd_conv3_nhwc_down_fwd(const void *x, const void *pf, float *t_out, int32_t B, int32_t C_in,
                                            int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream) {
  NH_COMMON("conv3_nhwc_down_fwd");
  LORA_AMD_CHECK(x && pf && t_out, LORA_AMD_EINVAL, "conv3_nhwc_down_fwd: null pointer");
  LORA_AMD_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)pf % 16) == 0 && ((uintptr_t)t_out % 16) == 0, LORA_AMD_EINVAL,
                 "conv3_nhwc_down_fwd: 16-byte aligned buffers");
  const int pt = nh_pick_pt(B, H, W);
  const NhGeom g = nh_geom(B, C_in, H, W, r, pt);
  const unsigned grid = (unsigned)((int64_t)B * g.nrg * g.ntc);
#define NH_LAUNCH_T(E)                                                                                         \
  using S = typename E::storage;                                                                               \
  if (pt == 4) hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 4>), dim3(grid), dim3(kNhThreads), 0,            \
                                  (hipStream_t)stream, (const S *)x, (const S *)pf, t_out, g);                 \
  else if (pt == 2) hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 2>), dim3(grid), dim3(kNhThreads), 0,       \
                                       (hipStream_t)stream, (const S *)x, (const S *)pf, t_out, g);            \
  else hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 1>), dim3(grid), dim3(kNhThreads), 0, (hipStream_t)stream, \
                          (const S *)x, (const S *)pf, t_out, g);
  NH_BY_DTYPE(NH_LAUNCH_T)
#undef NH_LAUNCH_T
  return check_launch("lora_amd_conv3_nhwc_down_fwd");
}

extern "C" int lora_amd_conv3_nhwc_bwd_dx(void *dx, const float *gt, const void *pd, int32_t B, int32_t C_in,
                                          int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream) {
  NH_COMMON("conv3_nhwc_bwd_dx");
  LORA_AMD_CHECK(dx && gt && pd, LORA_AMD_EINVAL, "conv3_nhwc_bwd_dx: null pointer");
  LORA_AMD_CHECK(((uintptr_t)dx % 16) == 0 && ((uintptr_t)pd % 16) == 0 && ((uintptr_t)gt % 16) == 0, LORA_AMD_EINVAL,
                 "conv3_nhwc_bwd_dx: 16-byte aligned buffers");
  const int pt = std::min(nh_pick_pt(B, H, W), 2);  // PT * KS Gt fragments stay in registers
  const NhGeom g = nh_geom(B, C_in, H, W, r, pt);
  const unsigned grid = (unsigned)((int64_t)B * g.nrg * g.ntc);
  const int KS = (9 * r + 31) / 32;
#define NH_LAUNCH_DX2(E, PT_, KS_)                                                                      \
  hipLaunchKernelGGL((conv3_dx_nhwc_kernel<E, PT_, KS_>), dim3(grid), dim3(kNhThreads), 0, (hipStream_t)stream, \
                     (typename E::storage *)dx, gt, (const typename E::storage *)pd, g);
#define NH_LAUNCH_DX(E)                                                         \
  if (pt == 2) {                                                                \
    switch (KS) {                                                               \
      case 2: NH_LAUNCH_DX2(E, 2, 2) break;                                     \
      case 3: NH_LAUNCH_DX2(E, 2, 3) break;                                     \
      case 4: NH_LAUNCH_DX2(E, 2, 4) break;                                     \
      default: NH_LAUNCH_DX2(E, 2, 5) break;                                    \
    }                                                                           \
  } else {                                                                      \
    switch (KS) {                                                               \
      case 2: NH_LAUNCH_DX2(E, 1, 2) break;                                     \
      case 3: NH_LAUNCH_DX2(E, 1, 3) break;                                     \
      case 4: NH_LAUNCH_DX2(E, 1, 4) break;                                     \
      default: NH_LAUNCH_DX2(E, 1, 5) break;                                    \
    }                                                                           \
  }
  NH_BY_DTYPE(NH_LAUNCH_DX)
#undef NH_LAUNCH_DX
#undef NH_LAUNCH_DX2
  return check_launch("lora_amd_conv3_nhwc_bwd_dx");
}

extern "C" int lora_amd_conv3_nhwc_bwd_down(const void *x, const float *gt, float *down_part, int32_t B, int32_t C_in,
                                            int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream) {
  NH_COMMON("conv3_nhwc_bwd_down");
  LORA_AMD_CHECK(x && gt && down_part, LORA_AMD_EINVAL, "conv3_nhwc_bwd_down: null pointer");
  const int64_t M = (int64_t)B * H * W;
  const NhGeom g = nh_geom(B, C_in, H, W, r, 1);
  const int nblk = (int)((M + 31) / 32);
  const int nsplit = nh_pick_split(M, C_in, r), rank_pad = nh_rank_pad(r);
  const dim3 grid((unsigned)(C_in / 32), (unsigned)nsplit);
  const int NT = (9 * r + 15) / 16;  // r = 4: 3, 8: 5, 12: 7, 16: 9
#define NH_LAUNCH_DD2(E, NT_)                                                                                   \
  hipLaunchKernelGGL((conv3_ddown_nhwc_kernel<E, NT_>), grid, dim3(kNhThreads), 0, (hipStream_t)stream,        \
                     (const typename E::storage *)x, gt, down_part, g, nblk, nsplit, rank_pad);
#define NH_LAUNCH_DD(E)                    \
  switch (NT) {                            \
    case 3: NH_LAUNCH_DD2(E, 3) break;     \
    case 5: NH_LAUNCH_DD2(E, 5) break;     \
    case 7: NH_LAUNCH_DD2(E, 7) break;     \
    default: NH_LAUNCH_DD2(E, 9) break;    \
  }
  NH_BY_DTYPE(NH_LAUNCH_DD)
#undef NH_LAUNCH_DD
#undef NH_LAUNCH_DD2
  return check_launch("lora_amd_conv3_nhwc_bwd_down");
}

extern "C" int lora_amd_sum_parts(const float *part, int32_t nparts, int64_t stride, float *out, int64_t n,
                                  void *stream) {
  LORA_AMD_CHECK(part && out && nparts >= 1, LORA_AMD_EINVAL, "sum_parts: null pointer");
  LORA_AMD_CHECK(n % 4 == 0 && stride % 4 == 0 && ((uintptr_t)part % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 LORA_AMD_EINVAL, "sum_parts: n, stride multiples of 4 and 16-byte aligned buffers");
  if (n == 0) return LORA_AMD_OK;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(nh_sum_parts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     nparts, stride / 4, out, n4);
  return check_launch("lora_amd_sum_parts");
}
