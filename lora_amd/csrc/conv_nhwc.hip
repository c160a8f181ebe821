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
//   * dDown contracts over PIXELS, which are strided in NHWC: image-row strips of X (with zero margins, so that a tap
//     is a pure shift) and the transposed Gt are staged in LDS, the X^T fragments are gathered from there with 2-byte
//     reads; the strips are dealt to `nsplit` workgroups per 64-channel chunk, whose partials are folded by the
//     trainer's batched reduce (same [parts][rank_pad][C*9] layout as conv.hip's).
// Algorithmic bytes per site: forward B*H*W*C*e (X once); backward 2*B*H*W*C*e (dX read + write) + B*H*W*C*e (X once);
// the [B*H*W, r] f32 tensors are 2r/C of that.  All three are HBM/L2-stream bound (9*r*2 flop per 2-byte element is
// 144 flop/B at r = 16, under the 312 flop/B MFMA balance).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "mfma16.hpp"

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
// grid = (B * nrg * ntc, ksplit); workgroup = pixel tile (PT rows x 16 columns) x channel share kz; wave w takes the
// 32-channel k-steps kc = kz*4 + w, + 4*ksplit, ...  (small maps: the packed factor a workgroup pulls, 9*C*16 elements,
// outweighs its 16*PT pixels of X, so the channels are split over `ksplit` workgroups as well; each then writes its own
// partial T and lora_amd_sum_parts folds them).  Per k-step and column shift dx: 3 weight fragments and the PT + 2 input rows the
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
  // consecutive tiles (row groups of one image: they share halo rows) go to the same XCD, i.e. the same L2
  int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
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

  const int ksplit = gridDim.y;
  t_out += (int64_t)blockIdx.y * ((int64_t)g.B * H * W * g.r);  // this share's partial (the output itself at ksplit 1)
  for (int kc = blockIdx.y * 4 + wave; kc < KC; kc += 4 * ksplit) {
    const S *xk = x + kc * 32;
    const S *pk = pf + ((int64_t)kc * 64 + lane) * 8;
    // every load of the k-step is issued before the first MFMA (9 weight fragments + 3 x (PT + 2) row fragments):
    // the waits are then counted ones, in issue order, and the MFMAs start as the data arrives
    nu32x4 praw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) praw[tap] = *reinterpret_cast<const nu32x4 *>(pk + (int64_t)(tap * KC) * 512);
    nu32x4 xr[3][PT + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) xr[d][q] = *reinterpret_cast<const nu32x4 *>(xk + rowoff[q] + coloff[d]);
    // opaque uses in issue order: the scheduler may not sink a load below them (left alone it keeps two loads in
    // flight and waits for each pair: 14 memory round trips per k-step instead of one)
    F pa[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("" : "+v"(praw[tap]));
      pa[tap] = nh_frag_bits<E>(praw[tap]);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) asm volatile("" : "+v"(xr[d][q]));
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) {
        const unsigned m = rowmask[q] & colmask[d];
        xr[d][q] = xr[d][q] & (nu32x4){m, m, m, m};
      }
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) acc[t] = NhMfma<E>::mma(pa[dy * 3 + d], nh_frag_bits<E>(xr[d][t + dy]), acc[t]);
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

// ============================================================================ round 6: one launch per forward
// VERDICT r2..r5 "K4 launch fusion": a 3x3 site's forward was pack + down-conv (+ fold of its channel-share partials on small
// maps) + up-projection = 3-4 launches of 5-10 us each on maps that hold a few hundred KB — launch-bound, 44 sites a step.
//   * conv3_site_pack_kernel: the fragment packs of EVERY conv site of a model in one launch, once per optimiser step
//     (the factors change once per step, not per forward): pf, pd as conv3_pack_kernel writes them, plus pu — `up` [C_out, r]
//     as the A operand of the up-projection's transposed product (csrc/rank16_mfma.hip's rank_update16 builds the same
//     fragment from f32 per workgroup): per 32-column group and interleaved tile t, lane (i, kq) holds (kq < 2 ? hi : lo) of
//     up[n(i, t)][8 (kq & 1) + e], n(i, t) = 32 cg + 8 (i / 4) + 4 t + i % 4.
//   * conv3_fwd_fused_kernel: T = conv3x3(X; down) for the workgroup's pixel tile exactly as conv3_down_nhwc_kernel, then —
//     in the SAME launch — Y[tile pixels, :] += scale * mask o (T up^T) on the matrix cores, dropout regenerated from
//     Philox(seed, offset) with rank_update16's chunk indexing.  Small maps split the channel loop over `ksplit`
//     workgroups per tile: each writes its partial T write-through and arrives at the tile's counter; the LAST arriver sums
//     the partials in share order (deterministic), and runs the up-projection for the tile (the hand-off of
//     csrc/svd_small.hip: sc1 stores + vmcnt(0) + relaxed agent-scope arrival, one acquire fence in the last arriver, the
//     counter reset by it — nothing to clear between launches, hipGraph-replayable).
struct NhPackSite {   // mirrors lora_amd_conv3_pack_site
  const float *down, *up;
  void *pf, *pd, *pu;
  int32_t r, C_in, C_out, KS;
  int64_t begin;
};

template <class E>
__global__ __launch_bounds__(256) void conv3_site_pack_kernel(const NhPackSite *__restrict__ sites, int n, int64_t total) {
  using S = typename E::storage;
  for (int64_t id0 = (int64_t)blockIdx.x * 256 + threadIdx.x; id0 < total; id0 += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sites[mid].begin <= id0) lo = mid; else hi = mid - 1;
    }
    const NhPackSite q = sites[lo];
    const int r = q.r, C = q.C_in, KS = q.KS, KC = C >> 5;
    const int64_t npf = (int64_t)9 * KC * 64, npd = (int64_t)(C >> 4) * KS * 64;
    int64_t id = id0 - q.begin;
    Chunk8<E> c;
    S *dst;
    if (id < npf) {
      const int lane = (int)(id & 63), f = (int)(id >> 6);
      const int kc = f % KC, tap = f / KC;
      const int row = lane & 15, c0 = kc * 32 + (lane >> 4) * 8;
      const bool lo_rows = r <= 8, lop = lo_rows && row >= 8;
      const int j = lop ? row - 8 : row;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = j < r ? gl(q.down)[((int64_t)j * C + c0 + e) * 9 + tap] : 0.f;
        c.v[e] = E::from_f(lop ? v - E::to_f(E::from_f(v)) : v);
      }
      dst = reinterpret_cast<S *>(q.pf) + id * 8;
    } else if (id < npf + npd) {
      id -= npf;
      const int lane = (int)(id & 63), f = (int)(id >> 6);
      const int ks = f % KS, ct = f / KS;
      const int ch = ct * 16 + (lane & 15), s0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sl = s0 + e;
        const int tap = sl / r, j = sl - tap * r;
        c.v[e] = E::from_f(sl < 9 * r ? gl(q.down)[((int64_t)j * C + ch) * 9 + tap] : 0.f);
      }
      dst = reinterpret_cast<S *>(q.pd) + id * 8;
    } else {
      id -= npf + npd;   // piece = (column group, tile, lane)
      const int lane = (int)(id & 63), tile = (int)((id >> 6) & 1), cg = (int)(id >> 7);
      const int i = lane & 15, kq = lane >> 4;
      const int nn = cg * 32 + 8 * (i >> 2) + 4 * tile + (i & 3), j0 = 8 * (kq & 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (nn < q.C_out && j0 + e < r) ? gl(q.up)[(int64_t)nn * r + j0 + e] : 0.f;
        const S h = E::from_f(v);
        c.v[e] = kq < 2 ? h : E::from_f(v - E::to_f(h));
      }
      dst = reinterpret_cast<S *>(q.pu) + id * 8;
    }
    union { Chunk8<E> c; nu32x4 u; } o;
    o.c = c;
    *gl(reinterpret_cast<nu32x4 *>(dst)) = o.u;
  }
}

template <class E, int PT, bool DROP>
__global__ __launch_bounds__(kNhThreads) void conv3_fwd_fused_kernel(
    const typename E::storage *__restrict__ x, const typename E::storage *__restrict__ pf,
    const typename E::storage *__restrict__ pu, typename E::storage *__restrict__ y, float *__restrict__ t_out,
    float *__restrict__ t_part, unsigned *__restrict__ counters, const NhGeom g, int Co, float scale, float p, uint64_t seed,
    uint64_t offset, const uint64_t *offset_dev) {
  using S = typename E::storage;
  using F = typename NhMfma<E>::frag;
  __shared__ __attribute__((aligned(16))) float red[4 * PT * 64 * 4];
  __shared__ __attribute__((aligned(16))) float s_T[PT * 16 * 16];       // [tile row][pixel][rank] of the finished tile
  __shared__ __attribute__((aligned(16))) nu32x4 s_tf[PT * 2 * 64];      // [tile row][hi, lo][lane]: B operands of the update
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int tile_id = (int)xcd_remap(blockIdx.x, gridDim.x);
  int bid = tile_id;
  const int tc = bid % g.ntc;
  bid /= g.ntc;
  const int rgi = bid % g.nrg, b = bid / g.nrg;
  const int H = g.H, W = g.W, C = g.C, KC = C >> 5, r = g.r;
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
  const int ksplit = gridDim.y;
  for (int kc = blockIdx.y * 4 + wave; kc < KC; kc += 4 * ksplit) {   // conv3_down_nhwc_kernel's k loop
    const S *xk = x + kc * 32;
    const S *pk = pf + ((int64_t)kc * 64 + lane) * 8;
    nu32x4 praw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) praw[tap] = *gl(reinterpret_cast<const nu32x4 *>(pk + (int64_t)(tap * KC) * 512));
    nu32x4 xr[3][PT + 2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) xr[d][q] = *gl(reinterpret_cast<const nu32x4 *>(xk + rowoff[q] + coloff[d]));
    F pa[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("" : "+v"(praw[tap]));
      pa[tap] = nh_frag_bits<E>(praw[tap]);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) asm volatile("" : "+v"(xr[d][q]));
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
      for (int q = 0; q < PT + 2; ++q) {
        const unsigned m = rowmask[q] & colmask[d];
        xr[d][q] = xr[d][q] & (nu32x4){m, m, m, m};
      }
#pragma unroll
      for (int t = 0; t < PT; ++t)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) acc[t] = NhMfma<E>::mma(pa[dy * 3 + d], nh_frag_bits<E>(xr[d][t + dy]), acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < PT; ++t) *reinterpret_cast<nf32x4 *>(red + ((wave * PT + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  // wave w finishes tile rows t = w, w + 4, ...: v[e] = (this workgroup's share of) T[pixel l15 of row t][rank 4 lg + e]
  const bool rank_live = lg * 4 < r;
  const int64_t Mr = (int64_t)g.B * H * W * r;
  for (int t = wave; t < PT; t += 4) {
    nf32x4 v = (nf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const nf32x4 *>(red + ((w * PT + t) * 64 + lane) * 4);
    if (r <= 8 && lg < 2) {
#pragma unroll
      for (int w = 0; w < 4; ++w) v += *reinterpret_cast<const nf32x4 *>(red + ((w * PT + t) * 64 + lane + 32) * 4);
    }
    const int yy = y0 + t;
    const bool live = yy < H && xx < W && rank_live;
    const int64_t pix = ((int64_t)b * H + nh_clamp(yy, H - 1)) * W + nh_clamp(xx, W - 1);
    if (ksplit > 1) {   // write-through partial of this channel share
      float *dst = t_part + (int64_t)blockIdx.y * Mr + pix * r + lg * 4;
      if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e) __hip_atomic_store(dst + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) s_T[(t * 16 + l15) * 16 + lg * 4 + e] = (live && lg * 4 + e < r) ? v[e] : 0.f;
    }
  }
  if (ksplit > 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned prev = __hip_atomic_fetch_add(counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)(ksplit - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int t = wave; t < PT; t += 4) {   // the shares in order: the same sum whichever workgroup arrives last
      const int yy = y0 + t;
      const bool live = yy < H && xx < W && rank_live;
      const int64_t pix = ((int64_t)b * H + nh_clamp(yy, H - 1)) * W + nh_clamp(xx, W - 1);
      nf32x4 v = (nf32x4){0.f, 0.f, 0.f, 0.f};
      if (live)
        for (int kz = 0; kz < ksplit; ++kz) v += *gl(reinterpret_cast<const nf32x4 *>(t_part + (int64_t)kz * Mr + pix * r + lg * 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) s_T[(t * 16 + l15) * 16 + lg * 4 + e] = (live && lg * 4 + e < r) ? v[e] : 0.f;
    }
  }
  __syncthreads();
  // T of the tile: to memory (the backward's dUp = (mask o G)^T T reads it) and into B-operand fragments (hi | hi, lo | lo)
  for (int i = tid; i < PT * 16 * (r >> 2); i += kNhThreads) {
    const int q4 = i % (r >> 2), px = (i / (r >> 2)) & 15, t = i / ((r >> 2) * 16);
    const int yy = y0 + t, xq = tc * 16 + px;
    if (yy < H && xq < W)
      *gl(reinterpret_cast<nf32x4 *>(t_out + (((int64_t)b * H + yy) * W + xq) * r + q4 * 4)) =
          *reinterpret_cast<const nf32x4 *>(s_T + (t * 16 + px) * 16 + q4 * 4);
  }
  for (int t = wave; t < PT; t += 4) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = s_T[(t * 16 + l15) * 16 + 8 * (lg & 1) + e];
    mu32x4 hi, lo;
    split_hi_lo<E>(v, hi, lo);
    s_tf[(t * 2 + 0) * 64 + lane] = hi;
    s_tf[(t * 2 + 1) * 64 + lane] = lo;
  }
  __syncthreads();
  // Y[tile pixels, :] += scale mask o (T up^T): rank_update16's transposed product, a wave = the 32-column groups wave, wave + 4, ..
  uint64_t off = 0;
  if constexpr (DROP) off = dropout_offset(offset, offset_dev);
  const int ncg = Co >> 5;
  bool pok[PT];
  int64_t prow[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int yy = y0 + t;
    pok[t] = yy < H && xx < W;
    prow[t] = ((int64_t)b * H + nh_clamp(yy, H - 1)) * W + nh_clamp(xx, W - 1);
  }
  auto ufrag = [&](int cg, int tile) -> mu32x4 {
    return *gl(reinterpret_cast<const mu32x4 *>(pu + (((int64_t)(cg < ncg ? cg : 0) * 2 + tile) * 64 + lane) * 8));
  };
  auto yload = [&](int cg, int t) -> mu32x4 {
    return *gl(reinterpret_cast<const mu32x4 *>(y + prow[t] * Co + (cg < ncg ? cg : 0) * 32 + 8 * lg));
  };
  mu32x4 ua0 = ufrag(wave, 0), ua1 = ufrag(wave, 1);
  mu32x4 yv[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) yv[t] = yload(wave, t);
  for (int cg = wave; cg < ncg; cg += 4) {
    const mu32x4 c0 = ua0, c1 = ua1;
    mu32x4 yc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) yc[t] = yv[t];
    ua0 = ufrag(cg + 4, 0); ua1 = ufrag(cg + 4, 1);   // the next group's operands are in flight while this one is multiplied
#pragma unroll
    for (int t = 0; t < PT; ++t) yv[t] = yload(cg + 4, t);
    const int col = cg * 32 + 8 * lg;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const mu32x4 th = s_tf[(t * 2 + 0) * 64 + lane], tl = s_tf[(t * 2 + 1) * 64 + lane];
      mf32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = FmMfma<E>::mma(fm_frag<E>(c0), fm_frag<E>(th), d0);
      d0 = FmMfma<E>::mma(fm_frag<E>(c0), fm_frag<E>(tl), d0);
      d1 = FmMfma<E>::mma(fm_frag<E>(c1), fm_frag<E>(th), d1);
      d1 = FmMfma<E>::mma(fm_frag<E>(c1), fm_frag<E>(tl), d1);
      float pr[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
      if constexpr (DROP) {
        float mk[8];
        dropout_mult8(seed, off, (uint64_t)((prow[t] * (int64_t)Co + col) >> 3), p, mk);
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] *= mk[e];
      }
      if (pok[t]) {
        union { mu32x4 u; Chunk8<E> c; } in, out;
        in.u = yc[t];
#pragma unroll
        for (int e = 0; e < 8; ++e) out.c.v[e] = E::from_f(fmaf(scale, pr[e], E::to_f(in.c.v[e])));
        *gl(reinterpret_cast<mu32x4 *>(y + prow[t] * Co + col)) = out.u;
      }
    }
  }
}

// ============================================================================ dX += conv_transpose3x3(Gt; down)
// Same pixel tiles.  The (tap, rank) contraction has 9r slots = KS k-steps of 32; the tile's Gt fragments (shifted by
// the tap of each slot group, zero outside the image) are built once and stay in registers; workgroup (tile, cz) walks
// the 64-channel blocks cz, cz + csplit, ..., wave w owning subtile w of each: KS weight fragments, KS MFMAs per tile
// row, one 8-byte read-modify-write of dX per lane (a lane owns 4 consecutive channels of one pixel; the four waves
// together cover the pixel's 128-byte line).
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
  // consecutive tiles (row groups of one image: they share halo rows) go to the same XCD, i.e. the same L2
  int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
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
#pragma unroll
  for (int t = 0; t < PT; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(graw[t][ks][h]));  // all PT*KS*2 loads in flight, one wait
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
        for (int e = 0; e < 4; ++e) {
          const unsigned qe = q[e];  // (a __builtin_bit_cast applied to q[e] itself reads element 0 for every e)
          v[h * 4 + e] = __builtin_bit_cast(float, qe);
        }
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
  // block cb + csplit's weight fragments and dX lines are requested before block cb is multiplied
  F pa[KS];
  nu32x2 prev[PT];
  auto fetch = [&](int cb) {
    const int ct = cb * 4 + wave;  // the four waves cover the pixel's 128-byte line of this block
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pa[ks] = *reinterpret_cast<const F *>(pd + (((int64_t)ct * KS + ks) * 64 + lane) * 8);
#pragma unroll
    for (int t = 0; t < PT; ++t) prev[t] = *reinterpret_cast<const nu32x2 *>(dx + poff[t] + ct * 16);
  };
  int cb = blockIdx.y;
  if (cb < NCB) fetch(cb);
  for (; cb < NCB; cb += gridDim.y) {
    const int ct = cb * 4 + wave;
    F pc[KS];
    nu32x2 pv[PT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pc[ks] = pa[ks];
#pragma unroll
    for (int t = 0; t < PT; ++t) pv[t] = prev[t];
    if (cb + (int)gridDim.y < NCB) fetch(cb + gridDim.y);
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      nf32x4 a = (nf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = NhMfma<E>::mma(pc[ks], bg[t][ks], a);
      union { nu32x2 v; S s[4]; } o;
      o.v = pv[t];
#pragma unroll
      for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(E::to_f(o.s[e]) + a[e]);
      if (pok[t]) *reinterpret_cast<nu32x2 *>(dx + poff[t] + ct * 16) = o.v;
    }
  }
}

// ============================================================================ dDown partials
// dDown[j, c, tap] = sum_p Gt[p, j] X[p + tap, c] contracts over PIXELS, which are strided in NHWC, so both operands
// are staged through LDS:
//   * a workgroup owns a 64-channel chunk `cc` and walks a run of consecutive image-row strips (PR rows each);
//     a strip's X rows y0-1 .. y0+PR (64 channels = one 128-byte line per pixel, coalesced 16-byte loads) are written
//     [pixel][136 B] with a zero pixel left and right of every row and zero rows outside the image, so that a tap is a
//     PURE SHIFT of the flat staged index (dy*(W+2) + dx) and the zero padding of the convolution needs no mask;
//   * Gt of the strip is staged transposed and in the activation dtype, gT[rank][padded pixel] (zero at the padding
//     positions), so the A operand (rows = ranks, k = 8 consecutive pixels) is one 16-byte LDS read per k-step;
//   * the B operand X^T (columns = channels, k = 8 consecutive pixels) is gathered from LDS with 2-byte reads at
//     compile-time offsets i*136 (the 4 lane groups are 8 pixels = 1088 B apart: disjoint bank groups); the 10 values
//     pixel-1 .. pixel+8 serve the three dx shifts of a row (two direct, one by v_alignbit);
//   * 8 waves: wave w owns channels (w & 3)*16 .. +15 of the chunk and the k-steps of parity w >> 2; its 9 accumulators
//     (one per tap) live in registers across ALL strips of the workgroup and the two parities meet once, at the end;
//     the next strip's global loads and the next k-step's LDS reads are issued before the current MFMAs.
// Output: part[sid][j][c*9 + tap] (the trainer's batched reduce folds the nsplit partials).
constexpr int kDdRowB = 136;      // bytes per staged pixel
constexpr int kDdMaxPix = 432;    // (PR + 2) * (W + 2) staged pixels at most
constexpr int kDdTail = 41;       // front margin pixel + k-step overrun behind the last row (zero)
constexpr int kDdGRow = 456;      // gT row length in elements (>= padded strip pixels rounded up to 32; multiple of 8)
constexpr int kDdThreads = 512;   // 8 waves: wave w owns channel tile (w & 3) and the k-steps of parity (w >> 2)
constexpr int kDdNld = 6;         // 16-byte chunks of X per thread and strip: (PR + 2) * W * 8 <= 6 * 512

template <class E, int RQ>
__global__ __launch_bounds__(kDdThreads) void conv3_ddown_nhwc_kernel(const typename E::storage *__restrict__ x,
                                                                      const float *__restrict__ gt,
                                                                      float *__restrict__ part, const NhGeom g,
                                                                      int PR, int nsplit, int rank_pad, int xcd_aware) {
  using S = typename E::storage;
  using F = typename NhMfma<E>::frag;
  __shared__ __attribute__((aligned(16))) unsigned char xs[(kDdMaxPix + kDdTail) * kDdRowB];
  __shared__ __attribute__((aligned(16))) unsigned short gT[16 * kDdGRow];
  static_assert(sizeof(xs) >= 4 * 9 * 64 * 16, "the cross-group reduction reuses the staging area");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int ctw = wave & 3, kg = wave >> 2;
  // 1-D grid, XCD-aware: the C/64 chunk workgroups of one strip run (they read the same Gt rows) share an L2
  const int nch = g.C >> 6;
  const int logical = xcd_aware ? (int)xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cc = logical % nch, sid = logical / nch;
  const int H = g.H, W = g.W, C = g.C, r = g.r, WP = W + 2;
  const int spi = (H + PR - 1) / PR, nstrips = g.B * spi;
  const int SPP = PR * WP, KSP = (SPP + 31) >> 5;
  const int nchunks = (PR + 2) * W * 8;

  // everything that is never written below stays zero: the pixel margins of every staged row, the tail, ranks >= r
  for (int i = tid; i < (int)(sizeof(xs) / 16); i += kDdThreads) reinterpret_cast<nu32x4 *>(xs)[i] = (nu32x4){0u, 0u, 0u, 0u};
  for (int i = tid; i < (int)(sizeof(gT) / 16); i += kDdThreads) reinterpret_cast<nu32x4 *>(gT)[i] = (nu32x4){0u, 0u, 0u, 0u};

  // strip-independent descriptors of this thread's X chunks and of its Gt pixel
  int xrel[kDdNld], xlds[kDdNld], xry[kDdNld];
#pragma unroll
  for (int u = 0; u < kDdNld; ++u) {
    const int q = tid + kDdThreads * u;
    const bool live = q < nchunks;
    const int f = live ? q >> 3 : 0, c16 = q & 7;
    const int ryp = f / W, xq = f - ryp * W;  // staged row 0 .. PR+1 (image row y0 - 1 + ryp), column
    xry[u] = live ? ryp - 1 : (1 << 20);      // a dead chunk is "outside the image" for every strip
    xrel[u] = ((ryp - 1) * W + xq) * C + cc * 64 + c16 * 8;
    xlds[u] = (ryp * WP + xq + 2) * kDdRowB + c16 * 16;  // + 1 column margin + 1 front pixel
  }
  const int gsp = tid;  // padded strip pixel whose Gt row this thread stages (KSP * 32 <= 456 < 512)
  const bool gin = gsp < KSP * 32;
  const int gry = gsp / WP, gxx = gsp - gry * WP;
  const bool gst = gsp < SPP && gxx >= 1 && gxx <= W;
  const int grel = (gry * W + gxx - 1) * r;

  nu32x4 xr[kDdNld];
  nf32x4 gr[RQ];
  auto prefetch = [&](int s) {
    const int b = s / spi, y0 = (s - b * spi) * PR;
    const int rows_valid = min(PR, H - y0);
    const S *xb = x + ((int64_t)(b * H + y0) * W) * C;
    const float *gb = gt + ((int64_t)(b * H + y0) * W) * r;
#pragma unroll
    for (int u = 0; u < kDdNld; ++u) {
      const int y = y0 + xry[u];
      const bool ok = y >= 0 && y < H;
      xr[u] = *reinterpret_cast<const nu32x4 *>(xb + (ok ? xrel[u] : cc * 64 + (tid & 7) * 8));
    }
    const bool ok = gst && gry < rows_valid;
#pragma unroll
    for (int qd = 0; qd < RQ; ++qd) gr[qd] = *reinterpret_cast<const nf32x4 *>(gb + (ok ? grel : 0) + 4 * qd);
  };

  nf32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (nf32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned short *xs16 = reinterpret_cast<const unsigned short *>(xs);
  const int col = ctw * 16 + l15;  // this lane's channel inside the chunk

  // this workgroup's strips are CONSECUTIVE ones (the halo rows two neighbours share are then re-read by the same
  // workgroup a moment later: an L2 hit instead of a second trip to HBM)
  const int per = nstrips / nsplit, extra = nstrips - per * nsplit;  // the first `extra` workgroups take one more
  const int s_begin = sid * per + min(sid, extra), s_end = s_begin + per + (sid < extra ? 1 : 0);
  if (s_begin < s_end) prefetch(s_begin);
  for (int s = s_begin; s < s_end; ++s) {
    const int b = s / spi, y0 = (s - b * spi) * PR;
    const int rows_valid = min(PR, H - y0);
    __syncthreads();  // the previous strip has been multiplied (first trip: the zero fill is complete)
#pragma unroll
    for (int u = 0; u < kDdNld; ++u) {
      const int y = y0 + xry[u];
      const unsigned m = (y >= 0 && y < H) ? 0xFFFFFFFFu : 0u;
      const nu32x4 v = xr[u] & (nu32x4){m, m, m, m};
      if (tid + kDdThreads * u < nchunks) {
        *reinterpret_cast<nu32x2 *>(xs + xlds[u]) = (nu32x2){v[0], v[1]};
        *reinterpret_cast<nu32x2 *>(xs + xlds[u] + 8) = (nu32x2){v[2], v[3]};
      }
    }
    if (gin) {
      const bool ok = gst && gry < rows_valid;
#pragma unroll
      for (int qd = 0; qd < RQ; ++qd)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          union { S s; unsigned short u; } cv;
          cv.s = E::from_f(ok ? gr[qd][e] : 0.f);
          gT[(qd * 4 + e) * kDdGRow + gsp] = cv.u;
        }
    }
    __syncthreads();
    if (s + 1 < s_end) prefetch(s + 1);  // in flight while this strip is multiplied

    // k-steps ks = kg, kg + 2, ...: the LDS reads of the next one are issued before the MFMAs of the current one
    unsigned v[3][10];
    nu32x4 gaw;
    auto fetch = [&](int ks) {
      const int sp0 = ks * 32 + lg * 8;
      gaw = *reinterpret_cast<const nu32x4 *>(gT + l15 * kDdGRow + sp0);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const unsigned short *src = xs16 + (sp0 + WP * d) * (kDdRowB / 2) + col;  // pixel (row + d - 1, column - 1)
#pragma unroll
        for (int i = 0; i < 10; ++i) v[d][i] = src[i * (kDdRowB / 2)];
      }
    };
    int ks = kg;
    if (ks < KSP) fetch(ks);
    for (; ks < KSP; ks += 2) {
      unsigned P[3][5];
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int k = 0; k < 5; ++k) P[d][k] = v[d][2 * k] | (v[d][2 * k + 1] << 16);
      const F ga = nh_frag_bits<E>(gaw);
      if (ks + 2 < KSP) fetch(ks + 2);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const nu32x4 fm = (nu32x4){P[d][0], P[d][1], P[d][2], P[d][3]};
        const nu32x4 fp = (nu32x4){P[d][1], P[d][2], P[d][3], P[d][4]};
        const nu32x4 f0 = (nu32x4){(P[d][0] >> 16) | (P[d][1] << 16), (P[d][1] >> 16) | (P[d][2] << 16),
                                   (P[d][2] >> 16) | (P[d][3] << 16), (P[d][3] >> 16) | (P[d][4] << 16)};
        acc[d * 3 + 0] = NhMfma<E>::mma(ga, nh_frag_bits<E>(fm), acc[d * 3 + 0]);
        acc[d * 3 + 1] = NhMfma<E>::mma(ga, nh_frag_bits<E>(f0), acc[d * 3 + 1]);
        acc[d * 3 + 2] = NhMfma<E>::mma(ga, nh_frag_bits<E>(fp), acc[d * 3 + 2]);
      }
    }
  }

  // the two k-step groups of a channel tile meet in LDS (the staging area is free now)
  __syncthreads();
  float *red = reinterpret_cast<float *>(xs);
  if (kg == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<nf32x4 *>(red + ((ctw * 9 + t) * 64 + lane) * 4) = acc[t];
  }
  __syncthreads();
  if (kg == 0) {
    // acc[tap][e] = dDown[rank lg*4 + e][channel cc*64 + ctw*16 + l15][tap]
    const int64_t row_len = (int64_t)C * 9;
    const int c = cc * 64 + ctw * 16 + l15;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const nf32x4 o = acc[t] + *reinterpret_cast<const nf32x4 *>(red + ((ctw * 9 + t) * 64 + lane) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = lg * 4 + e;
        if (j < r) part[((int64_t)sid * rank_pad + j) * row_len + (int64_t)c * 9 + t] = o[e];
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
static inline int64_t nh_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static int nh_pick_pt(int B, int H, int W) {
  const int64_t ntc = (W + 15) / 16;
  const int64_t t4 = (int64_t)B * ((H + 3) / 4) * ntc, t2 = (int64_t)B * ((H + 1) / 2) * ntc;
  if (t4 >= 200) return 4;
  if (t2 >= 100) return 2;
  return 1;
}
// image rows per dDown strip: as many as the staging budget holds ((PR+2)*W*8 chunks over 256 threads x kDdNld,
// (PR+2)*(W+2) staged pixels); 0 = the map is too wide
static int nh_pick_pr(int H, int W) {
  int pr = 0;
  for (int p = 1; p <= H; ++p)
    if ((p + 2) * W * 8 <= kDdNld * kDdThreads && (p + 2) * (W + 2) <= kDdMaxPix &&
        ((p * (W + 2) + 31) / 32) * 32 <= kDdGRow)
      pr = p;
  return pr;
}
static inline bool nh_native(int B, int C, int H, int W, int r) {
  return B >= 1 && H >= 1 && W >= 1 && C >= 64 && C % 64 == 0 && r >= 4 && r <= 16 && r % 4 == 0 &&
         (int64_t)B * H * W * C < ((int64_t)1 << 31) && nh_pick_pr(H, W) >= 1;
}
struct NhCuts { int pt, ksplit, pt_dx, csplit, pr, nsplit; };
static NhCuts nh_cuts(int B, int C, int H, int W, int r) {
  NhCuts q;
  const int64_t ntc = (W + 15) / 16, M = (int64_t)B * H * W;
  q.pt = nh_pick_pt(B, H, W);
  // forward: few pixel tiles (small maps) -> also split the channel k-steps (C/32; >= 4 per share, one per wave)
  const int64_t tiles = (int64_t)B * nh_cdiv(H, q.pt) * ntc;
  const int ks = tiles >= 192 ? 1 : (int)nh_cdiv(256, tiles);
  q.ksplit = (int)std::max<int64_t>(1, std::min<int64_t>(ks, (C / 32) / 4));
  // input gradient: PT * KS Gt fragments stay in registers -> at most 2 tile rows; channel blocks (C/64) over grid.y
  q.pt_dx = std::min(q.pt, 2);
  const int64_t tiles_dx = (int64_t)B * nh_cdiv(H, q.pt_dx) * ntc;
  const int cs = tiles_dx >= 192 ? 1 : (int)nh_cdiv(256, tiles_dx);
  q.csplit = (int)std::max<int64_t>(1, std::min<int64_t>(cs, C / 64));
  // factor gradient: strips of PR rows, dealt to nsplit workgroups per 64-channel chunk; the nsplit partials are
  // nsplit * 18 r / M of the X stream: at most ~30 % of it (or 4 MB, whichever is more), <= 256 workgroups in all
  q.pr = nh_pick_pr(H, W);
  const int64_t nstrips = (int64_t)B * nh_cdiv(H, std::max(q.pr, 1));
  int ns = 0;
  {
    // at most one workgroup per CU (a 257th would share a CU with another one and double the critical path), the cap
    // on the partial bytes, and then EQUAL shares: strips per workgroup = ceil(nstrips / that bound)
    const int64_t cap = std::max<int64_t>((int64_t)(0.30 * (double)M / (18.0 * r)),
                                          std::max<int64_t>(1, ((int64_t)4 << 20) / ((int64_t)nh_rank_pad(r) * C * 36)));
    const int64_t bound = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(256 / (C / 64), cap), nstrips));
    ns = (int)nh_cdiv(nstrips, nh_cdiv(nstrips, bound));
  }
  q.nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ns, nstrips));
  return q;
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
  const NhCuts q = nh_cuts(B, C_in, H, W, r);
  const int64_t M = (int64_t)B * H * W;
  out->native = 1;
  out->pt = q.pt;
  out->ksplit = q.ksplit;
  out->csplit = q.csplit;
  out->ks = (9 * r + 31) / 32;
  out->pr = q.pr;
  out->nsplit = q.nsplit;
  out->rank_pad = nh_rank_pad(r);
  out->pf_elems = (int64_t)9 * C_in * 16;
  out->pd_elems = (int64_t)C_in * out->ks * 32;
  out->t_part_floats = q.ksplit > 1 ? q.ksplit * M * r : 0;
  out->fwd_tiles = (int32_t)((int64_t)B * nh_cdiv(H, q.pt) * ((W + 15) / 16));
  out->down_part_floats = (int64_t)q.nsplit * out->rank_pad * C_in * 9;
  return LORA_AMD_OK;
}

#define NH_COMMON(name)                                                                                          \
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16, LORA_AMD_EINVAL,                       \
                 name ": activations must be bf16 or f16");                                                       \
  LORA_AMD_CHECK(r >= 1 && r <= 16, LORA_AMD_ERANK, name ": rank %d outside [1,16]", r);                          \
  LORA_AMD_CHECK(nh_native(B, C_in, H, W, r), LORA_AMD_EINVAL,                                                    \
                 name ": needs C_in %% 64 == 0, rank in {4, 8, 12, 16}, W <= 128, B*H*W*C_in < 2^31 "            \
                      "(lora_amd_conv3_nhwc_plan)")

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

static int nh_launch_sum(const float *part, int nparts, int64_t n, float *out, void *stream) {
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(nh_sum_parts_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     nparts, n4, out, n4);
  return check_launch("lora_amd_sum_parts");
}

extern "C" int lora_amd_conv3_nhwc_down_fwd(const void *x, const void *pf, float *t_part, float *t_out, int32_t B,
                                            int32_t C_in, int32_t H, int32_t W, int32_t r, int32_t act_dtype,
                                            void *stream) {
  NH_COMMON("conv3_nhwc_down_fwd");
  LORA_AMD_CHECK(x && pf && t_out, LORA_AMD_EINVAL, "conv3_nhwc_down_fwd: null pointer");
  LORA_AMD_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)pf % 16) == 0 && ((uintptr_t)t_out % 16) == 0 &&
                     ((uintptr_t)t_part % 16) == 0,
                 LORA_AMD_EINVAL, "conv3_nhwc_down_fwd: 16-byte aligned buffers");
  const NhCuts q = nh_cuts(B, C_in, H, W, r);
  LORA_AMD_CHECK(q.ksplit == 1 || t_part != nullptr, LORA_AMD_EWORKSPACE,
                 "conv3_nhwc_down_fwd: this geometry splits the channels over %d workgroups and needs t_part "
                 "(plan.t_part_floats)", q.ksplit);
  const int pt = q.pt;
  const NhGeom g = nh_geom(B, C_in, H, W, r, pt);
  const dim3 grid((unsigned)((int64_t)B * g.nrg * g.ntc), (unsigned)q.ksplit);
  float *dst = q.ksplit > 1 ? t_part : t_out;
#define NH_LAUNCH_T(E)                                                                                         \
  using S = typename E::storage;                                                                               \
  if (pt == 4) hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 4>), grid, dim3(kNhThreads), 0,                  \
                                  (hipStream_t)stream, (const S *)x, (const S *)pf, dst, g);                   \
  else if (pt == 2) hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 2>), grid, dim3(kNhThreads), 0,             \
                                       (hipStream_t)stream, (const S *)x, (const S *)pf, dst, g);              \
  else hipLaunchKernelGGL((conv3_down_nhwc_kernel<E, 1>), grid, dim3(kNhThreads), 0, (hipStream_t)stream,     \
                          (const S *)x, (const S *)pf, dst, g);
  NH_BY_DTYPE(NH_LAUNCH_T)
#undef NH_LAUNCH_T
  const int rc = check_launch("lora_amd_conv3_nhwc_down_fwd");
  if (rc != LORA_AMD_OK || q.ksplit == 1) return rc;
  return nh_launch_sum(t_part, q.ksplit, (int64_t)B * H * W * r, t_out, stream);
}

// ---------------------------------------------------------------------------- round 6: batched pack, fused forward
static inline int64_t nh_pack_pieces(int r, int C_in, int C_out) {
  const int KS = (9 * r + 31) / 32;
  return (int64_t)9 * (C_in / 32) * 64 + (int64_t)(C_in / 16) * KS * 64 + (int64_t)(C_out / 32) * 2 * 64;
}

extern "C" int lora_amd_conv3_nhwc_pack_plan(lora_amd_conv3_pack_site *sites, int32_t n, int64_t *total) {
  LORA_AMD_CHECK(sites && n >= 1 && total, LORA_AMD_EINVAL, "conv3_nhwc_pack_plan: bad argument");
  static_assert(sizeof(lora_amd_conv3_pack_site) == sizeof(NhPackSite), "site table layout");
  int64_t begin = 0;
  for (int i = 0; i < n; ++i) {
    lora_amd_conv3_pack_site &q = sites[i];
    LORA_AMD_CHECK(q.down && q.up && q.pf && q.pd && q.pu, LORA_AMD_EINVAL, "conv3_nhwc_pack_plan: site %d: null pointer", i);
    LORA_AMD_CHECK(q.r >= 4 && q.r <= 16 && q.r % 4 == 0, LORA_AMD_ERANK, "conv3_nhwc_pack_plan: site %d: rank %d not in {4, 8, 12, 16}", i, q.r);
    LORA_AMD_CHECK(q.C_in >= 64 && q.C_in % 64 == 0 && q.C_out >= 32 && q.C_out % 32 == 0, LORA_AMD_EINVAL,
                   "conv3_nhwc_pack_plan: site %d: C_in %% 64 == 0 and C_out %% 32 == 0 required", i);
    LORA_AMD_CHECK((((uintptr_t)q.pf | (uintptr_t)q.pd | (uintptr_t)q.pu) & 15u) == 0, LORA_AMD_EINVAL,
                   "conv3_nhwc_pack_plan: site %d: packs must be 16-byte aligned", i);
    q.KS = (9 * q.r + 31) / 32;
    q.begin = begin;
    begin += nh_pack_pieces(q.r, q.C_in, q.C_out);
  }
  *total = begin;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_conv3_nhwc_pack_batched(const lora_amd_conv3_pack_site *sites_dev, int32_t n, int64_t total,
                                                int32_t act_dtype, void *stream) {
  LORA_AMD_CHECK(sites_dev && n >= 1 && total >= 1, LORA_AMD_EINVAL, "conv3_nhwc_pack_batched: bad argument");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16, LORA_AMD_EINVAL,
                 "conv3_nhwc_pack_batched: activations must be bf16 or f16");
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  const NhPackSite *sd = reinterpret_cast<const NhPackSite *>(sites_dev);
  if (act_dtype == LORA_AMD_BF16)
    hipLaunchKernelGGL(conv3_site_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sd, n, total);
  else
    hipLaunchKernelGGL(conv3_site_pack_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, sd, n, total);
  return check_launch("lora_amd_conv3_nhwc_pack_batched");
}

extern "C" int lora_amd_conv3_nhwc_fwd_fused(const void *x, const void *pf, const void *pu, void *y, float *t_out,
                                             float *t_part, uint32_t *counters, int32_t B, int32_t C_in, int32_t C_out,
                                             int32_t H, int32_t W, int32_t r, int32_t act_dtype, float scale,
                                             float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev,
                                             void *stream) {
  NH_COMMON("conv3_nhwc_fwd_fused");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "conv3_nhwc_fwd_fused: bf16 activations (the (hi, lo) split of `up` ~ 1e-4 needs bf16's exponent range; "
                 "f16 runs conv3_nhwc_down_fwd + rank_update)");
  LORA_AMD_CHECK(x && pf && pu && y && t_out, LORA_AMD_EINVAL, "conv3_nhwc_fwd_fused: null pointer");
  LORA_AMD_CHECK(C_out >= 32 && C_out % 32 == 0, LORA_AMD_EINVAL, "conv3_nhwc_fwd_fused: C_out %% 32 == 0 required");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "conv3_nhwc_fwd_fused: dropout p=%f", dropout_p);
  LORA_AMD_CHECK((((uintptr_t)x | (uintptr_t)pf | (uintptr_t)pu | (uintptr_t)y | (uintptr_t)t_out | (uintptr_t)t_part) & 15u) == 0,
                 LORA_AMD_EINVAL, "conv3_nhwc_fwd_fused: 16-byte aligned buffers");
  const NhCuts q = nh_cuts(B, C_in, H, W, r);
  LORA_AMD_CHECK(q.ksplit == 1 || (t_part != nullptr && counters != nullptr), LORA_AMD_EWORKSPACE,
                 "conv3_nhwc_fwd_fused: this geometry splits the channels over %d workgroups per tile and needs t_part "
                 "(plan.t_part_floats) and one zeroed counter per tile (plan.fwd_tiles)", q.ksplit);
  const NhGeom g = nh_geom(B, C_in, H, W, r, q.pt);
  const dim3 grid((unsigned)((int64_t)B * g.nrg * g.ntc), (unsigned)q.ksplit);
  using S = bf16_t::storage;
  const bool drop = dropout_p > 0.f;
#define NH_FUSED(PT_, D_)                                                                                              \
  hipLaunchKernelGGL((conv3_fwd_fused_kernel<bf16_t, PT_, D_>), grid, dim3(kNhThreads), 0, (hipStream_t)stream,        \
                     (const S *)x, (const S *)pf, (const S *)pu, (S *)y, t_out, t_part, counters, g, C_out, scale,      \
                     dropout_p, seed, offset, offset_dev)
  if (q.pt == 4) { if (drop) NH_FUSED(4, true); else NH_FUSED(4, false); }
  else if (q.pt == 2) { if (drop) NH_FUSED(2, true); else NH_FUSED(2, false); }
  else { if (drop) NH_FUSED(1, true); else NH_FUSED(1, false); }
#undef NH_FUSED
  return check_launch("lora_amd_conv3_nhwc_fwd_fused");
}

extern "C" int lora_amd_conv3_nhwc_bwd_dx(void *dx, const float *gt, const void *pd, int32_t B, int32_t C_in,
                                          int32_t H, int32_t W, int32_t r, int32_t act_dtype, void *stream) {
  NH_COMMON("conv3_nhwc_bwd_dx");
  LORA_AMD_CHECK(dx && gt && pd, LORA_AMD_EINVAL, "conv3_nhwc_bwd_dx: null pointer");
  LORA_AMD_CHECK(((uintptr_t)dx % 16) == 0 && ((uintptr_t)pd % 16) == 0 && ((uintptr_t)gt % 16) == 0, LORA_AMD_EINVAL,
                 "conv3_nhwc_bwd_dx: 16-byte aligned buffers");
  const NhCuts q = nh_cuts(B, C_in, H, W, r);
  const int pt = q.pt_dx;
  const NhGeom g = nh_geom(B, C_in, H, W, r, pt);
  const dim3 grid((unsigned)((int64_t)B * g.nrg * g.ntc), (unsigned)q.csplit);
  const int KS = (9 * r + 31) / 32;
#define NH_LAUNCH_DX2(E, PT_, KS_)                                                                      \
  hipLaunchKernelGGL((conv3_dx_nhwc_kernel<E, PT_, KS_>), grid, dim3(kNhThreads), 0, (hipStream_t)stream, \
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
  LORA_AMD_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)gt % 16) == 0, LORA_AMD_EINVAL,
                 "conv3_nhwc_bwd_down: 16-byte aligned buffers");
  const NhCuts q = nh_cuts(B, C_in, H, W, r);
  const NhGeom g = nh_geom(B, C_in, H, W, r, 1);
  const int rank_pad = nh_rank_pad(r);
  const dim3 grid((unsigned)((C_in / 64) * q.nsplit));
#define NH_LAUNCH_DD2(E, RQ_)                                                                                   \
  hipLaunchKernelGGL((conv3_ddown_nhwc_kernel<E, RQ_>), grid, dim3(kDdThreads), 0, (hipStream_t)stream,        \
                     (const typename E::storage *)x, gt, down_part, g, q.pr, q.nsplit, rank_pad,          \
                     1);
#define NH_LAUNCH_DD(E)                    \
  switch (r / 4) {                         \
    case 1: NH_LAUNCH_DD2(E, 1) break;     \
    case 2: NH_LAUNCH_DD2(E, 2) break;     \
    case 3: NH_LAUNCH_DD2(E, 3) break;     \
    default: NH_LAUNCH_DD2(E, 4) break;    \
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
