// K4 — low-rank branch of LoraInjectedConv2d and its gradients, NCHW, gfx950.
//
// replaces: lora_diffusion/lora.py:130-135 (conv2d k x k in->r, conv2d 1x1 r->out, dropout, mul, add = 5 ATen
//           launches + 3 extra [B,C_out,H,W] round trips per site) and the autograd of that sequence.
//
// Layout/mapping.  NCHW keeps the pixel dimension contiguous, so a lane owns one 16-byte chunk of 8 consecutive
// pixels of an image plane and walks the CHANNEL dimension; lane l of a wave owns chunk (group*cpw + l) of the flat
// chunk space [B][H*W/8], `cpw` <= 64 chosen so that a wave covers whole image rows.  Consequences:
//   * every load/store of X, Y, G, dX is a coalesced 16 B/lane access of one channel plane;
//   * contractions over channels (T = down * X, Gt = up^T * G) accumulate in registers with no cross-lane traffic;
//     the four waves of a workgroup take interleaved channels and are summed through LDS once per rank group,
//     further channel splits across workgroups go through a small f32 partial buffer;
//   * the 3x3 taps need a +-1 pixel halo inside the row (two wave shuffles per loaded row; the wave owns whole
//     rows, so the halo never crosses a wave) and the rows above/below (two more aligned 16-byte loads, L2 hits);
//   * contractions over pixels (dUp, dDown) are wave reductions per channel (butterfly over the 4 ranks of a
//     rank group: 7 shuffles for 4 sums) written as per-group partials that the trainer's batched reduce folds
//     into the flat gradient buffer.
// The weights of the current channel (4 ranks x ks*ks taps) are wave-uniform and live in SGPRs.
// Ranks are processed in groups of 4 (outer loop; the re-read of the activations for r > 4 is served by L2).
//
// Roofline: HBM for 1x1 and for 3x3 at r <= 4 (3x3: 2*9*r flop per 2-byte element = 9r flop/B vs the f32 VALU
// balance of ~25 flop/B at 6.3 TB/s); 3x3 at r = 16 is f32-VALU bound (144 flop/B).  Algorithmic bytes per site:
// forward  B*C_in*HW*e (X once) + 2*B*C_out*HW*e (Y read+write);  backward  B*C_out*HW*e (G once) +
// B*C_in*HW*e (X once) + 2*B*C_in*HW*e (dX read+write); the [B,r,HW] f32 tensors are r/C of that.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace lora_amd {

constexpr int kCT = 256;  // 4 waves

struct ChunkPos {
  bool act;
  int b, p0, y, x0;
};

__device__ inline ChunkPos chunk_pos(int lane, int cpw, int64_t NP, int npix8, int W) {
  ChunkPos c;
  const int64_t q = (int64_t)blockIdx.x * cpw + lane;
  c.act = lane < cpw && q < NP;
  const int64_t qq = c.act ? q : 0;
  c.b = (int)(qq / npix8);
  c.p0 = (int)(qq - (int64_t)c.b * npix8) * 8;
  c.y = c.p0 / W;
  c.x0 = c.p0 - c.y * W;
  return c;
}

__device__ inline void ld8f(const float *p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void st8f(float *p, const float (&v)[8]) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// Row chunk of a plane with its +-1 halo: w[0] = pixel p0-1, w[1..8] = the chunk, w[9] = pixel p0+8 (zeros
// outside the row / when the row is outside the image).  All lanes of the wave must call this together.
template <class E>
__device__ inline void load_row_window(const typename E::storage *plane_chunk, bool ok, int x0, int W,
                                       float (&w)[10]) {
  float row[8];
  load8_sel<E>(plane_chunk, ok, row);
  float left = lane_from_prev(row[7]), right = lane_from_next(row[0]);
  if (x0 == 0) left = 0.f;
  if (x0 + 8 == W) right = 0.f;
  w[0] = left;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i + 1] = row[i];
  w[9] = right;
}
// The same from a row chunk that is already in registers (zeros when the row is outside the image).
__device__ inline void row_window(const float (&row)[8], int x0, int W, float (&w)[10]) {
  float left = lane_from_prev(row[7]), right = lane_from_next(row[0]);
  if (x0 == 0) left = 0.f;
  if (x0 + 8 == W) right = 0.f;
  w[0] = left;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i + 1] = row[i];
  w[9] = right;
}
__device__ inline void load_row_window_f32(const float *plane_chunk, bool ok, int x0, int W, float (&w)[10]) {
  float row[8];
  ld8f_sel(plane_chunk, ok, row);
  float left = lane_from_prev(row[7]), right = lane_from_next(row[0]);
  if (x0 == 0) left = 0.f;
  if (x0 + 8 == W) right = 0.f;
  w[0] = left;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i + 1] = row[i];
  w[9] = right;
}

// Sum acc[4][8] over the 4 waves of the workgroup; wave w ends up with rank w of the group in out[8].
__device__ inline void block_rank_reduce(float *s_red, const float (&acc)[4][8], int wave, int lane, float (&out)[8]) {
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) st8f(&s_red[((wave * 4 + j) * 64 + lane) * 8], acc[j]);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = 0.f;
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) {
    float v[8];
    ld8f(&s_red[((w4 * 4 + wave) * 64 + lane) * 8], v);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] += v[i];
  }
}

// Wave-uniform weights of one channel: w[j][t] = f[(rank0+j) * stride_j + t] for j < 4, zero beyond rank r.
// `f` is f32 in the constant/global address space and every index is uniform -> scalar (SMEM) loads.
template <int KK>
__device__ inline void load_weights(const float *__restrict__ f, int64_t stride_j, int rank0, int r, float (&w)[4][KK]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int jj = rank0 + j;
    const float *wp = f + (int64_t)(jj < r ? jj : r - 1) * stride_j;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float v = wp[t];
      w[j][t] = jj < r ? v : 0.f;
    }
  }
}

// ============================================================================ T partials = conv_kxk(X; down)
// grid (ngroups, split_in, rank groups).  t_part[s][B][r][HW].  Two channels per trip: all row loads of both are
// issued before the first FMA (6 x 16 B in flight per lane for 3x3).
template <class E, int KS>
__global__ __launch_bounds__(kCT) void conv_down_fwd_kernel(const typename E::storage *__restrict__ x,
                                                            const float *__restrict__ down,
                                                            float *__restrict__ t_part, int B, int C, int H, int W,
                                                            int r, int cpw, int64_t NP, int cps) {
  constexpr int KK = KS * KS;
  constexpr int U = 2;
  __shared__ __attribute__((aligned(16))) float s_red[4 * 4 * 64 * 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HW = H * W, npix8 = HW >> 3;
  const ChunkPos cp = chunk_pos(lane, cpw, NP, npix8, W);
  const int s = blockIdx.y, rg = blockIdx.z;
  const int c_begin = s * cps, c_end = min(C, c_begin + cps);
  const typename E::storage *xb = x + (int64_t)cp.b * C * HW + cp.p0;
  bool rok[KS];
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const int yy = cp.y + dy - (KS >> 1);
    rok[dy] = cp.act && yy >= 0 && yy < H;
  }

  float acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * U) {
    float rows[U][KS][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        const bool ok = c < c_end && rok[dy];  // clamped (always valid) address, zeroed afterwards: no branch
        load8_sel<E>(xb + (int64_t)(c < c_end ? c : c_end - 1) * HW + (rok[dy] ? (dy - (KS >> 1)) * W : 0), ok,
                     rows[u][dy]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      if (c >= c_end) continue;  // wave-uniform
      float w[4][KK];
      load_weights<KK>(down + (int64_t)c * KK, (int64_t)C * KK, rg * 4, r, w);
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        if (KS == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(w[j][0], rows[u][dy][i], acc[j][i]);
        } else {
          float xw[10];
          row_window(rows[u][dy], cp.x0, W, xw);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(w[j][(dy * 3 + dx) % KK], xw[i + dx], acc[j][i]);
        }
      }
    }
  }
  float o[8];
  block_rank_reduce(s_red, acc, wave, lane, o);
  const int jj = rg * 4 + wave;
  if (cp.act && jj < r) st8f(t_part + (((int64_t)s * B + cp.b) * r + jj) * HW + cp.p0, o);
}

// ============================================================================ finalize: out = Sel (sum_s part[s])
// sum kernel: 64 outputs (float4 each) x 4 split-slices per workgroup; every thread keeps 4 independent loads in
// flight and the slices meet in LDS.  Deterministic (fixed summation order), no atomics.
__global__ __launch_bounds__(kCT) void partial_sum_kernel(const float *__restrict__ part, int S, int64_t part_stride,
                                                          float *__restrict__ out, int64_t n4) {
  __shared__ float4 s_acc[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (i < n4) {
    const float4 *pp = reinterpret_cast<const float4 *>(part) + i;
    const int64_t st4 = part_stride >> 2;
    int s = sl;
    for (; s + 12 < S; s += 16) {
      const float4 v0 = pp[(int64_t)s * st4], v1 = pp[(int64_t)(s + 4) * st4], v2 = pp[(int64_t)(s + 8) * st4],
                   v3 = pp[(int64_t)(s + 12) * st4];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; s < S; s += 4) {
      const float4 v0 = pp[(int64_t)s * st4];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
  }
  s_acc[sl][o] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                             (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (sl == 0 && i < n4) {
    const float4 b0 = s_acc[0][o], b1 = s_acc[1][o], b2 = s_acc[2][o], b3 = s_acc[3][o];
    reinterpret_cast<float4 *>(out)[i] = make_float4((b0.x + b1.x) + (b2.x + b3.x), (b0.y + b1.y) + (b2.y + b3.y),
                                                     (b0.z + b1.z) + (b2.z + b3.z), (b0.w + b1.w) + (b2.w + b3.w));
  }
}

// selector (rare: set_lora_diag): io[b][a][p] <- sum_j Sel[a][j] io[b][j][p]  (transposed: Sel[j][a]), in place.
__global__ __launch_bounds__(kCT) void selector_mix_kernel(float *__restrict__ io, const float *__restrict__ sel,
                                                           int sel_transposed, int r, int HW, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * kCT + threadIdx.x;
  if (i >= n4) return;
  const int hw4 = HW >> 2;
  const int64_t b = i / hw4;
  const int p = (int)(i - b * hw4) * 4;
  float4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j)
    v[j] = j < r ? *reinterpret_cast<const float4 *>(io + (b * r + j) * HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    if (a >= r) continue;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int bb = 0; bb < 16; ++bb) {
      if (bb >= r) continue;
      const float sv = sel_transposed ? sel[bb * r + a] : sel[a * r + bb];
      o.x = fmaf(sv, v[bb].x, o.x); o.y = fmaf(sv, v[bb].y, o.y);
      o.z = fmaf(sv, v[bb].z, o.z); o.w = fmaf(sv, v[bb].w, o.w);
    }
    *reinterpret_cast<float4 *>(io + (b * r + a) * HW + p) = o;
  }
}

// ============================================================================ Y += scale * mask * up T
// grid (ngroups, split_out).  t: [B][r][HW] f32.
template <class E, int RT, bool DROP>
__global__ __launch_bounds__(kCT) void conv_up_fwd_kernel(typename E::storage *__restrict__ y,
                                                          const float *__restrict__ t,
                                                          const float *__restrict__ up, int B, int Co, int HW,
                                                          int r, int cpw, int64_t NP, int cps, float scale, float p,
                                                          uint64_t seed, uint64_t offset, const uint64_t *offset_dev) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npix8 = HW >> 3;
  const ChunkPos cp = chunk_pos(lane, cpw, NP, npix8, HW);
  if (!cp.act) return;  // no wave-collective operations below
  float tt[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    if (j < r) {
      ld8f(t + ((int64_t)cp.b * r + j) * HW + cp.p0, tt[j]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) tt[j][i] = 0.f;
    }
  }
  const int s = blockIdx.y;
  const int c_begin = s * cps, c_end = min(Co, c_begin + cps);
  typename E::storage *yb = y + (int64_t)cp.b * Co * HW + cp.p0;
  constexpr int U = 4;
  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * U) {
    float yv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      load8_sel<E>(yb + (int64_t)(c < c_end ? c : c_end - 1) * HW, c < c_end, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      if (c >= c_end) continue;
      float pr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float uv = up[(int64_t)c * r + (j < r ? j : r - 1)];
        const float uj = j < r ? uv : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) pr[i] = fmaf(uj, tt[j][i], pr[i]);
      }
      if (DROP) {
        float mk[8];
        const int64_t e = ((int64_t)cp.b * Co + c) * HW + cp.p0;
        dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
        for (int i = 0; i < 8; ++i) pr[i] *= mk[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) yv[u][i] = fmaf(scale, pr[i], yv[u][i]);
      store8<E>(yb + (int64_t)c * HW, yv[u]);
    }
  }
}

// ============================================================================ pass over G
// grid (ngroups, split_out, rank groups).  gt_part[s][B][r][HW] = scale * up^T (mask*G) over the split's channels;
// up_part[group][rank_pad][Co] = scale * sum over the group's pixels of (mask*G)[co] * T[j].
template <class E, bool DROP>
__global__ __launch_bounds__(kCT) void conv_bwd_g_kernel(const typename E::storage *__restrict__ g,
                                                         const float *__restrict__ t, const float *__restrict__ up,
                                                         float *__restrict__ gt_part, float *__restrict__ up_part,
                                                         int B, int Co, int HW, int r, int rank_pad, int cpw,
                                                         int64_t NP, int cps, float scale, float p, uint64_t seed,
                                                         uint64_t offset, const uint64_t *offset_dev) {
  constexpr int U = 4;
  __shared__ __attribute__((aligned(16))) float s_red[4 * 4 * 64 * 8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npix8 = HW >> 3;
  const ChunkPos cp = chunk_pos(lane, cpw, NP, npix8, HW);
  const int s = blockIdx.y, rg = blockIdx.z;
  const int c_begin = s * cps, c_end = min(Co, c_begin + cps);
  const typename E::storage *gb = g + (int64_t)cp.b * Co * HW + cp.p0;
  float *upp = up_part + (int64_t)blockIdx.x * rank_pad * Co;

  float tt[4][8], acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ld8f_sel(t + ((int64_t)cp.b * r + (rg * 4 + j < r ? rg * 4 + j : 0)) * HW + cp.p0, cp.act && rg * 4 + j < r, tt[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
  }
  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * U) {
    float gv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      load8_sel<E>(gb + (int64_t)(c < c_end ? c : c_end - 1) * HW, c < c_end && cp.act, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      if (c >= c_end) continue;  // wave-uniform
      if (DROP && cp.act) {
        float mk[8];
        const int64_t e = ((int64_t)cp.b * Co + c) * HW + cp.p0;
        dropout_mult8(seed, dropout_offset(offset, offset_dev), (uint64_t)(e >> 3), p, mk);
#pragma unroll
        for (int i = 0; i < 8; ++i) gv[u][i] *= mk[i];
      }
      float w[4][1];
      load_weights<1>(up + (int64_t)c * r, 1, rg * 4, r, w);
      float d[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float dj = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[j][i] = fmaf(w[j][0], gv[u][i], acc[j][i]);
          dj = fmaf(gv[u][i], tt[j][i], dj);
        }
        d[j] = dj;
      }
      const float tot = wave_sum4(d[0], d[1], d[2], d[3], lane);
      const int jj = rg * 4 + idx4(lane);
      if (lane < 4 && jj < r) upp[(int64_t)jj * Co + c] = scale * tot;
    }
  }
  float o[8];
  block_rank_reduce(s_red, acc, wave, lane, o);
  const int jj = rg * 4 + wave;
  if (cp.act && jj < r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= scale;
    st8f(gt_part + (((int64_t)s * B + cp.b) * r + jj) * HW + cp.p0, o);
  }
}

// ============================================================================ passes over X and over dX
// gt: [B][r][HW] f32 (already S^T-projected).  Two light kernels instead of one register-bound one: the dDown pass
// reads X, the dX pass reads/writes dX; they share only the tiny gt / down operands.

// One gt row window: gw[j][k] = gt[rank0+j][p0 + shift - 1 + k], k = 0..9 (zeros outside the row / image / rank).
template <int KS>
__device__ inline void load_gt_row(const float *__restrict__ gt, const ChunkPos &cp, int r, int rank0, bool row_ok,
                                   int shift, int W, int HW, float (&gw)[4][10]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = row_ok && rank0 + j < r;
    const float *gp = gt + ((int64_t)cp.b * r + (rank0 + j < r ? rank0 + j : 0)) * HW + cp.p0 + (row_ok ? shift : 0);
    if (KS == 1) {
      float row[8];
      ld8f_sel(gp, ok, row);
      gw[j][0] = gw[j][9] = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) gw[j][i + 1] = row[i];
    } else {
      load_row_window_f32(gp, ok, cp.x0, W, gw[j]);
    }
  }
}

// down_part[group][rank_pad][C*KK] = sum over the group's pixels of gt[j][p] * X[c][p + tap].
// grid (ngroups_in, split, rank groups).  The wave's gt chunk (4 ranks x 8 pixels) stays in registers; two channels
// per trip with all X row loads issued first; one butterfly wave reduction per tap.
template <class E, int KS>
__global__ __launch_bounds__(kCT) void conv_bwd_down_kernel(const typename E::storage *__restrict__ x,
                                                            const float *__restrict__ gt,
                                                            float *__restrict__ down_part, int B, int C, int H, int W,
                                                            int r, int rank_pad, int cpw, int64_t NP, int cps) {
  constexpr int KK = KS * KS;
  constexpr int U = 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HW = H * W, npix8 = HW >> 3;
  const ChunkPos cp = chunk_pos(lane, cpw, NP, npix8, W);
  const int s = blockIdx.y, rg = blockIdx.z;
  const int c_begin = s * cps, c_end = min(C, c_begin + cps);
  const typename E::storage *xb = x + (int64_t)cp.b * C * HW + cp.p0;
  float *dpp = down_part + (int64_t)blockIdx.x * rank_pad * C * KK;
  bool rok[KS];
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const int yy = cp.y + dy - (KS >> 1);
    rok[dy] = cp.act && yy >= 0 && yy < H;
  }
  float g0[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ld8f_sel(gt + ((int64_t)cp.b * r + (rg * 4 + j < r ? rg * 4 + j : 0)) * HW + cp.p0, cp.act && rg * 4 + j < r, g0[j]);
  }
  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * U) {
    float rows[U][KS][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        const bool ok = c < c_end && rok[dy];  // clamped (always valid) address, zeroed afterwards: no branch
        load8_sel<E>(xb + (int64_t)(c < c_end ? c : c_end - 1) * HW + (rok[dy] ? (dy - (KS >> 1)) * W : 0), ok,
                     rows[u][dy]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + 4 * u;
      if (c >= c_end) continue;  // wave-uniform
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) {
        float xw[10];
        if (KS == 1) {
          xw[0] = xw[9] = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) xw[i + 1] = rows[u][dy][i];
        } else {
          row_window(rows[u][dy], cp.x0, W, xw);
        }
#pragma unroll
        for (int dxi = 0; dxi < KS; ++dxi) {
          float d[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float dj = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) dj = fmaf(xw[i + (KS == 1 ? 1 : dxi)], g0[j][i], dj);
            d[j] = dj;
          }
          const float tot = wave_sum4(d[0], d[1], d[2], d[3], lane);
          const int jj = rg * 4 + idx4(lane);
          if (lane < 4 && jj < r) dpp[((int64_t)jj * C + c) * KK + dy * KS + dxi] = tot;
        }
      }
    }
  }
}

// dX[c][p] += sum_{j,tap} down[j][c][tap] * gt[j][p - tap].  grid (ngroups_in, split).
// A wave takes its channels four at a time: their dX increments stay in f32 registers across ALL rank groups and
// all three tap rows and are added to dX once (one rounding to the activation dtype, whatever the rank).  The gt
// row windows (4 ranks x 10 pixels) are fetched per (rank group, tap row) - L1/L2 hits, [B,r,HW] is tiny.
template <class E, int KS>
__global__ __launch_bounds__(kCT) void conv_bwd_dx_kernel(typename E::storage *__restrict__ dx,
                                                          const float *__restrict__ gt,
                                                          const float *__restrict__ down, int B, int C, int H, int W,
                                                          int r, int cpw, int64_t NP, int cps) {
  constexpr int KK = KS * KS;
  constexpr int CB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int HW = H * W, npix8 = HW >> 3;
  const ChunkPos cp = chunk_pos(lane, cpw, NP, npix8, W);
  const int s = blockIdx.y;
  const int c_begin = s * cps, c_end = min(C, c_begin + cps);
  typename E::storage *dxb = dx + (int64_t)cp.b * C * HW + cp.p0;

  for (int c0 = c_begin + wave; c0 < c_end; c0 += 4 * CB) {
    float dv[CB][8];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {  // issued now (branch-free, clamped), consumed after the rank loop
      const int c = c0 + 4 * cb;
      load8_sel<E>(dxb + (int64_t)(c < c_end ? c : c_end - 1) * HW, c < c_end, dv[cb]);
    }
    float dacc[CB][8];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 8; ++i) dacc[cb][i] = 0.f;
    for (int rg = 0; rg * 4 < r; ++rg) {
#pragma unroll 1  // keep ONE row window live (unrolled, the compiler hoists all three and drops to 1 wave/SIMD)
      for (int dy = 0; dy < KS; ++dy) {
        // source row of tap row dy: y - (dy - 1)
        const int yy = KS == 1 ? cp.y : cp.y - (dy - 1);
        float gw[4][10];
        load_gt_row<KS>(gt, cp, r, rg * 4, cp.act && yy >= 0 && yy < H, KS == 1 ? 0 : -(dy - 1) * W, W, HW, gw);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int c = c0 + 4 * cb;
          if (c >= c_end) continue;  // wave-uniform
          float w[4][KS];
          load_weights<KS>(down + (int64_t)c * KK + dy * KS, (int64_t)C * KK, rg * 4, r, w);
#pragma unroll
          for (int dxi = 0; dxi < KS; ++dxi)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < 8; ++i)
                dacc[cb][i] = fmaf(w[j][dxi], gw[j][KS == 1 ? i + 1 : i + 2 - dxi], dacc[cb][i]);
        }
      }
    }
    if (cp.act) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int c = c0 + 4 * cb;
        if (c >= c_end) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) dv[cb][i] += dacc[cb][i];
        store8<E>(dxb + (int64_t)c * HW, dv[cb]);
      }
    }
  }
}

// ---------------------------------------------------------------------------- host
struct ConvGeo {
  bool native;
  int cpw_in, ngroups_in, ngroups_out, split_in, split_out, rank_pad;
  int64_t NP;
};

static int pick_split(int ngroups, int C, int r) {
  int s = (512 + ngroups - 1) / ngroups;            // aim at >= 512 workgroups
  s = std::min(s, std::max(1, C / (4 * r)));        // partial-sum traffic <= ~50 % of the activation bytes
  s = std::min(s, std::max(1, C / 8));              // >= 2 channels per wave
  s = std::min(s, 32);                              // bounded depth of the partial-sum kernel
  return std::max(1, s);
}
static int stream_split(int ngroups, int C) {       // passes that own their output: split for parallelism only
  return std::max(1, std::min((1024 + ngroups - 1) / ngroups, C / 8));  // measured: the dX / dDown passes keep gaining to ~1000 workgroups
}

// C_in / C_out may be passed as 0 by entry points that do not touch that side.
static ConvGeo conv_geo(int B, int C_in, int C_out, int H, int W, int ks, int r) {
  ConvGeo q;
  memset(&q, 0, sizeof(q));
  const int64_t HW = (int64_t)H * W;
  q.native = B > 0 && C_in >= 0 && C_out >= 0 && H > 0 && W > 0 && (ks == 1 || ks == 3) && HW % 8 == 0 &&
             HW < (1ll << 24) && r >= 1 && r <= 16 && (ks == 1 || (W % 8 == 0 && W <= 512));
  if (!q.native) return q;
  q.NP = (int64_t)B * HW / 8;
  const int per_row = ks == 3 ? W / 8 : 1;
  q.cpw_in = (64 / per_row) * per_row;
  q.ngroups_in = (int)((q.NP + q.cpw_in - 1) / q.cpw_in);
  q.ngroups_out = (int)((q.NP + 63) / 64);
  q.split_in = C_in > 0 ? pick_split(q.ngroups_in, C_in, r) : 1;
  q.split_out = C_out > 0 ? pick_split(q.ngroups_out, C_out, r) : 1;
  q.rank_pad = (r + 3) & ~3;
  return q;
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_conv_plan(int32_t B, int32_t C_in, int32_t C_out, int32_t H, int32_t W, int32_t ks,
                                  int32_t r, lora_amd_conv_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr, LORA_AMD_EINVAL, "conv_plan: null output");
  LORA_AMD_CHECK(r >= 1 && r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK, "conv_plan: rank %d outside [1,%d]", r,
                 LORA_AMD_MAX_RANK);
  memset(out, 0, sizeof(*out));
  const ConvGeo q = conv_geo(B, C_in, C_out, H, W, ks, r);
  if (!q.native) return LORA_AMD_OK;
  const int64_t HW = (int64_t)H * W;
  out->native = 1;
  out->cpw_in = q.cpw_in;
  out->ngroups_in = q.ngroups_in;
  out->ngroups_out = q.ngroups_out;
  out->split_in = q.split_in;
  out->split_out = q.split_out;
  out->rank_pad = q.rank_pad;
  out->t_part_floats = (int64_t)q.split_in * B * r * HW;
  out->gt_part_floats = (int64_t)q.split_out * B * r * HW;
  out->up_part_floats = (int64_t)q.ngroups_out * q.rank_pad * C_out;
  out->down_part_floats = (int64_t)q.ngroups_in * q.rank_pad * C_in * ks * ks;
  return LORA_AMD_OK;
}

#define CONV_COMMON(name, Cchk)                                                                           \
  LORA_AMD_CHECK(dtype_ok(act_dtype), LORA_AMD_EINVAL, name ": bad dtype");                               \
  LORA_AMD_CHECK(factor_dtype == LORA_AMD_F32, LORA_AMD_EINVAL,                                           \
                 name ": the conv path takes f32 factors (the trainable masters); convert 16-bit factors first"); \
  LORA_AMD_CHECK(r >= 1 && r <= 16, LORA_AMD_ERANK, name ": native conv path needs rank in [1,16], got %d", r); \
  LORA_AMD_CHECK(q.native, LORA_AMD_EINVAL, name ": geometry not supported (see lora_amd_conv_plan)");

static inline bool al16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

static void launch_finalize(const float *part, int S, int64_t part_stride, const float *sel, int transposed,
                            float *out, int B, int r, int64_t HW, hipStream_t st) {
  if (S > 1) {
    const int64_t n4 = (int64_t)B * r * HW / 4;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(kCT), 0, st, part, S, part_stride,
                       out, n4);
  }
  if (sel != nullptr) {
    const int64_t n4 = (int64_t)B * HW / 4;
    hipLaunchKernelGGL(selector_mix_kernel, dim3((unsigned)((n4 + kCT - 1) / kCT)), dim3(kCT), 0, st, out, sel,
                       transposed, r, (int)HW, n4);
  }
}

extern "C" int lora_amd_conv_down_fwd(const void *x, const void *down, const float *sel, float *t_part, float *t_out,
                                      int32_t B, int32_t C_in, int32_t H, int32_t W, int32_t ks, int32_t r,
                                      int32_t act_dtype, int32_t factor_dtype, void *stream) {
  const ConvGeo q = conv_geo(B, C_in, 0, H, W, ks, r);
  CONV_COMMON("conv_down_fwd", C_in);
  LORA_AMD_CHECK(x && down && t_part && t_out && al16(x) && al16(t_part) && al16(t_out), LORA_AMD_EINVAL,
                 "conv_down_fwd: null or unaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  const int S = q.split_in, cps = (C_in + S - 1) / S;
  const bool direct = S == 1;  // single split: the partials ARE the result
  float *dst = direct ? t_out : t_part;
  const dim3 grid((unsigned)q.ngroups_in, (unsigned)S, (unsigned)((r + 3) / 4));
#define CD(E, KSV)                                                                                                 \
  hipLaunchKernelGGL((conv_down_fwd_kernel<E, KSV>), grid, dim3(kCT), 0, st,                                       \
                     reinterpret_cast<const typename E::storage *>(x), reinterpret_cast<const float *>(down), dst, \
                     B, C_in, H, W, r, q.cpw_in, q.NP, cps)
#define CD_E(E) do { if (ks == 1) CD(E, 1); else CD(E, 3); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: CD_E(f32_t); break;
    case LORA_AMD_F16: CD_E(f16_t); break;
    default: CD_E(bf16_t); break;
  }
#undef CD_E
#undef CD
  launch_finalize(t_part, S, (int64_t)B * r * H * W, sel, 0, t_out, B, r, (int64_t)H * W, st);
  return check_launch("lora_amd_conv_down_fwd");
}

extern "C" int lora_amd_conv_up_fwd(void *y, const float *t, const void *up, int32_t B, int32_t C_out, int32_t H,
                                    int32_t W, int32_t r, int32_t act_dtype, int32_t factor_dtype, float scale,
                                    float dropout_p, uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream) {
  const ConvGeo q = conv_geo(B, 0, C_out, H, W, 1, r);
  CONV_COMMON("conv_up_fwd", C_out);
  LORA_AMD_CHECK(y && t && up && al16(y) && al16(t), LORA_AMD_EINVAL, "conv_up_fwd: null or unaligned pointer");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "conv_up_fwd: dropout p=%f", dropout_p);
  hipStream_t st = (hipStream_t)stream;
  // the pass streams Y: no reduction across channels, so split purely for parallelism
  const int S = stream_split(q.ngroups_out, C_out);
  const int cps = (C_out + S - 1) / S;
  const dim3 grid((unsigned)q.ngroups_out, (unsigned)S);
  const int RT = r <= 4 ? 4 : r <= 8 ? 8 : 16;
  const bool drop = dropout_p > 0.f;
  const int HW = H * W;
#define CU(E, RTV, D)                                                                                            \
  hipLaunchKernelGGL((conv_up_fwd_kernel<E, RTV, D>), grid, dim3(kCT), 0, st,                                     \
                     reinterpret_cast<typename E::storage *>(y), t, reinterpret_cast<const float *>(up), B, C_out, \
                     HW, r, 64, q.NP, cps, scale, dropout_p, seed, offset, offset_dev)
#define CU_RT(E, D) do { if (RT == 4) CU(E, 4, D); else if (RT == 8) CU(E, 8, D); else CU(E, 16, D); } while (0)
#define CU_E(E) do { if (drop) CU_RT(E, true); else CU_RT(E, false); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: CU_E(f32_t); break;
    case LORA_AMD_F16: CU_E(f16_t); break;
    default: CU_E(bf16_t); break;
  }
#undef CU_E
#undef CU_RT
#undef CU
  return check_launch("lora_amd_conv_up_fwd");
}

extern "C" int lora_amd_conv_bwd_g(const void *g, const float *t, const void *up, const float *sel, float *gt_part,
                                   float *gt_out, float *up_part, int32_t B, int32_t C_out, int32_t H, int32_t W,
                                   int32_t r, int32_t act_dtype, int32_t factor_dtype, float scale, float dropout_p,
                                   uint64_t seed, uint64_t offset, const uint64_t *offset_dev, void *stream) {
  const ConvGeo q = conv_geo(B, 0, C_out, H, W, 1, r);
  CONV_COMMON("conv_bwd_g", C_out);
  LORA_AMD_CHECK(g && t && up && gt_part && gt_out && up_part && al16(g) && al16(t) && al16(gt_part) && al16(gt_out),
                 LORA_AMD_EINVAL, "conv_bwd_g: null or unaligned pointer");
  LORA_AMD_CHECK(dropout_p >= 0.f && dropout_p < 1.f, LORA_AMD_EINVAL, "conv_bwd_g: dropout p=%f", dropout_p);
  hipStream_t st = (hipStream_t)stream;
  const int S = q.split_out, cps = (C_out + S - 1) / S;
  const bool direct = S == 1;
  float *dst = direct ? gt_out : gt_part;
  const dim3 grid((unsigned)q.ngroups_out, (unsigned)S, (unsigned)((r + 3) / 4));
  const bool drop = dropout_p > 0.f;
  const int HW = H * W;
#define CG(E, D)                                                                                                  \
  hipLaunchKernelGGL((conv_bwd_g_kernel<E, D>), grid, dim3(kCT), 0, st,                                           \
                     reinterpret_cast<const typename E::storage *>(g), t, reinterpret_cast<const float *>(up), dst, \
                     up_part, B, C_out, HW, r, q.rank_pad, 64, q.NP, cps, scale, dropout_p, seed, offset, offset_dev)
#define CG_E(E) do { if (drop) CG(E, true); else CG(E, false); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: CG_E(f32_t); break;
    case LORA_AMD_F16: CG_E(f16_t); break;
    default: CG_E(bf16_t); break;
  }
#undef CG_E
#undef CG
  launch_finalize(gt_part, S, (int64_t)B * r * HW, sel, 1, gt_out, B, r, HW, st);
  return check_launch("lora_amd_conv_bwd_g");
}

extern "C" int lora_amd_conv_bwd_x(const void *x, void *dx, const float *gt, const void *down, float *down_part,
                                   int32_t B, int32_t C_in, int32_t H, int32_t W, int32_t ks, int32_t r,
                                   int32_t act_dtype, int32_t factor_dtype, void *stream) {
  const ConvGeo q = conv_geo(B, C_in, 0, H, W, ks, r);
  CONV_COMMON("conv_bwd_x", C_in);
  LORA_AMD_CHECK(x && gt && down && down_part && al16(x) && al16(gt) && (dx == nullptr || al16(dx)), LORA_AMD_EINVAL,
                 "conv_bwd_x: null or unaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  // both passes own their outputs per (channel, chunk) / (rank, channel): channel splits need no reduction
  const int S = stream_split(q.ngroups_in, C_in);
  const int cps = (C_in + S - 1) / S;
  const dim3 grid_dn((unsigned)q.ngroups_in, (unsigned)S, (unsigned)((r + 3) / 4));
  const dim3 grid_dx((unsigned)q.ngroups_in, (unsigned)S);
#define CX(E, KSV)                                                                                                  \
  do {                                                                                                              \
    hipLaunchKernelGGL((conv_bwd_down_kernel<E, KSV>), grid_dn, dim3(kCT), 0, st,                                   \
                       reinterpret_cast<const typename E::storage *>(x), gt, down_part, B, C_in, H, W, r, q.rank_pad, \
                       q.cpw_in, q.NP, cps);                                                                        \
    if (dx != nullptr)                                                                                              \
      hipLaunchKernelGGL((conv_bwd_dx_kernel<E, KSV>), grid_dx, dim3(kCT), 0, st,                                   \
                         reinterpret_cast<typename E::storage *>(dx), gt, reinterpret_cast<const float *>(down), B, \
                         C_in, H, W, r, q.cpw_in, q.NP, cps);                                                       \
  } while (0)
#define CX_E(E) do { if (ks == 1) CX(E, 1); else CX(E, 3); } while (0)
  switch (act_dtype) {
    case LORA_AMD_F32: CX_E(f32_t); break;
    case LORA_AMD_F16: CX_E(f16_t); break;
    default: CX_E(bf16_t); break;
  }
#undef CX_E
#undef CX
  return check_launch("lora_amd_conv_bwd_x");
}
