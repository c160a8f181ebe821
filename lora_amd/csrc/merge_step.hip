// K3 inside the training step: W_eff = W + scale up down AND its transpose from ONE read of W.
//
// replaces: lora_diffusion/lora.py:635-669 (collapse_lora's merge) applied to scratch copies once per optimiser step
//           (ops.MergedWeights: forward = frozen GEMM on W_eff, lora.py:53-58 without dropout; input gradient = frozen GEMM
//           on W_eff^T), and round 3's way of producing the two scratch weights: one merge_co_kernel site for W_eff (eight
//           40-row sub-sites for a head-padded output) plus a `transposed` site reading a frozen transposed COPY of W
//           (4 N K e of traffic per site and three extra weight copies; the in-step launch sat at 0.58 of the byte roof).
//
// One workgroup owns a tile of 128 rows (n) x 64 columns (k) of one site:
//   * column-owner mapping as in merge.hip: a thread owns one 16-byte column chunk, its r x 8 block of `down` lives in
//     registers, the tile's `up` rows in LDS; it walks rows slot, slot + 32, ... with all its W loads in flight;
//   * the merged value is computed ONCE in f32 (fma chain over the ranks in rank order, as merge_co_kernel's ROUND_ONCE),
//     rounded once to the weight dtype, stored to W_eff (16-byte lanes, full 128-byte row segments) and dropped into an
//     LDS image of the tile; after one barrier the image is read back column-wise (2-byte LDS reads, k fastest across
//     lanes) and stored to W_eff^T as 16-byte lanes that are contiguous along n: 256 bytes per k row per instruction;
//   * layouts the GEMMs want are written directly: a head-padded OUTPUT of the adapter (rows of W_eff / columns of W_eff^T
//     at (n / d) D + n % d), a head-padded INPUT (columns of W_eff / rows of W_eff^T at (k / d) D + k % d), and row /
//     column offsets into a wider buffer (ld_out, ld_out_t: q, k, v of an attention block share one scratch weight so
//     that their forward is ONE GEMM);
//   * rounding: ROUND_ONCE (nearest even) or ROUND_DITHER — nearest with a FIXED per-element dither, i.e. the result the
//     reference's bf16 cast would give if the frozen f32 master had a random sub-ulp residue: rounding W + delta to bf16
//     nearest-even on a weight that is already ON the bf16 grid deletes every delta below half an ulp (the adapter's
//     contribution for the first hundreds of steps from up = 0, lora.py:50-51) and quantises the rest in whole ulps;
//     with the dither P(round up) = frac(delta / ulp) per element, unbiased over the elements of a row (what a GEMM sums),
//     deterministic from step to step (the dither depends on (site, n, k) only), and exactly W where delta = 0.
// Algorithmic bytes per site: N K e (W once) + N K e (W_eff) [+ N K e (W_eff^T)] + (N + K) r 4: 3 instead of 4 N K e.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace lora_amd {

typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

constexpr int kMsThreads = 256;
constexpr int kMsMaxSitesLds = 256;   // site-table prefix kept in LDS for the tile search (larger tables: searched in L2)
// tile geometries (rows n x columns k); the LDS image row is TC * 2 + 4 bytes (the column gather is then 2-way at worst)
struct MsTile { int tr, tc; };
constexpr MsTile kMsTiles[4] = {{128, 64}, {64, 128}, {128, 128}, {256, 64}};
static int g_ms_tile = 2;     // lora_amd_merge_step_set_tuning; 128 x 128 measured best (profiles/r04_kbench_mstep.log)
static int g_ms_dither = 2;   // 1: one hash per element; 2: two hashes per 16-byte chunk, 16-bit windows of the 64 bits

__device__ __forceinline__ int ms_map(int i, int d, int D) { return d ? (i / d) * D + (i % d) : i; }

// 16 bits of hash of a 64-bit key (murmur-style finaliser): the element's fixed dither
__device__ __forceinline__ uint32_t ms_dither16(uint32_t lo, uint32_t hi) {
  uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA77u + 0x165667B1u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h >> 16;
}

// the dithers of the 8 elements of a 16-byte chunk from ONE hash chain (four 32-bit multiplies per chunk instead of
// twenty-four: v_mul_lo_u32 is a quarter-rate instruction and the per-element hash was half of the kernel's VALU time):
// element i takes bytes (i, i + 1 mod 8) of the 64 hashed bits — uniform 16-bit marginals, independent high bytes
struct MsDither8 { uint32_t h1, h2; };
__device__ __forceinline__ MsDither8 ms_dither_chunk(uint32_t lo, uint32_t hi) {
  uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA77u + 0x165667B1u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  MsDither8 d;
  d.h1 = h * 0x297A2D39u; d.h1 ^= d.h1 >> 15;
  d.h2 = h * 0xC2B2AE3Du; d.h2 ^= d.h2 >> 13;
  return d;
}
template <int I>
__device__ __forceinline__ uint32_t ms_dither_pick(const MsDither8 &d) {
  // v_perm_b32: selector bytes 0-3 = h1's bytes, 4-7 = h2's bytes, 0x0c = zero
  return __builtin_amdgcn_perm(d.h2, d.h1, 0x0c0c0000u | (uint32_t)(((I + 1) & 7) << 8) | (uint32_t)I);
}

// f32 -> 16-bit storage bits, nearest-even or dithered (see the header).  Finite inputs.
template <class EW, bool DITHER>
__device__ __forceinline__ uint32_t ms_round(float v, uint32_t u16) {
  if constexpr (!DITHER) {
    union { typename EW::storage s; unsigned short b; } c;
    c.s = EW::from_f(v);
    return c.b;
  } else if constexpr (EW::kCode == LORA_AMD_BF16) {
    // truncate(bits + u): rounds the magnitude up with probability (low 16 bits) / 65536; exact where they are zero
    return (__builtin_bit_cast(uint32_t, v) + u16) >> 16;
  } else {
    // f16: neighbours toward zero / away from zero of v, pick by the position of v between them
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    union { h2 v; unsigned short b[2]; } z;
    z.v = __builtin_amdgcn_cvt_pkrtz(v, 0.f);
    union { _Float16 s; unsigned short b; } c;
    c.b = z.b[0];
    const float flo = (float)c.s;
    if (flo == v) return c.b;
    union { _Float16 s; unsigned short b; } n;
    n.b = (unsigned short)(c.b + 1);  // sign-magnitude: the next value away from zero
    const float fhi = (float)n.s;
    const float frac = (v - flo) / (fhi - flo);
    return ((float)u16 < frac * 65536.f) ? n.b : c.b;
  }
}

template <class EW, int RT, int DITHER, int TR, int TC>
__global__ __launch_bounds__(kMsThreads) void merge_step_kernel(const lora_amd_mstep_site *__restrict__ sites, int n_sites,
                                                                float alpha) {
  using SW = typename EW::storage;
  constexpr int C8 = TC / 8, SLOTS = kMsThreads / C8, U = TR / SLOTS, PITCH = TC * 2 + 4, CH = TR / 8;
  static_assert(U >= 1 && U * SLOTS == TR && (CH & (CH - 1)) == 0, "tile geometry");
  __shared__ int64_t s_begin[kMsMaxSitesLds];
  __shared__ __attribute__((aligned(16))) float s_up[TR * RT];
  __shared__ __attribute__((aligned(16))) unsigned char s_img[TR * PITCH];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  int si;
  if (n_sites <= kMsMaxSitesLds) {
    for (int i = tid; i < n_sites; i += kMsThreads) s_begin[i] = sites[i].tile_begin;
    __syncthreads();
    int lo = 0, hi = n_sites - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_begin[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    si = lo;
  } else {
    int lo = 0, hi = n_sites - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sites[mid].tile_begin <= tile) lo = mid; else hi = mid - 1;
    }
    si = lo;
  }
  const lora_amd_mstep_site s = sites[si];
  const int r = s.r;
  const int64_t tl = tile - s.tile_begin;
  const int tn = (int)(tl / s.tiles_k), tk = (int)(tl - (int64_t)tn * s.tiles_k);
  const int row0 = tn * TR, col0 = tk * TC;
  const int nrows = min(TR, s.N - row0);
  const int ncol8 = min(TC, s.K - col0) >> 3;      // 16-byte chunks of this tile (K % 8 == 0)
  const int cl = tid % C8, slot = tid / C8;         // C8 chunk columns x SLOTS row slots
  const bool live = cl < ncol8;
  const int col = col0 + (live ? cl : 0) * 8;      // idle lanes of a ragged last column tile stay inside the row

  // this thread's r x 8 block of `down` (f32 [r, K])
  float fc[RT][8];
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    if (j < r && live) {
      const float4 a = gl_ld4(s.down + (int64_t)j * s.K + col);
      const float4 b = gl_ld4(s.down + (int64_t)j * s.K + col + 4);
      fc[j][0] = a.x; fc[j][1] = a.y; fc[j][2] = a.z; fc[j][3] = a.w;
      fc[j][4] = b.x; fc[j][5] = b.y; fc[j][6] = b.z; fc[j][7] = b.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) fc[j][i] = 0.f;
    }
  }
  // the tile's rows of `up` (f32 [N, r]) -> LDS [nrows][RT]
  for (int i = tid; i < nrows * RT; i += kMsThreads) {
    const int rl = i / RT, j = i - rl * RT;
    s_up[i] = j < r ? gl(s.up)[(int64_t)(row0 + rl) * r + j] : 0.f;
  }
  __syncthreads();

  const SW *win = reinterpret_cast<const SW *>(s.w) + (int64_t)row0 * s.K + col;
  SW *wout = reinterpret_cast<SW *>(s.out) + ms_map(col, s.col_d, s.col_D);
  su32x4 w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rl = slot + u * SLOTS;
    const bool ok = live && rl < nrows;
    w[u] = __builtin_nontemporal_load(gl(reinterpret_cast<const su32x4 *>(win + (int64_t)(ok ? rl : 0) * s.K)));
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rl = slot + u * SLOTS;
    if (!(live && rl < nrows)) continue;
    union { su32x4 v; SW e[8]; } in;
    in.v = w[u];
    const float *upr = s_up + rl * RT;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const float uj = upr[j];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = fmaf(uj, fc[j][i], p[i]);
    }
    const int n = row0 + rl;
    uint32_t d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (DITHER == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = ms_dither16((uint32_t)n * (uint32_t)s.K + (uint32_t)(col + i), (uint32_t)s.dither_key);
    } else if constexpr (DITHER == 2) {
      const MsDither8 h = ms_dither_chunk((uint32_t)n * (uint32_t)s.K + (uint32_t)col, (uint32_t)s.dither_key);
      d[0] = ms_dither_pick<0>(h); d[1] = ms_dither_pick<1>(h); d[2] = ms_dither_pick<2>(h); d[3] = ms_dither_pick<3>(h);
      d[4] = ms_dither_pick<4>(h); d[5] = ms_dither_pick<5>(h); d[6] = ms_dither_pick<6>(h); d[7] = ms_dither_pick<7>(h);
    }
    uint32_t b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = ms_round<EW, DITHER != 0>(fmaf(alpha, p[i], EW::to_f(in.e[i])), d[i]);
    su32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = b[2 * i] | (b[2 * i + 1] << 16);
    __builtin_nontemporal_store(o, gl(reinterpret_cast<su32x4 *>(wout + (int64_t)ms_map(n, s.row_d, s.row_D) * s.ld_out)));
    if (s.out_t != nullptr) {
      uint32_t *img = reinterpret_cast<uint32_t *>(s_img + rl * PITCH + cl * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) img[i] = o[i];
    }
  }
  if (s.out_t == nullptr) return;   // site-uniform
  __syncthreads();
  // the image column-wise: a task = (column k of the tile, chunk of 8 rows); lanes take k fastest (4 consecutive k, then
  // the row chunks) so that one instruction stores whole runs of 16-byte row chunks: 256 contiguous bytes per k row of
  // W_eff^T for a 128-row tile
  const int nk = ncol8 * 8, nchunks = (nrows + 7) >> 3;
  SW *wt = reinterpret_cast<SW *>(s.out_t);
  for (int t = tid; t < nk * CH; t += kMsThreads) {
    const int kq = t / (4 * CH), rem = t % (4 * CH);  // 4 columns per group of 4 CH tasks
    const int k = kq * 4 + (rem & 3), ch = rem >> 2;
    if (ch >= nchunks) continue;
    const unsigned char *src = s_img + (ch * 8) * PITCH + k * 2;
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = *reinterpret_cast<const unsigned short *>(src + i * PITCH);
    const int n0 = row0 + ch * 8;
    su32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (uint32_t)e[2 * i] | ((uint32_t)e[2 * i + 1] << 16);
    SW *dst = wt + (int64_t)ms_map(col0 + k, s.col_d, s.col_D) * s.ld_out_t + ms_map(n0, s.row_d, s.row_D);
    if (n0 + 8 <= s.N) {
      __builtin_nontemporal_store(o, gl(reinterpret_cast<su32x4 *>(dst)));
    } else {
      for (int i = 0; i < s.N - n0; ++i) gl(reinterpret_cast<unsigned short *>(dst))[i] = e[i];
    }
  }
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_merge_step_plan(lora_amd_mstep_site *sites, int32_t n, int32_t w_dtype, int64_t *total_tiles) {
  LORA_AMD_CHECK(sites && n >= 1 && total_tiles, LORA_AMD_EINVAL, "merge_step_plan: bad argument");
  LORA_AMD_CHECK(w_dtype == LORA_AMD_F16 || w_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "merge_step_plan: 16-bit weights only (f32 weights: lora_amd_merge_batched)");
  auto map_ok = [](int d, int D, int cols) { return d == 0 || (d > 0 && d % 8 == 0 && D % 8 == 0 && D >= d && cols % d == 0); };
  int64_t acc = 0;
  for (int i = 0; i < n; ++i) {
    lora_amd_mstep_site &s = sites[i];
    LORA_AMD_CHECK(s.N > 0 && s.K > 0 && s.K % 8 == 0 && s.N % 8 == 0, LORA_AMD_EINVAL,
                   "merge_step_plan: site %d: N = %d, K = %d must be multiples of 8", i, s.N, s.K);
    LORA_AMD_CHECK(s.r >= 1 && s.r <= 16, LORA_AMD_ERANK, "merge_step_plan: site %d: rank %d outside [1,16]", i, s.r);
    LORA_AMD_CHECK(s.w && s.up && s.down && s.out, LORA_AMD_EINVAL, "merge_step_plan: site %d: null pointer", i);
    LORA_AMD_CHECK(map_ok(s.row_d, s.row_D, s.N) && map_ok(s.col_d, s.col_D, s.K), LORA_AMD_EINVAL,
                   "merge_step_plan: site %d: bad head layout", i);
    const int64_t kp = s.col_d ? (int64_t)(s.K / s.col_d) * s.col_D : s.K, np = s.row_d ? (int64_t)(s.N / s.row_d) * s.row_D : s.N;
    LORA_AMD_CHECK((((uintptr_t)s.w | (uintptr_t)s.out | (uintptr_t)s.out_t | (uintptr_t)s.down) & 15u) == 0 &&
                       ((uintptr_t)s.up & 3u) == 0 && s.ld_out >= kp && s.ld_out % 8 == 0 &&
                       (s.out_t == nullptr || (s.ld_out_t >= np && s.ld_out_t % 8 == 0)),
                   LORA_AMD_EINVAL, "merge_step_plan: site %d: 16-byte aligned tensors and row strides expected", i);
    const MsTile tg = kMsTiles[g_ms_tile];
    s.tiles_k = (s.K + tg.tc - 1) / tg.tc;
    s.tile_begin = acc;
    acc += (int64_t)s.tiles_k * ((s.N + tg.tr - 1) / tg.tr);
  }
  LORA_AMD_CHECK(acc < (1ll << 31), LORA_AMD_EINVAL, "merge_step_plan: too many tiles");
  *total_tiles = acc | ((int64_t)g_ms_tile << 40);  // opaque to the caller: the tile count and the geometry it was planned for
  return LORA_AMD_OK;
}

extern "C" int lora_amd_merge_step(const lora_amd_mstep_site *sites_dev, int32_t n, int64_t plan_value, int32_t rank_max,
                                   int32_t w_dtype, float alpha, int32_t rounding, void *stream) {
  const int ms_tile = (int)(plan_value >> 40);
  const int64_t total_tiles = plan_value & ((1ll << 40) - 1);
  LORA_AMD_CHECK(ms_tile >= 0 && ms_tile < 4, LORA_AMD_EINVAL, "merge_step: not a value of lora_amd_merge_step_plan");
  LORA_AMD_CHECK(sites_dev && n >= 1 && total_tiles >= 1 && total_tiles < (1ll << 31), LORA_AMD_EINVAL, "merge_step: bad argument");
  LORA_AMD_CHECK(w_dtype == LORA_AMD_F16 || w_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL, "merge_step: 16-bit weights only");
  LORA_AMD_CHECK(rank_max >= 1 && rank_max <= 16, LORA_AMD_ERANK, "merge_step: rank %d outside [1,16]", rank_max);
  LORA_AMD_CHECK(rounding == LORA_AMD_ROUND_ONCE || rounding == LORA_AMD_ROUND_DITHER, LORA_AMD_EINVAL,
                 "merge_step: rounding must be ROUND_ONCE or ROUND_DITHER, got %d", rounding);
  hipStream_t st = (hipStream_t)stream;
  const int RT = rank_max <= 4 ? 4 : rank_max <= 8 ? 8 : 16;
  const bool dith = rounding == LORA_AMD_ROUND_DITHER;
  const int dmode = dith ? g_ms_dither : 0;
#define MS_K(E, RTV, DV, TRV, TCV)                                                                                     \
  hipLaunchKernelGGL((merge_step_kernel<E, RTV, DV, TRV, TCV>), dim3((unsigned)total_tiles), dim3(kMsThreads), 0, st, sites_dev, n, alpha)
#define MS_D(E, RTV, TRV, TCV)                                                                                         \
  do { if (dmode == 0) MS_K(E, RTV, 0, TRV, TCV); else if (dmode == 1) MS_K(E, RTV, 1, TRV, TCV); else MS_K(E, RTV, 2, TRV, TCV); } while (0)
#define MS(E, RTV)                                                                                                     \
  do {                                                                                                                 \
    if (ms_tile == 0) MS_D(E, RTV, 128, 64);                                                                           \
    else if (ms_tile == 1) MS_D(E, RTV, 64, 128);                                                                      \
    else if (ms_tile == 2) MS_D(E, RTV, 128, 128);                                                                     \
    else MS_D(E, RTV, 256, 64);                                                                                        \
  } while (0)
#define MS_E(E) do { if (RT == 4) MS(E, 4); else if (RT == 8) MS(E, 8); else MS(E, 16); } while (0)
  if (w_dtype == LORA_AMD_F16) MS_E(f16_t); else MS_E(bf16_t);
#undef MS_E
#undef MS
#undef MS_D
#undef MS_K
  return check_launch("lora_amd_merge_step");
}

// Tuning hook (kbench / tests): tile geometry 0..3 = 128x64, 64x128, 128x128, 256x64 (rows x columns; tables must be planned
// AFTER the call), dither form 1 = hash per element, 2 = hash per 16-byte chunk.  Negative keeps the current value.
extern "C" int lora_amd_merge_step_set_tuning(int32_t tile, int32_t dither) {
  LORA_AMD_CHECK(tile < 4 && dither < 3 && dither != 0, LORA_AMD_EINVAL, "merge_step_set_tuning: tile %d, dither %d", tile, dither);
  if (tile >= 0) g_ms_tile = tile;
  if (dither > 0) g_ms_dither = dither;
  return LORA_AMD_OK;
}
