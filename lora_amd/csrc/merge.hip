// K3 — fused merge  W' = W + alpha * (up @ down)  for ALL adapter sites in one launch.
//
// Replaces lora_diffusion/lora.py:635-669 (collapse_lora): the reference
// materialises up@down as a full [N,K] matrix, then runs a scale pass and an
// add pass and allocates a fresh Parameter per site (>= 3x the minimal HBM
// traffic, 144 x 3 launches).  Here W is streamed exactly once: every lane owns
// 8 consecutive elements of a row (one 16-byte load, one 16-byte store), the
// [r, cols] slab of `down` and the [rows, r] slab of `up` that a tile needs are
// staged in LDS as f32, and the rank-r dot product is an f32 fma chain.
//
// HBM-bound: algorithmic bytes per site = 2*N*K*e_w + (N+K)*r*e_ab, flops
// 2*N*K*r (AI = r/2 flop/B for bf16 — never MFMA work).
#include "common.hpp"

namespace lora_amd {

constexpr int kMergeThreads = 256;
constexpr int kMergeLdsDownFloats = 8192;  // 32 KiB: [r][cols_per_tile]
constexpr int kMergeLdsUpFloats = 2048;    // 8 KiB: [rows_per_tile][r]
constexpr int kMergeUnroll = 4;

static int64_t g_merge_tile_elems = 32768;  // tuning knob (lora_amd_set_tuning)
static int64_t g_merge_blocks_per_cu = 4;

template <class E>
__device__ inline float ld_as_f32(const void *p, int64_t i) {
  return E::to_f(reinterpret_cast<const typename E::storage *>(p)[i]);
}

// uniform binary search: last site whose tile_begin <= tile
__device__ inline int find_site(const lora_amd_merge_site *sites, int n_sites, int64_t tile) {
  int lo = 0, hi = n_sites - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (sites[mid].tile_begin <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <class EW, class EAB, int ROUND>
__device__ inline float merge_one(float w, float p, float alpha) {
  if (ROUND == LORA_AMD_ROUND_REFERENCE) {
    // (up @ down) in the factors' dtype, .type(W.dtype), alpha * (...), W + (...)
    // each torch op rounds its result to the tensor dtype (lora.py:646-655).
    float pw = round_to<EW>(round_to<EAB>(p));
    float q = round_to<EW>(alpha * pw);
    return w + q;  // final rounding happens in store8 / from_f
  } else {
    return fmaf(alpha, p, w);
  }
}

template <class EW, class EAB, int ROUND>
__global__ __launch_bounds__(kMergeThreads) void merge_kernel(
    const lora_amd_merge_site *__restrict__ sites, int n_sites, int64_t total_tiles, float alpha) {
  using SW = typename EW::storage;
  __shared__ __attribute__((aligned(16))) float s_down[kMergeLdsDownFloats];
  __shared__ __attribute__((aligned(16))) float s_up[kMergeLdsUpFloats];

  const int tid = threadIdx.x;
  for (int64_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int si = find_site(sites, n_sites, tile);
    const lora_amd_merge_site s = sites[si];
    const int r = s.r;
    const int64_t tl = tile - s.tile_begin;
    const int tr = (int)(tl / s.tiles_k), tc = (int)(tl % s.tiles_k);
    const int row0 = tr * s.rows_per_tile, col0 = tc * s.cols_per_tile;
    const int nrows = min(s.rows_per_tile, s.N - row0);
    const int ncols = min(s.cols_per_tile, s.K - col0);

    const bool vec = (s.flags & 1) != 0;
    __syncthreads();  // previous tile's LDS readers are done
    if (vec) {
      // s_down layout [r][2][ncols/8][4]: a lane's 8 columns are two conflict-free
      // 16-byte slots (stride 16 B across lanes) instead of one 32-byte-strided pair.
      const int c8 = ncols >> 3;
      for (int i = tid; i < r * ncols; i += kMergeThreads) {
        int j = i / ncols, c = i - j * ncols;
        float v = ld_as_f32<EAB>(s.down, (int64_t)j * s.K + col0 + c);
        s_down[((j * 2 + ((c >> 2) & 1)) * c8 + (c >> 3)) * 4 + (c & 3)] = v;
      }
    } else {
      for (int i = tid; i < r * ncols; i += kMergeThreads) {
        int j = i / ncols, c = i - j * ncols;
        s_down[i] = ld_as_f32<EAB>(s.down, (int64_t)j * s.K + col0 + c);
      }
    }
    for (int i = tid; i < nrows * r; i += kMergeThreads)
      s_up[i] = ld_as_f32<EAB>(s.up, (int64_t)row0 * r + i);
    __syncthreads();

    const SW *win = reinterpret_cast<const SW *>(s.w_in);
    SW *wout = reinterpret_cast<SW *>(s.w_out);

    if (vec) {
      const int c8 = ncols >> 3;
      const int nchunk = nrows * c8;
      const int dq = kMergeThreads / c8, dr = kMergeThreads % c8;
      int rl = tid / c8, cc = tid % c8;
      for (int c = tid; c < nchunk; c += kMergeThreads * kMergeUnroll) {
        float w[kMergeUnroll][8];
        int rls[kMergeUnroll], ccs[kMergeUnroll];
        bool ok[kMergeUnroll];
#pragma unroll
        for (int u = 0; u < kMergeUnroll; ++u) {
          ok[u] = (c + u * kMergeThreads) < nchunk;
          rls[u] = rl; ccs[u] = cc;
          if (ok[u]) load8<EW>(win + (int64_t)(row0 + rl) * s.K + col0 + cc * 8, w[u]);
          rl += dq; cc += dr;
          if (cc >= c8) { cc -= c8; ++rl; }
        }
#pragma unroll
        for (int u = 0; u < kMergeUnroll; ++u) {
          if (!ok[u]) continue;
          float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const float *upr = s_up + rls[u] * r;
          for (int j = 0; j < r; ++j) {
            const float uj = upr[j];
            const float4 d0 = *reinterpret_cast<const float4 *>(&s_down[((j * 2 + 0) * c8 + ccs[u]) * 4]);
            const float4 d1 = *reinterpret_cast<const float4 *>(&s_down[((j * 2 + 1) * c8 + ccs[u]) * 4]);
            p[0] = fmaf(uj, d0.x, p[0]); p[1] = fmaf(uj, d0.y, p[1]);
            p[2] = fmaf(uj, d0.z, p[2]); p[3] = fmaf(uj, d0.w, p[3]);
            p[4] = fmaf(uj, d1.x, p[4]); p[5] = fmaf(uj, d1.y, p[5]);
            p[6] = fmaf(uj, d1.z, p[6]); p[7] = fmaf(uj, d1.w, p[7]);
          }
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = merge_one<EW, EAB, ROUND>(w[u][i], p[i], alpha);
          store8<EW>(wout + (int64_t)(row0 + rls[u]) * s.K + col0 + ccs[u] * 8, o);
        }
      }
    } else {
      const int n = nrows * ncols;
      for (int i = tid; i < n; i += kMergeThreads) {
        int rl = i / ncols, c = i - rl * ncols;
        float p = 0.f;
        for (int j = 0; j < r; ++j) p = fmaf(s_up[rl * r + j], s_down[j * ncols + c], p);
        int64_t off = (int64_t)(row0 + rl) * s.K + col0 + c;
        wout[off] = EW::from_f(merge_one<EW, EAB, ROUND>(EW::to_f(win[off]), p, alpha));
      }
    }
  }
}

template <class EW, class EAB>
static void launch_merge(const lora_amd_merge_site *sites, int n_sites, int64_t total_tiles,
                         float alpha, int rounding, int grid, hipStream_t st) {
  if (rounding == LORA_AMD_ROUND_REFERENCE)
    hipLaunchKernelGGL((merge_kernel<EW, EAB, LORA_AMD_ROUND_REFERENCE>), dim3(grid), dim3(kMergeThreads), 0, st,
                       sites, n_sites, total_tiles, alpha);
  else
    hipLaunchKernelGGL((merge_kernel<EW, EAB, LORA_AMD_ROUND_ONCE>), dim3(grid), dim3(kMergeThreads), 0, st,
                       sites, n_sites, total_tiles, alpha);
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_merge_plan(lora_amd_merge_site *sites, int32_t n_sites, int32_t w_dtype,
                                   int64_t *total_tiles) {
  LORA_AMD_CHECK(sites && total_tiles && n_sites >= 0, LORA_AMD_EINVAL, "merge_plan: null argument");
  LORA_AMD_CHECK(dtype_ok(w_dtype), LORA_AMD_EINVAL, "merge_plan: bad w_dtype %d", w_dtype);
  int64_t acc = 0;
  for (int i = 0; i < n_sites; ++i) {
    lora_amd_merge_site &s = sites[i];
    LORA_AMD_CHECK(s.N > 0 && s.K > 0, LORA_AMD_EINVAL, "merge_plan: site %d has N=%d K=%d", i, s.N, s.K);
    LORA_AMD_CHECK(s.r >= 1 && s.r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK,
                   "merge_plan: site %d rank %d outside [1,%d]", i, s.r, LORA_AMD_MAX_RANK);
    int max_cols = (kMergeLdsDownFloats / s.r) & ~7;
    int k8 = (s.K + 7) & ~7;
    int cols = k8 < max_cols ? k8 : max_cols;
    int64_t rows = g_merge_tile_elems / cols;
    int max_rows = kMergeLdsUpFloats / s.r;
    if (rows > max_rows) rows = max_rows;
    if (rows > 128) rows = 128;
    if (rows < 8) rows = 8;
    if (rows > s.N) rows = s.N;
    s.cols_per_tile = cols;
    s.rows_per_tile = (int32_t)rows;
    s.tiles_k = (s.K + cols - 1) / cols;
    s.tile_begin = acc;
    bool aligned = (((uintptr_t)s.w_in | (uintptr_t)s.w_out) & 15u) == 0;
    s.flags = (s.K % 8 == 0 && aligned) ? 1 : 0;
    s.reserved = 0;
    acc += (int64_t)s.tiles_k * ((s.N + rows - 1) / rows);
  }
  *total_tiles = acc;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_merge_batched(const lora_amd_merge_site *sites_dev, int32_t n_sites,
                                      int64_t total_tiles, int32_t w_dtype, int32_t ab_dtype,
                                      float alpha, int32_t rounding, void *stream) {
  LORA_AMD_CHECK(sites_dev && n_sites > 0 && total_tiles > 0, LORA_AMD_EINVAL, "merge: empty site table");
  LORA_AMD_CHECK(dtype_ok(w_dtype) && dtype_ok(ab_dtype), LORA_AMD_EINVAL, "merge: bad dtype");
  LORA_AMD_CHECK(rounding == LORA_AMD_ROUND_REFERENCE || rounding == LORA_AMD_ROUND_ONCE, LORA_AMD_EINVAL,
                 "merge: bad rounding mode %d", rounding);
  hipStream_t st = (hipStream_t)stream;
  int64_t cap = 256 * g_merge_blocks_per_cu;
  int grid = (int)(total_tiles < cap ? total_tiles : cap);
#define DISPATCH_AB(EW)                                                                                   \
  switch (ab_dtype) {                                                                                     \
    case LORA_AMD_F32: launch_merge<EW, f32_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, st); break; \
    case LORA_AMD_F16: launch_merge<EW, f16_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, st); break; \
    default: launch_merge<EW, bf16_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, st); break;          \
  }
  switch (w_dtype) {
    case LORA_AMD_F32: DISPATCH_AB(f32_t); break;
    case LORA_AMD_F16: DISPATCH_AB(f16_t); break;
    default: DISPATCH_AB(bf16_t); break;
  }
#undef DISPATCH_AB
  return check_launch("lora_amd_merge_batched");
}

extern "C" int lora_amd_merge_set_tuning(int64_t tile_elems, int64_t blocks_per_cu) {
  if (tile_elems > 0) g_merge_tile_elems = tile_elems;
  if (blocks_per_cu > 0) g_merge_blocks_per_cu = blocks_per_cu;
  return LORA_AMD_OK;
}
