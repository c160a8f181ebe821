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

static int64_t g_merge_tile_elems = 16384;  // tuning knob (lora_amd_merge_set_tuning); 16K measured best on MI355X
static int64_t g_merge_blocks_per_cu = 4;
static int g_merge_nt = 1;  // non-temporal W loads/stores: W is streamed exactly once; measured +8 % on the 144-site
                            // UNet set (147.7 -> 135.6 us).  Tuning knob: blocks_per_cu >= 100 selects nt=1, below nt=0

template <class E>
__device__ inline float ld_as_f32(const void *p, int64_t i) {
  return E::to_f(gl(reinterpret_cast<const typename E::storage *>(p))[i]);
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
    if (s.flags & 2) continue;  // handled by merge_co_kernel
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
        gl(wout)[off] = EW::from_f(merge_one<EW, EAB, ROUND>(EW::to_f(gl(win)[off]), p, alpha));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Column-owner kernel (the fast path; flags bit 1).  One block per tile of rows_per_tile x (ct8*8) columns,
// ct8 = power of two dividing K/8.  A thread owns ONE 16-byte column chunk: its r x 8 block of `down` sits in
// registers for the whole tile (loaded once, all loads in flight together), the tile's `up` rows sit in LDS,
// and the thread walks rows slot, slot+nslots, ... with 4 independent 16-byte W loads in flight.  No LDS
// traffic per element except the r floats of the row's `up`, no integer division in the loop.
// ---------------------------------------------------------------------------
constexpr int kMergeMaxSitesLds = 1024;

template <class EW, class EAB, int RT, int ROUND, bool NT>
__global__ __launch_bounds__(kMergeThreads) void merge_co_kernel(
    const lora_amd_merge_site *__restrict__ sites, int n_sites, int64_t total_tiles, float alpha) {
  using SW = typename EW::storage;
  using SAB = typename EAB::storage;
  __shared__ int64_t s_begin[kMergeMaxSitesLds];
  __shared__ __attribute__((aligned(16))) float s_up[kMergeLdsUpFloats];
  const int tid = threadIdx.x;
  const int64_t tile = blockIdx.x;
  int si;
  if (n_sites <= kMergeMaxSitesLds) {
    for (int i = tid; i < n_sites; i += kMergeThreads) s_begin[i] = sites[i].tile_begin;
    __syncthreads();
    int lo = 0, hi = n_sites - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_begin[mid] <= tile) lo = mid; else hi = mid - 1;
    }
    si = lo;
  } else {
    si = find_site(sites, n_sites, tile);
  }
  const lora_amd_merge_site s = sites[si];
  if (!(s.flags & 2)) return;  // handled by the LDS-slab kernel
  const int r = s.r;
  const int ct8 = s.cols_per_tile >> 3;
  const int log_ct8 = __ffs(ct8) - 1;
  const int nslots = kMergeThreads >> log_ct8;
  const int slot = tid >> log_ct8, cl = tid & (ct8 - 1);
  const int64_t tl = tile - s.tile_begin;
  const int tr = (int)(tl / s.tiles_k), tc = (int)(tl - (int64_t)tr * s.tiles_k);
  const int row0 = tr * s.rows_per_tile;
  const int nrows = min(s.rows_per_tile, s.N - row0);
  const int col = (tc * ct8 + cl) * 8;

  // this thread's r x 8 block of `down`
  float fc[RT][8];
  const SAB *dn = reinterpret_cast<const SAB *>(s.down);
  const bool tvec = s.transposed != 0 && sizeof(SAB) == 4 && (r & 3) == 0 && (reinterpret_cast<uintptr_t>(s.down) & 15u) == 0;
  const bool tpo = s.transposed != 0;  // the transposed product: `down` holds the original up [K, r], `up` the original down [r, N]
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    if (j < r) {
      if (tpo) {
        // the 8 columns' ranks are 8 r consecutive factor elements ([K, r] row-major): the r % 4 == 0 f32 case below
        // fetches them as 16-byte loads; anything else element by element
        if (!tvec) {
#pragma unroll
          for (int i = 0; i < 8; ++i) fc[j][i] = EAB::to_f(gl(dn)[(int64_t)(col + i) * r + j]);
        }
      } else {
        load8<EAB>(dn + (int64_t)j * s.K + col, fc[j]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) fc[j][i] = 0.f;
    }
  }
  if (tvec) {
    const float *dnf = reinterpret_cast<const float *>(s.down) + (int64_t)col * r;  // 16-byte aligned: col % 8 == 0, r % 4 == 0
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int q = 0; q < RT / 4; ++q)
        if (q * 4 < r) {
          const float4 v = gl_ld4(dnf + i * r + q * 4);
          fc[q * 4 + 0][i] = v.x; fc[q * 4 + 1][i] = v.y; fc[q * 4 + 2][i] = v.z; fc[q * 4 + 3][i] = v.w;
        }
  }
  // the tile's rows of `up` -> LDS [nrows][RT]
  const SAB *upp = reinterpret_cast<const SAB *>(s.up);
  if (tpo) {  // rank-major source [r, N]: consecutive threads read consecutive rows of one rank
    for (int i = tid; i < nrows * RT; i += kMergeThreads) {
      const int j = i / nrows, rl = i - j * nrows;
      s_up[rl * RT + j] = j < r ? EAB::to_f(gl(upp)[(int64_t)j * s.N + row0 + rl]) : 0.f;
    }
  } else {
    for (int i = tid; i < nrows * RT; i += kMergeThreads) {
      const int rl = i / RT, j = i - rl * RT;
      s_up[i] = j < r ? EAB::to_f(gl(upp)[(int64_t)(row0 + rl) * r + j]) : 0.f;
    }
  }
  __syncthreads();

  const SW *win = reinterpret_cast<const SW *>(s.w_in) + (int64_t)row0 * s.K + col;
  // head-padded output layout (out_heads = d | D << 16): logical column k lives at (k/d)*D + k%d of a (K/d)*D-wide row
  const int hd = s.out_heads & 0xFFFF, hD = s.out_heads >> 16;
  const int64_t ldo = hd ? (int64_t)(s.K / hd) * hD : s.K;
  const int pcol = hd ? (col / hd) * hD + col % hd : col;
  SW *wout = reinterpret_cast<SW *>(s.w_out) + (int64_t)row0 * ldo + pcol;
  constexpr int U = kMergeUnroll;
  for (int rb = slot; rb < nrows; rb += nslots * U) {
    float w[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb + u * nslots;
      if (rl < nrows) { if (NT) load8_nt<EW>(win + (int64_t)rl * s.K, w[u]); else load8<EW>(win + (int64_t)rl * s.K, w[u]); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rl = rb + u * nslots;
      if (rl >= nrows) continue;
      const float *upr = s_up + rl * RT;
      float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const float uj = upr[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = fmaf(uj, fc[j][i], p[i]);
      }
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = merge_one<EW, EAB, ROUND>(w[u][i], p[i], alpha);
      if (NT) store8_nt<EW>(wout + (int64_t)rl * ldo, o); else store8<EW>(wout + (int64_t)rl * ldo, o);
    }
  }
}

template <class EW, class EAB>
static void launch_merge(const lora_amd_merge_site *sites, int n_sites, int64_t total_tiles, float alpha,
                         int rounding, int grid_slab, int n_fast, int rt_fast, hipStream_t st) {
  const bool ref = rounding == LORA_AMD_ROUND_REFERENCE;
  if (n_fast > 0) {
#define CO(RTV, R)                                                                                       \
  do {                                                                                                   \
    if (g_merge_nt)                                                                                      \
      hipLaunchKernelGGL((merge_co_kernel<EW, EAB, RTV, R, true>), dim3((unsigned)total_tiles),          \
                         dim3(kMergeThreads), 0, st, sites, n_sites, total_tiles, alpha);                \
    else                                                                                                 \
      hipLaunchKernelGGL((merge_co_kernel<EW, EAB, RTV, R, false>), dim3((unsigned)total_tiles),         \
                         dim3(kMergeThreads), 0, st, sites, n_sites, total_tiles, alpha);                \
  } while (0)
#define CO_R(RTV) do { if (ref) CO(RTV, LORA_AMD_ROUND_REFERENCE); else CO(RTV, LORA_AMD_ROUND_ONCE); } while (0)
    if (rt_fast <= 4) CO_R(4); else if (rt_fast <= 8) CO_R(8); else CO_R(16);
#undef CO_R
#undef CO
  }
  if (n_fast < n_sites) {
    if (ref)
      hipLaunchKernelGGL((merge_kernel<EW, EAB, LORA_AMD_ROUND_REFERENCE>), dim3(grid_slab), dim3(kMergeThreads), 0,
                         st, sites, n_sites, total_tiles, alpha);
    else
      hipLaunchKernelGGL((merge_kernel<EW, EAB, LORA_AMD_ROUND_ONCE>), dim3(grid_slab), dim3(kMergeThreads), 0, st,
                         sites, n_sites, total_tiles, alpha);
  }
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_merge_plan(lora_amd_merge_site *sites, int32_t n_sites, int32_t w_dtype,
                                   lora_amd_merge_summary *summary) {
  LORA_AMD_CHECK(sites && summary && n_sites >= 0, LORA_AMD_EINVAL, "merge_plan: null argument");
  int n_fast = 0, rt_fast = 4;
  LORA_AMD_CHECK(dtype_ok(w_dtype), LORA_AMD_EINVAL, "merge_plan: bad w_dtype %d", w_dtype);
  int64_t acc = 0;
  for (int i = 0; i < n_sites; ++i) {
    lora_amd_merge_site &s = sites[i];
    LORA_AMD_CHECK(s.N > 0 && s.K > 0, LORA_AMD_EINVAL, "merge_plan: site %d has N=%d K=%d", i, s.N, s.K);
    LORA_AMD_CHECK(s.r >= 1 && s.r <= LORA_AMD_MAX_RANK, LORA_AMD_ERANK,
                   "merge_plan: site %d rank %d outside [1,%d]", i, s.r, LORA_AMD_MAX_RANK);
    const bool aligned = (((uintptr_t)s.w_in | (uintptr_t)s.w_out) & 15u) == 0;
    const bool ab_aligned = (((uintptr_t)s.down) & 31u) == 0;
    int ct8 = 1;
    if (s.K % 8 == 0) {
      const int c8 = s.K / 8;
      while (ct8 < 256 && (c8 % (ct8 * 2)) == 0) ct8 *= 2;
    }
    const int hd = s.out_heads & 0xFFFF, hD = s.out_heads >> 16;
    LORA_AMD_CHECK(s.out_heads == 0 || (hd > 0 && hd % 8 == 0 && hD % 8 == 0 && hD >= hd && s.K % hd == 0),
                   LORA_AMD_EINVAL, "merge_plan: site %d: bad head layout d=%d D=%d for K=%d", i, hd, hD, s.K);
    LORA_AMD_CHECK(s.out_heads == 0 || (aligned && ab_aligned && ct8 >= 4 && s.r <= 16), LORA_AMD_EINVAL,
                   "merge_plan: site %d: a head-padded output needs the column-owner kernel (16-byte rows, rank <= 16)", i);
    LORA_AMD_CHECK(s.transposed == 0 || (aligned && s.K % 8 == 0 && ct8 >= 4 && s.r <= 16), LORA_AMD_EINVAL,
                   "merge_plan: site %d: the transposed product needs the column-owner kernel (16-byte rows, rank <= 16)", i);
    if (aligned && (ab_aligned || s.transposed) && s.K % 8 == 0 && ct8 >= 4 && s.r <= 16) {
      // column-owner tiles: (ct8*8) columns x rows_per_tile rows
      const int cols = ct8 * 8;
      const int rt = s.r <= 4 ? 4 : s.r <= 8 ? 8 : 16;
      int64_t rows = g_merge_tile_elems / cols;
      const int nslots = kMergeThreads / ct8;
      if (rows > kMergeLdsUpFloats / rt) rows = kMergeLdsUpFloats / rt;
      if (rows < nslots) rows = nslots;
      if (rows > s.N) rows = s.N;
      s.cols_per_tile = cols;
      s.rows_per_tile = (int32_t)rows;
      s.tiles_k = s.K / cols;
      s.flags = 3;
      ++n_fast;
      if (rt > rt_fast) rt_fast = rt;
    } else {
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
      s.flags = (s.K % 8 == 0 && aligned) ? 1 : 0;
    }
    s.tile_begin = acc;
    acc += (int64_t)s.tiles_k * ((s.N + s.rows_per_tile - 1) / s.rows_per_tile);
  }
  summary->total_tiles = acc;
  summary->n_fast_sites = n_fast;
  summary->rank_tile_fast = rt_fast;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_merge_batched(const lora_amd_merge_site *sites_dev, int32_t n_sites,
                                      const lora_amd_merge_summary *summary, int32_t w_dtype, int32_t ab_dtype,
                                      float alpha, int32_t rounding, void *stream) {
  LORA_AMD_CHECK(sites_dev && summary && n_sites > 0 && summary->total_tiles > 0, LORA_AMD_EINVAL,
                 "merge: empty site table");
  LORA_AMD_CHECK(dtype_ok(w_dtype) && dtype_ok(ab_dtype), LORA_AMD_EINVAL, "merge: bad dtype");
  LORA_AMD_CHECK(rounding == LORA_AMD_ROUND_REFERENCE || rounding == LORA_AMD_ROUND_ONCE, LORA_AMD_EINVAL,
                 "merge: bad rounding mode %d", rounding);
  LORA_AMD_CHECK(summary->total_tiles < (1ll << 31), LORA_AMD_EINVAL, "merge: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total_tiles = summary->total_tiles;
  const int64_t cap = 256 * g_merge_blocks_per_cu;
  const int grid = (int)(total_tiles < cap ? total_tiles : cap);
  const int n_fast = summary->n_fast_sites, rt_fast = summary->rank_tile_fast;
#define DISPATCH_AB(EW)                                                                                           \
  switch (ab_dtype) {                                                                                             \
    case LORA_AMD_F32: launch_merge<EW, f32_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, n_fast, rt_fast, st); break; \
    case LORA_AMD_F16: launch_merge<EW, f16_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, n_fast, rt_fast, st); break; \
    default: launch_merge<EW, bf16_t>(sites_dev, n_sites, total_tiles, alpha, rounding, grid, n_fast, rt_fast, st); break;          \
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
  if (blocks_per_cu >= 100) { g_merge_nt = 1; blocks_per_cu -= 100; } else if (blocks_per_cu > 0) { g_merge_nt = 0; }
  if (blocks_per_cu > 0) g_merge_blocks_per_cu = blocks_per_cu;
  return LORA_AMD_OK;
}
