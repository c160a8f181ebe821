// K2 on the merged-weight path, matrix-core form: both factor gradients of every Linear adapter in one launch, each
// row of G and X read from HBM exactly ONCE.
//
// replaces: the autograd of lora_diffusion/lora.py:53-58 for lora_up.weight / lora_down.weight
//           (dUp = s G^T (X down^T), dDown = (s G up)^T X) when the forward ran on W + s up down (ops.MergedWeights),
//           and csrc/linear_fused.hip's linear_bwd_factors_self_ragged_kernel for 16-bit activations: that VALU pass
//           reads every row block twice (row-dot phase, column-sum phase; the second read misses the L2: FETCH_SIZE
//           2.0x algorithmic, 0.25 of the byte roof).
//
// One workgroup (4 waves) owns R rows of one site.  With A = the narrower of (X, G) and B = the wider one, fa / fb the
// factor contracted against A's / B's columns (A = X: fa = down, fb = up):
//
//     TA = s A fa^T [R, r]        outB[j, c] = sum_m TA[m, j] B[m, c]       (A = X: T,  outB = dUp partial)
//     TB = s B fb^T [R, r]        outA[j, c] = sum_m TB[m, j] A[m, c]       (        Gt, outA = dDown partial)
//
//   * A's row block [R, Ca] stays RESIDENT in LDS (160 KiB per CU on gfx950: 64 rows x 320 columns = 42 KB, two
//     workgroups per CU; the 1280-wide sites take one workgroup per CU), B streams through a second LDS buffer in column
//     chunks [R, CW]: chunk c + 1 is in flight (registers) while chunk c is consumed.  Per chunk: TB accumulates
//     (phase 1) and outB's columns of the chunk are finished and stored (phase 2, TA is complete by then); after the
//     last chunk TB is complete and outA is computed from the resident block.  Nothing is read twice.
//   * both phases are v_mfma_f32_16x16x32 (the rank padded to the 16 of the tile):
//       phase 1  D[row, j]  += Data[row, 32 cols] . F[j, 32 cols]^T      A-operand = ds_read_b128 of a row, B-operand = a
//                packed factor fragment (lora_amd_factor_pack: 1 KB coalesced per wave, L2-resident);
//       phase 2  D[col, j]  += Data^T[col, 32 rows] . T[32 rows, j]      A-operand = two ds_read_b64_tr_b16 (the LDS
//                transpose read of gfx950: the row-major tile delivered column-major), B-operand = T from LDS.
//     f32 precision is kept by splitting every 16-bit operand that is not data: factor = hi + lo, T = hi + lo (two MFMAs
//     into the same accumulator) — the matrix pipe is < 20 % busy at the HBM rate, so the split is free.
//   * phase 1 is split over the waves by k-step (each packed fragment is fetched by ONE wave and reused for all R / 16
//     row tiles); the four partial [R, 16] blocks meet once in LDS.
//   * LDS rows are padded to pitch = 32 (mod 64) bytes: the ds_read_b128 of 16 rows and the transpose reads of 8 rows
//     x 32 B are both bank-conflict-free (scripts/lds_banks.py checks the lane groups of the microarchitecture guide).
// Algorithmic bytes per site: M (N + K) e (G and X once) + the partial slabs 2 RT 4 (N + K) M / R (written here, read by
// lora_amd_reduce_batched).  HBM-bound: 4 MFMA per KB against ~90 cycles per KB per CU.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace lora_amd {

typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int mu32x2 __attribute__((ext_vector_type(2)));
typedef short ms16x4 __attribute__((ext_vector_type(4)));

template <class E> struct FmMfma;
template <> struct FmMfma<bf16_t> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  __device__ static mf32x4 mma(frag a, frag b, mf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct FmMfma<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  __device__ static mf32x4 mma(frag a, frag b, mf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <class E>
__device__ __forceinline__ typename FmMfma<E>::frag fm_frag(mu32x4 v) {
  union { typename FmMfma<E>::frag f; mu32x4 u; } c;
  c.u = v;
  return c.f;
}

constexpr int kFmThreads = 256;   // 4 waves
constexpr int kFmNPA1 = 10, kFmNPA2 = 20;  // 16-byte pieces per thread of ONE resident block (R * Ca <= 20480 / 40960 elements)
constexpr int kFmNPB = 8;         // ... of one streamed chunk (R * CW <= 16384 elements)
constexpr int kFmMaxNB = 8;       // row blocks one workgroup walks (their partial sums meet in its slab)
constexpr int kFmMaxRT16 = 4;     // row tiles of 16 per block (R <= 64)
constexpr int kFmLdsSmall = 81920, kFmLdsLarge = 163840;  // two workgroups per CU / one

__host__ __device__ inline int fm_pitch(int cols) {  // bytes; smallest p >= 2 cols with p % 64 == 32
  const int b = cols * 2;
  return ((b + 31) / 64) * 64 + 32;
}
__host__ __device__ inline int fm_tpitch(int R) { return R * 2 + 16; }

// chunk index of a head-padded row, branch-free (a branch around the address of a load costs its own s_waitcnt): q = c / hc by
// a 32-bit magic multiply (exact for c < 2^16), hc = 0 (dense) has magic 0 and gives c back
struct FmHeads { uint32_t magic; int hc, hp; };
__device__ __forceinline__ FmHeads fm_heads(int hc, int hp) {
  FmHeads h;
  h.magic = hc ? (uint32_t)((0x100000000ull + hc - 1) / (uint32_t)hc) : 0u;
  h.hc = hc; h.hp = hp;
  return h;
}
__device__ __forceinline__ int fm_hchunk(int c, const FmHeads &h) {
  const int q = (int)__umulhi((uint32_t)c, h.magic);
  return q * h.hp + (c - q * h.hc);
}

// ---------------------------------------------------------------------------------------------------------- factor pack
// f32 masters -> MFMA fragment order in the activation dtype, hi and lo parts:
//   pk[split][c8][jj][e] = part_split(factor(jj, c8 * 8 + e))   jj < r, else 0;   16 x 8 elements = 256 B per c8
// factor(jj, c) = down[jj, c] (FACTOR_RK) or up[c, jj] (FACTOR_KR).  A k-step's fragment (4 consecutive c8) is 1 KB
// contiguous and lane l reads its 16 bytes at l * 16.
template <class E>
__global__ __launch_bounds__(256) void factor_pack_kernel(const lora_amd_pack_site *__restrict__ sites, int n, int64_t total) {
  using S = typename E::storage;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sites[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const lora_amd_pack_site q = sites[lo];
    int64_t p = i - q.begin;                 // piece = (side, c8, jj): down side first
    const int64_t nd = (int64_t)(q.K >> 3) * 16;
    const bool is_up = p >= nd;
    if (is_up) p -= nd;
    const int c8 = (int)(p >> 4), jj = (int)(p & 15);
    const int C = is_up ? q.N : q.K;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (jj < q.r) {
      if (is_up) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gl(q.up)[(int64_t)(c8 * 8 + e) * q.r + jj];
      } else {
        const float *src = q.down + (int64_t)jj * C + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gl(src)[e];
      }
    }
    Chunk8<E> h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h.v[e] = E::from_f(v[e]);
      l.v[e] = E::from_f(v[e] - E::to_f(h.v[e]));
    }
    S *dst = reinterpret_cast<S *>(is_up ? q.pk_up : q.pk_down);
    const int64_t split_stride = (int64_t)(C >> 3) * 128;  // elements per split
    union { Chunk8<E> c; mu32x4 u; } hb, lb;
    hb.c = h; lb.c = l;
    *gl(reinterpret_cast<mu32x4 *>(dst + ((int64_t)c8 * 16 + jj) * 8)) = hb.u;
    *gl(reinterpret_cast<mu32x4 *>(dst + split_stride + ((int64_t)c8 * 16 + jj) * 8)) = lb.u;
  }
}

// ---------------------------------------------------------------------------------------------------------- the pass
template <class E, int NP>
struct FmStage { mu32x4 v[NP]; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + s_barrier: hipcc drains EVERY outstanding
// global load (s_waitcnt vmcnt(0)) before it, which would end the flight of the next chunk's / next row block's loads at
// the first barrier after they were issued — the whole point of this kernel's pipeline is that they stay in flight across
// the barriers until their registers are written to LDS.  All cross-wave communication here goes through LDS
// (lgkmcnt(0) = this wave's ds_write / ds_read have completed); global stores are never read by another wave.
__device__ __forceinline__ void fm_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Pieces (16 bytes) of a [R, c8w] tile, row-major, dealt to the threads with stride 256: piece p0 + tid + i * 256 is
// (row, c).  One division per tile instead of one per piece.
struct FmPieces {
  int row, c, dr, dc, c8w;
  __device__ __forceinline__ FmPieces(int p0, int c8w_) : c8w(c8w_) {
    // an opaque zero: the pieces' (row, c) depend only on the thread and the tile shape, and hipcc otherwise hoists all of
    // them out of the row-block loop and keeps them live (+54 registers: spills at two workgroups per CU)
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    const int p = p0 + threadIdx.x + z;
    row = p / c8w; c = p - row * c8w;
    dr = kFmThreads / c8w; dc = kFmThreads - dr * c8w;
  }
  __device__ __forceinline__ void next() {
    row += dr; c += dc;
    if (c >= c8w) { c -= c8w; ++row; }
  }
};

// Issue the loads of a thread's pieces of the tile whose first column chunk is c8_0 — all in flight.  Rows past the end
// of the matrix (and pieces past the end of the tile) read a valid address and are zeroed.
template <class E, int NP>
__device__ __forceinline__ void fm_issue(FmStage<E, NP> &st, const typename E::storage *data, int64_t ld, int64_t m0,
                                         int nrows, int p0, int c8w, int c8_0, const FmHeads &hd) {
  FmPieces it(p0, c8w);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const bool ok = it.row < nrows;
    const typename E::storage *src = data + (m0 + (ok ? it.row : 0)) * ld + (int64_t)fm_hchunk(c8_0 + it.c, hd) * 8;
    const mu32x4 v = *gl(reinterpret_cast<const mu32x4 *>(src));
    st.v[i] = ok ? v : mu32x4{0u, 0u, 0u, 0u};
    it.next();
  }
}
// nn.Dropout on the branch (lora.py:45, 57): G enters both contractions as mask (.) G.  The keep pattern of the forward is
// regenerated here from (seed, offset) — Philox chunk = 8 consecutive elements of the dense [M, N] output, as every other
// kernel of the library indexes it — and applied as a bit mask when the piece goes to LDS; the 1 / (1 - p) factor is folded
// into the site's scale (both outputs are linear in G).
struct FmDrop { bool on; uint64_t seed, off; uint32_t thr; int64_t row0; int n8, c8_0; };

template <class E, int NP, bool DROP>
__device__ __forceinline__ void fm_write(const FmStage<E, NP> &st, unsigned char *buf, int pitch, int R, int p0, int c8w,
                                         const FmDrop &dr) {
  FmPieces it(p0, c8w);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (it.row < R) {
      mu32x4 v = st.v[i];
      if constexpr (DROP) {  // straight-line: `on` only selects between the mask and all-ones
        uint32_t rr[4];
        Philox ph(dr.seed);
        ph((uint64_t)((dr.row0 + it.row) * dr.n8 + dr.c8_0 + it.c), dr.off, rr);
        const uint32_t all = dr.on ? 0u : 0xFFFFFFFFu;
#pragma unroll
        for (int w = 0; w < 4; ++w)
          v[w] &= all | ((rr[w] & 0xFFFFu) >= dr.thr ? 0x0000FFFFu : 0u) | ((rr[w] >> 16) >= dr.thr ? 0xFFFF0000u : 0u);
      }
      *reinterpret_cast<mu32x4 *>(buf + it.row * pitch + it.c * 16) = v;
    }
    it.next();
  }
}

// The first fragment pair (hi, lo) of a wave's share of a tile's k-steps: issued BEFORE the barrier that publishes the tile,
// so that the L2 trip of the factors is not on the path from "tile in LDS" to "first MFMA".
struct FmFrag2 { mu32x4 h, l; };
template <class E>
__device__ __forceinline__ FmFrag2 fm_first_frags(const typename E::storage *pk, int64_t split_stride, int nks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ks = wave < nks ? wave : 0;
  FmFrag2 f;
  f.h = *gl(reinterpret_cast<const mu32x4 *>(pk + (int64_t)ks * 512 + lane * 8));
  f.l = *gl(reinterpret_cast<const mu32x4 *>(pk + split_stride + (int64_t)ks * 512 + lane * 8));
  return f;
}

// phase 1 of one tile [R, ncols] in LDS: acc[t] += Data[tile t rows, k-step] . (hi + lo of the packed factor), for the
// k-steps ks = wave, wave + 4, ... of the tile.  `pk` points at the fragment of the tile's first column (split 0),
// `split_stride` = elements between the hi and the lo pack, `f0` = fm_first_frags of the same arguments.
template <class E>
__device__ __forceinline__ void fm_phase1(mf32x4 (&acc)[kFmMaxRT16], const unsigned char *buf, int pitch, int nrt, int nks,
                                          const typename E::storage *pk, int64_t split_stride, FmFrag2 f0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char *rowp = buf + (lane & 15) * pitch + (lane >> 4) * 16;
  if (wave >= nks) return;
  mu32x4 fh = f0.h, fl = f0.l;
#pragma unroll 1
  for (int ks = wave; ks < nks; ks += 4) {
    const int kn = ks + 4 < nks ? ks + 4 : ks;  // next fragment in flight while this one is used
    const mu32x4 nh = *gl(reinterpret_cast<const mu32x4 *>(pk + (int64_t)kn * 512 + lane * 8));
    const mu32x4 nl = *gl(reinterpret_cast<const mu32x4 *>(pk + split_stride + (int64_t)kn * 512 + lane * 8));
    const typename FmMfma<E>::frag bh = fm_frag<E>(fh), bl = fm_frag<E>(fl);
#pragma unroll
    for (int t = 0; t < kFmMaxRT16; ++t) {
      if (t < nrt) {
        const typename FmMfma<E>::frag a =
            fm_frag<E>(*reinterpret_cast<const mu32x4 *>(rowp + t * 16 * pitch + ks * 64));
        acc[t] = FmMfma<E>::mma(a, bh, acc[t]);
        acc[t] = FmMfma<E>::mma(a, bl, acc[t]);
      }
    }
    fh = nh; fl = nl;
  }
}

// The four waves' partial [R, 16] blocks -> T = s * sum, split hi / lo, stored [split][jj][position] so that phase 2
// reads a lane's 8 contraction rows with one 16-byte load.  Position of row ro (0..31) inside its 32-row k-step:
//   ro < 16: 8 (ro / 4) + ro % 4        ro >= 16: 8 ((ro - 16) / 4) + 4 + ro % 4
// (the rows a transpose read hands to lane group q: 4q..4q+3 with the first read, 16+4q..16+4q+3 with the second).
template <class E>
__device__ __forceinline__ void fm_combine_put(const mf32x4 (&acc)[kFmMaxRT16], float *scratch, int nrt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < kFmMaxRT16; ++t) {
    if (t < nrt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) scratch[((wave * nrt + t) * 4 + g) * 64 + lane] = acc[t][g];
    }
  }
}
// ... after a barrier: rows >= nrows (past the end of the matrix: the engine kernel's tiles repeat the last row there)
// give T = 0
template <class E>
__device__ __forceinline__ void fm_combine_get(const float *scratch, unsigned char *tt, int nrt, int R, float scale, int nrows) {
  using S = typename E::storage;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tp = fm_tpitch(R);
  for (int t = wave; t < nrt; t += 4) {
    union { S s[4]; mu32x2 u; } h, l;
    const int jj = lane & 15, q = lane >> 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += scratch[((w * nrt + t) * 4 + g) * 64 + lane];
      v = (t * 16 + 4 * q + g) < nrows ? v * scale : 0.f;
      h.s[g] = E::from_f(v);
      l.s[g] = E::from_f(v - E::to_f(h.s[g]));
    }
    const int pos = (t >> 1) * 32 + 8 * q + 4 * (t & 1);
    *reinterpret_cast<mu32x2 *>(tt + jj * tp + pos * 2) = h.u;
    *reinterpret_cast<mu32x2 *>(tt + (16 + jj) * tp + pos * 2) = l.u;
  }
}
template <class E>
__device__ __forceinline__ void fm_combine(const mf32x4 (&acc)[kFmMaxRT16], float *scratch, unsigned char *tt, int nrt,
                                           int R, float scale, int nrows) {
  fm_combine_put<E>(acc, scratch, nrt);
  fm_barrier();
  fm_combine_get<E>(scratch, tt, nrt, R, scale, nrows);
}

// 8 contraction rows x 1 column of a row-major LDS tile as an MFMA operand: lane (q = lane / 16, i = lane % 16) gets
// rows {32 ks + 4q + e, e < 4} and {32 ks + 16 + 4q + e} of column c0 + i: two ds_read_b64_tr_b16 (each 16-lane
// group reads a [4 rows][16 columns] block; lane s of the group supplies the address of row s / 4, columns 4 (s % 4) ..).
template <class E>
__device__ __forceinline__ typename FmMfma<E>::frag fm_colfrag(const unsigned char *buf, int pitch, int ks, int c0) {
  const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
  const unsigned char *p = buf + (ks * 32 + 4 * q + (i >> 2)) * pitch + (c0 + 4 * (i & 3)) * 2;
  union { ms16x4 h[2]; mu32x4 u; } r;
  r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(p));
  r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(p + 16 * pitch));
  return fm_frag<E>(r.u);
}

// phase 2 of one tile [R, ncols] in LDS: out[jj][col0 + c] (+)= sum_rows T[row, jj] Data[row, c] for the tile's columns;
// column tiles of 16 are dealt to the waves.  `tf` = the T fragments (hi, lo per 32-row k-step) in registers.
// `accumulate`: the workgroup's earlier row blocks already left their sums in `out` (its own slab: the same lane wrote
// the same 16 bytes; read back past the L1 with a non-temporal load, one column tile ahead of its use).
template <class E>
__device__ __forceinline__ void fm_phase2(const unsigned char *buf, int pitch, int nk2, int ncols,
                                          const mu32x4 (&tf)[kFmMaxRT16], float *out, int64_t ldo, int RT, bool accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jj = lane & 15, q = lane >> 4;
  const int nct = ncols >> 4;
  const bool owner = jj < RT;
  float *base = out + (owner ? jj : 0) * ldo + 4 * q;
  mf32x4 old = {0.f, 0.f, 0.f, 0.f};
  if (accumulate && owner && wave < nct) old = __builtin_nontemporal_load(gl(reinterpret_cast<const mf32x4 *>(base + wave * 16)));
#pragma unroll 1
  for (int ct = wave; ct < nct; ct += 4) {
    mf32x4 nxt = {0.f, 0.f, 0.f, 0.f};
    if (accumulate && owner && ct + 4 < nct) nxt = __builtin_nontemporal_load(gl(reinterpret_cast<const mf32x4 *>(base + (ct + 4) * 16)));
    mf32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < kFmMaxRT16 / 2; ++k2) {
      if (k2 < nk2) {
        const typename FmMfma<E>::frag a = fm_colfrag<E>(buf, pitch, k2, ct * 16);
        d = FmMfma<E>::mma(a, fm_frag<E>(tf[2 * k2]), d);
        d = FmMfma<E>::mma(a, fm_frag<E>(tf[2 * k2 + 1]), d);
      }
    }
    if (owner) *gl(reinterpret_cast<mf32x4 *>(base + ct * 16)) = d + old;
    old = nxt;
  }
}

template <class E>
__device__ __forceinline__ void fm_load_tfrags(mu32x4 (&tf)[kFmMaxRT16], const unsigned char *tt, int R, int nk2) {
  const int lane = threadIdx.x & 63, jj = lane & 15, q = lane >> 4;
  const int tp = fm_tpitch(R);
#pragma unroll
  for (int k2 = 0; k2 < kFmMaxRT16 / 2; ++k2) {
    if (k2 < nk2) {
      tf[2 * k2] = *reinterpret_cast<const mu32x4 *>(tt + jj * tp + (k2 * 32 + 8 * q) * 2);
      tf[2 * k2 + 1] = *reinterpret_cast<const mu32x4 *>(tt + (16 + jj) * tp + (k2 * 32 + 8 * q) * 2);
    }
  }
}

template <class E, int LDSB, bool DROP>
__global__ __launch_bounds__(kFmThreads, LDSB <= 81920 ? 2 : 1) void factors_mfma_kernel(const lora_amd_fm_site *__restrict__ sites, int n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
  using S = typename E::storage;
  constexpr int NPA = LDSB <= 81920 ? kFmNPA1 : kFmNPA2;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (sites[mid].block_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_fm_site q = sites[lo];
  const int64_t sb_idx = (int64_t)blockIdx.x - q.block_begin;   // this workgroup's slab = its run of row blocks
  const int R = q.rows_per_block, nrt = R >> 4, nk2 = R >> 5;
  const int64_t nrb = (q.M + R - 1) / R;
  const int64_t rb0 = sb_idx * q.blocks_per_wg;
  const int nblk = (int)min((int64_t)q.blocks_per_wg, nrb - rb0);
  const bool ax = q.resident_is_x != 0;
  const int RT = q.r <= 4 ? 4 : q.r <= 8 ? 8 : 16;
  // A = resident operand, B = streamed operand
  const S *da = reinterpret_cast<const S *>(ax ? q.x : q.g), *db = reinterpret_cast<const S *>(ax ? q.g : q.x);
  const int64_t lda = ax ? q.ldx : q.ldg, ldb = ax ? q.ldg : q.ldx;
  const int Ca = ax ? q.K : q.N, Cb = ax ? q.N : q.K;
  const FmHeads hda = fm_heads((ax ? q.x_head_dim : q.g_head_dim) >> 3, (ax ? q.x_head_pad : q.g_head_pad) >> 3);
  const FmHeads hdb = fm_heads((ax ? q.g_head_dim : q.x_head_dim) >> 3, (ax ? q.g_head_pad : q.x_head_pad) >> 3);
  const S *pka = reinterpret_cast<const S *>(ax ? q.pk_down : q.pk_up), *pkb = reinterpret_cast<const S *>(ax ? q.pk_up : q.pk_down);
  float *outa = (ax ? q.down_part : q.up_part) + sb_idx * RT * (int64_t)Ca;   // = sum over the blocks of TB^T A
  float *outb = (ax ? q.up_part : q.down_part) + sb_idx * RT * (int64_t)Cb;   // = ... TA^T B
  const int pa = q.pitch_a, pb = q.pitch_b, CW = q.cw, nch = q.nchunk;
  unsigned char *bufA = lds, *bufB = bufA + R * pa;
  unsigned char *ttA = bufB + R * pb, *ttB = ttA + 32 * fm_tpitch(R);
  float *scratch = reinterpret_cast<float *>(bufB);  // [4 waves][nrt][4][64] f32 <= R * pitch_b (planner)
  const int c8a = Ca >> 3, cw0 = min(CW, Cb);
  const int64_t splita = (int64_t)c8a * 128, splitb = (int64_t)(Cb >> 3) * 128;
  // dropout: G is the streamed operand when X is resident, else the resident one
  const bool drop = DROP && q.dropout_p > 0.f;
  FmDrop dra, drb;
  dra.on = drop && !ax; drb.on = drop && ax;
  dra.seed = drb.seed = q.seed;
  dra.off = drb.off = drop ? dropout_offset(q.offset, q.offset_dev) : 0;
  dra.thr = drb.thr = (uint32_t)(q.dropout_p * 65536.0f + 0.5f);
  dra.n8 = drb.n8 = q.N >> 3;
  dra.c8_0 = 0;

  // ---- the first block's resident rows and first chunk: in flight before the first wait.  From here on the NEXT block's
  // resident rows (sa) and the next chunk (sb) are always in flight while the current ones are consumed.
  FmStage<E, NPA> sa;
  FmStage<E, kFmNPB> sb;
  {
    const int64_t m0 = rb0 * R;
    const int nrows = (int)min((int64_t)R, q.M - m0);
    fm_issue<E, NPA>(sa, da, lda, m0, nrows, 0, c8a, 0, hda);
    fm_issue<E, kFmNPB>(sb, db, ldb, m0, nrows, 0, cw0 >> 3, 0, hdb);
  }
  mf32x4 acc[kFmMaxRT16];
  mu32x4 tf[kFmMaxRT16];
#pragma unroll 1
  for (int blk = 0; blk < nblk; ++blk) {
    const int64_t m0 = (rb0 + blk) * R, m1 = m0 + R;
    const int nrows = (int)min((int64_t)R, q.M - m0);
    const int nrows1 = (int)min((int64_t)R, q.M - m1);   // of the next block (if any)
    const bool more = blk + 1 < nblk;
    const bool accum = blk > 0;
    FmFrag2 f0 = fm_first_frags<E>(pka, splita, Ca >> 5);
    dra.row0 = drb.row0 = m0;
    fm_write<E, NPA, DROP>(sa, bufA, pa, R, 0, c8a, dra);
    fm_barrier();
    if (more) fm_issue<E, NPA>(sa, da, lda, m1, nrows1, 0, c8a, 0, hda);
    // ---- TA
#pragma unroll
    for (int t = 0; t < kFmMaxRT16; ++t) acc[t] = mf32x4{0.f, 0.f, 0.f, 0.f};
    fm_phase1<E>(acc, bufA, pa, nrt, Ca >> 5, pka, splita, f0);
    fm_combine<E>(acc, scratch, ttA, nrt, R, q.scale, nrows);
    f0 = fm_first_frags<E>(pkb, splitb, cw0 >> 5);
    fm_barrier();  // TA visible; the scratch (= chunk buffer) is free
    fm_load_tfrags<E>(tf, ttA, R, nk2);
    // ---- B in column chunks
#pragma unroll
    for (int t = 0; t < kFmMaxRT16; ++t) acc[t] = mf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
      const int col0 = c * CW, cw = min(CW, Cb - col0);
      drb.c8_0 = col0 >> 3;
      fm_write<E, kFmNPB, DROP>(sb, bufB, pb, R, 0, cw >> 3, drb);
      fm_barrier();
      if (c + 1 < nch) {
        const int col1 = col0 + CW, cw1 = min(CW, Cb - col1);
        fm_issue<E, kFmNPB>(sb, db, ldb, m0, nrows, 0, cw1 >> 3, col1 >> 3, hdb);
      } else if (more) {
        fm_issue<E, kFmNPB>(sb, db, ldb, m1, nrows1, 0, cw0 >> 3, 0, hdb);
      }
      fm_phase1<E>(acc, bufB, pb, nrt, cw >> 5, pkb + (int64_t)(col0 >> 3) * 128, splitb, f0);
      if (c + 1 < nch) {
        const int col1 = col0 + CW;
        f0 = fm_first_frags<E>(pkb + (int64_t)(col1 >> 3) * 128, splitb, min(CW, Cb - col1) >> 5);
      }
      fm_phase2<E>(bufB, pb, nk2, cw, tf, outb + col0, Cb, RT, accum);
      fm_barrier();  // the chunk buffer is free
    }
    // ---- TB, then the resident block's column sums
    fm_combine<E>(acc, scratch, ttB, nrt, R, q.scale, nrows);
    fm_barrier();
    fm_load_tfrags<E>(tf, ttB, R, nk2);
    fm_phase2<E>(bufA, pa, nk2, Ca, tf, outa, Ca, RT, accum);
    fm_barrier();  // the resident buffer is free for the next block
  }
}

// ---- whole-step fragment sets (the engine's consumers) --------------------------------------------------------------
// Every dependent trip to L2 / HBM on a consumer wave's path costs 2-3 us while the chip streams (measured: with the
// fragments fetched inside phase 1 and the slab read back in phase 2 a 64-row block took ~28 us in EVERY variant of this
// pass: eight to ten such trips, not bytes).  The engine's consumers therefore touch global memory only through loads that
// were issued a whole pipeline step earlier: a wave's packed fragments of ALL its k-steps of the next tile are loaded into
// a register set while the current tile is processed.
constexpr int kFmNF = 5;  // k-steps per wave and tile (tiles of <= 640 columns)
struct FmFragSet { mu32x4 h[kFmNF], l[kFmNF]; };
template <class E>
__device__ __forceinline__ void fm_load_fragset(FmFragSet &f, const typename E::storage *pk, int64_t split_stride, int nks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kFmNF; ++i) {
    const int ks = wave + 4 * i;
    const int kc = ks < nks ? ks : 0;  // straight-line: surplus slots re-read fragment 0 (never used)
    f.h[i] = *gl(reinterpret_cast<const mu32x4 *>(pk + (int64_t)kc * 512 + lane * 8));
    f.l[i] = *gl(reinterpret_cast<const mu32x4 *>(pk + split_stride + (int64_t)kc * 512 + lane * 8));
  }
}
template <class E>
__device__ __forceinline__ void fm_phase1_set(mf32x4 (&acc)[kFmMaxRT16], const unsigned char *buf, int pitch, int nrt, int nks,
                                              const FmFragSet &f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char *rowp = buf + (lane & 15) * pitch + (lane >> 4) * 16;
#pragma unroll
  for (int i = 0; i < kFmNF; ++i) {
    const int ks = wave + 4 * i;
    if (ks < nks) {
      const typename FmMfma<E>::frag bh = fm_frag<E>(f.h[i]), bl = fm_frag<E>(f.l[i]);
#pragma unroll
      for (int t = 0; t < kFmMaxRT16; ++t) {
        if (t < nrt) {
          const typename FmMfma<E>::frag a = fm_frag<E>(*reinterpret_cast<const mu32x4 *>(rowp + t * 16 * pitch + ks * 64));
          acc[t] = FmMfma<E>::mma(a, bh, acc[t]);
          acc[t] = FmMfma<E>::mma(a, bl, acc[t]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------- the engine
// The same pass with the HBM stream decoupled from the arithmetic: one workgroup per CU = four consumer waves (phase 1 /
// combine / phase 2 exactly as above) + ONE loader wave that does nothing but `global_load_lds_dwordx4` (16 bytes per lane
// straight into LDS, 1 KB per instruction, no registers) and runs ahead of the consumers.  Why a separate wave: every wave
// has ONE in-order vmcnt counter, so in the register-staged kernel a wait for a packed factor fragment (issued late) also
// waits for every prefetched tile issued before it — the prefetch can never stay in flight across the fragment loads.  The
// loader wave's counter sees only tile loads; the consumers' only fragments and slab accesses.
//   LDS (<= 160 KiB): resident block x 2 (block i + 1 lands while block i is consumed; x 1 for the 1280-wide sites),
//   a 2-slot ring of column chunks of the streamed operand, the combine scratch, the two T images.
//   Barrier protocol per row block (all five waves execute the same s_barrier sequence):
//     C1 resident block landed | C2 phase-1 partials in scratch | C3 T image visible |
//     per chunk: C4 chunk landed, C5 chunk consumed (its slot may be overwritten) | C6, C7 as C2, C3 | C8 resident consumed
//   Loader between C4(c) and C5(c): issues the NEXT chunk (of this block, or chunk 0 of the next) into the slot freed by
//   C5(c - 1), then 1 / nchunk of the next resident block.  Before C4(c) it waits vmcnt(n) with n = the instructions issued
//   after chunk c (that part of the next resident block): the chunk has landed, the part stays in flight.
//   Tiles are written lane-linearly (LDS address = base + 1024 k + 16 lane): the padded row pitch is produced by
//   computing each lane's SOURCE address from its LDS offset (pad bytes and rows past the end of the matrix re-read a valid
//   row; the combine step zeroes T for those rows, so they contribute nothing).
constexpr int kFeThreads = 320;

// LORA_AMD_FM_TRACE (scripts/fm_trace.py): cycle stamps of one workgroup's barriers — consumer wave 0 and the loader wave —
// into a caller buffer passed through `offset_dev` of site 0.  A diagnostic: TRACE = false compiles it out.
template <bool TRACE>
__device__ __forceinline__ void fe_stamp(unsigned long long *tr, int &idx, int tag) {
  if constexpr (TRACE) {
    if (tr != nullptr && (threadIdx.x & 63) == 0 && idx < 250) {
      tr[idx * 2] = (unsigned long long)tag;
      tr[idx * 2 + 1] = clock64();
      ++idx;
    }
  }
}

__device__ __forceinline__ void fe_glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                   (void __attribute__((address_space(3))) *)lds_wave_base, 16, 0, 0);
}
// s_waitcnt vmcnt(n) for a run-time n (0..63): the operand is an immediate
__device__ __forceinline__ void fe_wait_vmcnt(int n) {
#define FE_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    FE_W(0) FE_W(1) FE_W(2) FE_W(3) FE_W(4) FE_W(5) FE_W(6) FE_W(7) FE_W(8) FE_W(9) FE_W(10) FE_W(11) FE_W(12) FE_W(13)
    FE_W(14) FE_W(15) FE_W(16) FE_W(17) FE_W(18) FE_W(19) FE_W(20) FE_W(21) FE_W(22) FE_W(23) FE_W(24) FE_W(25) FE_W(26)
    FE_W(27) FE_W(28) FE_W(29) FE_W(30) FE_W(31) FE_W(32) FE_W(33) FE_W(34) FE_W(35) FE_W(36) FE_W(37) FE_W(38) FE_W(39)
    FE_W(40) FE_W(41) FE_W(42) FE_W(43) FE_W(44) FE_W(45) FE_W(46) FE_W(47) FE_W(48) FE_W(49) FE_W(50) FE_W(51) FE_W(52)
    FE_W(53) FE_W(54) FE_W(55) FE_W(56) FE_W(57) FE_W(58) FE_W(59) FE_W(60) FE_W(61) FE_W(62)
    default: break;  // 63 or more: nothing to wait for (the counter saturates at 63)
  }
#undef FE_W
}

// One tile [R rows][pitch bytes] of LDS, instructions [k0, k1): lane's LDS offset o = 1024 k + 16 lane -> (row, byte in row)
// -> its source piece.  Columns: `ncol8` 16-byte pieces starting at piece c8_0 of the (possibly head-padded) row.
template <class E>
__device__ __forceinline__ void fe_issue(unsigned char *tile, int pitch, uint32_t pitch_magic, int R, int k0, int k1,
                                         const typename E::storage *data, int64_t ld, int64_t m0, int nrows, int ncol8, int c8_0,
                                         const FmHeads &hd) {
  const int lane = threadIdx.x & 63;
  const int total = R * pitch;
#pragma unroll 1
  for (int k = k0; k < k1; ++k) {
    const int o = k * 1024 + lane * 16;
    if (o < total) {
      const int row = (int)__umulhi((uint32_t)o, pitch_magic);
      const int cb = o - row * pitch;
      const int c8 = (cb >> 4) < ncol8 ? (cb >> 4) : 0;
      const int rowc = row < nrows ? row : nrows - 1;
      fe_glds16(data + (m0 + rowc) * ld + (int64_t)fm_hchunk(c8_0 + c8, hd) * 8, tile + k * 1024);
    }
  }
}

// The loader's inner loop must be SHORT: measured with the per-instruction address arithmetic above (two magic divisions,
// a 64-bit row multiply, the exec mask: ~50 instructions, several quarter-rate) the loader wave needed ~27 us per 82 KB row
// block and the whole engine ran at 0.18 of the byte roof — the loader, not HBM, was the bottleneck.  A lane's piece of
// instruction k is the same in every full row block: its element offset from the block's first row is computed ONCE per
// workgroup into registers; issuing is then one 64-bit add + the DMA per instruction.  (Rows past the end of the matrix —
// the last block of a site only — and chunks that do not start on a head boundary take the slow path above.)
template <int NK>
struct FeOffs { uint32_t o[NK]; };

template <int NK>
__device__ __forceinline__ void fe_offsets(FeOffs<NK> &f, int pitch, uint32_t pitch_magic, int R, int64_t ld, int ncol8,
                                           const FmHeads &hd) {
  const int lane = threadIdx.x & 63;
  const int total = R * pitch;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int o = k * 1024 + lane * 16;
    const int row = (int)__umulhi((uint32_t)o, pitch_magic);
    const int cb = o - row * pitch;
    const int c8 = (cb >> 4) < ncol8 ? (cb >> 4) : 0;
    // lanes past the end of the tile (last instruction only) re-read piece 0: their 16 bytes land in the region's 1 KB
    // rounding, which nothing reads — no per-lane predicate in the issue loop
    f.o[k] = o < total ? (uint32_t)(row * (int)ld + fm_hchunk(c8, hd) * 8) * (uint32_t)sizeof(uint16_t) : 0u;  // bytes
  }
}
template <class E, int NK>
__device__ __forceinline__ void fe_issue_fast(unsigned char *tile, const FeOffs<NK> &f, int k0, int k1,
                                              const typename E::storage *base) {
  k0 = __builtin_amdgcn_readfirstlane(k0);  // wave-uniform by construction: keep the loop control on the scalar unit
  k1 = __builtin_amdgcn_readfirstlane(k1);
  // the block's base address as a SCALAR pair + the lane's 32-bit byte offset: the saddr form of global_load_lds (no 64-bit
  // per-lane address arithmetic, the offset table stays 32-bit)
  const uint64_t b64 = reinterpret_cast<uint64_t>(base);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
  const char *sbase = reinterpret_cast<const char *>(sb);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    if (k >= k0 && k < k1) fe_glds16(sbase + f.o[k], tile + k * 1024);
  }
}

constexpr int kFeNKA = 44, kFeNKB = 32;  // most DMA instructions of a resident block / of a chunk (planner): their offsets
                                         // live in the loader's registers (5 waves per workgroup: 256 registers per wave)

// The loader wave's program, kept out of line: its register file (the two offset tables) is allocated apart from the
// consumers' accumulators (inlined into the kernel the two lived side by side: 256 registers + spills).
template <class E, bool TRACE>
__device__ __attribute__((noinline)) void fe_loader(const lora_amd_fm_site &q, unsigned char *lds, int64_t sb_idx,
                                                    unsigned long long *tr) {
  int ti = 0;
  using S = typename E::storage;
  const int R = q.rows_per_block;
  const int64_t nrb = (q.M + R - 1) / R;
  const int64_t rb0 = sb_idx * q.blocks_per_wg;
  const int nblk = (int)min((int64_t)q.blocks_per_wg, nrb - rb0);
  const bool ax = q.resident_is_x != 0;
  const S *da = reinterpret_cast<const S *>(ax ? q.x : q.g), *db = reinterpret_cast<const S *>(ax ? q.g : q.x);
  const int64_t lda = ax ? q.ldx : q.ldg, ldb = ax ? q.ldg : q.ldx;
  const int Ca = ax ? q.K : q.N, Cb = ax ? q.N : q.K;
  const FmHeads hda = fm_heads((ax ? q.x_head_dim : q.g_head_dim) >> 3, (ax ? q.x_head_pad : q.g_head_pad) >> 3);
  const FmHeads hdb = fm_heads((ax ? q.g_head_dim : q.x_head_dim) >> 3, (ax ? q.g_head_pad : q.x_head_pad) >> 3);
  const int pa = q.pitch_a, pb = q.pitch_b, CW = q.cw, nch = q.nchunk;
  const int szA = (R * pa + 1023) & ~1023, szB = (R * pb + 1023) & ~1023, nbufA = q.a_bufs;
  unsigned char *bufA0 = lds, *ring = lds + nbufA * szA;
  const int nA = (R * pa + 1023) >> 10, nApart = (nA + nch - 1) / nch, c8a = Ca >> 3;
  const uint32_t mga = (uint32_t)((0x100000000ull + pa - 1) / (uint32_t)pa), mgb = (uint32_t)((0x100000000ull + pb - 1) / (uint32_t)pb);
    const int nB = (R * pb + 1023) >> 10;
    FeOffs<kFeNKA> offa;
    FeOffs<kFeNKB> offb;
    fe_offsets<kFeNKA>(offa, pa, mga, R, lda, c8a, hda);
    fe_offsets<kFeNKB>(offb, pb, mgb, R, ldb, CW >> 3, hdb);
    // a chunk's pieces are the first chunk's shifted by the chunk's first column — if the chunk starts on a head boundary
    // (or the rows are dense); the last chunk may be narrower than CW: its surplus columns would read past the row
    const bool bfast = (hdb.hc == 0 || (CW >> 3) % hdb.hc == 0) && Cb % CW == 0;
    auto issue_a = [&](unsigned char *tile, int k0, int k1, int64_t m0, int nrows) {
      if (nrows == R) fe_issue_fast<E, kFeNKA>(tile, offa, k0, k1, da + m0 * lda);
      else fe_issue<E>(tile, pa, mga, R, k0, k1, da, lda, m0, nrows, c8a, 0, hda);
    };
    auto issue_b = [&](unsigned char *tile, int64_t m0, int nrows, int col0, int cw) {
      if (nrows == R && bfast) fe_issue_fast<E, kFeNKB>(tile, offb, 0, nB, db + m0 * ldb + (int64_t)fm_hchunk(col0 >> 3, hdb) * 8);
      else fe_issue<E>(tile, pb, mgb, R, 0, nB, db, ldb, m0, nrows, cw >> 3, col0 >> 3, hdb);
    };
    {
      const int64_t m0 = rb0 * R;
      const int nrows = (int)min((int64_t)R, q.M - m0);
      issue_a(bufA0, 0, nA, m0, nrows);
      issue_b(ring, m0, nrows, 0, min(CW, Cb));
      fe_wait_vmcnt(min(nB, 63));  // the resident block has landed; chunk 0 may still fly
    }
#pragma unroll 1
    for (int blk = 0; blk < nblk; ++blk) {
      const int64_t m0 = (rb0 + blk) * R, m1 = m0 + R;
      const int nrows = (int)min((int64_t)R, q.M - m0);
      const int nrows1 = (int)min((int64_t)R, q.M - m1);
      const bool more = blk + 1 < nblk;
      unsigned char *bufAn = bufA0 + (nbufA == 2 ? ((blk + 1) & 1) * szA : 0);
      fe_stamp<TRACE>(tr, ti, 100);
      fm_barrier();  // C1
      fm_barrier();  // C2
      fm_barrier();  // C3
      fe_stamp<TRACE>(tr, ti, 103);
      int newer = 0;   // instructions issued after the chunk the consumers wait for next
#pragma unroll 1
      for (int c = 0; c < nch; ++c) {
        fe_stamp<TRACE>(tr, ti, 110);
        fe_wait_vmcnt(min(newer, 63));
        fe_stamp<TRACE>(tr, ti, 111);
        fm_barrier();  // C4(c): chunk c has landed
        fe_stamp<TRACE>(tr, ti, 104);
        // ring slot = running chunk index & 1; the slot of the previous chunk (= of the next one) is free since its C5:
        // the next tile of the stream goes there
        unsigned char *nslot = ring + ((blk * nch + c + 1) & 1) * szB;
        if (c + 1 < nch) {
          const int col1 = (c + 1) * CW;
          issue_b(nslot, m0, nrows, col1, min(CW, Cb - col1));
        } else if (more) {
          issue_b(nslot, m1, nrows1, 0, min(CW, Cb));
        }
        newer = 0;
        if (more && nbufA == 2) {  // this step's share of the next resident block
          const int k0 = c * nApart, k1 = min(nA, k0 + nApart);
          issue_a(bufAn, k0, k1, m1, nrows1);
          newer = max(k1 - k0, 0);
        }
        fe_stamp<TRACE>(tr, ti, 112);
        fm_barrier();  // C5(c)
        fe_stamp<TRACE>(tr, ti, 105);
      }
      fm_barrier();  // C6
      fm_barrier();  // C7
      fm_barrier();  // C8: the resident buffer of block blk is free
      fe_stamp<TRACE>(tr, ti, 108);
      if (more && nbufA == 1) issue_a(bufA0, 0, nA, m1, nrows1);
      // the next resident block must have landed before C1; the next block's chunk 0 was issued BEFORE its last share
      // (nbufA == 2) or before the whole block (nbufA == 1), so it has landed too — its first C4 waits for nothing
      if (more) fe_wait_vmcnt(0);
      fe_stamp<TRACE>(tr, ti, 109);
    }
}

template <class E, bool TRACE>
__global__ __launch_bounds__(kFeThreads, 1) void factors_mfma_engine_kernel(const lora_amd_fm_site *__restrict__ sites, int n) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kFmLdsLarge];
  using S = typename E::storage;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (sites[mid].block_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const lora_amd_fm_site q = sites[lo];
  const int64_t sb_idx = (int64_t)blockIdx.x - q.block_begin;
  const int R = q.rows_per_block, nrt = R >> 4, nk2 = R >> 5;
  const int64_t nrb = (q.M + R - 1) / R;
  const int64_t rb0 = sb_idx * q.blocks_per_wg;
  const int nblk = (int)min((int64_t)q.blocks_per_wg, nrb - rb0);
  const bool ax = q.resident_is_x != 0;
  const int RT = q.r <= 4 ? 4 : q.r <= 8 ? 8 : 16;
  const int Ca = ax ? q.K : q.N, Cb = ax ? q.N : q.K;
  const int pa = q.pitch_a, pb = q.pitch_b, CW = q.cw, nch = q.nchunk;
  // LDS carve (every region starts on a 1 KB boundary: DMA instructions never straddle two regions)
  const int szA = (R * pa + 1023) & ~1023, szB = (R * pb + 1023) & ~1023, nbufA = q.a_bufs;
  unsigned char *bufA0 = lds, *ring = lds + nbufA * szA;
  float *scratch = reinterpret_cast<float *>(ring + 2 * szB);
  unsigned char *ttA = reinterpret_cast<unsigned char *>(scratch) + R * 256, *ttB = ttA + 32 * fm_tpitch(R);
  const int c8a = Ca >> 3;
  const int wave = threadIdx.x >> 6;

  unsigned long long *tr = nullptr;
  if constexpr (TRACE) {  // site 0's offset_dev carries the trace buffer; the traced workgroup = a middle one of the grid
    if (blockIdx.x == gridDim.x / 2) tr = reinterpret_cast<unsigned long long *>(const_cast<uint64_t *>(sites[0].offset_dev));
  }
  if (wave == 4) {  // the loader wave
    fe_loader<E, TRACE>(q, lds, sb_idx, tr != nullptr ? tr + 512 : nullptr);
    return;
  }
  if (wave != 0) tr = nullptr;
  int ti = 0;

  // ==================================================================== consumer waves (threads 0..255)
  const S *pka = reinterpret_cast<const S *>(ax ? q.pk_down : q.pk_up), *pkb = reinterpret_cast<const S *>(ax ? q.pk_up : q.pk_down);
  const int64_t splita = (int64_t)c8a * 128, splitb = (int64_t)(Cb >> 3) * 128;
  const int cw0 = min(CW, Cb);
  mf32x4 acc[kFmMaxRT16];
  mu32x4 tf[kFmMaxRT16];
  FmFragSet fcur;  // the fragments of the next tile of the stream: reloaded as soon as phase 1 has consumed them
  fm_load_fragset<E>(fcur, pka, splita, Ca >> 5);
#pragma unroll 1
  for (int blk = 0; blk < nblk; ++blk) {
    const int64_t rb = rb0 + blk;
    const int64_t m0 = rb * R;
    const int nrows = (int)min((int64_t)R, q.M - m0);
    // one partial slab per ROW BLOCK (no read-modify-write of a shared slab: that read is a trip to L2 per column tile)
    float *outa = (ax ? q.down_part : q.up_part) + rb * RT * (int64_t)Ca;
    float *outb = (ax ? q.up_part : q.down_part) + rb * RT * (int64_t)Cb;
    const unsigned char *bufA = bufA0 + (nbufA == 2 ? (blk & 1) * szA : 0);
    fe_stamp<TRACE>(tr, ti, 0);
    fm_barrier();  // C1
    fe_stamp<TRACE>(tr, ti, 1);
#pragma unroll
    for (int t = 0; t < kFmMaxRT16; ++t) acc[t] = mf32x4{0.f, 0.f, 0.f, 0.f};
    fm_phase1_set<E>(acc, bufA, pa, nrt, Ca >> 5, fcur);        // waits for fcur (issued a step ago), then ...
    fm_load_fragset<E>(fcur, pkb, splitb, cw0 >> 5);             // ... chunk 0's fragments start their trip
    fm_combine_put<E>(acc, scratch, nrt);
    fe_stamp<TRACE>(tr, ti, 20);
    fm_barrier();  // C2
    fe_stamp<TRACE>(tr, ti, 2);
    fm_combine_get<E>(scratch, ttA, nrt, R, q.scale, nrows);
    fe_stamp<TRACE>(tr, ti, 30);
    fm_barrier();  // C3
    fe_stamp<TRACE>(tr, ti, 3);
    fm_load_tfrags<E>(tf, ttA, R, nk2);
#pragma unroll
    for (int t = 0; t < kFmMaxRT16; ++t) acc[t] = mf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
      const int col0 = c * CW, cw = min(CW, Cb - col0);
      const unsigned char *slot = ring + ((blk * nch + c) & 1) * szB;
      fe_stamp<TRACE>(tr, ti, 40);
      fm_barrier();  // C4(c)
      fe_stamp<TRACE>(tr, ti, 4);
      fm_phase1_set<E>(acc, slot, pb, nrt, cw >> 5, fcur);
      if (c + 1 < nch) {
        const int col1 = col0 + CW;
        fm_load_fragset<E>(fcur, pkb + (int64_t)(col1 >> 3) * 128, splitb, min(CW, Cb - col1) >> 5);
      } else {
        fm_load_fragset<E>(fcur, pka, splita, Ca >> 5);          // the next block's resident tile (same factor)
      }
      fe_stamp<TRACE>(tr, ti, 45);
      fm_phase2<E>(slot, pb, nk2, cw, tf, outb + col0, Cb, RT, false);
      fe_stamp<TRACE>(tr, ti, 50);
      fm_barrier();  // C5(c)
      fe_stamp<TRACE>(tr, ti, 5);
    }
    fm_combine_put<E>(acc, scratch, nrt);
    fe_stamp<TRACE>(tr, ti, 60);
    fm_barrier();  // C6
    fe_stamp<TRACE>(tr, ti, 6);
    fm_combine_get<E>(scratch, ttB, nrt, R, q.scale, nrows);
    fe_stamp<TRACE>(tr, ti, 70);
    fm_barrier();  // C7
    fe_stamp<TRACE>(tr, ti, 7);
    fm_load_tfrags<E>(tf, ttB, R, nk2);
    fm_phase2<E>(bufA, pa, nk2, Ca, tf, outa, Ca, RT, false);
    fe_stamp<TRACE>(tr, ti, 80);
    fm_barrier();  // C8
    fe_stamp<TRACE>(tr, ti, 8);
  }
}

// ---------------------------------------------------------------------------------------------------------- host side
struct FmGeom { int R, resident_is_x, cw, nchunk, pitch_a, pitch_b, lds; };

// Geometry of a site at R rows per block inside `lds_cap` bytes of LDS: the widest chunk of B that fits.
static bool fm_fit(int64_t M, int K, int N, int r, int act_dtype, int R, int lds_cap, FmGeom *g) {
  if (act_dtype == LORA_AMD_F32 || M <= 0 || r < 1 || r > 16 || K % 32 || N % 32 || K < 32 || N < 32) return false;
  if (R != 32 && R != 64) return false;
  g->resident_is_x = K <= N;
  const int Ca = std::min(K, N), Cb = std::max(K, N);
  g->pitch_a = fm_pitch(Ca);
  // the NEXT block's resident rows wait in registers while the current block is consumed: NPA pieces per thread
  if ((int64_t)R * Ca > (int64_t)(lds_cap <= kFmLdsSmall ? kFmNPA1 : kFmNPA2) * kFmThreads * 8) return false;
  const int fixed = R * g->pitch_a + 2 * 32 * fm_tpitch(R);
  // pieces per thread <= kFmNPB; the chunk buffer also holds the [4][R/16][4][64] f32 scratch of the phase-1 combine
  for (int cwmax = std::min(kFmNPB * kFmThreads * 8 / R, 512); cwmax >= 32; cwmax -= 32) {
    const int nch = (Cb + cwmax - 1) / cwmax;
    const int cw = std::min(((Cb + nch - 1) / nch + 31) / 32 * 32, cwmax);
    const int pb = std::max(fm_pitch(cw), 288);  // >= 256: the chunk buffer doubles as the combine scratch
    if (fixed + R * pb > lds_cap) continue;
    g->R = R; g->cw = cw; g->nchunk = (Cb + cw - 1) / cw; g->pitch_b = pb; g->lds = fixed + R * pb;
    return true;
  }
  return false;
}

// The engine kernel's geometry: resident block x 2 (x 1 if two do not fit), a 2-slot ring of chunks, scratch, T images in
// 160 KiB.  The chunk as wide as fits (fewer barriers), at most 63 DMA instructions per chunk and per share of the next
// resident block (the loader's counted waits).
static bool fm_fit_engine(int64_t M, int K, int N, int r, int act_dtype, int R, FmGeom *g, int *a_bufs, int nb_min = 1) {
  if (act_dtype == LORA_AMD_F32 || M <= 0 || r < 1 || r > 16 || K % 32 || N % 32 || K < 32 || N < 32) return false;
  if (R != 32 && R != 64) return false;
  g->resident_is_x = K <= N;
  const int Ca = std::min(K, N), Cb = std::max(K, N);
  g->pitch_a = fm_pitch(Ca);
  auto up1k = [](int b) { return (b + 1023) & ~1023; };
  const int szA = up1k(R * g->pitch_a), fixed = R * 256 + 2 * 32 * fm_tpitch(R);
  for (int nb = 2; nb >= nb_min; --nb) {
    for (int cwmax = 512; cwmax >= 32; cwmax -= 32) {
      const int nch = (Cb + cwmax - 1) / cwmax;
      const int cw = std::min(((Cb + nch - 1) / nch + 31) / 32 * 32, cwmax);
      const int pb = fm_pitch(cw), szB = up1k(R * pb);
      const int nchunk = (Cb + cw - 1) / cw;
      if (nb * szA + 2 * szB + fixed > kFmLdsLarge) continue;
      if ((szB >> 10) > kFeNKB || (szA >> 10) > kFeNKA || Ca > 128 * kFmNF || cw > 128 * kFmNF) continue;
      g->R = R; g->cw = cw; g->nchunk = nchunk; g->pitch_b = pb; g->lds = nb * szA + 2 * szB + fixed;
      *a_bufs = nb;
      return true;
    }
  }
  return false;
}

// rows per block and LDS class (1: two workgroups per CU, 2: one) of a site.  A caller's / LORA_AMD_FM_ROWS' row count is
// tried first in both classes; default: 64 rows, then 32, two workgroups per CU before one.
static bool fm_choose(int64_t M, int K, int N, int r, int act_dtype, int hint, FmGeom *g, int *cls) {
  const int caps[2] = {kFmLdsSmall, kFmLdsLarge};
  if (hint > 0)
    for (int c = 0; c < 2; ++c)
      if (fm_fit(M, K, N, r, act_dtype, hint, caps[c], g)) { *cls = c + 1; return true; }
  for (int c = 0; c < 2; ++c)
    for (int R = 64; R >= 32; R >>= 1)
      if (fm_fit(M, K, N, r, act_dtype, R, caps[c], g)) { *cls = c + 1; return true; }
  return false;
}

static int fm_rows_env() {
  static const int v = getenv("LORA_AMD_FM_ROWS") ? atoi(getenv("LORA_AMD_FM_ROWS")) : 0;
  return v;
}
// row blocks one workgroup walks (LORA_AMD_FM_NB): their partial sums meet in the workgroup's slab.  Register-staged
// kernel: 1 (measured: with its loads, its factor fragments and its slab reads on ONE in-order vmcnt counter a longer run
// only serialises: 1006 / 1174 / 1657 us at 1 / 2 / 4 blocks); the engine kernel (loader wave): 8
static int fm_blocks_per_wg(int64_t nrb, int engine, int64_t block_bytes) {
  static const int env = getenv("LORA_AMD_FM_NB") ? atoi(getenv("LORA_AMD_FM_NB")) : 0;
  // engine: about 640 KB of G + X per workgroup (8 blocks of a 320-wide attention site, 2 of a GEGLU site): long enough to
  // amortise the exposed first resident block, short enough that the last workgroups do not leave the chip idle
  const int v = env > 0 ? env : (engine ? (int)std::max<int64_t>(1, (655360 + block_bytes / 2) / block_bytes) : 1);
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(std::max(v, 1), kFmMaxNB), nrb));
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_factors_mfma_plan(int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype, int32_t rows,
                                          int32_t flags, lora_amd_factors_mfma_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr && rows >= 0 && dtype_ok(act_dtype), LORA_AMD_EINVAL, "factors_mfma_plan: bad argument");
  memset(out, 0, sizeof(*out));
  FmGeom g;
  int cls = 0, a_bufs = 0;
  const int hint = rows > 0 ? rows : fm_rows_env();
  static const bool no_engine = getenv("LORA_AMD_FM_ENGINE") && atoi(getenv("LORA_AMD_FM_ENGINE")) == 0;
  // the engine (class 3) unless the site has dropout (the mask is applied in registers: register-staged kernel) or the
  // caller / LORA_AMD_FM_ENGINE=0 turns it off; 64 rows first (half the partial slabs), then 32
  bool eng = false;
  if (!(flags & 3) && !no_engine) {
    if (hint > 0) eng = fm_fit_engine(M, K, N, r, act_dtype, hint, &g, &a_bufs);
    // a double-buffered resident block first (the next block lands while this one is consumed), 64 rows before 32
    for (int nb = 2; !eng && nb >= 1; --nb)
      for (int R = 64; !eng && R >= 32; R >>= 1) eng = fm_fit_engine(M, K, N, r, act_dtype, R, &g, &a_bufs, nb) && a_bufs >= nb;
  }
  if (eng) cls = 3;
  else if (!fm_choose(M, K, N, r, act_dtype, hint, &g, &cls)) return LORA_AMD_OK;
  out->a_bufs = a_bufs;
  out->supported = 1;
  out->lds_class = cls;
  out->rank_tile = r <= 4 ? 4 : r <= 8 ? 8 : 16;
  out->rows_per_block = g.R;
  const int64_t nrb = (M + g.R - 1) / g.R;
  out->blocks_per_wg = fm_blocks_per_wg(nrb, cls == 3, (int64_t)g.R * (N + K) * 2);
  // one partial slab per row block on the engine (its workgroups walk several blocks for the pipeline only), one per run of
  // blocks on the register-staged kernel
  out->nparts = cls == 3 ? (int32_t)nrb : (int32_t)((nrb + out->blocks_per_wg - 1) / out->blocks_per_wg);
  out->lds_bytes = g.lds;
  out->up_part_floats = (int64_t)out->nparts * out->rank_tile * N;
  out->down_part_floats = (int64_t)out->nparts * out->rank_tile * K;
  out->pack_up_elems = (int64_t)N * 32;    // [2][N/8][16][8]
  out->pack_down_elems = (int64_t)K * 32;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_factor_pack_plan(lora_amd_pack_site *sites, int32_t n, int64_t *total) {
  LORA_AMD_CHECK(sites && n >= 1 && total, LORA_AMD_EINVAL, "factor_pack_plan: bad argument");
  int64_t begin = 0;
  for (int i = 0; i < n; ++i) {
    lora_amd_pack_site &q = sites[i];
    LORA_AMD_CHECK(q.down && q.up && q.pk_down && q.pk_up, LORA_AMD_EINVAL, "factor_pack_plan: site %d: null pointer", i);
    LORA_AMD_CHECK(q.r >= 1 && q.r <= 16, LORA_AMD_ERANK, "factor_pack_plan: site %d: rank %d outside [1,16]", i, q.r);
    LORA_AMD_CHECK(q.N >= 32 && q.K >= 32 && q.N % 32 == 0 && q.K % 32 == 0 && ((uintptr_t)q.pk_down % 16) == 0 &&
                       ((uintptr_t)q.pk_up % 16) == 0,
                   LORA_AMD_EINVAL, "factor_pack_plan: site %d: N, K must be multiples of 32, packs 16-byte aligned", i);
    q.begin = begin;
    begin += (int64_t)((q.N + q.K) >> 3) * 16;
  }
  *total = begin;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_factor_pack(const lora_amd_pack_site *sites_dev, int32_t n, int64_t total, int32_t act_dtype,
                                    void *stream) {
  LORA_AMD_CHECK(sites_dev && n >= 1 && total >= 1, LORA_AMD_EINVAL, "factor_pack: bad argument");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_F16 || act_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "factor_pack: the matrix-core factor pass takes f16 / bf16 activations");
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == LORA_AMD_F16) hipLaunchKernelGGL(factor_pack_kernel<f16_t>, dim3(grid), dim3(256), 0, st, sites_dev, n, total);
  else hipLaunchKernelGGL(factor_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, sites_dev, n, total);
  return check_launch("lora_amd_factor_pack");
}

extern "C" int lora_amd_factors_mfma_ragged_plan(lora_amd_fm_site *sites, int32_t n, int32_t act_dtype, int32_t lds_class,
                                                 int64_t *grid) {
  LORA_AMD_CHECK(sites && n >= 1 && grid && lds_class >= 1 && lds_class <= 3, LORA_AMD_EINVAL,
                 "factors_mfma_ragged_plan: bad argument");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_F16 || act_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "factors_mfma_ragged_plan: f16 / bf16 activations only");
  auto heads_ok = [](int d, int D, int cols, int64_t ld) {
    return d == 0 || (d > 0 && D >= d && d % 8 == 0 && D % 8 == 0 && cols % d == 0 && ld >= (int64_t)(cols / d) * D);
  };
  int64_t begin = 0;
  const int rt0 = sites[0].r <= 4 ? 4 : sites[0].r <= 8 ? 8 : 16;
  for (int i = 0; i < n; ++i) {
    lora_amd_fm_site &q = sites[i];
    FmGeom g;
    LORA_AMD_CHECK(q.r >= 1 && q.r <= 16 && (q.r <= 4 ? 4 : q.r <= 8 ? 8 : 16) == rt0, LORA_AMD_ERANK,
                   "factors_mfma_ragged_plan: site %d: rank %d (one rank tile per table)", i, q.r);
    LORA_AMD_CHECK(q.g && q.x && q.pk_up && q.pk_down && q.up_part && q.down_part, LORA_AMD_EINVAL,
                   "factors_mfma_ragged_plan: site %d: null pointer", i);
    int a_bufs = 0;
    const bool ok = lds_class == 3 ? fm_fit_engine(q.M, q.K, q.N, q.r, act_dtype, q.rows_per_block, &g, &a_bufs)
                                   : fm_fit(q.M, q.K, q.N, q.r, act_dtype, q.rows_per_block, lds_class == 1 ? kFmLdsSmall : kFmLdsLarge, &g);
    LORA_AMD_CHECK(lds_class != 3 || q.dropout_p == 0.f, LORA_AMD_EINVAL,
                   "factors_mfma_ragged_plan: site %d: a dropout site in an engine table (plan it with flags = 1)", i);
    q.a_bufs = a_bufs;
    LORA_AMD_CHECK(ok && ((uintptr_t)q.g % 16) == 0 && ((uintptr_t)q.x % 16) == 0 && q.ldg % 8 == 0 && q.ldx % 8 == 0 &&
                       ((uintptr_t)q.pk_up % 16) == 0 && ((uintptr_t)q.pk_down % 16) == 0 &&
                       heads_ok(q.g_head_dim, q.g_head_pad, q.N, q.ldg) && heads_ok(q.x_head_dim, q.x_head_pad, q.K, q.ldx),
                   LORA_AMD_EINVAL, "factors_mfma_ragged_plan: site %d: shape / alignment / head layout / LDS class not supported", i);
    q.rows_per_block = g.R; q.resident_is_x = g.resident_is_x; q.cw = g.cw; q.nchunk = g.nchunk;
    q.pitch_a = g.pitch_a; q.pitch_b = g.pitch_b; q.lds_bytes = g.lds;
    const int64_t nrb = (q.M + g.R - 1) / g.R;
    LORA_AMD_CHECK(q.blocks_per_wg >= 1 && q.blocks_per_wg <= kFmMaxNB, LORA_AMD_EINVAL,
                   "factors_mfma_ragged_plan: site %d: blocks_per_wg %d outside [1, %d]", i, q.blocks_per_wg, kFmMaxNB);
    q.block_begin = begin;
    begin += (nrb + q.blocks_per_wg - 1) / q.blocks_per_wg;
  }
  LORA_AMD_CHECK(begin < (1ll << 31), LORA_AMD_EINVAL, "factors_mfma_ragged_plan: too many blocks");
  *grid = begin;
  return LORA_AMD_OK;
}

extern "C" int lora_amd_linear_bwd_factors_mfma_ragged(const lora_amd_fm_site *sites_dev, int32_t n, int64_t grid,
                                                       int32_t lds_class, int32_t act_dtype, int32_t masked, void *stream) {
  LORA_AMD_CHECK(sites_dev && n >= 1 && grid >= 1 && grid < (1ll << 31) && lds_class >= 1 && lds_class <= 3,
                 LORA_AMD_EINVAL, "linear_bwd_factors_mfma_ragged: bad argument");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_F16 || act_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "linear_bwd_factors_mfma_ragged: f16 / bf16 activations only");
  hipStream_t st = (hipStream_t)stream;
  if (lds_class == 3) {
    LORA_AMD_CHECK(!masked, LORA_AMD_EINVAL, "linear_bwd_factors_mfma_ragged: the engine kernel takes no dropout sites");
    const bool trace = getenv("LORA_AMD_FM_TRACE") && atoi(getenv("LORA_AMD_FM_TRACE")) == 1;
    if (act_dtype == LORA_AMD_F16)
      hipLaunchKernelGGL((factors_mfma_engine_kernel<f16_t, false>), dim3((unsigned)grid), dim3(kFeThreads), 0, st, sites_dev, n);
    else if (trace)
      hipLaunchKernelGGL((factors_mfma_engine_kernel<bf16_t, true>), dim3((unsigned)grid), dim3(kFeThreads), 0, st, sites_dev, n);
    else
      hipLaunchKernelGGL((factors_mfma_engine_kernel<bf16_t, false>), dim3((unsigned)grid), dim3(kFeThreads), 0, st, sites_dev, n);
    return check_launch("lora_amd_linear_bwd_factors_mfma_ragged");
  }
  const bool drop = masked != 0;  // a table of dropout sites: the kernel with the Philox mask on G (straight-line, no per-site branch)
#define FM(E, L)                                                                                                      \
  do {                                                                                                                \
    if (drop) hipLaunchKernelGGL((factors_mfma_kernel<E, L, true>), dim3((unsigned)grid), dim3(kFmThreads), 0, st, sites_dev, n); \
    else hipLaunchKernelGGL((factors_mfma_kernel<E, L, false>), dim3((unsigned)grid), dim3(kFmThreads), 0, st, sites_dev, n); \
  } while (0)
  if (act_dtype == LORA_AMD_F16) {
    if (lds_class == 1) FM(f16_t, kFmLdsSmall); else FM(f16_t, kFmLdsLarge);
  } else {
    if (lds_class == 1) FM(bf16_t, kFmLdsSmall); else FM(bf16_t, kFmLdsLarge);
  }
#undef FM
  return check_launch("lora_amd_linear_bwd_factors_mfma_ragged");
}

