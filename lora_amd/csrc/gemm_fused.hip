// K1 fully fused — Y = X W^T + b + scale * (X down^T) up^T in ONE kernel on the matrix cores.
//
// replaces: lora_diffusion/lora.py:53-58 — the frozen addmm AND the low-rank branch (SURVEY §8d "K1 fwd (fused)":
//           read W, X, down, up once, write Y once: (N*K + M*K + M*N)*e + (N+K)*r*4 algorithmic bytes).
//
// Structure (wave64, 4 waves = 2(M) x 2(N) per workgroup, bf16/f16 operands, f32 accumulation):
//   * the frozen contraction is a v_mfma_f32_16x16x32 tile loop over K in steps of 64: X and W slabs stream
//     HBM -> LDS with global_load_lds (16 B per lane, no VGPR round trip) through a 3-slot ring, two steps in flight
//     behind the one being multiplied, counted vmcnt + one raw s_barrier per step;
//     the LDS image is the linear [row][128 B] tile with its 16-byte chunks XOR-swizzled by (row & 7) — applied on the
//     SOURCE address of the DMA and on the fragment read — so that every ds_read_b128 service group hits 16 distinct
//     16-byte slots (conflict-free) while the global side still reads whole 128-byte lines;
//   * T = X down^T rides along as one extra 16-wide MFMA column per row subtile: the f32 `down` slab of the step is
//     DMA'd into the same ring slot and enters as hi + lo 16-bit fragments (T as precise as with f32 factors), so T
//     costs 1/CS more MFMAs and no extra pass over X.  T (f32) is written for the backward by the first column block only;
//   * epilogue: T is rounded to the activation dtype (as the reference's autocast does for lora_down's output), parked
//     in LDS in A-fragment order, and ONE more MFMA per output subtile adds (scale*up) T^T-style rank-r term straight
//     into the accumulators; bias is added, the tile is transposed through LDS and leaves as 16-byte row stores.
// Dropout (p > 0) and the selector keep the two-launch path (linear_fused.hip): the mask applies to the low-rank term
// alone, which this kernel never materialises.
#include <algorithm>

#include "common.hpp"

namespace lora_amd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class E> struct MfmaT;
template <> struct MfmaT<bf16_t> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaT<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int kGT = 256;   // threads
constexpr int kBK = 64;    // K step (elements) = one 128-byte LDS row of 16-bit elements

__device__ inline void glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                   (void __attribute__((address_space(3))) *)lds_wave_base, 16, 0, 0);
}

// One [rows][64] slab HBM -> LDS: wave-instruction q covers rows [8q, 8q+8); lane l lands at 8q*128 + l*16 and
// fetches chunk (l & 7) ^ (row & 7) of its row (the swizzle lives on the source side).  Rows past `limit` re-read the
// last valid row (their results are never stored).
// Head-padded operands: a tensor whose logical row is `heads` runs of d elements may be stored with every run padded
// to D elements (what the attention kernels want for d = 40 / 80).  hc = d / 8 and hp = D / 8 chunks; logical 16-byte
// chunk c lives at physical chunk (c / hc) * hp + c % hc.  hc == 0: dense rows.
__device__ inline int head_chunk(int c, int hc, int hp) { return hc ? (c / hc) * hp + (c % hc) : c; }

template <class S>
__device__ inline void stage_slab(const S *g, int64_t ld, int64_t row0, int64_t limit, int k0, char *lds, int nrows,
                                  int wave, int lane, int hc = 0, int hp = 0) {
  for (int q = wave; q * 8 < nrows; q += 4) {
    const int rl = q * 8 + (lane >> 3);
    int64_t row = row0 + rl;
    if (row >= limit) row = limit - 1;
    const int c = (lane & 7) ^ (rl & 7);
    glds16(g + row * ld + head_chunk((k0 >> 3) + c, hc, hp) * 8, lds + q * 1024);
  }
}

// 8 f32 values -> one MFMA operand fragment of the activation dtype
template <class E>
__device__ inline typename MfmaT<E>::frag make_frag(const float (&v)[8]) {
  union { typename MfmaT<E>::frag f; typename E::storage s[8]; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.s[i] = E::from_f(v[i]);
  return u.f;
}

template <class F>
__device__ inline F lds_frag(const char *slab, int row, int c) {
  return *reinterpret_cast<const F *>(slab + row * 128 + ((c ^ (row & 7)) << 4));
}

template <int N>
__device__ inline void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// grid (column blocks, row blocks).  RS/CS: 16x16 subtiles per wave along M / N  ->  BM = 32*RS, BN = 32*CS.
// NS-stage LDS ring (NS = 3): the slabs of steps i+1 and i+2 are in flight while step i is multiplied.  Every VMEM
// operation of the K loop is an LDS-DMA (X, W and the f32 `down` slab), issued in the same number L per wave and
// step, so "step i has landed" is the counted `s_waitcnt vmcnt((NS-2)*L)` of each wave followed by ONE raw s_barrier
// per step (which also says that everybody is done reading the slot about to be refilled).
template <class E, int RS, int CS, int NS, int RG>
__global__ __launch_bounds__(kGT) void linear_gemm_fwd_kernel(
    const typename E::storage *__restrict__ x, int64_t ldx, const typename E::storage *__restrict__ w, int64_t ldw,
    const typename E::storage *__restrict__ bias, typename E::storage *__restrict__ y, int64_t ldy,
    const float *__restrict__ down, const float *__restrict__ up, float *__restrict__ t_out, int64_t M, int K, int N,
    int r, float scale, float t_scale, int flayout, int xhc, int xhp, int yhc, int yhp) {
  using S = typename E::storage;
  using F = typename MfmaT<E>::frag;
  constexpr int BM = 32 * RS, BN = 32 * CS;
  constexpr int XB = BM * 128, WB = BN * 128, DB = RG * 1024;  // bytes per slab: X, W (16-bit), down (f32 [4*RG][64])
  constexpr int SB = XB + WB + DB;
  constexpr int L = BM / 32 + BN / 32 + 1;                    // LDS-DMA instructions per wave per step
  static_assert(RS == 1 || RS == 2 || RS == 4, "row subtiles per wave");
  constexpr int OUT_LD = BN + 8;                              // padded output tile row (elements)
  constexpr int STAGE = NS * SB;
  constexpr int OUTB = BM * OUT_LD * 2;
  constexpr int TSB = BM * 16 * 2;                            // T tile [BM][16] in the activation dtype
  constexpr int SMEM = (STAGE > OUTB ? STAGE : OUTB) + TSB;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  char *ts = smem + (STAGE > OUTB ? STAGE : OUTB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nk = K / kBK;

  f32x4 acc[RS][CS];
#pragma unroll
  for (int a = 0; a < RS; ++a)
#pragma unroll
    for (int b = 0; b < CS; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // T row subtiles of a wave row are shared out between its two waves: wave (wm, wn) accumulates subtiles i = wn, wn+2
  constexpr int TS = (RS + 1) / 2;
  f32x4 tacc[TS];
#pragma unroll
  for (int i = 0; i < TS; ++i) tacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool t_owner = wn < RS;

  // one step's slabs -> ring slot: X rows, W rows, and 4 rank rows of `down` per wave (f32, 16-byte chunks swizzled by
  // the rank; ranks past r re-read row r-1 and are zeroed when the fragment is built)
  // flayout bit 0: `down` is given as [K, r] (k-major): the 64 x r block of a step is one contiguous run, DMA'd
  // linearly ([k][r] in LDS, scalar fragment reads); else [r, K]: 4 rank rows per wave, 16-byte chunks swizzled by rank.
  // flayout bit 1: `up` is given as [r, N].  (The backward dX = G W + (G up) down reuses this kernel on W^T with
  // down' = up [N,r] read k-major and up' = down [r,K]: no transposed factor copies.)
  const bool dn_kr = flayout & 1, up_rk = flayout & 2;
  const int dn_w = wave < RG ? wave : 0;  // waves past the rank groups re-stage group 0 (keeps L uniform per wave)
  const int dn_rank = min(dn_w * 4 + (lane >> 4), r - 1);
  const int kr_chunk = min(dn_w * 64 + lane, 16 * r - 1);  // 16-byte chunk of the contiguous 64*r*4-byte block
  const float *dn_src = dn_kr ? down + (kr_chunk << 2)
                              : down + (int64_t)dn_rank * K + (((lane & 15) ^ ((dn_w * 4 + (lane >> 4)) & 15)) << 2);
  const int64_t dn_step = dn_kr ? (int64_t)kBK * r : kBK;
#define ISSUE(step, slot)                                                               \
  do {                                                                                  \
    char *xs_ = smem + (slot) * SB, *ws_ = xs_ + XB, *ds_ = ws_ + WB;                   \
    stage_slab<S>(x, ldx, m0, M, (step) * kBK, xs_, BM, wave, lane, xhc, xhp);          \
    stage_slab<S>(w, ldw, n0, N, (step) * kBK, ws_, BN, wave, lane);                    \
    glds16(dn_src + (step) * dn_step, ds_ + dn_w * 1024);                               \
  } while (0)

#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < nk) ISSUE(p, p);

  // operands of the epilogue, fetched while the first slabs fly: (scale * up) B fragments and the bias
  F ub[CS];
  float bv[CS];
#pragma unroll
  for (int j = 0; j < CS; ++j) {
    const int n = n0 + (wn * CS + j) * 16 + l15;
    const int nc = n < N ? n : N - 1;  // clamped addresses, branch-free loads (all in flight together), then select
    float uv[8];
    if (up_rk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rank = lg * 8 + e;
        const float v = up[(int64_t)(rank < r ? rank : r - 1) * N + nc];
        uv[e] = (rank < r && n < N) ? scale * v : 0.f;
      }
    } else if ((r & 3) == 0) {  // whole 16-byte groups of ranks: r = 4 -> one load in lane group 0, r = 16 -> two in groups 0, 1
      const int g0 = lg * 8 < r ? lg * 8 : 0, g1 = lg * 8 + 4 < r ? lg * 8 + 4 : 0;
      const float4 a = *reinterpret_cast<const float4 *>(up + (int64_t)nc * r + g0);
      const float4 b = *reinterpret_cast<const float4 *>(up + (int64_t)nc * r + g1);
      const bool la = lg * 8 < r && n < N, lb = lg * 8 + 4 < r && n < N;
      uv[0] = la ? scale * a.x : 0.f; uv[1] = la ? scale * a.y : 0.f; uv[2] = la ? scale * a.z : 0.f;
      uv[3] = la ? scale * a.w : 0.f; uv[4] = lb ? scale * b.x : 0.f; uv[5] = lb ? scale * b.y : 0.f;
      uv[6] = lb ? scale * b.z : 0.f; uv[7] = lb ? scale * b.w : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rank = lg * 8 + e;
        const float v = up[(int64_t)nc * r + (rank < r ? rank : r - 1)];
        uv[e] = (rank < r && n < N) ? scale * v : 0.f;
      }
    }
    ub[j] = make_frag<E>(uv);
    const float b = bias != nullptr ? E::to_f(bias[nc]) : 0.f;
    bv[j] = n < N ? b : 0.f;
  }

  for (int step = 0; step < nk; ++step) {
    const int rem = nk - 1 - step;  // steps already issued beyond this one: min(rem, NS - 2)
    if (NS > 2 && rem >= NS - 2) wait_vmcnt<(NS > 2 ? (NS - 2) * L : 0)>(); else wait_vmcnt<0>();
    raw_barrier();
    if (step + NS - 1 < nk) ISSUE(step + NS - 1, (step + NS - 1) % NS);
    const char *xs = smem + (step % NS) * SB, *ws = xs + XB, *ds = ws + WB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      F a[RS], b[CS];
#pragma unroll
      for (int i = 0; i < RS; ++i) a[i] = lds_frag<F>(xs, (wm * RS + i) * 16 + l15, ks * 4 + lg);
#pragma unroll
      for (int j = 0; j < CS; ++j) b[j] = lds_frag<F>(ws, (wn * CS + j) * 16 + l15, ks * 4 + lg);
#pragma unroll
      for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int j = 0; j < CS; ++j) acc[i][j] = MfmaT<E>::mma(a[i], b[j], acc[i][j]);
      if (t_owner) {
        const bool live = l15 < r;
        float dv[8];
        if (dn_kr) {
          const int jc = live ? l15 : r - 1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = *reinterpret_cast<const float *>(ds + (((ks * 32 + lg * 8 + e) * r + jc) << 2));
            dv[e] = live ? v : 0.f;
          }
        } else {
          const int c0 = ks * 8 + lg * 2;
          const f32x4 p0 = *reinterpret_cast<const f32x4 *>(ds + l15 * 256 + (((c0) ^ l15) << 4));
          const f32x4 p1 = *reinterpret_cast<const f32x4 *>(ds + l15 * 256 + (((c0 + 1) ^ l15) << 4));
          dv[0] = live ? p0[0] : 0.f; dv[1] = live ? p0[1] : 0.f; dv[2] = live ? p0[2] : 0.f; dv[3] = live ? p0[3] : 0.f;
          dv[4] = live ? p1[0] : 0.f; dv[5] = live ? p1[1] : 0.f; dv[6] = live ? p1[2] : 0.f; dv[7] = live ? p1[3] : 0.f;
        }
        // the f32 master factor enters as hi + lo 16-bit parts (two MFMAs): T keeps the precision of the two-launch
        // path (f32 factors) instead of rounding `down` to the activation dtype
        float dl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dl[e] = dv[e] - E::to_f(E::from_f(dv[e]));
        const F dhi = make_frag<E>(dv), dlo = make_frag<E>(dl);
#pragma unroll
        for (int q = 0; q < TS; ++q) {
          const int i = 2 * q + wn;  // wave-uniform; the select below keeps register indexing static
#pragma unroll
          for (int ii = 0; ii < RS; ++ii)
            if (ii == i) {
              tacc[q] = MfmaT<E>::mma(a[ii], dhi, tacc[q]);
              tacc[q] = MfmaT<E>::mma(a[ii], dlo, tacc[q]);
            }
        }
      }
    }
  }
#undef ISSUE
  raw_barrier();  // everybody is done with the ring: it is reused for the output tile below

  // ---- T: out (f32, first column block) and into LDS in the activation dtype: ts[row][16 ranks]
  if (t_owner) {
#pragma unroll
    for (int q = 0; q < TS; ++q) {
      if (2 * q + wn >= RS) continue;
      const int row_base = (wm * RS + 2 * q + wn) * 16 + lg * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = row_base + i;
        reinterpret_cast<S *>(ts)[rl * 16 + l15] = E::from_f(tacc[q][i]);
        if (blockIdx.x == 0 && l15 < r && m0 + rl < M) t_out[(m0 + rl) * r + l15] = t_scale * tacc[q][i];
      }
    }
  }
  __syncthreads();
  // ---- rank-r term: acc += T_tile (A operand, K = ranks padded to 32) x (scale * up) (B operand)
  {
    F ta[RS];
    const float zeros[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      const int row = (wm * RS + i) * 16 + l15;
      ta[i] = lg < 2 ? *reinterpret_cast<const F *>(ts + row * 32 + lg * 16) : make_frag<E>(zeros);
    }
#pragma unroll
    for (int j = 0; j < CS; ++j)
#pragma unroll
      for (int i = 0; i < RS; ++i) acc[i][j] = MfmaT<E>::mma(ta[i], ub[j], acc[i][j]);
  }
  // ---- bias, conversion, transpose through LDS, 16-byte row stores
  S *os = reinterpret_cast<S *>(smem);
#pragma unroll
  for (int j = 0; j < CS; ++j) {
    const int cl = (wn * CS + j) * 16 + l15;
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      const int rb = (wm * RS + i) * 16 + lg * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) os[(rb + e) * OUT_LD + cl] = E::from_f(acc[i][j][e] + bv[j]);
    }
  }
  __syncthreads();
  constexpr int CH = BN / 8;
  for (int id = tid; id < BM * CH; id += kGT) {
    const int rl = id / CH, cc = id - rl * CH;
    if (m0 + rl < M && n0 + cc * 8 < N)
      *reinterpret_cast<Chunk8<E> *>(y + (m0 + rl) * ldy + head_chunk((n0 >> 3) + cc, yhc, yhp) * 8) =
          *reinterpret_cast<const Chunk8<E> *>(os + rl * OUT_LD + cc * 8);
  }
  if (yhc) {  // head-padded output: this block owns whole heads (host checks CH % yhc == 0): zero their pad chunks
    const int head0 = (n0 >> 3) / yhc;
    const int nheads = min(CH / yhc, (N >> 3) / yhc - head0), padc = yhp - yhc;
    Chunk8<E> z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = E::from_f(0.f);
    for (int id = tid; id < BM * nheads * padc; id += kGT) {
      const int rl = id / (nheads * padc), rem = id - rl * (nheads * padc);
      const int hh = rem / padc, pc = rem - hh * padc;
      if (m0 + rl < M)
        *reinterpret_cast<Chunk8<E> *>(y + (m0 + rl) * ldy + ((head0 + hh) * yhp + yhc + pc) * 8) = z;
    }
  }
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_linear_gemm_supported(int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype) {
  return (act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16) && M > 0 && K >= kBK && K % kBK == 0 && N >= 16 &&
         N % 8 == 0 && r >= 1 && r <= 16;
}

extern "C" int lora_amd_linear_gemm_fwd(const void *x, int64_t ldx, const void *w, int64_t ldw, const void *bias,
                                        void *y, int64_t ldy, const float *down, const float *up, float *t_out,
                                        int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype, float scale,
                                        float t_scale, int32_t factor_layout, int32_t tile, void *stream) {
  return lora_amd_linear_gemm_fwd_heads(x, ldx, w, ldw, bias, y, ldy, down, up, t_out, M, K, N, r, act_dtype, scale,
                                        t_scale, factor_layout, tile, 0, 0, 0, 0, stream);
}

extern "C" int lora_amd_linear_gemm_fwd_heads(const void *x, int64_t ldx, const void *w, int64_t ldw,
                                              const void *bias, void *y, int64_t ldy, const float *down,
                                              const float *up, float *t_out, int64_t M, int32_t K, int32_t N,
                                              int32_t r, int32_t act_dtype, float scale, float t_scale,
                                              int32_t factor_layout, int32_t tile, int32_t x_head_dim,
                                              int32_t x_head_pad, int32_t y_head_dim, int32_t y_head_pad,
                                              void *stream) {
  LORA_AMD_CHECK(lora_amd_linear_gemm_supported(M, K, N, r, act_dtype), LORA_AMD_EINVAL,
                 "linear_gemm_fwd: needs bf16/f16 activations, K %% 64 == 0, N %% 8 == 0, rank <= 16");
  auto heads_ok = [](int d, int D, int cols, int64_t ld) {
    return d == 0 || (d > 0 && D >= d && d % 8 == 0 && D % 8 == 0 && cols % d == 0 && ld >= (int64_t)(cols / d) * D);
  };
  LORA_AMD_CHECK(heads_ok(x_head_dim, x_head_pad, K, ldx) && heads_ok(y_head_dim, y_head_pad, N, ldy) &&
                     (y_head_dim == 0 || 160 % y_head_dim == 0),
                 LORA_AMD_EINVAL, "linear_gemm_fwd: head layout (d=%d D=%d | d=%d D=%d) not supported for K=%d N=%d",
                 x_head_dim, x_head_pad, y_head_dim, y_head_pad, K, N);
  const int xhc = x_head_dim / 8, xhp = x_head_pad / 8, yhc = y_head_dim / 8, yhp = y_head_pad / 8;
  LORA_AMD_CHECK(x && w && y && down && up && t_out, LORA_AMD_EINVAL, "linear_gemm_fwd: null pointer");
  auto al = [](const void *p, int64_t ld) { return ((uintptr_t)p % 16) == 0 && ld % 8 == 0; };
  LORA_AMD_CHECK(al(x, ldx) && al(w, ldw) && al(y, ldy) && ((uintptr_t)down % 16) == 0, LORA_AMD_EINVAL,
                 "linear_gemm_fwd: rows must be 16-byte aligned");
  // tile = 10 * stages + shape;  shape: 1 = 64x320, 2 = 64x160, 3 = 32x160, 4 = 128x160 output tile per workgroup,
  // stages: 2 or 3 LDS ring slots (0 -> 2).  tile == 0: the largest tile that still gives every CU a workgroup.
  int shape = tile % 10, stages = tile / 10;
  if (stages != 3) stages = 2;
  if (shape <= 0 || shape > 4) {
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (int64_t)((N + bn - 1) / bn); };
    shape = blocks(128, 160) >= 512 ? 4 : blocks(64, 160) >= 256 ? 2 : 3;
  }
  const int rg = r <= 4 ? 1 : 4;
  hipStream_t st = (hipStream_t)stream;
#define GF(E, RSV, CSV, NSV, RGV)                                                                                  \
  hipLaunchKernelGGL((linear_gemm_fwd_kernel<E, RSV, CSV, NSV, RGV>),                                              \
                     dim3((unsigned)((N + 32 * CSV - 1) / (32 * CSV)), (unsigned)((M + 32 * RSV - 1) / (32 * RSV))), \
                     dim3(kGT), 0, st, reinterpret_cast<const typename E::storage *>(x), ldx,                      \
                     reinterpret_cast<const typename E::storage *>(w), ldw,                                        \
                     reinterpret_cast<const typename E::storage *>(bias), reinterpret_cast<typename E::storage *>(y), \
                     ldy, down, up, t_out, M, K, N, r, scale, t_scale, factor_layout, xhc, xhp, yhc, yhp)
#define GF_R(E, RSV, CSV, NSV) do { if (rg == 1) GF(E, RSV, CSV, NSV, 1); else GF(E, RSV, CSV, NSV, 4); } while (0)
#define GF_S(E, RSV, CSV) do { if (stages == 3) GF_R(E, RSV, CSV, 3); else GF_R(E, RSV, CSV, 2); } while (0)
#define GF_T(E)                                                                        \
  do {                                                                                 \
    if (shape == 1) { if (stages == 3 && rg == 4) GF(E, 2, 10, 2, 4); else GF_S(E, 2, 10); } \
    else if (shape == 2) GF_S(E, 2, 5);                                                \
    else if (shape == 3) GF_S(E, 1, 5);                                                \
    else GF_S(E, 4, 5);                                                                \
  } while (0)
  if (act_dtype == LORA_AMD_BF16) GF_T(bf16_t); else GF_T(f16_t);
#undef GF_T
#undef GF_S
#undef GF_R
#undef GF
  return check_launch("lora_amd_linear_gemm_fwd");
}
