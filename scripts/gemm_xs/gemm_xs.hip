// K1/K2, input-stationary form — OUT = IN B^T + bias + scale * mask o ((IN down^T) up^T) in ONE launch, wave-autonomous.
//
// replaces: lora_diffusion/lora.py:53-58 (and its input gradient dX = G W + scale (mask o G up) down) at the sites with a
//           SHORT contraction and MANY rows — (16384, 320, *), (4096, 640, *), (9216, 320, *) — where csrc/gemm_ws.hip (weight
//           in registers, input through an LDS ring) sat at 0.18-0.19 of the byte roof for four rounds: every workgroup pulled
//           the whole [N, K] panel into registers BEFORE its first MFMA and then had one or two input tiles to spend it on.
//
// The roles are turned round (the structure that took the factor pass from 0.22 to 0.45 of the roof, csrc/factor_mfma.hip):
//   * a wave OWNS SL 16-row slabs of the input and keeps them in REGISTERS for the whole kernel: the 16 bytes a lane loads
//     from a row-major row (row = lane & 15, k = 32 s + 8 (lane >> 4) ..+7) ARE the B operand of v_mfma_f32_16x16x32 with
//     the operands swapped (A = weight fragment): no LDS staging of the input, no barrier in the k loop, K / 32 x SL
//     independent 16-byte loads in flight per lane from the first instruction on;
//   * the frozen weight comes PRE-PACKED in fragment order (lora_amd_ws_pack's layout: [16-column tile][k step][lane][8]);
//     a workgroup's panel of NCT column tiles (50-96 KB) goes global -> LDS by LDS-DMA while the input loads are in flight,
//     and is read back with conflict-free ds_read_b128 (one per MFMA pair); LDS <= 80 KB, 2 workgroups per CU overlap
//     one's streaming with the other's MFMAs;
//   * the swapped product leaves 4 CONSECUTIVE output columns of one row in a lane: 8-byte stores straight from the
//     accumulators, one Philox call per lane and column tile for the dropout mask;
//   * the low-rank branch never leaves the matrix pipe: T^T = down X^T (down as hi + lo 16-bit fragments, built once per
//     workgroup into LDS) lands as lane (row, g) -> T[row][4g .. 4g+3], which IS the B operand of the rank product against
//     the up fragment {up_hi[4g..], up_hi[4g..]} x {T_hi[4g..], T_lo[4g..]} (+ one MFMA for up_lo T_hi): f32-grade factors;
//   * a workgroup walks PG consecutive panels with its rows RESIDENT (the input is read from HBM once per column group, not
//     once per panel): two LDS buffers, panel p + 1 arrives by LDS-DMA while panel p is multiplied; one barrier per panel and a
//     COUNTED vmcnt wait (the only younger operations are the NCT x SL output stores of the previous panel, always issued —
//     rows / columns out of range store to a trash line — so the next panel's DMA is never drained behind them);
//   * blocks that share rows are 8 apart in the grid (same XCD by the observed b % 8 placement): the other column groups'
//     input comes out of that XCD's L2.
// Same entry contract as lora_amd_linear_ws (one site), same packed weight (or the row-major weight itself: site.reserved = 1);
// the two kernels are routed per shape.
// Measured (profiles/r05_kbench_xs.log): at (16384, 320, 320) the library GEMM, the weight-stationary kernel and this one all
// take 12-13 us for 21 MB — launch, load, multiply and store phases of ONE wave of workgroups do not overlap, the problem is
// two bandwidth-delay products small; the wide sites (N >= 4 K) are where the input-resident loop pays.
#include <algorithm>

#include "common.hpp"
#include "gemm_xs.h"

namespace lora_amd {
namespace {

typedef float xf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int xu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int xu32x2 __attribute__((ext_vector_type(2)));

template <class E> struct XsMfma;
template <> struct XsMfma<bf16_t> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  __device__ static xf32x4 mma(frag a, frag b, xf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct XsMfma<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  __device__ static xf32x4 mma(frag a, frag b, xf32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <class E>
__device__ __forceinline__ typename XsMfma<E>::frag xs_frag(xu32x4 v) {
  union { typename XsMfma<E>::frag f; xu32x4 u; } c;
  c.u = v;
  return c.f;
}

__device__ __forceinline__ void xs_glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                   (void __attribute__((address_space(3))) *)lds_wave_base, 16, 0, 0);
}

// 4 f32 -> (hi, lo) 16-bit parts packed as two dwords each
template <class E>
__device__ __forceinline__ void xs_split4(const float (&v)[4], xu32x2 &hi, xu32x2 &lo) {
  union { typename E::storage s[4]; xu32x2 u; } h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h.s[e] = E::from_f(v[e]);
    l.s[e] = E::from_f(v[e] - E::to_f(h.s[e]));
  }
  hi = h.u;
  lo = l.u;
}

// the factor / bias values of a workgroup's columns live in LDS for the whole kernel (no global load inside the panel loop: a
// value loaded there and carried to the next iteration made the compiler drain vmcnt to 0 at the loop's back edge)
constexpr int kXsUpFloats = 8192;   // columns x r4 (r4 = rank rounded up to 4) per workgroup
constexpr int kXsBiasCols = 2048;

struct XsArgs {
  const void *x;
  int64_t ldx, M;
  int32_t npanels, nrb, pg, ncg;   // panels of the site, row blocks, panels per workgroup, column groups
  int32_t nct_total, reserved;     // 16-column tiles the packed operand holds (its zero padding included)
  lora_amd_ws_site site;
};

// Stores of rows / columns out of range go here: a store that is always issued keeps the per-panel VMEM operation count exact,
// which the counted s_waitcnt of the panel loop relies on; every lane may write the same 8 bytes.
__device__ char g_xs_trash[64];

// KF = K / 32; NCT = 16-column tiles per panel; SL = 16-row slabs per wave; FL = factor layout (0: down [r, K], up [N, r];
// 3: down [K, r], up [r, N]: the input-gradient call); DROP as in csrc/gemm_ws.hip: forward — the rank term keeps its own
// accumulator and is multiplied by the lane's mask values; backward — the input fragments that feed T are ANDed with the
// forward's mask, 1 / (1 - p) goes into T.  LORA = false: plain Y = X B^T + bias (the merged-weight sites).
template <class E, int KF, int NCT, int SL, int FL, bool DROP, bool LORA>
__global__ __launch_bounds__(256, 1) void linear_xs_kernel(const XsArgs a) {
  using S = typename E::storage;
  constexpr int K = KF * 32;
  constexpr int PANEL = NCT * KF * 1024;             // bytes of one panel image
  constexpr int TF = (KF + 3) / 4;                   // k-steps of `down` one wave converts
  constexpr int NDMA = (NCT * KF + 3) / 4;           // 1 KB pieces of a panel per wave
  constexpr int NST = NCT * SL;                      // output stores per wave and panel (always issued)
  static_assert(2 * PANEL + 2 * KF * 1024 + kXsUpFloats * 4 + kXsBiasCols * 2 <= 160 * 1024, "LDS budget");
  static_assert(NST <= 63, "vmcnt field");
  // separate objects: the LDS-DMA writes `smem` only, and the compiler puts an s_waitcnt vmcnt(0) in front of every LDS read
  // it cannot prove disjoint from a DMA destination
  __shared__ __attribute__((aligned(1024))) char smem[2 * PANEL];                     // the two panel images
  __shared__ __attribute__((aligned(16))) char sdown[LORA ? 2 * KF * 1024 : 16];      // [KF][hi 1 KB | lo 1 KB]
  __shared__ __attribute__((aligned(16))) float sup[LORA ? kXsUpFloats : 4];          // [columns of this workgroup][r4]: scale * up
  __shared__ __attribute__((aligned(16))) S sbias[kXsBiasCols];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  // blocks b and b + 8 share their rows (same XCD): b = (rb / 8) * 8 ncg + cg * 8 + rb % 8
  const int ncg = a.ncg;
  const int grp = blockIdx.x / (8 * ncg), rem = blockIdx.x - grp * 8 * ncg;
  const int cg = rem >> 3, rb = grp * 8 + (rem & 7);
  if (rb >= a.nrb) return;
  const lora_amd_ws_site &st = a.site;
  const int N = st.N, r = st.r;
  const int64_t M = a.M, ldx = a.ldx;
  const S *x = reinterpret_cast<const S *>(a.x);
  const int p_begin = cg * a.pg, p_end = min(p_begin + a.pg, a.npanels);
  const int64_t row0 = ((int64_t)rb * 4 + wave) * (16 * SL);
  const bool rowmajor = st.reserved == 1;            // the weight itself ([N, K] rows) instead of the packed operand

  // panel pn -> LDS buffer `buf`: 1 KB fragment images dealt to the waves
  const S *wbase = reinterpret_cast<const S *>(st.wp);
  auto issue_panel = [&](int pn, int buf) {
    char *dst = smem + buf * PANEL;
    if (!rowmajor) {
#pragma unroll
      for (int q = 0; q < NDMA; ++q) {
        const int piece = wave + 4 * q;                // = ct * KF + s
        if (piece < NCT * KF) {
          const int ct = piece / KF, s = piece - ct * KF;
          const int ctg = min(pn * NCT + ct, a.nct_total - 1);   // a tile past the packed operand: its columns are >= N
          xs_glds16(wbase + ((int64_t)ctg * KF + s) * 512 + lane * 8, dst + piece * 1024);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NDMA; ++q) {
        const int piece = wave + 4 * q;                // = ct * KF + s
        if (piece < NCT * KF) {
          const int ct = piece / KF, s = piece - ct * KF;
          const int n = pn * NCT * 16 + ct * 16 + l15;
          xs_glds16(wbase + (int64_t)(n < N ? n : N - 1) * K + s * 32 + lg * 8, dst + piece * 1024);
        }
      }
    }
  };
  issue_panel(p_begin, 0);
  // ---- this wave's rows: every 16-byte piece of its SL slabs, all in flight at once
  xu32x4 xr[SL][KF];
  bool rok[SL];
#pragma unroll
  for (int sl = 0; sl < SL; ++sl) {
    const int64_t row = row0 + sl * 16 + l15;
    rok[sl] = row < M;
    const S *xp = x + (rok[sl] ? row : M - 1) * ldx + lg * 8;
#pragma unroll
    for (int s = 0; s < KF; ++s) xr[sl][s] = *gl(reinterpret_cast<const xu32x4 *>(xp + s * 32));
  }
  // ---- small operands
  constexpr bool dn_kr = FL & 1, up_rk = FL & 2;
  const float scale = st.scale, t_scale = st.t_scale;
  const uint64_t doff = DROP ? dropout_offset(st.offset, st.offset_dev) : 0;
  const uint32_t dthr = (uint32_t)(st.dropout_p * 65536.0f + 0.5f);
  const float dkeep = DROP ? 1.0f / (1.0f - st.dropout_p) : 1.0f;
  const float *upp = st.up;
  const S *biasp = reinterpret_cast<const S *>(st.bias);
  float draw[LORA ? TF : 1][8];
  const int r4 = (r + 3) & ~3;
  const int c_begin = p_begin * NCT * 16, ncols = (p_end - p_begin) * NCT * 16;   // this workgroup's columns
  if (LORA) {
    const float *downp = st.down;
    const int rank = l15 < r ? l15 : r - 1;
#pragma unroll
    for (int q = 0; q < TF; ++q) {
      const int s = wave + 4 * q, sc = s < KF ? s : KF - 1;
      if (dn_kr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) draw[q][e] = *gl(downp + (int64_t)(sc * 32 + lg * 8 + e) * r + rank);
      } else {
        const float4 p0 = gl_ld4(downp + (int64_t)rank * K + sc * 32 + lg * 8), p1 = gl_ld4(downp + (int64_t)rank * K + sc * 32 + lg * 8 + 4);
        draw[q][0] = p0.x; draw[q][1] = p0.y; draw[q][2] = p0.z; draw[q][3] = p0.w;
        draw[q][4] = p1.x; draw[q][5] = p1.y; draw[q][6] = p1.z; draw[q][7] = p1.w;
      }
    }
  }
  // ---- scale * up and the bias of this workgroup's columns -> LDS
  if (LORA) {
    for (int i = tid; i < ncols * r4; i += 256) {
      const int c = i / r4, rk = i - c * r4, nn = c_begin + c;
      float u = 0.f;
      if (rk < r && nn < N) u = (up_rk ? *gl(upp + (int64_t)rk * N + nn) : *gl(upp + (int64_t)nn * r + rk)) * scale;
      sup[i] = u;
    }
  }
  for (int i = tid; i < ncols; i += 256) {
    const int nn = c_begin + i;
    sbias[i] = (biasp != nullptr && nn < N) ? *gl(biasp + nn) : E::from_f(0.f);
  }
  // ---- `down` -> hi / lo fragments in LDS (each wave its share of the k-steps)
  if (LORA) {
#pragma unroll
    for (int q = 0; q < TF; ++q) {
      const int s = wave + 4 * q;
      if (s < KF) {
        union { Chunk8<E> c; xu32x4 u; } h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = l15 < r ? draw[q][e] : 0.f;
          h.c.v[e] = E::from_f(v);
          l.c.v[e] = E::from_f(v - E::to_f(h.c.v[e]));
        }
        *reinterpret_cast<xu32x4 *>(sdown + s * 2048 + lane * 16) = h.u;
        *reinterpret_cast<xu32x4 *>(sdown + s * 2048 + 1024 + lane * 16) = l.u;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first panel's pieces of this wave have landed, its rows are here
  __syncthreads();

  // ---- T^T = down X^T: lane (row, g) ends with T[row][4g .. 4g+3]
  xu32x4 bt1[LORA ? SL : 1], bt2[LORA ? SL : 1];
  if (LORA) {
    xf32x4 tacc[SL];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) tacc[sl] = xf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KF; ++s) {
      const xu32x4 dh = *reinterpret_cast<const xu32x4 *>(sdown + s * 2048 + lane * 16);
      const xu32x4 dl = *reinterpret_cast<const xu32x4 *>(sdown + s * 2048 + 1024 + lane * 16);
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        xu32x4 xv = xr[sl][s];
        if (DROP && FL == 3) {  // the forward's mask on G: the fragment's 8 k-slots are one Philox chunk of row `row`
          const int64_t row = row0 + sl * 16 + l15;
          uint32_t rr[4];
          Philox ph(st.seed);
          ph((uint64_t)(row * (int64_t)(K >> 3) + 4 * s + lg), doff, rr);
#pragma unroll
          for (int w = 0; w < 4; ++w)
            xv[w] &= ((rr[w] & 0xFFFFu) >= dthr ? 0x0000FFFFu : 0u) | ((rr[w] >> 16) >= dthr ? 0xFFFF0000u : 0u);
        }
        tacc[sl] = XsMfma<E>::mma(xs_frag<E>(dh), xs_frag<E>(xv), tacc[sl]);
        tacc[sl] = XsMfma<E>::mma(xs_frag<E>(dl), xs_frag<E>(xv), tacc[sl]);
      }
    }
    const float tm = (DROP && FL == 3) ? dkeep : 1.0f;
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      float tv[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) tv[v] = tacc[sl][v] * tm;
      if (cg == 0 && st.t_out != nullptr && rok[sl]) {
        float *tp = st.t_out + (row0 + sl * 16 + l15) * r + 4 * lg;
        if ((r & 3) == 0) {
          if (4 * lg < r) gl_st4(tp, tv[0] * t_scale, tv[1] * t_scale, tv[2] * t_scale, tv[3] * t_scale);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) if (4 * lg + v < r) *gl(tp + v) = tv[v] * t_scale;
        }
      }
      xu32x2 hi, lo;
      xs_split4<E>(tv, hi, lo);
      bt1[sl] = xu32x4{hi[0], hi[1], lo[0], lo[1]};
      bt2[sl] = xu32x4{hi[0], hi[1], 0u, 0u};
    }
  }

  // ---- the column group's panels
  S *y = reinterpret_cast<S *>(st.y);
  const int64_t ldy = st.ldy;
  for (int pn = p_begin; pn < p_end; ++pn) {
    const int buf = (pn - p_begin) & 1;
    if (pn > p_begin) {
      // the loads issued one iteration ago (this panel's image, its factor / bias pieces) are OLDER than the NST stores that
      // followed them: a counted wait leaves the stores in flight.  (No accumulate form here: a load of y between the stores —
      // even one under a branch that is never taken — put an s_waitcnt vmcnt(0) in front of EVERY store: 63 us instead of
      // ~25 for 96 MB, profiles/r05_kbench_xs_first.log.  Grouped input gradients stay on lora_amd_linear_ws.)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
      // every wave's pieces are in, and every wave is done with the buffer the next DMA overwrites.  An LDS-only barrier:
      // __syncthreads() drains vmcnt to 0 — the previous panel's stores, a memory round trip per panel (measured: 63 us for
      // 96 MB with it, profiles/r05_kbench_xs_first.log)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (pn + 1 < p_end) issue_panel(pn + 1, buf ^ 1);
    asm volatile("" ::: "memory");
    const char *pan = smem + buf * PANEL;
    const int n0 = pn * NCT * 16;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int ncol = n0 + ct * 16 + 4 * lg;  // this lane's 4 output columns
      xf32x4 acc[SL];
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) acc[sl] = xf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KF; ++s) {
        const xu32x4 wf = *reinterpret_cast<const xu32x4 *>(pan + (ct * KF + s) * 1024 + lane * 16);
#pragma unroll
        for (int sl = 0; sl < SL; ++sl) acc[sl] = XsMfma<E>::mma(xs_frag<E>(wf), xs_frag<E>(xr[sl][s]), acc[sl]);
      }
      xu32x4 ua1 = xu32x4{0u, 0u, 0u, 0u}, ua2 = ua1;
      const int cl = n0 - c_begin + ct * 16;   // column of this tile inside the workgroup's range
      if (LORA) {
        float uv[4] = {0.f, 0.f, 0.f, 0.f};
        if (4 * lg < r4) {
          const float4 u4 = *reinterpret_cast<const float4 *>(sup + (cl + l15) * r4 + 4 * lg);
          uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w;
        }
        xu32x2 hi, lo;
        xs_split4<E>(uv, hi, lo);
        ua1 = xu32x4{hi[0], hi[1], hi[0], hi[1]};
        ua2 = xu32x4{lo[0], lo[1], 0u, 0u};
      }
      union { S s[4]; xu32x2 v; } bb;
      bb.v = *reinterpret_cast<const xu32x2 *>(sbias + cl + 4 * lg);
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const int64_t row = row0 + sl * 16 + l15;
        xf32x4 val = acc[sl];
        if (LORA) {
          if (DROP && FL == 0) {
            xf32x4 br = XsMfma<E>::mma(xs_frag<E>(ua1), xs_frag<E>(bt1[sl]), xf32x4{0.f, 0.f, 0.f, 0.f});
            br = XsMfma<E>::mma(xs_frag<E>(ua2), xs_frag<E>(bt2[sl]), br);
            uint32_t rr[4];
            Philox ph(st.seed);
            ph((uint64_t)((row * (int64_t)N + ncol) >> 3), doff, rr);   // chunk = 8 consecutive columns of a row of [M, N]
            const int h4 = (ncol >> 2) & 1;                              // which half of the chunk
            const uint32_t w0 = h4 ? rr[2] : rr[0], w1 = h4 ? rr[3] : rr[1];
            val[0] += ((w0 & 0xFFFFu) >= dthr) ? br[0] * dkeep : 0.f;
            val[1] += ((w0 >> 16) >= dthr) ? br[1] * dkeep : 0.f;
            val[2] += ((w1 & 0xFFFFu) >= dthr) ? br[2] * dkeep : 0.f;
            val[3] += ((w1 >> 16) >= dthr) ? br[3] * dkeep : 0.f;
          } else {
            val = XsMfma<E>::mma(xs_frag<E>(ua1), xs_frag<E>(bt1[sl]), val);
            val = XsMfma<E>::mma(xs_frag<E>(ua2), xs_frag<E>(bt2[sl]), val);
          }
        }
        const bool ok = rok[sl] && ncol < N;
        S *yp = ok ? y + row * ldy + ncol : reinterpret_cast<S *>(g_xs_trash);
        union { S s[4]; xu32x2 v; } o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(val[e] + E::to_f(bb.s[e]));
        *gl(reinterpret_cast<xu32x2 *>(yp)) = o.v;   // always issued: exactly NST stores per wave and panel
      }
    }
  }
}

struct XsCfg { int KF, NCT; };
inline bool xs_cfg(int K, XsCfg *c) {
  switch (K) {
    case 320: *c = {10, 5}; return true;   // 2 x 50 KB panel images + 20 KB of down fragments
    case 640: *c = {20, 2}; return true;   // 2 x 40 KB + 40 KB
    default: return false;
  }
}

template <class E, int KF, int NCT, int SL>
void xs_launch(const XsArgs &a, int fl, bool drop, bool lora, dim3 grid, hipStream_t st) {
#define XS_GO(FL, DROP, LORA) hipLaunchKernelGGL((linear_xs_kernel<E, KF, NCT, SL, FL, DROP, LORA>), grid, dim3(256), 0, st, a)
  if (!lora) XS_GO(0, false, false);
  else if (fl == 0) { if (drop) XS_GO(0, true, true); else XS_GO(0, false, true); }
  else { if (drop) XS_GO(3, true, true); else XS_GO(3, false, true); }
#undef XS_GO
}

}  // namespace
}  // namespace lora_amd

using namespace lora_amd;

static int g_xs_sl = 0, g_xs_pg = 0;   // kbench overrides (0 = choose): slabs per wave, panels per workgroup

extern "C" void lora_amd_xs_set_tuning(int32_t slabs, int32_t panels_per_group) {
  g_xs_sl = slabs;
  g_xs_pg = panels_per_group;
}

extern "C" int lora_amd_xs_config(int32_t K, int32_t *panel_cols, int32_t *block_rows) {
  XsCfg c;
  if (!xs_cfg(K, &c)) return 0;
  if (panel_cols) *panel_cols = c.NCT * 16;
  if (block_rows) *block_rows = 64 * 2;
  return 1;
}

extern "C" int lora_amd_linear_xs(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t act_dtype,
                                  const lora_amd_ws_site *site, void *stream) {
  XsCfg c;
  LORA_AMD_CHECK(xs_cfg(K, &c), LORA_AMD_EINVAL, "linear_xs: contraction length %d has no input-stationary kernel", K);
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "linear_xs: bf16/f16 only");
  LORA_AMD_CHECK(x && site && M > 0, LORA_AMD_EINVAL, "linear_xs: bad argument");
  LORA_AMD_CHECK(((uintptr_t)x % 16) == 0 && ldx % 8 == 0 && ldx >= K, LORA_AMD_EINVAL,
                 "linear_xs: input rows must be 16-byte aligned");
  const lora_amd_ws_site &q = *site;
  const bool lora = q.down != nullptr && q.up != nullptr && q.r > 0;
  LORA_AMD_CHECK(q.wp && q.y && q.N > 0, LORA_AMD_EINVAL, "linear_xs: null pointer");
  LORA_AMD_CHECK(!lora || (q.r >= 1 && q.r <= 16), LORA_AMD_ERANK, "linear_xs: rank %d outside [1,16]", q.r);
  LORA_AMD_CHECK(q.N % 4 == 0 && q.ldy % 4 == 0 && ((uintptr_t)q.y % 8) == 0 && ((uintptr_t)q.wp % 16) == 0 &&
                     ((uintptr_t)q.bias % 8) == 0 && ((uintptr_t)q.t_out % 16) == 0 &&
                     (!lora || (((uintptr_t)q.down % 16) == 0 && ((uintptr_t)q.up % 16) == 0)),
                 LORA_AMD_EINVAL, "linear_xs: N, ldy must be multiples of 4, pointers aligned");
  LORA_AMD_CHECK(q.dropout_p >= 0.f && q.dropout_p < 1.f, LORA_AMD_EINVAL, "linear_xs: dropout p=%f", q.dropout_p);
  LORA_AMD_CHECK(q.reserved == 0 || q.reserved == 1, LORA_AMD_EINVAL, "linear_xs: reserved = 0 (packed weight) or 1 (row-major)");
  const int fl = q.flayout;
  LORA_AMD_CHECK(fl == 0 || fl == 3, LORA_AMD_EINVAL,
                 "linear_xs: factor layout 0 (forward) or 3 (input gradient); no accumulate form (use lora_amd_linear_ws)");
  const bool drop = lora && q.dropout_p > 0.f;
  LORA_AMD_CHECK(!drop || fl == 3 || q.N % 8 == 0, LORA_AMD_EINVAL, "linear_xs: dropout needs N %% 8 == 0");
  XsArgs a;
  a.x = x; a.ldx = ldx; a.M = M; a.site = q;
  const int pcols = c.NCT * 16;
  a.npanels = (q.N + pcols - 1) / pcols;
  {
    // the packed operand is padded to lora_amd_ws_packed_elems' panel (the kernel clamps the tiles of its last panel to it)
    int bn = 0;
    lora_amd_ws_config(K, &bn, nullptr);
    a.nct_total = (int)((int64_t)(q.N + bn - 1) / bn * bn / 16);
    a.reserved = 0;
  }
  // slabs per wave and panels per workgroup (profiles/r05_kbench_xs.log): ONE round of workgroups (LDS holds one per CU), the
  // input read as few times as that allows; 64-row waves only where the site is wide enough to amortise their registers
  const int sl_max = K == 320 ? 4 : 2;
  int sl = g_xs_sl > 0 ? g_xs_sl : (K == 320 ? (q.N >= 2048 && M >= 12288 ? 4 : 2) : 2);
  if (g_xs_sl == 0) while (sl > 1 && (M + 64 * sl - 1) / (64 * sl) < 32) sl >>= 1;   // few rows: smaller blocks
  if (sl > sl_max) sl = sl_max;
  if (sl != 1 && sl != 2 && sl != 4) sl = 2;
  a.nrb = (int)((M + 64 * sl - 1) / (64 * sl));
  int pg = g_xs_pg;
  if (pg <= 0) {
    const int groups = std::max(1, 256 / std::max(1, a.nrb));   // column groups that keep the grid within one round
    pg = (a.npanels + groups - 1) / groups;
  }
  {  // the workgroup's factor / bias columns must fit their LDS arrays
    const int r4 = lora ? (q.r + 3) & ~3 : 4;
    const int cap = std::min(kXsUpFloats / (pcols * r4), kXsBiasCols / pcols);
    if (pg > cap) pg = cap;
  }
  if (pg > a.npanels) pg = a.npanels;
  if (pg < 1) pg = 1;
  a.pg = pg;
  a.ncg = (a.npanels + pg - 1) / pg;
  const dim3 grid((unsigned)(((a.nrb + 7) / 8) * 8 * a.ncg));
  hipStream_t st = (hipStream_t)stream;
#define XS_DISPATCH(E)                                                                         \
  do {                                                                                         \
    if (K == 320) {                                                                            \
      if (sl == 4) xs_launch<E, 10, 5, 4>(a, fl, drop, lora, grid, st);                        \
      else if (sl == 2) xs_launch<E, 10, 5, 2>(a, fl, drop, lora, grid, st);                   \
      else xs_launch<E, 10, 5, 1>(a, fl, drop, lora, grid, st);                                \
    } else {                                                                                   \
      if (sl >= 2) xs_launch<E, 20, 2, 2>(a, fl, drop, lora, grid, st);                        \
      else xs_launch<E, 20, 2, 1>(a, fl, drop, lora, grid, st);                                \
    }                                                                                          \
  } while (0)
  if (act_dtype == LORA_AMD_BF16) XS_DISPATCH(bf16_t); else XS_DISPATCH(f16_t);
#undef XS_DISPATCH
  return check_launch("lora_amd_linear_xs");
}
