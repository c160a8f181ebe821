// K1/K2 weight-stationary fused GEMM — OUT = IN B^T + bias + scale * (IN down^T) up^T for one or several sites that
// read the SAME input, in ONE launch, with the frozen operand resident in registers.
//
// replaces: lora_diffusion/lora.py:53-58 called on to_q / to_k / to_v with one tensor (three addmm + three low-rank
//           branches = 18 ATen launches, X read 6 times), and the per-site two-operand-streaming kernel of
//           gemm_fused.hip at the sites whose contraction is short (K = 320 / 640 / 768 / 1280).
//
// Why this shape.  At SD1.5's attention sites the contraction is 320..1280 deep: a K loop of 5..20 slab steps is all
// prologue and epilogue, and a workgroup that streams BOTH operands through an LDS ring spends its life waiting for
// the first slabs (gemm_fused.hip: 0.17 of the byte roof at 16384 x 320 x 320).  Here
//   * the frozen weight is pre-packed ONCE into MFMA fragment order (lora_amd_ws_pack: frozen weights never change
//     while training; 288 GB of HBM hold a second layout of the adapted sites) so a wave loads its whole
//     [64*CS/4 columns][K] panel with fully coalesced 1 KB instructions straight into VGPRs — W fragments stay in
//     registers for every row tile the workgroup processes (1 wave per SIMD, 512-register budget);
//   * only the activation streams: whole [BM][K] row tiles go HBM/L2 -> LDS by LDS-DMA (full 128-byte lines,
//     XOR-swizzled on the source address) into a 3-slot ring, two tiles ahead of the one being multiplied;
//   * the MFMA is issued with the operands SWAPPED (A = W fragment, B = X fragment), so an accumulator register holds
//     4 CONSECUTIVE output columns of one row: the epilogue stores 8 bytes per lane straight from registers, no LDS
//     transpose, no barrier;
//   * T = IN down^T is split over the 4 waves by k-step (each keeps its share of the f32 `down` as hi + lo 16-bit
//     fragments in registers) and meets in LDS once per tile; the rank-r term is ONE more MFMA per accumulator;
//   * sites that share the input (attn1 q/k/v, attn2 k/v) are panels of the same grid: grid = (panels of all sites) x
//     (row groups); consecutive blocks of a row group land on the same XCD (b % 8), so the input tile is fetched from
//     HBM once and re-read from that XCD's L2.
// Head-padded activations (round 6; configs[3]'s dropout sites around the attention core): the input's 16-byte pieces are
// fetched from their padded position (one magic multiply on the source address: the LDS tile is the dense one), the output's
// 8-byte column groups go to their padded position and every live group of a head also writes one ZERO group into the head's
// pad (D - d <= d: the first D - d live groups own one pad group each, the others repeat one of them — same value, and the
// store count per tile stays exact for the counted waits).  Removes the pad / slice copies lora.forward_heads made around
// this kernel (0.75 ms of a 35 ms configs[3] step).
// The same entry point computes a site's input gradient dX = G W + scale (G up) down, Gt = scale G up when given the
// weight packed in the transposed orientation (contraction over N) and the factors with their roles swapped.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace lora_amd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <class E> struct WsMfma;
template <> struct WsMfma<bf16_t> {
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WsMfma<f16_t> {
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int kWsThreads = 256;
constexpr int kWsSlots = 3;

__device__ inline void ws_glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)gsrc,
                                   (void __attribute__((address_space(3))) *)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ inline void ws_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class E>
__device__ inline typename WsMfma<E>::frag ws_make_frag(const float (&v)[8]) {
  union { typename WsMfma<E>::frag f; typename E::storage s[8]; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.s[i] = E::from_f(v[i]);
  return u.f;
}

struct WsArgs {
  const void *x;
  int64_t ldx, M;
  int32_t nsites, row_groups, ntiles, total_panels;
  int32_t x_hc, x_hp;      // head-padded input: 16-byte chunks per head (live, padded); 0 = dense rows
  uint32_t x_magic;        // ceil(2^32 / x_hc), 0 = dense
  int32_t reserved;
  lora_amd_ws_site site[LORA_AMD_WS_MAX_SITES];
};

// Stores of rows past M go here (a store that is always issued keeps the per-tile VMEM operation count exact, which
// is what the counted s_waitcnt of the input ring relies on); every lane may write the same 16 bytes.
__device__ char g_ws_trash[64];

// An opaque use: the optimiser may not sink the load that produced `v` into a later branch (where every such load would
// wait for its own round trip: 60 serialized L2/HBM latencies in the first version of this kernel's prologue).
__device__ inline float ws_pin(float v) { asm volatile("" : "+v"(v)); return v; }

// KF = K / 32 (k-steps of one MFMA), CS = 16-column subtiles per wave (BN = 64 * CS), RS = 16-row subtiles per tile
// (BM = 16 * RS).  One workgroup = 4 waves side by side along the output columns, all sharing the tile's rows.
//
// Load schedule.  Prologue: input tiles 0 and 1 (LDS-DMA), the small operands, the panel; one vmcnt(0).  Steady state:
// tile it + 2 is fetched at the end of iteration it into the slot iteration it - 1 read; at the top of iteration `it`
// the operations younger than tile it's DMA are exactly the previous tile's RS*CS output stores and the next tile's
// NDMA pieces (vmcnt retires in issue order; rows past M store to a trash line so that the count is exact), so the
// wait is a counted one and the next tile stays in flight across it.
// FL = factor layout of every site of the launch (0: down [r,K], up [N,r]; 3: down [K,r], up [r,N] — the input-gradient
// call); RM = how a column's ranks are fetched from `up` when FL = 0: 1: r == 4 (one 16-byte load), 2: r % 4 == 0 (two),
// 0: element by element.  Compile-time so that no load sits behind a branch whose join would drain the queue.
// DROP: nn.Dropout on the low-rank branch (lora.py:45, 56), mask regenerated from (seed, offset) exactly as
// lora_amd_linear_fwd / lora_amd_linear_bwd_g index it (Philox chunk = 8 consecutive columns of a row of the [M, N] output):
//   forward  — the rank-r term goes through its own MFMA, is multiplied by the lane's 4 mask values and added;
//   backward — the G fragments that feed Gt = scale (mask*G) up are ANDed with the mask bits (one Philox call per
//              fragment: its 8 k-slots are one chunk), 1/(1-p) is applied to T; the frozen product G W reads G unmasked.
// HD: head-padded activations — 0 none, 1 the input's rows, 2 the output's (own instantiations: the address arithmetic costs
// ~28 registers, which the dense kernels at 500+ of 512 do not have; only the dropout sites are routed here with heads).
template <class E, int KF, int CS, int RS, int FL, int RM, bool DROP, int HD = 0>
__global__ __launch_bounds__(kWsThreads, 1) void linear_ws_kernel(const WsArgs a) {
  using S = typename E::storage;
  using F = typename WsMfma<E>::frag;
  constexpr int K = KF * 32, BM = RS * 16, BN = CS * 64;
  constexpr int NC = K / 64;                    // 128-byte chunks per input row
  constexpr int SLOT = BM * K * 2;              // bytes of one input tile in LDS: [NC][BM][128 B]
  constexpr int NDMA = (BM / 8) * NC / 4;       // 1 KB pieces (8 rows x 128 B) per wave and tile
  constexpr int TF = (KF + 3) / 4;              // k-steps of T owned by one wave
  constexpr int NY = RS * CS;                   // output stores per wave and tile
  static_assert(K % 64 == 0 && (BM / 8) * NC % 4 == 0, "tile geometry");
  static_assert(kWsSlots * SLOT + 4 * BM * 16 * 4 <= 160 * 1024, "LDS budget");
  static_assert((HD == 2 ? 2 : 1) * NY + NDMA <= 63, "vmcnt field (a head-padded output doubles the stores of a tile)");
  __shared__ __attribute__((aligned(1024))) char smem[kWsSlots * SLOT + 4 * BM * 16 * 4];
  float *tred = reinterpret_cast<float *>(smem + kWsSlots * SLOT);  // [4 waves][RS][64 lanes][4]: partial T^T

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int bid = blockIdx.x;
  const int pg = bid / a.row_groups, rg = bid - pg * a.row_groups;  // consecutive blocks: same rows, other panels
  int si = 0;
#pragma unroll
  for (int s = 1; s < LORA_AMD_WS_MAX_SITES; ++s)
    if (s < a.nsites && pg >= a.site[s].panel_begin) si = s;
  const lora_amd_ws_site &st = a.site[si];
  const int panel = pg - st.panel_begin;
  const int N = st.N, r = st.r;
  const int n_wave = panel * BN + wave * (CS * 16);  // first output column of this wave
  const S *x = reinterpret_cast<const S *>(a.x);
  const int64_t M = a.M, ldx = a.ldx;
  const int my_tiles = rg < a.ntiles ? (a.ntiles - rg + a.row_groups - 1) / a.row_groups : 0;
  if (my_tiles == 0) return;

  // piece u of a tile for this wave: chunk column c, 8-row group g; lane -> row g*8 + lane/8, 16-byte slot lane%8
  // (source chunk (lane%8) ^ (row%8): the XOR swizzle lives on the global side, LDS stays linear).  Rows past M re-read
  // the last valid row.  Addressing: one 64-bit tile base, 32-bit in-tile offsets.
  const int ldx32 = (int)ldx;
  const int lane_row = lane >> 3, lane_sw = lane & 7;
  const uint32_t x_magic = a.x_magic;
  const int x_hc = a.x_hc, x_hp = a.x_hp;
  auto issue_tile = [&](int tile, int slot) {
    const int64_t m0t = (int64_t)tile * BM;
    const S *tbase = x + m0t * ldx;
    const int rows_valid = (int)min((int64_t)BM, M - m0t);
    char *sbase = smem + slot * SLOT;
#pragma unroll
    for (int u = 0; u < NDMA; ++u) {
      const int q = wave + 4 * u, c = q / (BM / 8), g = q - c * (BM / 8);  // wave-uniform
      const int rl = g * 8 + lane_row;
      const int rc = rl < rows_valid ? rl : rows_valid - 1;
      // dense 16-byte chunk c * 8 + (swizzled slot) of the row -> its position in a head-padded row (x_magic = 0: unchanged)
      if constexpr (HD == 1) {
        const int cl = c * 8 + (lane_sw ^ (rl & 7));
        const int qh = (int)__umulhi((uint32_t)cl, x_magic);
        ws_glds16(tbase + rc * ldx32 + ((qh * x_hp + (cl - qh * x_hc)) << 3), sbase + (c * BM + g * 8) * 128);
      } else {
        ws_glds16(tbase + rc * ldx32 + c * 64 + ((lane_sw ^ (rl & 7)) << 3), sbase + (c * BM + g * 8) * 128);
      }
    }
  };

  // ---- input tiles 0 and 1 -> LDS ring slots 0, 1 (oldest operations: they have the longest way to come)
  issue_tile(rg, 0);
  if (my_tiles > 1) issue_tile(rg + a.row_groups, 1);

  // ---- small operands: raw and unconditional (clamped addresses, nothing under a branch), as wide as
  // their layout allows — forward layouts take 16-byte / 8-byte loads (16 instructions, 54 registers at r = 4)
  constexpr bool dn_kr = FL & 1, up_rk = FL & 2;
  const float scale = st.scale, t_scale = st.t_scale;
  const Philox ph(st.seed);
  const uint64_t doff = DROP ? dropout_offset(st.offset, st.offset_dev) : 0;
  const uint32_t dthr = (uint32_t)(st.dropout_p * 65536.0f + 0.5f);
  const float dkeep = 1.0f / (1.0f - st.dropout_p);
  const float *downp = st.down, *upp = st.up;
  float *t_out = st.t_out;
  constexpr bool up_vec = !up_rk && RM != 0;  // ranks of one column are contiguous 16-byte groups
  float draw[TF][8], uraw[CS][8];
  u32x2 braw[CS];
#pragma unroll
  for (int q = 0; q < TF; ++q) {
    const int kf = wave + 4 * q;
    const int rank = l15 < r ? l15 : r - 1, kfc = kf < KF ? kf : KF - 1;
    if (dn_kr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) draw[q][e] = downp[(int64_t)(kfc * 32 + lg * 8 + e) * r + rank];
    } else {
      const f32x4 p0 = *reinterpret_cast<const f32x4 *>(downp + (int64_t)rank * K + kfc * 32 + lg * 8);
      const f32x4 p1 = *reinterpret_cast<const f32x4 *>(downp + (int64_t)rank * K + kfc * 32 + lg * 8 + 4);
      draw[q][0] = p0[0]; draw[q][1] = p0[1]; draw[q][2] = p0[2]; draw[q][3] = p0[3];
      draw[q][4] = p1[0]; draw[q][5] = p1[1]; draw[q][6] = p1[2]; draw[q][7] = p1[3];
    }
  }
  const S *biasp = reinterpret_cast<const S *>(st.bias != nullptr ? st.bias : st.wp);  // any valid address when absent
#pragma unroll
  for (int j = 0; j < CS; ++j) {
    const int n = n_wave + j * 16 + l15;
    const int nc = n < N ? n : N - 1;
    if (up_vec) {
      const int g0 = lg * 8 < r ? lg * 8 : 0, g1 = lg * 8 + 4 < r ? lg * 8 + 4 : 0;
      const f32x4 p0 = *reinterpret_cast<const f32x4 *>(upp + (int64_t)nc * r + g0);
      uraw[j][0] = p0[0]; uraw[j][1] = p0[1]; uraw[j][2] = p0[2]; uraw[j][3] = p0[3];
      if (RM == 2) {
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(upp + (int64_t)nc * r + g1);
        uraw[j][4] = p1[0]; uraw[j][5] = p1[1]; uraw[j][6] = p1[2]; uraw[j][7] = p1[3];
      } else {
        uraw[j][4] = uraw[j][5] = uraw[j][6] = uraw[j][7] = 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rank = lg * 8 + e;
        const int rc = rank < r ? rank : r - 1;
        uraw[j][e] = up_rk ? upp[(int64_t)rc * N + nc] : upp[(int64_t)nc * r + rc];
      }
    }
    const int nb = n_wave + j * 16 + lg * 4;  // 4 consecutive columns: all in range or all out (N % 4 == 0)
    braw[j] = *reinterpret_cast<const u32x2 *>(biasp + (nb < N ? nb : N - 4));
  }
  // ---- the frozen panel: 1 KB per (subtile, k-step), fully coalesced, straight into registers
  const S *wp = reinterpret_cast<const S *>(st.wp) + ((int64_t)(panel * 4 + wave) * CS * KF) * 512 + lane * 8;
  F wreg[CS][KF];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int j = 0; j < CS; ++j) wreg[j][kf] = *reinterpret_cast<const F *>(wp + (j * KF + kf) * 512);

  F dhi[TF], dlo[TF], ub[CS];
  float bv[CS][4];
  S *y = reinterpret_cast<S *>(st.y);
  const int64_t ldy = st.ldy;
  const bool cols8 = n_wave + CS * 16 <= N;  // every column of this wave is in range: unconditional 8-byte stores
  const bool accum = st.flayout & 4;          // OUT += ... (input gradients of several sites meeting in one dX)
  const bool has_bias = st.bias != nullptr;
  // head-padded output: 4-column groups per head, live (y_hd4) and padded (y_hp4); the launcher guarantees whole panels
  // (cols8 everywhere), no accumulation, D - d <= d <= 2 (D - d)
  const int y_hd4 = (st.y_heads & 0xffff) >> 2, y_hp4 = (int)((uint32_t)st.y_heads >> 16) >> 2;
  constexpr bool yh = HD == 2;
  const uint32_t y_magic = yh ? 0xFFFFFFFFu / (uint32_t)max(y_hd4, 1) + 1u : 0u;
  const int y_pp = y_hp4 - y_hd4;

  // the small operands (oldest loads: hipcc's counted wait leaves the input pieces and the panel in flight): pin,
  // select, convert — before the k-loop, so that only fragments (64 registers), not raw values, live through it
#pragma unroll
  for (int q = 0; q < TF; ++q) {
    const bool live = l15 < r && wave + 4 * q < KF;
    float dv[8], dl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = ws_pin(draw[q][e]);
      dv[e] = live ? v : 0.f;
      dl[e] = dv[e] - E::to_f(E::from_f(dv[e]));  // f32 master = hi + lo 16-bit parts: T as precise as f32 factors
    }
    dhi[q] = ws_make_frag<E>(dv);
    dlo[q] = ws_make_frag<E>(dl);
  }
#pragma unroll
  for (int j = 0; j < CS; ++j) {
    const int n = n_wave + j * 16 + l15;
    float uv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = ws_pin(uraw[j][e]);
      uv[e] = (lg * 8 + e < r && n < N) ? scale * v : 0.f;
    }
    ub[j] = ws_make_frag<E>(uv);
    {
      union { u32x2 v; S s[4]; } bb;
      bb.v = braw[j];
      asm volatile("" : "+v"(bb.v));
      const bool live = has_bias && n_wave + j * 16 + lg * 4 < N;
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = live ? E::to_f(bb.s[e]) : 0.f;
    }
  }

  for (int it = 0; it < my_tiles; ++it) {
    const int tile = rg + it * a.row_groups;
    const int64_t m0 = (int64_t)tile * BM;
    if (it == 0) {
      ws_wait_vmcnt<0>();  // prologue: panel, small operands and the first tiles all have to be here
    } else {
      // tile `it` has landed for this wave: younger operations = previous tile's output stores + next tile's pieces
      const bool has_next = it + 1 < my_tiles;
      if constexpr (yh) { if (has_next) ws_wait_vmcnt<2 * NY + NDMA>(); else ws_wait_vmcnt<2 * NY>(); }
      else if (cols8) { if (has_next) ws_wait_vmcnt<NY + NDMA>(); else ws_wait_vmcnt<NY>(); }
      else { if (has_next) ws_wait_vmcnt<NDMA>(); else ws_wait_vmcnt<0>(); }  // edge panel: store count not fixed
    }
    ws_barrier();  // ... and for every wave
    const char *xs = smem + (it % kWsSlots) * SLOT;
    f32x4 acc[RS][CS];
    f32x4 lr[DROP && FL == 0 ? RS : 1][DROP && FL == 0 ? CS : 1];
#pragma unroll
    for (int i = 0; i < RS; ++i)
#pragma unroll
      for (int j = 0; j < CS; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) {
      F xf[RS];
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int row = i * 16 + l15, c = (kf & 1) * 4 + lg;
        xf[i] = *reinterpret_cast<const F *>(xs + ((kf >> 1) * BM + row) * 128 + ((c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int j = 0; j < CS; ++j) acc[i][j] = WsMfma<E>::mma(wreg[j][kf], xf[i], acc[i][j]);
    }
    // T^T = down X^T: this wave's k-steps (kf = wave, wave + 4, ...), hi + lo; partials meet in LDS
    f32x4 tacc[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) tacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < TF; ++q) {
      const int kf = wave + 4 * q;  // wave-uniform
      if (kf < KF) {
#pragma unroll
        for (int i = 0; i < RS; ++i) {
          const int row = i * 16 + l15, c = (kf & 1) * 4 + lg;
          F xf = *reinterpret_cast<const F *>(xs + ((kf >> 1) * BM + row) * 128 + ((c ^ (row & 7)) << 4));
          if (DROP && FL == 3) {  // Gt = scale (mask * G) up: the fragment's 8 columns are one Philox chunk
            uint32_t rr[4];
            ph((uint64_t)(((m0 + row) * (int64_t)K + kf * 32 + lg * 8) >> 3), doff, rr);
            union { F f; u32x4 u; } mx;
            mx.f = xf;
#pragma unroll
            for (int w = 0; w < 4; ++w)
              mx.u[w] &= ((rr[w] & 0xFFFFu) >= dthr ? 0x0000FFFFu : 0u) | ((rr[w] >> 16) >= dthr ? 0xFFFF0000u : 0u);
            xf = mx.f;
          }
          tacc[i] = WsMfma<E>::mma(dhi[q], xf, tacc[i]);
          tacc[i] = WsMfma<E>::mma(dlo[q], xf, tacc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RS; ++i) *reinterpret_cast<f32x4 *>(tred + ((wave * RS + i) * 64 + lane) * 4) = tacc[i];
    ws_barrier();  // all partials written AND every wave is done reading this tile's slot
    // T^T as B operand: lane (m = l15, lg) needs ranks lg*8 .. lg*8+7 = partial lanes (2 lg, m) and (2 lg + 1, m);
    // at r == 4 (RM == 1) only lane group 0 and only its first four ranks are live
    const bool t_writer = panel == 0 && t_out != nullptr;
#pragma unroll
    for (int i = 0; i < RS; ++i) {
      float tv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (lg < (RM == 1 ? 1 : 2)) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const f32x4 lo = *reinterpret_cast<const f32x4 *>(tred + ((w * RS + i) * 64 + (2 * lg) * 16 + l15) * 4);
          tv[0] += lo[0]; tv[1] += lo[1]; tv[2] += lo[2]; tv[3] += lo[3];
          if (RM != 1) {
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(tred + ((w * RS + i) * 64 + (2 * lg + 1) * 16 + l15) * 4);
            tv[4] += hi[0]; tv[5] += hi[1]; tv[6] += hi[2]; tv[7] += hi[3];
          }
        }
      }
      if (DROP && FL == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tv[e] *= dkeep;
      }
      // T (f32) for the backward: row m0 + i*16 + l15; written by wave i % 4 of the site's first panel
      if (t_writer && (i & 3) == wave) {
        const int64_t m = m0 + i * 16 + l15;
        if (RM == 1) {
          if (lg == 0 && m < M)
            *reinterpret_cast<f32x4 *>(t_out + m * 4) = (f32x4){t_scale * tv[0], t_scale * tv[1], t_scale * tv[2], t_scale * tv[3]};
        } else if (lg < 2 && m < M) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (lg * 8 + e < r) t_out[m * r + lg * 8 + e] = t_scale * tv[e];
        }
      }
      const F tb = ws_make_frag<E>(tv);  // rounded to the activation dtype, as the reference's autocast does
      if (DROP && FL == 0) {
        // the rank-r term keeps its own accumulators (AGPRs, like acc): the mask is applied in the epilogue, where
        // the values pass through VGPRs anyway (VALU updates of acc here would pull all of it out of the AGPRs: spills)
#pragma unroll
        for (int j = 0; j < CS; ++j) lr[i][j] = WsMfma<E>::mma(ub[j], tb, (f32x4){0.f, 0.f, 0.f, 0.f});
      } else {
#pragma unroll
        for (int j = 0; j < CS; ++j) acc[i][j] = WsMfma<E>::mma(ub[j], tb, acc[i][j]);
      }
    }
    // mask * rank-r term of accumulator (i, j): the lane's 4 consecutive columns are half a Philox chunk
    auto dropped = [&](int i, int j) -> f32x4 {
      const int n0 = n_wave + j * 16 + lg * 4;
      uint32_t rr[4];
      ph((uint64_t)(((m0 + i * 16 + l15) * (int64_t)N + n0) >> 3), doff, rr);
      const uint32_t ra = (n0 & 4) ? rr[2] : rr[0], rb = (n0 & 4) ? rr[3] : rr[1];
      f32x4 o;
      o[0] = ((ra & 0xFFFFu) >= dthr ? dkeep : 0.f) * lr[i][j][0];
      o[1] = ((ra >> 16) >= dthr ? dkeep : 0.f) * lr[i][j][1];
      o[2] = ((rb & 0xFFFFu) >= dthr ? dkeep : 0.f) * lr[i][j][2];
      o[3] = ((rb >> 16) >= dthr ? dkeep : 0.f) * lr[i][j][3];
      return o;
    };
    // ---- epilogue: lane holds OUT[m = i*16 + l15][n_wave + j*16 + lg*4 + 0..3]
    if constexpr (yh) {
      // head-padded rows: group g of the dense row -> head g / y_hd4, position g % y_hd4 of its y_hp4 slots; the same lane
      // zeroes one pad group of that head (the first y_pp live groups own one each, the rest repeat one: 2 NY stores a tile)
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int64_t m = m0 + i * 16 + l15;
        const bool row_ok = m < M;
        S *yrow = y + (row_ok ? m : 0) * ldy;
#pragma unroll
        for (int j = 0; j < CS; ++j) {
          union { u32x2 v; S s[4]; } o;
          f32x4 val = acc[i][j];
          if (DROP && FL == 0) val += dropped(i, j);
#pragma unroll
          for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(val[e] + bv[j][e]);
          const int g4 = (n_wave + j * 16 + lg * 4) >> 2;
          const int qh = (int)__umulhi((uint32_t)g4, y_magic), u = g4 - qh * y_hd4;
          const int hb = qh * y_hp4, v = u < y_pp ? u : u - y_pp;
          u32x2 *live = row_ok ? reinterpret_cast<u32x2 *>(yrow + ((hb + u) << 2)) : reinterpret_cast<u32x2 *>(g_ws_trash);
          u32x2 *pad = row_ok ? reinterpret_cast<u32x2 *>(yrow + ((hb + y_hd4 + v) << 2)) : reinterpret_cast<u32x2 *>(g_ws_trash);
          *live = o.v;
          *pad = (u32x2){0u, 0u};
        }
      }
    } else if (cols8 && !accum) {  // the common case, branch-free: 8-byte stores, always issued (rows past M -> trash line)
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int64_t m = m0 + i * 16 + l15;
        S *yr = m < M ? y + m * ldy + n_wave + lg * 4 : reinterpret_cast<S *>(g_ws_trash);
        const int jstep = m < M ? 16 : 0;
#pragma unroll
        for (int j = 0; j < CS; ++j) {
          union { u32x2 v; S s[4]; } o;
          f32x4 val = acc[i][j];
          if (DROP && FL == 0) val += dropped(i, j);
#pragma unroll
          for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(val[e] + bv[j][e]);
          *reinterpret_cast<u32x2 *>(yr + j * jstep) = o.v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int64_t m = m0 + i * 16 + l15;
        const bool row_ok = m < M;
        S *yr = y + (row_ok ? m : M - 1) * ldy + n_wave + lg * 4;
#pragma unroll
        for (int j = 0; j < CS; ++j) {
          union { u32x2 v; S s[4]; } o, prev;
          float add[4] = {0.f, 0.f, 0.f, 0.f};
          if (accum) {
            if (cols8) {
              prev.v = *reinterpret_cast<const u32x2 *>(yr + j * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) add[e] = E::to_f(prev.s[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n_wave + j * 16 + lg * 4 + e < N) add[e] = E::to_f(yr[j * 16 + e]);
            }
          }
          f32x4 val = acc[i][j];
          if (DROP && FL == 0) val += dropped(i, j);
#pragma unroll
          for (int e = 0; e < 4; ++e) o.s[e] = E::from_f(val[e] + bv[j][e] + add[e]);
          if (cols8) {  // always issued: exactly NY stores per tile
            u32x2 *dst = row_ok ? reinterpret_cast<u32x2 *>(yr + j * 16) : reinterpret_cast<u32x2 *>(g_ws_trash);
            *dst = o.v;
          } else if (row_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n_wave + j * 16 + lg * 4 + e < N) yr[j * 16 + e] = o.s[e];
          }
        }
      }
    }
    // slot (it + 2) % 3 was read in iteration it - 1 (every wave is past the barrier above): fetch the tile two ahead
    // into it — the youngest operation of this iteration
    if (it + 2 < my_tiles) issue_tile(rg + (it + 2) * a.row_groups, (it + 2) % kWsSlots);
  }
}

// Pack B[n][k] (n < N rows of length K; element (n, k) at w[n * sn + k * sk]) into fragment order:
// out[(((panel*4 + wave)*CS + j)*KF + kf)*512 + lane*8 + e] = B[panel*BN + wave*CS*16 + j*16 + (lane & 15)][kf*32 + (lane >> 4)*8 + e]
// (zero beyond N).  One thread per 16-byte piece.
template <class E>
__global__ __launch_bounds__(256) void ws_pack_kernel(const typename E::storage *__restrict__ w, int64_t sn, int64_t sk,
                                                      int N, int KF, int CS, int64_t pieces,
                                                      typename E::storage *__restrict__ out) {
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < pieces; id += (int64_t)gridDim.x * 256) {
    const int lane = (int)(id & 63);
    int64_t f = id >> 6;
    const int kf = (int)(f % KF); f /= KF;
    const int j = (int)(f % CS); f /= CS;
    const int wave = (int)(f & 3);
    const int64_t panel = f >> 2;
    const int64_t n = (panel * 4 + wave) * CS * 16 + j * 16 + (lane & 15);
    const int k0 = kf * 32 + (lane >> 4) * 8;
    Chunk8<E> c;
#pragma unroll
    for (int e = 0; e < 8; ++e) c.v[e] = n < N ? w[n * sn + (int64_t)(k0 + e) * sk] : E::from_f(0.f);
    *reinterpret_cast<Chunk8<E> *>(out + id * 8) = c;
  }
}

struct WsCfg { int KF, CS, RS; };
static inline bool ws_cfg(int K, WsCfg *c) {
  switch (K) {
    case 320: *c = {10, 5, 4}; return true;
    case 640: *c = {20, 2, 2}; return true;
    case 768: *c = {24, 2, 2}; return true;
    case 1280: *c = {40, 1, 1}; return true;
    default: return false;
  }
}

}  // namespace lora_amd

using namespace lora_amd;

extern "C" int lora_amd_ws_config(int32_t K, int32_t *bn, int32_t *bm) {
  WsCfg c;
  if (!ws_cfg(K, &c)) return 0;
  if (bn) *bn = c.CS * 64;
  if (bm) *bm = c.RS * 16;
  return 1;
}

extern "C" int64_t lora_amd_ws_packed_elems(int32_t N, int32_t K) {
  WsCfg c;
  if (!ws_cfg(K, &c) || N <= 0) return 0;
  const int BN = c.CS * 64;
  return (int64_t)((N + BN - 1) / BN) * BN * K;
}

extern "C" int lora_amd_ws_pack(const void *w, int64_t stride_n, int64_t stride_k, int32_t N, int32_t K,
                                int32_t dtype, void *out, void *stream) {
  WsCfg c;
  LORA_AMD_CHECK(ws_cfg(K, &c), LORA_AMD_EINVAL, "ws_pack: contraction length %d has no weight-stationary kernel", K);
  LORA_AMD_CHECK(w && out && N > 0, LORA_AMD_EINVAL, "ws_pack: bad argument");
  LORA_AMD_CHECK(dtype == LORA_AMD_BF16 || dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "ws_pack: bf16/f16 weights only");
  const int64_t pieces = lora_amd_ws_packed_elems(N, K) / 8;
  const int grid = (int)std::min<int64_t>((pieces + 255) / 256, 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LORA_AMD_BF16)
    hipLaunchKernelGGL((ws_pack_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const __bf16 *>(w),
                       stride_n, stride_k, N, c.KF, c.CS, pieces, reinterpret_cast<__bf16 *>(out));
  else
    hipLaunchKernelGGL((ws_pack_kernel<f16_t>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const _Float16 *>(w),
                       stride_n, stride_k, N, c.KF, c.CS, pieces, reinterpret_cast<_Float16 *>(out));
  return check_launch("lora_amd_ws_pack");
}

extern "C" int lora_amd_linear_ws(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t act_dtype,
                                  const lora_amd_ws_site *sites, int32_t nsites, int32_t row_groups, void *stream) {
  return lora_amd_linear_ws_heads(x, ldx, M, K, 0, 0, act_dtype, sites, nsites, row_groups, stream);
}

// may a site's output (heads of d columns in slots of D) go through the padded epilogue: 8-byte groups, one pad group per live
// group at most, at most two live groups per pad group
static inline bool ws_y_heads_ok(int d, int D) { return d > 0 && d % 4 == 0 && D % 4 == 0 && D > d && D - d <= d && d <= 2 * (D - d) && D < 65536; }

extern "C" int lora_amd_linear_ws_heads(const void *x, int64_t ldx, int64_t M, int32_t K, int32_t x_head_dim, int32_t x_head_pad,
                                        int32_t act_dtype, const lora_amd_ws_site *sites, int32_t nsites, int32_t row_groups,
                                        void *stream) {
  WsCfg c;
  LORA_AMD_CHECK(ws_cfg(K, &c), LORA_AMD_EINVAL, "linear_ws: contraction length %d has no weight-stationary kernel", K);
  LORA_AMD_CHECK(act_dtype == LORA_AMD_BF16 || act_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "linear_ws: bf16/f16 only");
  LORA_AMD_CHECK(x && sites && nsites >= 1 && nsites <= LORA_AMD_WS_MAX_SITES && M > 0, LORA_AMD_EINVAL,
                 "linear_ws: bad argument (1..%d sites)", LORA_AMD_WS_MAX_SITES);
  LORA_AMD_CHECK(((uintptr_t)x % 16) == 0 && ldx % 8 == 0 && ldx >= K, LORA_AMD_EINVAL,
                 "linear_ws: input rows must be 16-byte aligned");
  LORA_AMD_CHECK((x_head_dim == 0 && x_head_pad == 0) ||
                     (x_head_dim >= 8 && x_head_dim % 8 == 0 && x_head_pad % 8 == 0 && x_head_pad >= x_head_dim &&
                      K % x_head_dim == 0 && ldx >= (int64_t)(K / x_head_dim) * x_head_pad),
                 LORA_AMD_EINVAL, "linear_ws: input head layout (%d in %d) does not fit K = %d, ldx = %lld", x_head_dim,
                 x_head_pad, K, (long long)ldx);
  bool drop = false;
  for (int s = 0; s < nsites; ++s) {
    LORA_AMD_CHECK(sites[s].dropout_p >= 0.f && sites[s].dropout_p < 1.f, LORA_AMD_EINVAL,
                   "linear_ws: site %d: dropout p=%f", s, sites[s].dropout_p);
    drop = drop || sites[s].dropout_p > 0.f;  // all sites of a launch or none (p = 0 sites would pay the Philox calls)
  }
  // forward with dropout at K = 320: the masked rank-r term keeps its own accumulators, which the 64-row tile has no
  // registers left for (495 of 512 in use) -> 32-row tiles
  // head-padded activations (own instantiations, dropout sites only): 1 = the input's rows, 2 = every site's output
  int hd = x_head_dim ? 1 : 0;
  for (int s = 0; s < nsites; ++s) {
    LORA_AMD_CHECK((sites[s].y_heads != 0) == (sites[0].y_heads != 0), LORA_AMD_EINVAL,
                   "linear_ws: site %d: head-padded outputs for every site of a launch or for none", s);
    if (sites[s].y_heads != 0) {
      LORA_AMD_CHECK(hd != 1, LORA_AMD_EINVAL, "linear_ws: head-padded input AND output in one launch is not built");
      hd = 2;
    }
  }
  LORA_AMD_CHECK(hd == 0 || drop, LORA_AMD_EINVAL,
                 "linear_ws: head-padded activations are built for the dropout sites (p > 0) only; p = 0 sites with heads take "
                 "lora_amd_linear_gemm_fwd_heads");
  // ... and with head layouts in either direction (their address arithmetic costs ~28 registers)
  const bool half_rows = drop && K == 320 && ((sites[0].flayout & 3) == 0 || hd != 0);
  if (half_rows) c.RS = 2;
  const int BN = c.CS * 64, BM = c.RS * 16;
  WsArgs a;
  a.x = x; a.ldx = ldx; a.M = M; a.nsites = nsites;
  a.x_hc = x_head_dim >> 3; a.x_hp = x_head_pad >> 3; a.reserved = 0;
  a.x_magic = x_head_dim ? (uint32_t)((0x100000000ull + (uint32_t)a.x_hc - 1) / (uint32_t)a.x_hc) : 0u;
  a.ntiles = (int)((M + BM - 1) / BM);
  int panels = 0;
  for (int s = 0; s < nsites; ++s) {
    a.site[s] = sites[s];
    const lora_amd_ws_site &q = sites[s];
    LORA_AMD_CHECK(q.wp && q.y && q.down && q.up && q.N > 0, LORA_AMD_EINVAL, "linear_ws: site %d: null pointer", s);
    LORA_AMD_CHECK(q.r >= 1 && q.r <= 16, LORA_AMD_ERANK, "linear_ws: site %d: rank %d outside [1,16]", s, q.r);
    LORA_AMD_CHECK(q.N % 4 == 0 && q.ldy % 4 == 0 && ((uintptr_t)q.y % 8) == 0 && ((uintptr_t)q.wp % 16) == 0 &&
                       ((uintptr_t)q.down % 16) == 0 && ((uintptr_t)q.up % 16) == 0 && ((uintptr_t)q.bias % 8) == 0 &&
                       ((uintptr_t)q.t_out % 16) == 0,
                   LORA_AMD_EINVAL, "linear_ws: site %d: N, ldy must be multiples of 4, pointers aligned", s);
    if (q.y_heads != 0) {
      const int d = q.y_heads & 0xffff, D = (int)((uint32_t)q.y_heads >> 16);
      LORA_AMD_CHECK(ws_y_heads_ok(d, D) && q.N % d == 0 && q.N % BN == 0 && !(q.flayout & 4) &&
                         q.ldy >= (int64_t)(q.N / d) * D,
                     LORA_AMD_EINVAL,
                     "linear_ws: site %d: head-padded output (%d in %d) needs whole %d-column panels, no accumulation, D - d <= d <= "
                     "2 (D - d), ldy >= heads * D", s, d, D, BN);
    }
    a.site[s].panel_begin = panels;
    panels += (q.N + BN - 1) / BN;
  }
  for (int s = nsites; s < LORA_AMD_WS_MAX_SITES; ++s) a.site[s] = a.site[0];
  a.total_panels = panels;
  // row groups (measured, scripts/kbench.py --what ws): a workgroup pays ~6 us to pull its panel into registers, so ONE
  // round of at most 256 workgroups (1 per CU: the panel occupies the register file) beats two uneven ones; when there
  // are more panels than that allows, whole rounds (workgroups % 256 == 0).  A multiple of 8, so that the panels of one
  // row group share an XCD (block b runs on XCD b % 8) and re-read the input tile from that XCD's L2.
  if (row_groups <= 0) {
    if (panels <= 32) {
      row_groups = std::max(8, (256 / panels) & ~7);
    } else {
      row_groups = 8;
      while ((panels * row_groups) % 256 != 0 && row_groups < 256) row_groups += 8;
    }
  }
  if (row_groups > a.ntiles) row_groups = a.ntiles;
  a.row_groups = row_groups;
  const unsigned grid = (unsigned)(panels * row_groups);
  hipStream_t st = (hipStream_t)stream;
  int fl = sites[0].flayout & 3, rm = 1;
  for (int s = 0; s < nsites; ++s) {
    LORA_AMD_CHECK((sites[s].flayout & 3) == fl && (fl == 0 || fl == 3), LORA_AMD_EINVAL,
                   "linear_ws: every site of a launch must use factor layout 0 (forward) or 3 (input gradient)");
    const int m = sites[s].r == 4 ? 1 : (sites[s].r % 4 == 0 ? 2 : 0);
    rm = m == 0 ? 0 : (rm == 0 ? 0 : std::max(rm, m));
  }
  if (fl == 3) rm = 0;
  for (int s = 0; s < nsites; ++s)
    LORA_AMD_CHECK(!drop || sites[s].N % 8 == 0, LORA_AMD_EINVAL,
                   "linear_ws: site %d: dropout needs N %% 8 == 0 (mask chunks of 8 columns)", s);
#define WS_L(E, KFV, CSV, RSV, FLV, RMV, DV, HV) \
  hipLaunchKernelGGL((linear_ws_kernel<E, KFV, CSV, RSV, FLV, RMV, DV, HV>), dim3(grid), dim3(kWsThreads), 0, st, a)
#define WS_D(E, KFV, CSV, RSV, DV, HV)                          \
  do {                                                          \
    if (fl == 3) WS_L(E, KFV, CSV, RSV, 3, 0, DV, HV);          \
    else if (rm == 1) WS_L(E, KFV, CSV, RSV, 0, 1, DV, HV);     \
    else if (rm == 2) WS_L(E, KFV, CSV, RSV, 0, 2, DV, HV);     \
    else WS_L(E, KFV, CSV, RSV, 0, 0, DV, HV);                  \
  } while (0)
#define WS_H(E, KFV, CSV, RSV) /* dropout sites: dense, head-padded input, head-padded output */ \
  do {                                                          \
    if (hd == 1) WS_D(E, KFV, CSV, RSV, true, 1);               \
    else if (hd == 2) WS_D(E, KFV, CSV, RSV, true, 2);          \
    else WS_D(E, KFV, CSV, RSV, true, 0);                       \
  } while (0)
#define WS(E, KFV, CSV, RSV)                                    \
  do {                                                          \
    if (drop) WS_H(E, KFV, CSV, RSV);                           \
    else WS_D(E, KFV, CSV, RSV, false, 0);                      \
  } while (0)
#define WS_K(E)                                                 \
  do {                                                          \
    if (K == 320 && half_rows) WS_H(E, 10, 5, 2);               \
    else if (K == 320 && drop) /* input gradient, dense */ WS_L(E, 10, 5, 4, 3, 0, true, 0); \
    else if (K == 320) WS_D(E, 10, 5, 4, false, 0);             \
    else if (K == 640) WS(E, 20, 2, 2);                         \
    else if (K == 768) WS(E, 24, 2, 2);                         \
    else WS(E, 40, 1, 1);                                       \
  } while (0)
  if (act_dtype == LORA_AMD_BF16) WS_K(bf16_t); else WS_K(f16_t);
#undef WS_K
#undef WS
#undef WS_H
#undef WS_D
#undef WS_L
  return check_launch("lora_amd_linear_ws");
}
