// K2 on the merged-weight path, matrix-core form: both factor gradients of every Linear adapter in one launch, each
// row of G and X read from HBM exactly ONCE.
//
// replaces: the autograd of lora_diffusion/lora.py:53-58 for lora_up.weight / lora_down.weight
//           (dUp = s G^T (X down^T), dDown = (s G up)^T X) when the forward ran on W + s up down (ops.MergedWeights),
//           and csrc/linear_fused.hip's linear_bwd_factors_self_ragged_kernel for 16-bit activations: that VALU pass
//           reads every row block twice (row-dot phase, column-sum phase; the second read misses the L2: FETCH_SIZE
//           2.0x algorithmic, 0.29 of the byte roof).
//
// One workgroup (4 waves) owns R = 64 (32) rows of one site.  With A = the narrower of (X, G) and B = the wider one, fa / fb
// the factor contracted against A's / B's columns (A = X: fa = down, fb = up):
//
//     TA = s A fa^T [R, r]        outB[j, c] = sum_m TA[m, j] B[m, c]       (A = X: T,  outB = dUp partial)
//     TB = s B fb^T [R, r]        outA[j, c] = sum_m TB[m, j] A[m, c]       (        Gt, outA = dDown partial)
//
//   * A's row block stays RESIDENT IN REGISTERS (<= 20 16-byte pieces per lane: 64 rows x 640 columns or 32 x 1280 over the
//     four waves), B streams through a ring of column-group slots; a wave owns the 32-column groups wave, wave + 4, ... of both
//     operands and is autonomous between four barriers (factors_reg_kernel below).
//   * a piece (16 rows x 64 bytes) is FETCHED with four consecutive lanes on one row (round 6: the MFMA operand order is 64
//     scattered 16-byte accesses per instruction, one lane per cycle through the L1's tag pipeline) and held in that layout;
//     every piece crosses the wave's own 32 x 32 LDS tile, which hands out both operand forms;
//   * both contractions are v_mfma_f32_16x16x32 (the rank padded to the 16 of the tile):
//       phase 1  D[row, j]  += Data[row, 32 cols] . F[j, 32 cols]^T      A operand = ds_read_b128 of the tile (row l & 15,
//                chunk l >> 4), B operand = a packed factor fragment (lora_amd_factor_pack: 1 KB per wave, L2-resident);
//       phase 2  D[col, j]  += Data^T[col, 32 rows] . T[32 rows, j]      A operand = two ds_read_b64_tr_b16 of the same tile
//                (the LDS transpose read of gfx950), B operand = T fragments from LDS.
//   * f32 precision is kept by splitting every 16-bit operand that is not data: factor = hi + lo, T = hi + lo (two MFMAs
//     into the same accumulator) — the matrix pipe is < 20 % busy at the HBM rate, so the split is free.
//   * f16 (round 6): f16 has 5 exponent bits — `up` starts at 0 (lora.py:50-51) and sits at ~1e-4 for hundreds of steps,
//     Gt = s G up at ~1e-8 .. 1e-3: their hi parts land near or in f16's subnormals and the lo parts vanish.  Every split
//     operand is therefore PRE-SCALED BY A POWER OF TWO and the scale folded back into the f32 result (exact):
//       factors  per (site, side): factor_absmax_kernel puts 2^e, e = 11 - floor(log2 max|f|), into the 16-byte tail of
//                the pack; the pack kernel multiplies before the split, the pass multiplies T by 2^-e;
//       T / Gt   per row block: every wave leaves the largest |partial| of its share in LDS; the sum of the four is an
//                upper bound of max|T|, 2^e' brings it into [2^14, 2^15), the slab values leave multiplied by 2^-e'.
//     bf16 has f32's exponent range: its instantiation carries none of this.
// Algorithmic bytes per site: M (N + K) e (G and X once) + the partial slabs 2 RT 4 (N + K) M / R (written here, read by
// lora_amd_reduce_batched).
// History: round 4's first form kept the row block of the narrower operand resident in LDS (0.22-0.30 of the roof: ten
// barriers and LDS round trips in front of serially dependent MFMAs; profiles/r04_kbench_fm_variants.log); it was removed
// in round 5, its geometry rules (fm_fit: supported shapes, rows per block) still size the tables.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "mfma16.hpp"

namespace lora_amd {

constexpr int kFmThreads = 256;   // 4 waves
constexpr int kFmNPA1 = 10, kFmNPA2 = 20;  // 16-byte pieces per thread of ONE resident block (R * Ca <= 20480 / 40960 elements)
constexpr int kFmNPB = 8;         // ... of one streamed chunk (R * CW <= 16384 elements)
constexpr int kFmMaxNB = 8;       // row blocks one workgroup walks (their partial sums meet in its slab)
constexpr int kFmLdsSmall = 81920, kFmLdsLarge = 163840;  // two workgroups per CU / one

__host__ __device__ inline int fm_pitch(int cols) {  // bytes; smallest p >= 2 cols with p % 64 == 32
  const int b = cols * 2;
  return ((b + 31) / 64) * 64 + 32;
}
__host__ __device__ inline int fm_tpitch(int R) { return R * 2 + 16; }

// chunk index of a head-padded row, branch-free (a branch around the address of a load costs its own s_waitcnt): q = c / hc by
// a 32-bit magic multiply (exact for c < 2^16), hc = 0 (dense) has magic 0 and gives c back
struct FmHeads { uint32_t magic; int hc, hp; };
__host__ __device__ inline uint32_t fm_head_magic(int hc) {   // ceil(2^32 / hc); the plan computes it (a 64-bit division)
  return hc ? (uint32_t)((0x100000000ull + hc - 1) / (uint32_t)hc) : 0u;
}
__device__ __forceinline__ int fm_hchunk(int c, const FmHeads &h) {
  const int q = (int)__umulhi((uint32_t)c, h.magic);
  return q * h.hp + (c - q * h.hc);
}

// ---------------------------------------------------------------------------------------------------------- factor pack
// f32 masters -> MFMA fragment order in the activation dtype, hi and lo parts:
//   pk[split][c8][jj][e] = part_split(factor(jj, c8 * 8 + e))   jj < RT (the rank tile 4 / 8 / 16; 0 for r <= jj < RT):
//   RT x 8 elements = 16 RT bytes per c8
// factor(jj, c) = down[jj, c] (FACTOR_RK) or up[c, jj] (FACTOR_KR).  A k-step's fragment (4 consecutive c8) is 64 RT bytes
// contiguous; lane (jj = l & 15, q = l >> 4) reads its 16 bytes at (q RT + (jj & (RT - 1))) * 16: rank rows beyond the tile
// REPEAT the tile's rows (third session of round 6: a rank-4 fragment is 256 B instead of 1 KB with three quarters of zeros —
// the fragments were half as many L2 -> L1 bytes as the data).  The repeated rows give T columns j >= RT that copy live ones:
// never stored (slabs hold RT rows), and of the same magnitude, which is what the f16 block scale looks at.
// f16 only: the power of two that brings the largest |factor| of a (site, side) into [2^11, 2^12) and its inverse, as two
// floats in the 16-byte tail of the pack ([2][C/8][RT][8] elements, then the tail).  One workgroup per (site, side).
__device__ __forceinline__ float fm_wave_max(float m) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}
// 2^e with max * 2^e in [2^top, 2^(top + 1)) (max > 0, finite, inside the clamp), as (scale, 1 / scale); (1, 1) for 0
__device__ __forceinline__ void fm_pow2_scale(float mx, int top, float &scale, float &inv) {
  int ex = (int)((__float_as_uint(mx) >> 23) & 0xffu);   // mx in [2^(ex - 127), 2^(ex - 126))
  if (!(mx > 0.f)) ex = 127 + top;                        // zero (or NaN): no scaling
  ex = min(max(ex, 40 + top), 210 + top);                 // both powers stay normal f32 numbers
  scale = __uint_as_float((uint32_t)(254 + top - ex) << 23);
  inv = __uint_as_float((uint32_t)(ex - top) << 23);
}
constexpr int kFmFactorTop = 11, kFmTTop = 14;
__host__ __device__ inline int fm_rank_tile(int r) { return r <= 4 ? 4 : r <= 8 ? 8 : 16; }
__global__ __launch_bounds__(256) void factor_absmax_kernel(const lora_amd_pack_site *__restrict__ sites, int n) {
  __shared__ float s_m[4];
  const lora_amd_pack_site q = sites[blockIdx.x >> 1];
  const bool is_up = blockIdx.x & 1;
  const int C = is_up ? q.N : q.K;
  const float *src = is_up ? q.up : q.down;
  const int64_t cnt = (int64_t)C * q.r;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < cnt; i += 256) m = fmaxf(m, fabsf(gl(src)[i]));
  m = fm_wave_max(m);
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    float sc, inv;
    fm_pow2_scale(m, kFmFactorTop, sc, inv);
    const int RT = fm_rank_tile(q.r);
    float *tail = reinterpret_cast<float *>(reinterpret_cast<uint16_t *>(is_up ? q.pk_up : q.pk_down) + (int64_t)C * 2 * RT);
    gl(tail)[0] = sc; gl(tail)[1] = inv; gl(tail)[2] = 0.f; gl(tail)[3] = 0.f;
  }
}

template <class E>
__global__ __launch_bounds__(256) void factor_pack_kernel(const lora_amd_pack_site *__restrict__ sites, int n, int64_t total) {
  using S = typename E::storage;
  constexpr bool kScaled = E::kCode == LORA_AMD_F16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sites[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const lora_amd_pack_site q = sites[lo];
    const int RT = fm_rank_tile(q.r), rts = RT == 4 ? 2 : RT == 8 ? 3 : 4;
    int64_t p = i - q.begin;                 // piece = (side, c8, jj < RT): down side first
    const int64_t nd = (int64_t)(q.K >> 3) * RT;
    const bool is_up = p >= nd;
    if (is_up) p -= nd;
    const int c8 = (int)(p >> rts), jj = (int)(p & (RT - 1));
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
    S *dst = reinterpret_cast<S *>(is_up ? q.pk_up : q.pk_down);
    if constexpr (kScaled) {   // the tail factor_absmax_kernel wrote in the launch before this one
      const float fs = gl(reinterpret_cast<const float *>(dst + (int64_t)C * 2 * RT))[0];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= fs;
    }
    Chunk8<E> h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h.v[e] = E::from_f(v[e]);
      l.v[e] = E::from_f(v[e] - E::to_f(h.v[e]));
    }
    const int64_t split_stride = (int64_t)(C >> 3) * RT * 8;  // elements per split
    union { Chunk8<E> c; mu32x4 u; } hb, lb;
    hb.c = h; lb.c = l;
    *gl(reinterpret_cast<mu32x4 *>(dst + ((int64_t)c8 * RT + jj) * 8)) = hb.u;
    *gl(reinterpret_cast<mu32x4 *>(dst + split_stride + ((int64_t)c8 * RT + jj) * 8)) = lb.u;
  }
}

// ---------------------------------------------------------------------------------------------------------- register form
// The same pass with the resident operand in REGISTERS and every wave autonomous between four barriers (the structure of
// csrc/rank16_mfma.hip's bwd_g16): a wave owns the 32-column groups cg = wave, wave + 4, ... of both operands.
//   1. all of the block's A pieces are issued at once (<= 20 x 16 bytes per lane: 64 rows x 640 columns or 32 x 1280 over
//      four waves) and stay in registers until step 5;
//   2. phase 1: a row step's two pieces -> the wave's own 32 x 32 LDS tile -> ds_read_b128 in operand order (B operand = the
//      packed factor fragment): TA partials -> LDS, barrier, the four waves' partials are summed into T fragments (hi, lo),
//      barrier;
//   3. B streams through a ring of column-group slots; per row step the tile gives the rows (phase 1 for TB) and, column-major
//      (ds_read_b64_tr_b16), the operand of outB += B^T TA; a group's [16, 32] slab columns are complete after its row steps
//      and stored at once;
//   4. TB partials -> LDS, barrier, T fragments, barrier;
//   5. outA += A^T TB from the resident pieces through the tile the same way.
// No wave waits for another inside a step, the LDS queue of a wave is in order (write -> read -> transpose read needs no wait),
// and the only loads on the critical path of a block are its first ones.
// resident (row step, group) units per wave: PAIRS = 10 serves every supported site (64 rows x 640 columns or 32 x 1280 of the
// narrower operand: 80 registers of pieces, ~200 in all, two workgroups per CU); PAIRS = 6 serves the sites of register class 1
// (R x Ca <= 64 x 320 or 32 x 640 — every M = 16384 site of SD1.5 and the cross-attention k / v: 56 % of the step's bytes) in
// ~140 registers, THREE workgroups per CU (four, at 128 registers, measured no faster)
constexpr int kFrPairsWide = 10, kFrPairsNarrow = 6;
constexpr int kFrPitch = 96;   // bytes per row of a wave's 32 x 32 staging tile (conflict-free, scripts/lds_banks.py)
constexpr int kFrSitesLds = 512;  // block prefix of the site table kept in LDS for the lookup

// -DFM_TRACE (scripts/fm_trace/: a second library, never the product's): every wave leaves 100 MHz wall-clock stamps of its
// block's stages in a buffer — [block][wave][16]: 0 kernel entry, 1 block prefix in LDS, 2 site index known, 3 site record in
// SGPRs and row offsets computed, 4 A's loads issued, 5 every first load issued, 6 A multiplied (its data landed), 7 T ready (two
// barriers), 8 B streamed, 9 Gt ready (two barriers), 10 block done
#ifdef FM_TRACE
__device__ unsigned long long *g_fm_trace = nullptr;
__device__ long long g_fm_trace_cap = 0;
#define FM_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); fm_ts[k] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FM_STAMP(k) do { } while (0)
#endif

// Round 6: NO register copy of a value a load is still writing.  The round-4/5 loop rotated its ring by assignment (slot k <-
// slot k + 1, `fh = nfh` for the prefetched factor fragments): every such v_mov of an in-flight destination made hipcc put
// `s_waitcnt vmcnt(0)` in front of it — the youngest load — so the "ring" drained completely once per unit and once more per
// column group, whatever its depth (rings of 2 / 4 / 6 / 8 / 12 units all ran within 3 % of each other,
// profiles/r06_kbench_fm_ring_depths.log; with the stores, phase 2 AND the fragment loads switched off the skeleton still took
// 85 % of the time, r06_kbench_fm_attribution.log).  Now a ring SLOT is a whole column group — its 2 or 4 pieces and its two
// factor fragments, loaded together — the slots are indexed statically in a loop unrolled over them, a slot is refilled in
// place right after its last use, and every load is unconditional (clamped address, value discarded) so that no load sits
// behind a branch: the waits are counted ones and RG groups stay in flight per wave.
//   RS2: two 32-row steps per block (R = 64) / one (R = 32) — compile-time, so that the resident pieces' (row step, group)
//   order is static too.
//   The block's work is a device function (fm_block) so that ONE launch can hold sites of both block heights: the kernel looks
//   its site up and enters the instantiation of the site's height (a table of one height compiles to that one alone).
struct FmSmem {
  unsigned char *stage;   // [4 waves][32 rows][kFrPitch]
  float *part;            // [4 waves][64 rows][16]
  mu32x4 *tf;             // [2 row steps][hi, lo][64 lanes]
  float *pmax;            // [4]
};

template <class E, bool DROP, int kFrPairs, int MINB, int RG, bool RS2>
__device__ __forceinline__ void fm_block(const lora_amd_fm_site &sd, const FmSmem &sm, int64_t blk, unsigned long long t_entry = 0,
                                         unsigned long long t_prefix = 0) {
  using S = typename E::storage;
#ifdef FM_TRACE
  unsigned long long fm_ts[16] = {t_entry, t_prefix, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  FM_STAMP(2);
#endif
  constexpr bool kScaled = E::kCode == LORA_AMD_F16;   // power-of-two pre-scaling of the split operands (file header)
  constexpr int NRS = RS2 ? 2 : 1;                     // row steps per block
  constexpr int kGroupsA = kFrPairs / NRS;             // resident column groups per wave
  unsigned char *s_stage = sm.stage;
  float *s_part = sm.part, *s_pmax = sm.pmax;
  mu32x4 *s_tf = sm.tf;
  const int tid = threadIdx.x, lane = tid & 63, jj = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: group indices and their masks live in SGPRs
  // Load layout (round 6, call c25/c26): a piece (16 rows x 64 bytes) is FETCHED with four consecutive lanes on one row's 64
  // bytes — lane l: row lr = l >> 2, 16-byte chunk lc = l & 3 — not in the MFMA operand order (row l & 15, chunk l >> 4), whose
  // 64 lanes are 64 scattered 16-byte accesses: scripts/ld_shape_probe.hip measures what a CU's load path sustains from
  // cache-resident data at 9.7 TB/s (chip) for the operand order — one lane per cycle through the L1's tag pipeline, 64 cycles
  // per instruction — against 23-28 TB/s for this one and 31-34 TB/s for full 128-byte lines; the pass ran that path at 50-70 %
  // (stage stamps: ~0.2 us to ISSUE one such load).  The registers hold a piece in the load layout; the MFMA operands come
  // from the wave's LDS tile, which every piece crosses anyway for the transposed read of phase 2.
  const int lr = lane >> 2, lc = lane & 3;
  constexpr int R = 32 * NRS;          // == sd.rows_per_block
  const int64_t rb = blk - sd.block_begin;
  const int64_t m0 = rb * R;
  const int nrows = (int)min((int64_t)R, sd.M - m0);
  const bool ax = sd.resident_is_x != 0;
  const int RT = fm_rank_tile(sd.r);
  const S *da = reinterpret_cast<const S *>(ax ? sd.x : sd.g), *db = reinterpret_cast<const S *>(ax ? sd.g : sd.x);
  const int64_t lda = ax ? sd.ldx : sd.ldg, ldb = ax ? sd.ldg : sd.ldx;
  const int Ca = ax ? sd.K : sd.N, Cb = ax ? sd.N : sd.K;
  const FmHeads hda{(uint32_t)(ax ? sd.x_head_magic : sd.g_head_magic), (ax ? sd.x_head_dim : sd.g_head_dim) >> 3,
                    (ax ? sd.x_head_pad : sd.g_head_pad) >> 3};
  const FmHeads hdb{(uint32_t)(ax ? sd.g_head_magic : sd.x_head_magic), (ax ? sd.g_head_dim : sd.x_head_dim) >> 3,
                    (ax ? sd.g_head_pad : sd.x_head_pad) >> 3};
  const S *pka = reinterpret_cast<const S *>(ax ? sd.pk_down : sd.pk_up), *pkb = reinterpret_cast<const S *>(ax ? sd.pk_up : sd.pk_down);
  const int64_t splita = (int64_t)(Ca >> 3) * RT * 8, splitb = (int64_t)(Cb >> 3) * RT * 8;   // elements per split
  float finv_a = 1.f, finv_b = 1.f;   // f16: 1 / (the power of two the pack multiplied the factor by)
  if constexpr (kScaled) {
    finv_a = gl(reinterpret_cast<const float *>(pka + 2 * splita))[1];
    finv_b = gl(reinterpret_cast<const float *>(pkb + 2 * splitb))[1];
  }
  const int nga = Ca >> 5, ngb = Cb >> 5;
  const int nga_w = wave < nga ? (nga - wave + 3) >> 2 : 0, ngb_w = wave < ngb ? (ngb - wave + 3) >> 2 : 0;   // this wave's groups
  const bool drop = DROP && sd.dropout_p > 0.f;
  const bool mask_a = drop && !ax, mask_b = drop && ax;   // G is the masked operand
  const uint64_t seed = sd.seed, off = drop ? dropout_offset(sd.offset, sd.offset_dev) : 0;
  const uint32_t thr = (uint32_t)(sd.dropout_p * 65536.0f + 0.5f);
  const int n8 = sd.N >> 3;

  // Addresses (round 6, after the stage stamps of scripts/fm_trace showed 4-9 us of a 15-23 us block going by BEFORE its first
  // load left: ~650 instructions of per-piece 64-bit multiplies, a 64-bit division for the head layout and the site lookup):
  // everything wave-uniform — the block's first row, the factor packs, the slabs — is a scalar base; a lane keeps ONE 32-bit byte
  // offset per (row step, row half) and operand (row clamped into the block: every load is issued whatever the indices, the
  // values of rows >= nrows and of dead groups are zeroed by finish()) and adds a group's column offset to it: one VALU add per
  // piece, and the load takes the base + 32-bit offset form.  A row block spans < 2^31 bytes (the plan checks ld < 2^23).
  const unsigned char *basea = reinterpret_cast<const unsigned char *>(da + m0 * lda);
  const unsigned char *baseb = reinterpret_cast<const unsigned char *>(db + m0 * ldb);
  const uint32_t ldab = (uint32_t)lda * 2u, ldbb = (uint32_t)ldb * 2u;
  uint32_t rowa[NRS][2], rowb[NRS][2];
#pragma unroll
  for (int rs = 0; rs < NRS; ++rs)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t rc = (uint32_t)min(rs * 32 + h * 16 + lr, nrows - 1);
      rowa[rs][h] = __umul24(rc, ldab);
      rowb[rs][h] = __umul24(rc, ldbb);
    }
  FM_STAMP(3);
  auto coff = [&](const FmHeads &h, int cg) -> uint32_t { return (uint32_t)fm_hchunk(cg * 4 + lc, h) * 16u; };
  // plain loads: the non-temporal form measured 7 % slower in the step (profiles/r04_kbench_fm_register_form_nt.log)
  auto load_piece = [&](const unsigned char *base, uint32_t ro, uint32_t co) -> mu32x4 {
    return *gl(reinterpret_cast<const mu32x4 *>(base + (ro + co)));
  };
  auto finish = [&](mu32x4 v, int cg, int row, bool live, bool masked) -> mu32x4 {
    if (!live || row >= nrows) return mu32x4{0u, 0u, 0u, 0u};
    if (DROP && masked) v &= dropout_and8(seed, off, (uint64_t)((m0 + row) * (int64_t)n8 + cg * 4 + lc), thr);
    return v;
  };
  const unsigned char *pkbh = reinterpret_cast<const unsigned char *>(pkb), *pkbl = reinterpret_cast<const unsigned char *>(pkb + splitb);
  const unsigned char *pkah = reinterpret_cast<const unsigned char *>(pka), *pkal = reinterpret_cast<const unsigned char *>(pka + splita);
  // a k-step's fragment: 64 RT bytes; lane (jj, q) its 16 bytes at (q RT + jj mod RT) * 16 (RT = 16: 16 l)
  const uint32_t lane16 = (uint32_t)(q * RT + (jj & (RT - 1))) * 16u, fragb = (uint32_t)RT * 64u;
  auto frag = [&](const unsigned char *pk, int cg) -> mu32x4 {
    return *gl(reinterpret_cast<const mu32x4 *>(pk + ((uint32_t)cg * fragb + lane16)));
  };
  // slabs: [RT][C] floats per block and operand; lane (j = jj, columns 4 q ..) of a group's two 16-column tiles
  float *outa = (ax ? sd.down_part : sd.up_part) + rb * RT * (int64_t)Ca;
  float *outb = (ax ? sd.up_part : sd.down_part) + rb * RT * (int64_t)Cb;
  const uint32_t jrow = jj < RT ? (uint32_t)jj : 0u;
  const uint32_t slaba = (jrow * (uint32_t)Ca + 4u * q) * 4u, slabb = (jrow * (uint32_t)Cb + 4u * q) * 4u;
  unsigned char *stage = s_stage + wave * 32 * kFrPitch;
  // kTfLds (the three-per-CU kernel): the T fragments of a row step are read from LDS at every use instead of living in 16
  // registers for the whole stream (two more ds_read_b128 per unit buy the third workgroup)
  constexpr bool kTfLds = MINB >= 3;
  // the wave's 32 x 32 tile: a row step's two pieces go in as they were loaded (row lr / 16 + lr, chunk lc) ...
  auto tile_put = [&](mu32x4 p0, mu32x4 p1) {
    *reinterpret_cast<mu32x4 *>(stage + lr * kFrPitch + lc * 16) = p0;
    *reinterpret_cast<mu32x4 *>(stage + (16 + lr) * kFrPitch + lc * 16) = p1;
    asm volatile("" ::: "memory");
  };
  // ... come back as the A operands of phase 1 (row jj / 16 + jj, k chunk q: conflict-free at the 96-byte pitch) ...
  auto tile_rows = [&](mu32x4 &f0, mu32x4 &f1) {
    f0 = *reinterpret_cast<const mu32x4 *>(stage + jj * kFrPitch + q * 16);
    f1 = *reinterpret_cast<const mu32x4 *>(stage + (16 + jj) * kFrPitch + q * 16);
  };
  // ... and column-major (ds_read_b64_tr_b16) for phase 2: out[j][cg * 32 + 16 nt + 4 q ..] += (the row step)^T T-fragment
  auto tile_cols_mma = [&](mu32x4 th, mu32x4 tl, int rs, mf32x4 (&acc)[2]) {
    if constexpr (kTfLds) {
      th = s_tf[(rs * 2 + 0) * 64 + lane];
      tl = s_tf[(rs * 2 + 1) * 64 + lane];
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned char *pp = stage + (4 * q + (jj >> 2)) * kFrPitch + (16 * nt + 4 * (jj & 3)) * 2;
      union { ms16x4 h[2]; mu32x4 u; } a;
      a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(pp));
      a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ms16x4 __attribute__((address_space(3))) *)(pp + 16 * kFrPitch));
      acc[nt] = FmMfma<E>::mma(fm_frag<E>(a.u), fm_frag<E>(th), acc[nt]);
      acc[nt] = FmMfma<E>::mma(fm_frag<E>(a.u), fm_frag<E>(tl), acc[nt]);
    }
    asm volatile("" ::: "memory");
  };
  float tinv = 1.f;   // f16: 1 / (the power of two the T fragments in use were multiplied by)
  auto store_group = [&](float *out, uint32_t slab, int cg, bool live, mf32x4 (&acc)[2]) {
    unsigned char *o = reinterpret_cast<unsigned char *>(out) + (slab + (uint32_t)(live ? cg : 0) * 128u);
    if (live && jj < RT) {
      if constexpr (kScaled) { acc[0] *= tinv; acc[1] *= tinv; }
      // non-temporal: the slabs are read again only by the fold, after the pass (call c43: 574-577 -> 568-569 us in the step)
      __builtin_nontemporal_store(acc[0], gl(reinterpret_cast<mf32x4 *>(o)));
      __builtin_nontemporal_store(acc[1], gl(reinterpret_cast<mf32x4 *>(o + 64)));
    }
    acc[0] = acc[1] = mf32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_parts = [&](const mf32x4 (&d)[2 * NRS]) {   // D1 lane (j = jj, rows 4 q + reg) of (row step, 16-row group)
#pragma unroll
    for (int x = 0; x < 2 * NRS; ++x)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) s_part[(wave * 64 + (x >> 1) * 32 + (x & 1) * 16 + 4 * q + reg) * 16 + jj] = d[x][reg];
    if constexpr (kScaled) {
      float m = 0.f;
#pragma unroll
      for (int x = 0; x < 2 * NRS; ++x)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) m = fmaxf(m, fabsf(d[x][reg]));
      m = fm_wave_max(m);
      if (lane == 0) s_pmax[wave] = m;
    }
  };
  // T = tscale * (sum of the four waves' shares) -> (hi, lo) fragments.  f16: tscale carries 1 / (factor scale) and the
  // block's own power of two, chosen from the bound |T| <= |tscale| * sum_w max|share_w|; every wave keeps its inverse
  auto build_tf = [&](float tscale) {   // k slot 8 q + e <-> row 4 q + e (e < 4) / 16 + 4 q + e - 4: the order of the transposed reads
    if constexpr (kScaled) {
      float up2, dn2;
      fm_pow2_scale(fabsf(tscale) * (s_pmax[0] + s_pmax[1] + s_pmax[2] + s_pmax[3]), kFmTTop, up2, dn2);
      tscale *= up2;
      tinv = dn2;
    }
    if (wave < NRS) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = wave * 32 + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
        v[e] = tscale * (s_part[(0 * 64 + row) * 16 + jj] + s_part[(1 * 64 + row) * 16 + jj] +
                         s_part[(2 * 64 + row) * 16 + jj] + s_part[(3 * 64 + row) * 16 + jj]);
      }
      mu32x4 h, l;
      split_hi_lo<E>(v, h, l);
      s_tf[(wave * 2 + 0) * 64 + lane] = h;
      s_tf[(wave * 2 + 1) * 64 + lane] = l;
    }
  };

  // ---- 1. the resident pieces of A: group gi of this wave = column group wave + 4 gi, pieces [gi][row step][row half]
  mu32x4 pa[kGroupsA][NRS][2];
#pragma unroll
  for (int gi = 0; gi < kGroupsA; ++gi) {
    const uint32_t co = coff(hda, gi < nga_w ? wave + 4 * gi : 0);
#pragma unroll
    for (int rs = 0; rs < NRS; ++rs)
#pragma unroll
      for (int h = 0; h < 2; ++h) pa[gi][rs][h] = load_piece(basea, rowa[rs][h], co);
  }
  FM_STAMP(4);
  // ---- 3 (issued here): the first RG column groups of B and their factor fragments
  mu32x4 pb[RG][NRS][2], fb[RG][2];
  auto load_group = [&](int gi, mu32x4 (&p)[NRS][2], mu32x4 (&f)[2]) {
    const int cg = gi < ngb_w ? wave + 4 * gi : 0;
    f[0] = frag(pkbh, cg);
    f[1] = frag(pkbl, cg);
    const uint32_t co = coff(hdb, cg);
#pragma unroll
    for (int rs = 0; rs < NRS; ++rs)
#pragma unroll
      for (int h = 0; h < 2; ++h) p[rs][h] = load_piece(baseb, rowb[rs][h], co);
  };
  // ---- 2. TA = A fa^T: the fragments of A's groups in a two-deep buffer indexed by the (static) parity of the group
  mf32x4 d1[2 * NRS];
#pragma unroll
  for (int x = 0; x < 2 * NRS; ++x) d1[x] = mf32x4{0.f, 0.f, 0.f, 0.f};
  {
    mu32x4 fa[2][2];
    fa[0][0] = frag(pkah, nga_w > 0 ? wave : 0);
    fa[0][1] = frag(pkal, nga_w > 0 ? wave : 0);
#pragma unroll
    for (int s = 0; s < RG; ++s) load_group(s, pb[s], fb[s]);
    // every load above leaves before the first wait: with cheap addresses hipcc otherwise sinks half of them behind the
    // first MFMAs (shorter live ranges), i.e. behind an s_waitcnt for the block's first piece
    __builtin_amdgcn_sched_barrier(0);
    FM_STAMP(5);
#pragma unroll
    for (int gi = 0; gi < kGroupsA; ++gi) {
      const bool live = gi < nga_w;
      const int cg = wave + 4 * gi;
      if (gi + 1 < kGroupsA) {
        const int cn = gi + 1 < nga_w ? cg + 4 : 0;
        fa[(gi + 1) & 1][0] = frag(pkah, cn);
        fa[(gi + 1) & 1][1] = frag(pkal, cn);
      }
#pragma unroll
      for (int rs = 0; rs < NRS; ++rs) {
#pragma unroll
        for (int h = 0; h < 2; ++h) pa[gi][rs][h] = finish(pa[gi][rs][h], cg, rs * 32 + h * 16 + lr, live, mask_a);
        tile_put(pa[gi][rs][0], pa[gi][rs][1]);
        mu32x4 f[2];
        tile_rows(f[0], f[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          d1[rs * 2 + h] = FmMfma<E>::mma(fm_frag<E>(f[h]), fm_frag<E>(fa[gi & 1][0]), d1[rs * 2 + h]);
          d1[rs * 2 + h] = FmMfma<E>::mma(fm_frag<E>(f[h]), fm_frag<E>(fa[gi & 1][1]), d1[rs * 2 + h]);
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  FM_STAMP(6);
  store_parts(d1);
  __syncthreads();
  build_tf(sd.scale * finv_a);
  __syncthreads();
  FM_STAMP(7);
  mu32x4 tfh[2], tfl[2];
  tfh[0] = tfl[0] = tfh[1] = tfl[1] = mu32x4{0u, 0u, 0u, 0u};
  if constexpr (!kTfLds) {
    tfh[0] = s_tf[0 * 64 + lane]; tfl[0] = s_tf[1 * 64 + lane];
    if constexpr (RS2) { tfh[1] = s_tf[2 * 64 + lane]; tfl[1] = s_tf[3 * 64 + lane]; }
  }
#pragma unroll
  for (int x = 0; x < 2 * NRS; ++x) d1[x] = mf32x4{0.f, 0.f, 0.f, 0.f};
  mf32x4 acc[2] = {mf32x4{0.f, 0.f, 0.f, 0.f}, mf32x4{0.f, 0.f, 0.f, 0.f}};
  // ---- 3. B: slot s holds group g0 + s; consumed, then refilled in place with group g0 + s + RG.  Branch-free on purpose: a
  // wave-uniform `if (live)` around the work of a dead slot (call c20's build) brought the register copies of in-flight slots
  // and their s_waitcnt vmcnt(0) back (the compiler rotates the slots through the join)
#pragma unroll 1
  for (int g0 = 0; g0 < ngb_w; g0 += RG) {
#pragma unroll
    for (int s = 0; s < RG; ++s) {
      const int gi = g0 + s;
      const bool live = gi < ngb_w;
      const int cg = wave + 4 * gi;
      mu32x4 qv[NRS][2];
#pragma unroll
      for (int rs = 0; rs < NRS; ++rs)
#pragma unroll
        for (int h = 0; h < 2; ++h) qv[rs][h] = finish(pb[s][rs][h], cg, rs * 32 + h * 16 + lr, live, mask_b);
      const mu32x4 fh = fb[s][0], fl = fb[s][1];
      load_group(gi + RG, pb[s], fb[s]);   // the slot's registers are free: its next group goes out before the LDS work
#pragma unroll
      for (int rs = 0; rs < NRS; ++rs) {   // a row step through the tile: rows for TB (phase 1), columns for outB (phase 2)
        tile_put(qv[rs][0], qv[rs][1]);
        mu32x4 f[2];
        tile_rows(f[0], f[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          d1[rs * 2 + h] = FmMfma<E>::mma(fm_frag<E>(f[h]), fm_frag<E>(fh), d1[rs * 2 + h]);
          d1[rs * 2 + h] = FmMfma<E>::mma(fm_frag<E>(f[h]), fm_frag<E>(fl), d1[rs * 2 + h]);
        }
        tile_cols_mma(tfh[rs], tfl[rs], rs, acc);
      }
      store_group(outb, slabb, cg, live, acc);
    }
  }
  // ---- 4. TB -> fragments
  FM_STAMP(8);
  store_parts(d1);
  __syncthreads();
  build_tf(sd.scale * finv_b);
  __syncthreads();
  FM_STAMP(9);
  if constexpr (!kTfLds) {
    tfh[0] = s_tf[0 * 64 + lane]; tfl[0] = s_tf[1 * 64 + lane];
    if constexpr (RS2) { tfh[1] = s_tf[2 * 64 + lane]; tfl[1] = s_tf[3 * 64 + lane]; }
  }
  // ---- 5. outA = A^T TB from the resident pieces
#pragma unroll
  for (int gi = 0; gi < kGroupsA; ++gi) {
    const bool live = gi < nga_w;
#pragma unroll
    for (int rs = 0; rs < NRS; ++rs) {
      tile_put(pa[gi][rs][0], pa[gi][rs][1]);
      tile_cols_mma(tfh[rs], tfl[rs], rs, acc);
    }
    store_group(outa, slaba, wave + 4 * gi, live, acc);
  }
#ifdef FM_TRACE
  FM_STAMP(10);
  if (g_fm_trace && lane < 16 && (blk * 4 + wave + 1) * 16 <= g_fm_trace_cap) {
    unsigned long long v = fm_ts[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = lane == k ? fm_ts[k] : v;
    g_fm_trace[(blk * 4 + wave) * 16 + lane] = v;
  }
#endif
}

// HEIGHTS: 1 = every site of the table has 64-row blocks, 0 = 32-row blocks, 2 = both (dispatch per workgroup)
template <class E, bool DROP, int kFrPairs, int MINB, int RG64, int RG32, int HEIGHTS>
__global__ __launch_bounds__(kFmThreads, MINB) void factors_reg_kernel(const lora_amd_fm_site *__restrict__ sites, int n,
                                                                              const int32_t *__restrict__ block_map) {
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[4 * 32 * kFrPitch];
  __shared__ __attribute__((aligned(16))) float s_part[4 * 64 * 16];   // [wave][row][j]
  __shared__ __attribute__((aligned(16))) mu32x4 s_tf[2 * 2 * 64];     // [row step][hi, lo][lane]
  __shared__ int64_t s_begin[kFrSitesLds];
  __shared__ float s_pmax[4];                                          // f16: largest |partial| of each wave's share
  const int tid = threadIdx.x;
#ifdef FM_TRACE
  const unsigned long long t_entry = wall_clock64();
#else
  const unsigned long long t_entry = 0;
#endif
  unsigned long long t_prefix = t_entry;
  // which site: the table's block prefix is fetched once, in parallel, and searched in LDS (a search over the table in
  // memory is eight DEPENDENT trips to L2 in front of the block's first load).  A workgroup per row block: walking runs of
  // 2 / 4 / 8 consecutive blocks per workgroup (launch and lookup paid once per run) measured 2-60 % SLOWER
  // (profiles/r06_kbench_fm_span.log), and so did a persistent launch — one workgroup per resident slot drawing its next
  // block from a counter behind the current block's loads: 5-28 % slower, parity-green (profiles/r06_kbench_fm_persistent.log,
  // the patch beside it) — although the stage stamps show a slot empty for ~3.7 us between two workgroups.
  // (round 6, third session) ... or not searched at all: the plan's block -> site map (lora_amd_factors_mfma_block_map) is ONE
  // scalar load off blockIdx — the prefix copy, its barrier and nine dependent LDS reads were 1.2 us of a 12.8 us block
  int lo = 0, hi = n - 1;
  if (block_map != nullptr) {
    lo = block_map[blockIdx.x];
  } else if (n <= kFrSitesLds) {
    for (int i = tid; i < n; i += kFmThreads) s_begin[i] = sites[i].block_begin;
    __syncthreads();
#ifdef FM_TRACE
    t_prefix = wall_clock64();
#endif
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_begin[mid] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  } else {
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sites[mid].block_begin <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
  }
  lo = __builtin_amdgcn_readfirstlane(lo);   // wave-uniform: the site record arrives by scalar loads and lives in SGPRs
  const lora_amd_fm_site sd = sites[lo];
  const FmSmem sm{s_stage, s_part, s_tf, s_pmax};
  const int64_t blk = blockIdx.x;
  if constexpr (HEIGHTS == 1) fm_block<E, DROP, kFrPairs, MINB, RG64, true>(sd, sm, blk, t_entry, t_prefix);
  else if constexpr (HEIGHTS == 0) fm_block<E, DROP, kFrPairs, MINB, RG32, false>(sd, sm, blk, t_entry, t_prefix);
  else {
    if (sd.rows_per_block == 64) fm_block<E, DROP, kFrPairs, MINB, RG64, true>(sd, sm, blk, t_entry, t_prefix);   // block-uniform
    else fm_block<E, DROP, kFrPairs, MINB, RG32, false>(sd, sm, blk, t_entry, t_prefix);
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

// rows per block and LDS class (1: two workgroups per CU, 2: one) of a site.  A caller's row count is
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

// row blocks one workgroup walks: their partial sums meet in the workgroup's slab (the kernel supports up to kFmMaxNB).
// 1: with its tile loads, its factor fragments and its slab reads on ONE in-order vmcnt counter a longer run only
// serialises (measured 1006 / 1174 / 1657 us at 1 / 2 / 4 blocks, profiles/r04_kbench_fm_variants.log)
static int fm_blocks_per_wg(int64_t nrb) {
  const int v = 1;
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(std::max(v, 1), kFmMaxNB), nrb));
}

static int g_fm_narrow = 1;   // 0: class-1 tables run the wide kernel too (A/B hook, lora_amd_factors_mfma_set_tuning)
static int g_fm_wide = 0;     // measurement variants of the 10-pair kernel's ring (same hook, bits 4..7; bf16, maskless, rows = 0)

}  // namespace lora_amd

using namespace lora_amd;

#ifdef FM_TRACE
extern "C" int lora_amd_fm_trace_set(void *buf, int64_t cap_words) {
  unsigned long long *p = (unsigned long long *)buf;
  long long c = cap_words;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_fm_trace), &p, sizeof(p)) != hipSuccess) return LORA_AMD_EINVAL;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_fm_trace_cap), &c, sizeof(c)) != hipSuccess) return LORA_AMD_EINVAL;
  return LORA_AMD_OK;
}
#endif

// Tuning / test hook: tables of register class 1 (64-row blocks) run 0 = the 10-pair kernel (two workgroups per CU, two column
// groups in flight per wave: rounds 4-5's geometry), 1 / 2 = the 6-pair kernel at three workgroups per CU with 1 / 2 groups in
// flight, 3 / 4 / 5 = the 6-pair kernel at two per CU with 2 / 3 / 4 groups in flight; < 0 only reads.  Returns the previous value.
extern "C" int lora_amd_factors_mfma_set_tuning(int32_t narrow) {
  const int prev = g_fm_narrow | (g_fm_wide << 4);
  if (narrow >= 0 && (narrow & 15) <= 5 && (narrow >> 4) <= 4) {
    g_fm_narrow = narrow & 15;
    g_fm_wide = narrow >> 4;
  }
  return prev;
}

extern "C" int lora_amd_factors_mfma_plan(int64_t M, int32_t K, int32_t N, int32_t r, int32_t act_dtype, int32_t rows,
                                          int32_t flags, lora_amd_factors_mfma_plan_t *out) {
  LORA_AMD_CHECK(out != nullptr && rows >= 0 && dtype_ok(act_dtype), LORA_AMD_EINVAL, "factors_mfma_plan: bad argument");
  memset(out, 0, sizeof(*out));
  FmGeom g;
  int cls = 0;
  (void)flags;  // bit 0 (dropout site) selects the masked kernel at launch time; the geometry is the same
  // the register-resident kernel holds 20 pieces per lane: 64 rows up to 640 columns of the narrower operand, 32 beyond
  if (rows == 0) rows = 64;
  if (!fm_choose(M, K, N, r, act_dtype, rows, &g, &cls)) return LORA_AMD_OK;
  out->supported = 1;
  out->lds_class = cls;
  out->rank_tile = r <= 4 ? 4 : r <= 8 ? 8 : 16;
  out->rows_per_block = g.R;
  const int64_t nrb = (M + g.R - 1) / g.R;
  out->blocks_per_wg = fm_blocks_per_wg(nrb);
  out->nparts = (int32_t)((nrb + out->blocks_per_wg - 1) / out->blocks_per_wg);
  out->lds_bytes = g.lds;
  out->up_part_floats = (int64_t)out->nparts * out->rank_tile * N;
  out->down_part_floats = (int64_t)out->nparts * out->rank_tile * K;
  out->pack_up_elems = (int64_t)N * 2 * out->rank_tile + 8;    // [2][N/8][RT][8] + the 16-byte tail (f16: the pack's power-of-two scale)
  out->pack_down_elems = (int64_t)K * 2 * out->rank_tile + 8;
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
    begin += (int64_t)((q.N + q.K) >> 3) * fm_rank_tile(q.r);
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
  if (act_dtype == LORA_AMD_F16) {
    hipLaunchKernelGGL(factor_absmax_kernel, dim3(2u * (unsigned)n), dim3(256), 0, st, sites_dev, n);
    hipLaunchKernelGGL(factor_pack_kernel<f16_t>, dim3(grid), dim3(256), 0, st, sites_dev, n, total);
  } else {
    hipLaunchKernelGGL(factor_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, sites_dev, n, total);
  }
  return check_launch("lora_amd_factor_pack");
}

extern "C" int lora_amd_factors_mfma_ragged_plan(lora_amd_fm_site *sites, int32_t n, int32_t act_dtype, int32_t lds_class,
                                                 int64_t *grid) {
  LORA_AMD_CHECK(sites && n >= 1 && grid && (lds_class == 1 || lds_class == 2), LORA_AMD_EINVAL,
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
    LORA_AMD_CHECK(lds_class == 2 || q.rows_per_block == sites[0].rows_per_block, LORA_AMD_EINVAL,
                   "factors_mfma_ragged_plan: site %d: %d rows per block, site 0: %d (a class-1 table has one block height)", i,
                   q.rows_per_block, sites[0].rows_per_block);
    LORA_AMD_CHECK(q.r >= 1 && q.r <= 16 && (q.r <= 4 ? 4 : q.r <= 8 ? 8 : 16) == rt0, LORA_AMD_ERANK,
                   "factors_mfma_ragged_plan: site %d: rank %d (one rank tile per table)", i, q.r);
    LORA_AMD_CHECK(q.g && q.x && q.pk_up && q.pk_down && q.up_part && q.down_part, LORA_AMD_EINVAL,
                   "factors_mfma_ragged_plan: site %d: null pointer", i);
    const bool ok = fm_fit(q.M, q.K, q.N, q.r, act_dtype, q.rows_per_block, lds_class == 1 ? kFmLdsSmall : kFmLdsLarge, &g);
    LORA_AMD_CHECK(ok && ((uintptr_t)q.g % 16) == 0 && ((uintptr_t)q.x % 16) == 0 && q.ldg % 8 == 0 && q.ldx % 8 == 0 &&
                       ((uintptr_t)q.pk_up % 16) == 0 && ((uintptr_t)q.pk_down % 16) == 0 &&
                       heads_ok(q.g_head_dim, q.g_head_pad, q.N, q.ldg) && heads_ok(q.x_head_dim, q.x_head_pad, q.K, q.ldx),
                   LORA_AMD_EINVAL, "factors_mfma_ragged_plan: site %d: shape / alignment / head layout / LDS class not supported", i);
    q.rows_per_block = g.R; q.resident_is_x = g.resident_is_x; q.cw = g.cw; q.nchunk = g.nchunk;
    q.lds_bytes = g.lds;
    LORA_AMD_CHECK(q.ldg < (1ll << 23) && q.ldx < (1ll << 23) && (q.g_head_dim == 0 || q.g_head_dim >= 16) &&
                       (q.x_head_dim == 0 || q.x_head_dim >= 16),
                   LORA_AMD_EINVAL, "factors_mfma_ragged_plan: site %d: row pitch >= 2^23 elements or head_dim 8", i);
    q.g_head_magic = (int32_t)fm_head_magic(q.g_head_dim >> 3);
    q.x_head_magic = (int32_t)fm_head_magic(q.x_head_dim >> 3);
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

extern "C" int lora_amd_factors_mfma_block_map(const lora_amd_fm_site *sites, int32_t n, int64_t grid, int32_t *map) {
  LORA_AMD_CHECK(sites && map && n >= 1 && grid >= 1, LORA_AMD_EINVAL, "factors_mfma_block_map: bad argument");
  int64_t b = 0;
  for (int i = 0; i < n; ++i) {
    LORA_AMD_CHECK(sites[i].block_begin == b, LORA_AMD_EINVAL,
                   "factors_mfma_block_map: site %d begins at block %lld, expected %lld (plan the table first)", i,
                   (long long)sites[i].block_begin, (long long)b);
    const int64_t end = i + 1 < n ? sites[i + 1].block_begin : grid;
    LORA_AMD_CHECK(end > b && end <= grid, LORA_AMD_EINVAL, "factors_mfma_block_map: site %d: blocks [%lld, %lld) outside the grid", i,
                   (long long)b, (long long)end);
    for (; b < end; ++b) map[b] = i;
  }
  return LORA_AMD_OK;
}

extern "C" int lora_amd_linear_bwd_factors_mfma_ragged(const lora_amd_fm_site *sites_dev, int32_t n, int64_t grid,
                                                       int32_t lds_class, int32_t rows_per_block, int32_t act_dtype,
                                                       int32_t masked, void *stream) {
  return lora_amd_linear_bwd_factors_mfma_ragged_mapped(sites_dev, n, grid, nullptr, lds_class, rows_per_block, act_dtype, masked,
                                                        stream);
}

extern "C" int lora_amd_linear_bwd_factors_mfma_ragged_mapped(const lora_amd_fm_site *sites_dev, int32_t n, int64_t grid,
                                                              const int32_t *block_map_dev, int32_t lds_class,
                                                              int32_t rows_per_block, int32_t act_dtype, int32_t masked,
                                                              void *stream) {
  LORA_AMD_CHECK(((uintptr_t)block_map_dev % 4) == 0, LORA_AMD_EINVAL, "linear_bwd_factors_mfma_ragged: block map not aligned");
  LORA_AMD_CHECK(sites_dev && n >= 1 && grid >= 1 && grid < (1ll << 31) && (lds_class == 1 || lds_class == 2),
                 LORA_AMD_EINVAL, "linear_bwd_factors_mfma_ragged: bad argument");
  LORA_AMD_CHECK(rows_per_block == 32 || rows_per_block == 64 || rows_per_block == 0, LORA_AMD_EINVAL,
                 "linear_bwd_factors_mfma_ragged: rows_per_block = the 32 or 64 every site of the table was planned with, or 0 "
                 "for a table that holds both (class 2 only)");
  LORA_AMD_CHECK(rows_per_block != 0 || lds_class == 2, LORA_AMD_EINVAL,
                 "linear_bwd_factors_mfma_ragged: a class-1 table has one block height");
  LORA_AMD_CHECK(act_dtype == LORA_AMD_F16 || act_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL,
                 "linear_bwd_factors_mfma_ragged: f16 / bf16 activations only");
  hipStream_t st = (hipStream_t)stream;
  const bool drop = masked != 0;  // a table of dropout sites: the kernel with the Philox mask on G (straight-line, no per-site branch)
  // register class 1 with 64-row blocks (<= 3 resident column groups per wave): the 6-pair kernels, chosen by
  // lora_amd_factors_mfma_set_tuning; everything else: the 10-pair kernel of its block height
  const int narrow = (lds_class == 1 && rows_per_block == 64) ? g_fm_narrow : 0;
  if (lds_class == 1 && rows_per_block != 64 && rows_per_block != 32) rows_per_block = 64;
#define FML(E, D, P, B, G64, G32, H) hipLaunchKernelGGL((factors_reg_kernel<E, D, P, B, G64, G32, H>), dim3((unsigned)grid), dim3(kFmThreads), 0, st, sites_dev, n, block_map_dev)
#define FM2(E, D)                                                            \
  do {                                                                       \
    if (narrow == 1) FML(E, D, kFrPairsNarrow, 3, 1, 1, 1);                  \
    else if (narrow == 2) FML(E, D, kFrPairsNarrow, 3, 2, 2, 1);             \
    else if (narrow == 3) FML(E, D, kFrPairsNarrow, 2, 2, 2, 1);             \
    else if (narrow == 4) FML(E, D, kFrPairsNarrow, 2, 3, 3, 1);             \
    else if (narrow == 5) FML(E, D, kFrPairsNarrow, 2, 4, 4, 1);             \
    else if (rows_per_block == 64) FML(E, D, kFrPairsWide, 2, 2, 4, 1);      \
    else if (rows_per_block == 32) FML(E, D, kFrPairsWide, 2, 2, 4, 0);      \
    else FML(E, D, kFrPairsWide, 2, 2, 4, 2);                                \
  } while (0)
#define FM(E) do { if (drop) FM2(E, true); else FM2(E, false); } while (0)
  if (g_fm_wide != 0 && narrow == 0 && rows_per_block == 0 && !drop && act_dtype == LORA_AMD_BF16) {
    // ring depths of the 10-pair kernel (64-row sites, 32-row sites): 1 = (3, 4), 2 = (4, 4), 3 = (2, 6), 4 = (3, 6); default (2, 4)
    if (g_fm_wide == 1) FML(bf16_t, false, kFrPairsWide, 2, 3, 4, 2);
    else if (g_fm_wide == 2) FML(bf16_t, false, kFrPairsWide, 2, 4, 4, 2);
    else if (g_fm_wide == 3) FML(bf16_t, false, kFrPairsWide, 2, 2, 6, 2);
    else FML(bf16_t, false, kFrPairsWide, 2, 3, 6, 2);
  } else if (act_dtype == LORA_AMD_F16) FM(f16_t); else FM(bf16_t);
#undef FM2
#undef FML
#undef FM
  return check_launch("lora_amd_linear_bwd_factors_mfma_ragged");
}
