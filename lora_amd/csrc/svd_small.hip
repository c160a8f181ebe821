// K5 (cli_svd.py:24-92): the small dense steps of the batched subspace iteration, fused.
//
// Everything between two passes over the residuals works on THIN matrices [rows][16] f32 (the sketch of one site: Y = dW F
// with rows = N, Z = dW^T Q with rows = K), 224 of them in a model, of 31 shapes.  Rounds 2-4 ran these steps as ~2000 small
// launches per distillation (Gram / Cholesky / triangular apply x 3 per orthonormalisation, 224 rocSOLVER SVDs of 16 x 16,
// per-group sign fixes, quantile sorts): 19 of the 27 ms.  Here one launch serves EVERY site of the model (a table with one
// entry per site, 256-row blocks, a block -> site map), and the reduction that follows a launch is finished inside it by the
// LAST-ARRIVING workgroup of each site (arrival counter, agent-scope release / acquire as the guide's inter-workgroup recipe
// prescribes; no spinning — a workgroup never waits for another):
//
//   thin_gram      G = A^T B over the rows (A = B: the Gram matrix of a sketch; A != B: the 16 x 16 core b Qb);
//                  finish: shifted Cholesky -> L^{-1} (next launch's operand) [+ the top-r Ritz energy of G], or the
//                  one-sided Jacobi SVD of the core -> Ub^T, S, Vb
//   thin_apply     Y' = Y L^{-T} and, in the same sweep, the Gram matrix of Y' -> finish as above: CholeskyQR3 = 4 launches
//   thin_rotate    U = Q Ub[:, :r] diag(scale), V = Qb Vb[:r]^T, and for V the sign rule of cli_svd (largest-|.| entry of a
//                  down row positive) as a per-block arg-max finished by the last arriver
//   thin_select    exact order statistics of a site's joint (up, down) values by 3 radix passes (11 + 11 + 10 bits), each a
//                  launch whose last arriver narrows the prefix: the two statistics torch.quantile interpolates between
//   thin_clamp     clamp at the quantile and write up [N][r], down [r][K]
//
// The products are v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact f32 FMA chains): a lane's 16-byte piece of a
// row-major [rows][16] row IS an operand (k-slots permuted the same way on both operands), so there is no LDS staging for the
// apply; the Gram of the output goes through the wave's own 1 KB LDS tile.  All of it is latency, not bandwidth (13-30 MB per
// launch over the whole model): what matters is one launch instead of hundreds and ~1000 workgroups in flight.
#include <math.h>

#include "common.hpp"
#include "mfma16.hpp"

namespace lora_amd {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRowsPerBlock = 256;  // 4 waves x 4 tiles x 16 rows

struct Finish {
  float *part;         // [total_blocks][256] per-block partial 16 x 16 sums
  unsigned *counters;  // [nsites], zero between launches (the last arriver resets its site's word)
  int mode;            // 0: none, 1: Cholesky inverse (+ Ritz energy), 2: Jacobi SVD of the sum
  float shift_rel;
  float *linv_out;     // [nsites][256]                       (mode 1)
  float *ritz_out;     // [nsites][2] or null: (sum of the `rank` largest eigenvalues of the un-shifted sum, sum of the rest)
  float *ubt;          // [nsites][rank][16] = U[:, :rank]^T  (mode 2)
  float *vb;           // [nsites][rank][16] = V[:, :rank]^T  (mode 2)
  float *s_out;        // [nsites][16] singular values, descending (mode 2)
  int rank;
};

#define WSYNC()                                                 \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      \
    __builtin_amdgcn_wave_barrier();                            \
  } while (0)

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, m, 64);
  hi = __shfl_xor(hi, m, 64);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_d(double v, int src) {  // `src` wave-uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// L^{-1} of the shifted sum G + shift I = L L^T (the arithmetic of csrc/linear.hip's chol_inverse_kernel, l = 16) with the
// matrix in REGISTERS: lane i < 16 holds row i, L[k][j] travels by v_readlane.  (Round-5's first form walked the factor in LDS:
// every update a load-wait-store round trip, ~50 us of a launch whose row sweep takes 10.)
__device__ void chol_inverse_regs(double (*G)[17], double shift, float *out, int lane) {
  double dmax = 0.0;
  for (int i = 0; i < 16; ++i) dmax = fmax(dmax, 0.5 * (G[i][i] + G[i][i]) + shift);
  const double floor_ = 1e-30 + 1e-6 * dmax;  // a pivot below the noise floor of an f32 Gram: a direction the block lacks
  const int row = lane & 15;
  double a[16], x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = 0.5 * (G[row][k] + G[k][row]) + (k == row ? shift : 0.0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double djj = readlane_d(a[j], j);
    const double d = djj > floor_ ? sqrt(djj) : (double)INFINITY;  // L[j][j] = inf: row / column j of L^{-1} become zero
    if (row == j) a[j] = d;
    else if (row > j) a[j] = a[j] / d;
#pragma unroll
    for (int k = j + 1; k < 16; ++k) {
      const double lkj = readlane_d(a[j], k);
      if (row >= k) a[k] -= a[j] * lkj;
    }
  }
  // forward substitution, lane = column c of the inverse: L x = e_c
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    double sacc = i == row ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) sacc -= readlane_d(a[k], i) * x[k];
    const double lii = readlane_d(a[i], i);
    x[i] = i < row ? 0.0 : sacc / lii;
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) *gl(out + i * 16 + lane) = (float)x[i];
  }
}

// One-sided (Hestenes) Jacobi on the 16 x 16 matrix held column-major in LDS (A[c][r]); one wave, f64.  Eight disjoint column
// pairs per round (round-robin tournament), eight lanes per pair, two rows per lane.  On return the columns of A are
// U diag(sigma) in some order and, if V != null, A_in V = A_out.
template <bool WANT_V>
__device__ void jacobi16(double (*A)[17], double (*V)[17], int lane) {
  const int pr = lane >> 3, sub = lane & 7;
  for (int sweep = 0; sweep < 14; ++sweep) {
    double off = 0.0;
    for (int round = 0; round < 15; ++round) {
      int p, q;
      if (pr == 0) { p = 15; q = round; }
      else { p = (round + pr) % 15; q = (round + 15 - pr) % 15; }
      const double ap0 = A[p][sub], ap1 = A[p][sub + 8], aq0 = A[q][sub], aq1 = A[q][sub + 8];
      double al = ap0 * ap0 + ap1 * ap1, be = aq0 * aq0 + aq1 * aq1, ga = ap0 * aq0 + ap1 * aq1;
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) { al += shfl_xor_d(al, m); be += shfl_xor_d(be, m); ga += shfl_xor_d(ga, m); }
      const double lim = sqrt(al * be);
      double c = 1.0, s = 0.0;
      if (fabs(ga) > 1e-16 * lim && lim > 0.0) {
        off = fmax(off, fabs(ga) / lim);
        const double zeta = (be - al) / (2.0 * ga);
        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        c = 1.0 / sqrt(1.0 + t * t);
        s = c * t;
      }
      A[p][sub] = c * ap0 - s * aq0;  A[p][sub + 8] = c * ap1 - s * aq1;
      A[q][sub] = s * ap0 + c * aq0;  A[q][sub + 8] = s * ap1 + c * aq1;
      if (WANT_V) {
        const double vp0 = V[p][sub], vp1 = V[p][sub + 8], vq0 = V[q][sub], vq1 = V[q][sub + 8];
        V[p][sub] = c * vp0 - s * vq0;  V[p][sub + 8] = c * vp1 - s * vq1;
        V[q][sub] = s * vp0 + c * vq0;  V[q][sub + 8] = s * vp1 + c * vq1;
      }
      WSYNC();
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) off = fmax(off, shfl_xor_d(off, m));
    if (off < 1e-15) break;
  }
}

// What the LAST-ARRIVING workgroup of a site does with the site's summed 16 x 16 matrix (in `G`, f64, row-major).  Called by
// wave 0 only (the other waves have left); `lane` = threadIdx.x.
__device__ void finish_site(const Finish &f, int site, double (*G)[17], double (*W1)[17], double (*W2)[17], double *sig,
                            int lane) {
  if (f.mode == 1) {
    if (f.ritz_out != nullptr) {  // eigenvalues of the (symmetric, PSD) sum = its singular values: one-sided Jacobi on a copy
      for (int idx = lane; idx < 256; idx += 64) W1[idx & 15][idx >> 4] = 0.5 * (G[idx >> 4][idx & 15] + G[idx & 15][idx >> 4]);
      WSYNC();
      jacobi16<false>(W1, nullptr, lane);
      if (lane < 16) {
        double n2 = 0.0;
        for (int r = 0; r < 16; ++r) n2 += W1[lane][r] * W1[lane][r];
        sig[lane] = sqrt(n2);
      }
      WSYNC();
      if (lane == 0) {
        double tot = 0.0, all = 0.0;
        for (int j = 0; j < 16; ++j) all += sig[j];
        for (int k = 0; k < f.rank; ++k) {  // the `rank` largest (selection; 16 values)
          int best = 0;
          for (int j = 1; j < 16; ++j) if (sig[j] > sig[best]) best = j;
          tot += sig[best];
          sig[best] = -1.0;
        }
        *gl(f.ritz_out + 2 * site) = (float)tot;
        *gl(f.ritz_out + 2 * site + 1) = (float)(all - tot);
      }
      WSYNC();
    }
    double tr = 0.0;
    for (int i = 0; i < 16; ++i) tr += G[i][i];
    chol_inverse_regs(G, (double)f.shift_rel * tr / 16.0, f.linv_out + (int64_t)site * 256, lane);
    return;
  }
  // mode 2: SVD of the core C = U S V^T.  Columns of W1 <- columns of C, W2 <- I.
  for (int idx = lane; idx < 256; idx += 64) {
    const int i = idx >> 4, j = idx & 15;
    W1[j][i] = G[i][j];
    W2[j][i] = i == j ? 1.0 : 0.0;
  }
  WSYNC();
  jacobi16<true>(W1, W2, lane);
  if (lane < 16) {
    double n2 = 0.0;
    for (int r = 0; r < 16; ++r) n2 += W1[lane][r] * W1[lane][r];
    sig[lane] = sqrt(n2);
  }
  WSYNC();
  if (lane < 16) {
    const double mine = sig[lane];
    int pos = 0;
    for (int j = 0; j < 16; ++j) pos += (sig[j] > mine || (sig[j] == mine && j < lane)) ? 1 : 0;
    *gl(f.s_out + (int64_t)site * 16 + pos) = (float)mine;
    if (pos < f.rank) {
      const double inv = mine > 0.0 ? 1.0 / mine : 0.0;
      for (int r = 0; r < 16; ++r) {
        *gl(f.ubt + ((int64_t)site * f.rank + pos) * 16 + r) = (float)(W1[lane][r] * inv);
        *gl(f.vb + ((int64_t)site * f.rank + pos) * 16 + r) = (float)W2[lane][r];
      }
    }
  }
}

// The block's 16 x 16 partial (element `t` in thread t < 256) -> the site's partial slab; the last-arriving block of the site
// sums the slabs in block order (deterministic) and finishes.  Returns after the finish; every wave but wave 0 of the last
// block, and every other block, simply leaves.
__device__ void arrive_and_finish(const Finish &f, int site, int64_t block_begin, int nblocks, int local_block, float mine) {
  __shared__ int s_last;
  __shared__ double G[16][17], W1[16][17], W2[16][17], sig[16];
  __shared__ double dsum[4][256];
  const int t = threadIdx.x;
  if (nblocks > 1) {
    // the slab goes out WRITE-THROUGH (agent-scope relaxed store = sc1): no release fence — a buffer_wbl2 per block, ~1500
    // blocks per launch, was most of this kernel's time in the round's first form (68 us against 10 without the hand-off).
    // This is the hand-off cdna_hip_programming.md gives for in-launch reductions on gfx950 ("equally valid ...: sc1
    // (write-through) slab stores -> every wave s_waitcnt vmcnt(0) -> __syncthreads() -> lane 0 relaxed agent fetch_add; the
    // reducer then reads the slabs ... with an acquire fence + plain loads": the fence's buffer_inv sc1 drops the L1 of the CU
    // all waves of the last block share).  It is NOT a release/acquire pair of the HIP memory model on other targets (ADVICE
    // r5): the library builds for gfx950 only (csrc/Makefile's ARCH is a cross-compile knob of the same family, and
    // lora_amd_target_arch() answers "gfx950"); tests/test_gpu_svd_small.py repeats the hand-off thousands of times.
    __hip_atomic_store(f.part + (block_begin + local_block) * 256 + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains before the arrival
    __syncthreads();
    if (t == 0) {
      const unsigned prev = __hip_atomic_fetch_add(f.counters + site, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)(nblocks - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(f.counters + site, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // the site's slabs in block order (fixed: deterministic), 16 bytes per lane, four interleaved block subsets, 8 loads in flight
    const int sb = t >> 6, e4 = t & 63;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const float *pp = f.part + block_begin * 256 + 4 * e4;
    int b = sb;
    for (; b + 28 < nblocks; b += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = gl_ld4(pp + (int64_t)(b + 4 * u) * 256);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
    }
    for (; b < nblocks; b += 4) {
      const float4 v = gl_ld4(pp + (int64_t)b * 256);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    dsum[sb][4 * e4] = a0; dsum[sb][4 * e4 + 1] = a1; dsum[sb][4 * e4 + 2] = a2; dsum[sb][4 * e4 + 3] = a3;
    __syncthreads();
    G[t >> 4][t & 15] = (dsum[0][t] + dsum[1][t]) + (dsum[2][t] + dsum[3][t]);
  } else {
    G[t >> 4][t & 15] = (double)mine;
  }
  __syncthreads();
  if (t >= 64) return;
  finish_site(f, site, G, W1, W2, sig, t);
}

struct Site { int64_t off, rows, block_begin; int32_t blocks, reserved; };
static_assert(sizeof(Site) == sizeof(lora_amd_thin_site), "thin site layout");
// (a class type cannot be copied through an address-space-qualified pointer: field by field, global loads)
__device__ __forceinline__ Site ld_site(const Site *p) {
  Site s;
  s.off = *gl(&p->off); s.rows = *gl(&p->rows); s.block_begin = *gl(&p->block_begin); s.blocks = *gl(&p->blocks);
  s.reserved = 0;
  return s;
}

// ---- G = A^T B over the rows ------------------------------------------------------------------------------------------
// lane l: k-slot l >> 4 (a row of the 4-row step), column l & 15: one coalesced 256-byte load per operand and step.
__global__ __launch_bounds__(256) void thin_gram_kernel(const Site *__restrict__ sites, const int32_t *__restrict__ blockmap,
                                                        const float *__restrict__ a, const float *__restrict__ b, Finish f) {
  __shared__ float red[4][256];
  const int site = gl(blockmap)[blockIdx.x];
  const Site st = ld_site(sites + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row0 = (int64_t)lb * kRowsPerBlock + wave * 64;
  const float *pa = a + st.off, *pb = b + st.off;
  const bool self = a == b;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  float va[16], vb[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int64_t row = row0 + 4 * s + (lane >> 4);
    const bool ok = row < st.rows;
    const int64_t idx = (ok ? row : 0) * 16 + (lane & 15);
    va[s] = ok ? *gl(pa + idx) : 0.f;
    vb[s] = self ? va[s] : (ok ? *gl(pb + idx) : 0.f);
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], vb[s], acc, 0, 0, 0);
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][(4 * (lane >> 4) + v) * 16 + (lane & 15)] = acc[v];
  __syncthreads();
  const int t = threadIdx.x;
  const float mine = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  arrive_and_finish(f, site, st.block_begin, st.blocks, lb, mine);
}

// ---- Y' = Y M^T (M = L^{-1} of the site, 16 x 16) [+ Gram of Y' -> finish] ----------------------------------------------
// D^T form: A operand = M (lane (i, g): M[i][4g .. 4g+3], one per k-step), B operand = the lane's 16-byte piece of row m
// (lane (m, g): Y[m][4g .. 4g+3]); k-slot g of step s stands for column 4g + s on both.  The result lane (m, g) holds
// Y'[m][4g .. 4g+3]: one 16-byte store.
template <bool GRAM>
__global__ __launch_bounds__(256) void thin_apply_kernel(const Site *__restrict__ sites, const int32_t *__restrict__ blockmap,
                                                         const float *__restrict__ src, const float *__restrict__ mats,
                                                         float *__restrict__ dst, Finish f) {
  __shared__ float red[4][256];
  __shared__ float tile[4][16][17];
  const int site = gl(blockmap)[blockIdx.x];
  const Site st = ld_site(sites + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const float4 mv = gl_ld4(mats + (int64_t)site * 256 + m * 16 + 4 * g);
  const float mm[4] = {mv.x, mv.y, mv.z, mv.w};
  const int64_t row0 = (int64_t)lb * kRowsPerBlock + wave * 64;
  float4 y[4];
  bool ok[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int64_t row = row0 + 16 * tl + m;
    ok[tl] = row < st.rows;
    y[tl] = gl_ld4(src + st.off + (ok[tl] ? row : 0) * 16 + 4 * g);
  }
  v4f gacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const float yy[4] = {ok[tl] ? y[tl].x : 0.f, ok[tl] ? y[tl].y : 0.f, ok[tl] ? y[tl].z : 0.f, ok[tl] ? y[tl].w : 0.f};
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mm[s], yy[s], acc, 0, 0, 0);
    if (ok[tl]) gl_st4(dst + st.off + (row0 + 16 * tl + m) * 16 + 4 * g, acc[0], acc[1], acc[2], acc[3]);
    if (GRAM) {
      // the wave's own tile: row m, columns 4g + v; read back with the row on the k-slot (rows beyond the site are zero)
#pragma unroll
      for (int v = 0; v < 4; ++v) tile[wave][m][4 * g + v] = ok[tl] ? acc[v] : 0.f;
      WSYNC();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float e = tile[wave][4 * q + g][m];
        gacc = __builtin_amdgcn_mfma_f32_16x16x4f32(e, e, gacc, 0, 0, 0);
      }
      WSYNC();
    }
  }
  if (!GRAM) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][(4 * g + v) * 16 + m] = gacc[v];
  __syncthreads();
  const int t = threadIdx.x;
  const float mine = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  arrive_and_finish(f, site, st.block_begin, st.blocks, lb, mine);
}

// ---- out[rows][r] = (src M^T)[:, :r] * colscale  (M = [r][16] per site), optional sign rule ------------------------------
struct SignOut {
  float *part;         // [total_blocks][32]: per column (|v| max, signed value at it) ; rows in `rowpart`
  int32_t *rowpart;    // [total_blocks][16]
  unsigned *counters;  // [nsites]
  float *sign;         // [nsites][16]
};

template <bool SIGN>
__global__ __launch_bounds__(256) void thin_rotate_kernel(const Site *__restrict__ sites, const int32_t *__restrict__ blockmap,
                                                          const float *__restrict__ src, const float *__restrict__ mats,
                                                          const float *__restrict__ scale_a, const float *__restrict__ scale_b,
                                                          float *__restrict__ dst, int r, SignOut so) {
  __shared__ float s_abs[4][16], s_val[4][16];
  __shared__ int s_row[4][16];
  __shared__ int s_last;
  const int site = gl(blockmap)[blockIdx.x];
  const Site st = ld_site(sites + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  float mm[4] = {0.f, 0.f, 0.f, 0.f};
  if (m < r) {
    const float4 mv = gl_ld4(mats + ((int64_t)site * r + m) * 16 + 4 * g);
    mm[0] = mv.x; mm[1] = mv.y; mm[2] = mv.z; mm[3] = mv.w;
  }
  float cs[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int c = 4 * g + v;
    float s = 1.f;
    if (c < r) {
      if (scale_a != nullptr) s *= *gl(scale_a + (int64_t)site * 16 + c);
      if (scale_b != nullptr) s *= *gl(scale_b + (int64_t)site * 16 + c);
    }
    cs[v] = s;
  }
  const int64_t row0 = (int64_t)lb * kRowsPerBlock + wave * 64;
  const int64_t doff = (st.off >> 4) * r;
  float4 y[4];
  bool ok[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int64_t row = row0 + 16 * tl + m;
    ok[tl] = row < st.rows;
    y[tl] = gl_ld4(src + st.off + (ok[tl] ? row : 0) * 16 + 4 * g);
  }
  float best_abs[4] = {-1.f, -1.f, -1.f, -1.f}, best_val[4] = {0.f, 0.f, 0.f, 0.f};
  int best_row[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const float yy[4] = {y[tl].x, y[tl].y, y[tl].z, y[tl].w};
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mm[s], yy[s], acc, 0, 0, 0);
    const int64_t row = row0 + 16 * tl + m;
    if (ok[tl]) {
      float *o = dst + doff + row * r + 4 * g;
      if ((r & 3) == 0) {
        if (4 * g < r) gl_st4(o, acc[0] * cs[0], acc[1] * cs[1], acc[2] * cs[2], acc[3] * cs[3]);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) if (4 * g + v < r) *gl(o + v) = acc[v] * cs[v];
      }
      if (SIGN) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float av = fabsf(acc[v]);
          if (av > best_abs[v]) { best_abs[v] = av; best_val[v] = acc[v]; best_row[v] = (int)(row - (int64_t)lb * kRowsPerBlock); }
        }
      }
    }
  }
  if (!SIGN) return;
  // arg-max of |.| per column over the block's rows, first row on ties: over the 16 lanes of a group (same g), then the waves
#pragma unroll
  for (int v = 0; v < 4; ++v) {
#pragma unroll
    for (int msk = 1; msk < 16; msk <<= 1) {
      const float oa = __shfl_xor(best_abs[v], msk, 64), ov = __shfl_xor(best_val[v], msk, 64);
      const int orow = __shfl_xor(best_row[v], msk, 64);
      if (oa > best_abs[v] || (oa == best_abs[v] && orow < best_row[v])) { best_abs[v] = oa; best_val[v] = ov; best_row[v] = orow; }
    }
    if (m == 0) { s_abs[wave][4 * g + v] = best_abs[v]; s_val[wave][4 * g + v] = best_val[v]; s_row[wave][4 * g + v] = best_row[v]; }
  }
  __syncthreads();
  const int t = threadIdx.x;
  const int64_t blk = st.block_begin + lb;
  if (t < 16) {
    float ba = s_abs[0][t], bv = s_val[0][t];
    int br = s_row[0][t];
    for (int w = 1; w < 4; ++w)
      if (s_abs[w][t] > ba || (s_abs[w][t] == ba && s_row[w][t] < br)) { ba = s_abs[w][t]; bv = s_val[w][t]; br = s_row[w][t]; }
    __hip_atomic_store(so.part + blk * 32 + t, ba, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through, as above
    __hip_atomic_store(so.part + blk * 32 + 16 + t, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(so.rowpart + blk * 16 + t, br, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    const unsigned prev = __hip_atomic_fetch_add(so.counters + site, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == (unsigned)(st.blocks - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(so.counters + site, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last || t >= 16) return;
  float ba = -1.f, bv = 0.f;
  for (int b2 = 0; b2 < st.blocks; ++b2) {  // blocks in row order: the first block holding the maximum wins a tie
    const float a2 = *gl(so.part + (st.block_begin + b2) * 32 + t);
    if (a2 > ba) { ba = a2; bv = *gl(so.part + (st.block_begin + b2) * 32 + 16 + t); }
  }
  *gl(so.sign + (int64_t)site * 16 + t) = bv < 0.f ? -1.f : 1.f;
}

// ---- order statistics of a site's joint values (cli_svd.py:39-47's quantile) --------------------------------------------
// values of site i: up[i] = U [N][r] (already scaled and signed) followed by down[i] = V [K][r] * sign[c]; n = (N + K) r.
// key = order-preserving map of the f32 bits; three radix passes (bits 31..21, 20..10, 9..0) each narrow (prefix, k).
struct QSite { int64_t off_u, off_v, n_u, n_v, block_begin; int32_t blocks, reserved; };
static_assert(sizeof(QSite) == sizeof(lora_amd_thin_qsite), "thin qsite layout");
struct QState { unsigned prefix, k_lo, want_next, next_key, nan_seen, done_lo_key, r0, r1; };  // per site, 32 bytes
__device__ __forceinline__ QSite ld_qsite(const QSite *p) {
  QSite s;
  s.off_u = *gl(&p->off_u); s.off_v = *gl(&p->off_v); s.n_u = *gl(&p->n_u); s.n_v = *gl(&p->n_v);
  s.block_begin = *gl(&p->block_begin); s.blocks = *gl(&p->blocks); s.reserved = 0;
  return s;
}

__device__ __forceinline__ unsigned f2key(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int kQElems = 8192;  // values per block

__global__ __launch_bounds__(256) void thin_select_kernel(const QSite *__restrict__ qs, const int32_t *__restrict__ blockmap,
                                                          const float *__restrict__ u, const float *__restrict__ v,
                                                          const float *__restrict__ sign, int r, int pass,
                                                          unsigned *__restrict__ hist, unsigned *__restrict__ counters,
                                                          QState *__restrict__ state, float *__restrict__ out2) {
  __shared__ unsigned h[2048];
  __shared__ unsigned s_min[4];
  __shared__ int s_last;
  const int site = gl(blockmap)[blockIdx.x];
  const QSite st = ld_qsite(qs + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
  const int t = threadIdx.x;
  const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
  const unsigned bins = pass == 2 ? 1024u : 2048u;
  const unsigned prefix = *gl(&state[site].prefix);  // the bits above this pass's digit, already right-aligned
  for (int i = t; i < 2048; i += 256) h[i] = 0u;
  __syncthreads();
  const int64_t n = st.n_u + st.n_v;
  unsigned mn = 0xffffffffu;
  unsigned nan_here = 0u;
  const int64_t e0 = (int64_t)lb * kQElems;
  for (int it = 0; it < kQElems / 256; ++it) {
    const int64_t e = e0 + it * 256 + t;
    if (e < n) {
      float val;
      if (e < st.n_u) val = *gl(u + st.off_u + e);
      else {
        const int64_t ev = e - st.n_u;
        val = *gl(v + st.off_v + ev) * *gl(sign + (int64_t)site * 16 + (int)(ev % r));
      }
      if (val != val) nan_here = 1u;
      const unsigned key = f2key(val);
      const unsigned hi = pass == 0 ? 0u : (key >> (shift + (pass == 1 ? 11 : 10)));
      if (pass == 0 || hi == prefix) atomicAdd(&h[(key >> shift) & (bins - 1)], 1u);
      else if (pass == 2 && hi > prefix) mn = min(mn, key);  // candidates for "the next larger value" beyond the prefix
    }
  }
  __syncthreads();
  unsigned *gh = hist + (int64_t)site * 2048;
  for (int i = t; i < (int)bins; i += 256)
    if (h[i]) __hip_atomic_fetch_add(gh + i, h[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (pass == 2) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mn = min(mn, (unsigned)__shfl_xor((int)mn, m, 64));
    if ((t & 63) == 0) s_min[t >> 6] = mn;
  }
  if (nan_here) __hip_atomic_fetch_or(&state[site].nan_seen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    if (pass == 2) {
      const unsigned m4 = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
      if (m4 != 0xffffffffu) __hip_atomic_fetch_min(&state[site].next_key, m4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // (everything this block published went through device-scope atomics: nothing to write back, no release fence)
    const unsigned prev = __hip_atomic_fetch_add(counters + site, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == (unsigned)(st.blocks - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counters + site, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the last arriver: take the site's histogram (device-scope atomics wrote it) and find the bin of k_lo
  for (int i = t; i < (int)bins; i += 256)  // an atomic exchange reads where the atomic adds landed, and clears for the next pass
    h[i] = __hip_atomic_exchange(gh + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  // which bin holds position k: thread t owns bins 8t .. 8t+7; inclusive scan of the 256 sums (Hillis-Steele in LDS), then the
  // one thread whose range straddles k walks its eight bins
  __shared__ unsigned sc[2][256];
  unsigned own = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) own += (8 * t + i < (int)bins) ? h[8 * t + i] : 0u;
  sc[0][t] = own;
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int d = 1; d < 256; d <<= 1) {
    sc[cur ^ 1][t] = sc[cur][t] + (t >= d ? sc[cur][t - d] : 0u);
    cur ^= 1;
    __syncthreads();
  }
  const unsigned incl = sc[cur][t], excl = incl - own;
  const unsigned k0 = *gl(&state[site].k_lo);
  if (!(excl <= k0 && k0 < incl)) return;   // exactly one thread goes on (k0 < n)
  unsigned k = k0, bin = 8 * t, cum = excl;
  for (; bin < (unsigned)(8 * t + 7); ++bin) {
    if (cum + h[bin] > k) break;
    cum += h[bin];
  }
  k -= cum;
  const unsigned new_prefix = pass == 0 ? bin : ((prefix << (pass == 1 ? 11 : 10)) | bin);
  if (pass < 2) {
    *gl(&state[site].prefix) = new_prefix;
    *gl(&state[site].k_lo) = k;
    return;
  }
  // pass 2: new_prefix is the full key of order statistic k_lo; the next one is the same value if the bin holds more
  // elements beyond position k, else the next non-empty bin of this histogram, else the smallest key beyond the prefix
  const unsigned key_lo = new_prefix;
  unsigned key_hi = key_lo;
  if (k + 1 >= h[bin]) {
    unsigned b2 = bin + 1;
    while (b2 < bins && h[b2] == 0u) ++b2;
    if (b2 < bins) key_hi = (prefix << 10) | b2;
    else {
      const unsigned nk = __hip_atomic_load(&state[site].next_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      key_hi = nk != 0xffffffffu ? nk : key_lo;  // k_lo is the maximum: torch's hi index is clamped to n - 1
    }
  }
  const bool bad = __hip_atomic_load(&state[site].nan_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  const float qnan = __uint_as_float(0x7fc00000u);
  *gl(out2 + (int64_t)site * 2) = bad ? qnan : key2f(key_lo);
  *gl(out2 + (int64_t)site * 2 + 1) = bad ? qnan : key2f(key_hi);
}

// up[i][n][c] = clamp(U[n][c], -hi, hi) in place; down[i][c][k] = clamp(V[k][c] * sign[c], -hi, hi)  ([K][r] -> [r][K])
__global__ __launch_bounds__(256) void thin_clamp_kernel(const QSite *__restrict__ qs, const int32_t *__restrict__ blockmap,
                                                         float *__restrict__ u, const float *__restrict__ v,
                                                         const float *__restrict__ sign, const float *__restrict__ hi,
                                                         float *__restrict__ down, int r) {
  const int site = gl(blockmap)[blockIdx.x];
  const QSite st = ld_qsite(qs + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
  const int t = threadIdx.x;
  const float h = *gl(hi + site);
  const int64_t n = st.n_u + st.n_v, K = st.n_v / r;
  const int64_t e0 = (int64_t)lb * kQElems;
  for (int it = 0; it < kQElems / 256; ++it) {
    const int64_t e = e0 + it * 256 + t;
    if (e >= n) break;
    if (e < st.n_u) {
      const float x = *gl(u + st.off_u + e);
      const float y = fminf(fmaxf(x, -h), h);  // torch.minimum(torch.maximum(x, -h), h): NaN in either operand propagates
      *gl(u + st.off_u + e) = (x != x || h != h) ? __uint_as_float(0x7fc00000u) : y;
    } else {
      // the block's elements in OUTPUT order (c major): consecutive threads write consecutive k
      const int64_t ev = e - st.n_u;
      const int64_t c = ev / K, k = ev - c * K;
      const float x = *gl(v + st.off_v + k * r + c) * *gl(sign + (int64_t)site * 16 + (int)c);
      const float y = fminf(fmaxf(x, -h), h);
      *gl(down + st.off_v + ev) = (x != x || h != h) ? __uint_as_float(0x7fc00000u) : y;
    }
  }
}

// ---- a thin matrix F [rows][16] f32 -> the (hi, lo) 16-bit MFMA fragments the planes product reads ----------------------------
// dst (16-bit elements, at element offset 2 off): [k step = 32 rows][hi 512 | lo 512], lane l = (j = l & 15, kq = l >> 4) holds
// F[32 ks + 8 kq + e][j], e = 0..7 (rows past the site: zero).  One workgroup = 256 rows = 8 k-steps, two pieces per thread.
template <class E>
__global__ __launch_bounds__(256) void thin_pack_kernel(const Site *__restrict__ sites, const int32_t *__restrict__ blockmap,
                                                        const float *__restrict__ src, typename E::storage *__restrict__ dst) {
  using S = typename E::storage;
  const int site = gl(blockmap)[blockIdx.x];
  const Site st = ld_site(sites + site);
  const int lb = (int)(blockIdx.x - st.block_begin);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int id = u * 256 + threadIdx.x, ksl = id >> 6, lane = id & 63;
    const int64_t ks = (int64_t)lb * 8 + ksl;
    if (ks * 32 >= st.rows) continue;
    const int j = lane & 15, kq = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t row = ks * 32 + 8 * kq + e;
      v[e] = row < st.rows ? *gl(src + st.off + row * 16 + j) : 0.f;
    }
    mu32x4 hi, lo;
    split_hi_lo<E>(v, hi, lo);
    S *o = dst + 2 * st.off + ks * 1024 + lane * 8;
    *gl(reinterpret_cast<mu32x4 *>(o)) = hi;
    *gl(reinterpret_cast<mu32x4 *>(o + 512)) = lo;
  }
}

// ---- dW = W_tuned - W_base (cli_svd.py:30-32) -> the (hi, lo) 16-bit planes of dW AND of dW^T, and |dW|_F^2 ----------------
// One read of the two weights, no f32 residual in memory (rounds 2-4: sub_ragged wrote it, split16_transpose read it back:
// 2 x 2.9 GB of the distillation's traffic).  One 64 x 64 tile per workgroup; the transposed planes through an LDS image of the
// tile (2-byte column gathers, 16-byte stores along n).  The tile's sum of squares goes to norm_part[tile] (summed per site on
// the host side of the launch: the stopping rule of the iteration needs |dW|^2 - E_r).
struct RDesc { const void *tuned, *base; void *hi, *lo, *thi, *tlo; int32_t N, K; int64_t tile_begin; };
static_assert(sizeof(RDesc) == sizeof(lora_amd_resid_desc), "resid desc layout");
constexpr int kRT = 64, kRPitch = kRT * 2 + 4;   // bytes per image row

template <class EIN, class E>
__global__ __launch_bounds__(256) void split16_residual_kernel(const RDesc *__restrict__ descs, int n, float *__restrict__ norm_part) {
  using SI = typename EIN::storage;
  using S = typename E::storage;
  __shared__ __attribute__((aligned(16))) unsigned char s_hi[kRT * kRPitch], s_lo[kRT * kRPitch];
  __shared__ float s_sq[4];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (*gl(&descs[mid].tile_begin) <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RDesc *dp = descs + lo;
  const int N = *gl(&dp->N), K = *gl(&dp->K);
  const int tiles_k = (K + kRT - 1) / kRT;
  const int64_t t = (int64_t)blockIdx.x - *gl(&dp->tile_begin);
  const int tn = (int)(t / tiles_k), tk = (int)(t - (int64_t)tn * tiles_k);
  const int n0 = tn * kRT, k0 = tk * kRT;
  const int tid = threadIdx.x;
  const SI *pa = reinterpret_cast<const SI *>(*gl(&dp->tuned)), *pb = reinterpret_cast<const SI *>(*gl(&dp->base));
  S *ph = reinterpret_cast<S *>(*gl(&dp->hi)), *pl = reinterpret_cast<S *>(*gl(&dp->lo));
  S *th = reinterpret_cast<S *>(*gl(&dp->thi)), *tl = reinterpret_cast<S *>(*gl(&dp->tlo));
  float sq = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rl = u * 32 + (tid >> 3), c8 = tid & 7;
    const int row = n0 + rl, col = k0 + c8 * 8;
    mu32x4 vh = mu32x4{0u, 0u, 0u, 0u}, vl = vh;
    if (row < N && col < K) {   // K % 8 == 0
      float a[8], b[8], v[8];
      load8<EIN>(pa + (int64_t)row * K + col, a);
      load8<EIN>(pb + (int64_t)row * K + col, b);
#pragma unroll
      // the reference subtracts in the weights' own dtype and only then calls .float() (cli_svd.py:30-33, 57-60: fp16
      // pipelines): a 16-bit input pair gives the difference ROUNDED to that dtype; f32 inputs subtract in f32
      for (int e = 0; e < 8; ++e) { v[e] = round_to<EIN>(a[e] - b[e]); sq = fmaf(v[e], v[e], sq); }
      split_hi_lo<E>(v, vh, vl);
      *gl(reinterpret_cast<mu32x4 *>(ph + (int64_t)row * K + col)) = vh;
      *gl(reinterpret_cast<mu32x4 *>(pl + (int64_t)row * K + col)) = vl;
    }
    uint32_t *ih = reinterpret_cast<uint32_t *>(s_hi + rl * kRPitch + c8 * 16), *il = reinterpret_cast<uint32_t *>(s_lo + rl * kRPitch + c8 * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ih[i] = vh[i]; il[i] = vl[i]; }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) sq += __shfl_xor(sq, m, 64);
  if ((tid & 63) == 0) s_sq[tid >> 6] = sq;
  __syncthreads();
  if (tid == 0) *gl(norm_part + blockIdx.x) = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
  // columns of the tile: task = (column k, chunk of 8 rows): 64 x 8 tasks, k fastest in groups of 4
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int task = u * 256 + tid;
    const int kq = task >> 5, rem = task & 31;
    const int k = kq * 4 + (rem & 3), ch = rem >> 2;
    const int col = k0 + k, row0 = n0 + ch * 8;
    if (col >= K || row0 >= N) continue;   // N % 8 == 0
    const unsigned char *sh = s_hi + (ch * 8) * kRPitch + k * 2, *sl = s_lo + (ch * 8) * kRPitch + k * 2;
    mu32x4 oh, ol;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      oh[i] = (uint32_t)*reinterpret_cast<const unsigned short *>(sh + (2 * i) * kRPitch) |
              ((uint32_t)*reinterpret_cast<const unsigned short *>(sh + (2 * i + 1) * kRPitch) << 16);
      ol[i] = (uint32_t)*reinterpret_cast<const unsigned short *>(sl + (2 * i) * kRPitch) |
              ((uint32_t)*reinterpret_cast<const unsigned short *>(sl + (2 * i + 1) * kRPitch) << 16);
    }
    *gl(reinterpret_cast<mu32x4 *>(th + (int64_t)col * N + row0)) = oh;
    *gl(reinterpret_cast<mu32x4 *>(tl + (int64_t)col * N + row0)) = ol;
  }
}

}  // namespace
}  // namespace lora_amd

using namespace lora_amd;

static Finish make_finish(const lora_amd_thin_finish *fin) {
  Finish f = {};
  if (fin != nullptr) {
    f.part = fin->part; f.counters = fin->counters; f.mode = fin->mode; f.shift_rel = fin->shift_rel;
    f.linv_out = fin->linv_out; f.ritz_out = fin->ritz_out; f.ubt = fin->ubt; f.vb = fin->vb; f.s_out = fin->s_out;
    f.rank = fin->rank;
  }
  return f;
}

static int check_finish(const lora_amd_thin_finish *fin, const char *who) {
  LORA_AMD_CHECK(fin != nullptr && fin->part && fin->counters && (fin->mode == 1 || fin->mode == 2), LORA_AMD_EINVAL,
                 "%s: finish needs part, counters and mode 1 (Cholesky) or 2 (SVD)", who);
  LORA_AMD_CHECK(fin->mode != 1 || fin->linv_out, LORA_AMD_EINVAL, "%s: mode 1 needs linv_out", who);
  LORA_AMD_CHECK(fin->mode != 2 || (fin->ubt && fin->vb && fin->s_out), LORA_AMD_EINVAL, "%s: mode 2 needs ubt, vb, s_out", who);
  LORA_AMD_CHECK(fin->rank >= 1 && fin->rank <= 16, LORA_AMD_EINVAL, "%s: rank in [1, 16]", who);
  return LORA_AMD_OK;
}

extern "C" int lora_amd_thin_gram(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                  const float *a, const float *b, const lora_amd_thin_finish *fin, void *stream) {
  LORA_AMD_CHECK(sites_dev && blockmap_dev && a && total_blocks >= 1, LORA_AMD_EINVAL, "thin_gram: null argument");
  if (int rc = check_finish(fin, "thin_gram")) return rc;
  hipLaunchKernelGGL(thin_gram_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const Site *)sites_dev, blockmap_dev, a, b ? b : a, make_finish(fin));
  return check_launch("lora_amd_thin_gram");
}

extern "C" int lora_amd_thin_apply(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                   const float *src, const float *mats, float *dst, const lora_amd_thin_finish *fin,
                                   void *stream) {
  LORA_AMD_CHECK(sites_dev && blockmap_dev && src && mats && dst && total_blocks >= 1, LORA_AMD_EINVAL,
                 "thin_apply: null argument");
  LORA_AMD_CHECK(src != dst, LORA_AMD_EINVAL, "thin_apply: in place is not supported");
  if (fin != nullptr) {
    if (int rc = check_finish(fin, "thin_apply")) return rc;
    hipLaunchKernelGGL(thin_apply_kernel<true>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const Site *)sites_dev, blockmap_dev, src, mats, dst, make_finish(fin));
  } else {
    hipLaunchKernelGGL(thin_apply_kernel<false>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const Site *)sites_dev, blockmap_dev, src, mats, dst, make_finish(nullptr));
  }
  return check_launch("lora_amd_thin_apply");
}

extern "C" int lora_amd_thin_rotate(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                    const float *src, const float *mats, int32_t rank, const float *scale_a,
                                    const float *scale_b, float *dst, float *sign_part, int32_t *sign_rows,
                                    uint32_t *counters, float *sign_out, void *stream) {
  LORA_AMD_CHECK(sites_dev && blockmap_dev && src && mats && dst && total_blocks >= 1 && rank >= 1 && rank <= 16,
                 LORA_AMD_EINVAL, "thin_rotate: null argument or rank outside [1, 16]");
  SignOut so = {sign_part, sign_rows, counters, sign_out};
  if (sign_out != nullptr) {
    LORA_AMD_CHECK(sign_part && sign_rows && counters, LORA_AMD_EINVAL, "thin_rotate: the sign rule needs its workspaces");
    hipLaunchKernelGGL(thin_rotate_kernel<true>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const Site *)sites_dev, blockmap_dev, src, mats, scale_a, scale_b, dst, (int)rank, so);
  } else {
    hipLaunchKernelGGL(thin_rotate_kernel<false>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const Site *)sites_dev, blockmap_dev, src, mats, scale_a, scale_b, dst, (int)rank, so);
  }
  return check_launch("lora_amd_thin_rotate");
}

extern "C" int lora_amd_thin_select(const lora_amd_thin_qsite *qsites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                    const float *u, const float *v, const float *sign, int32_t rank, int32_t pass,
                                    uint32_t *hist, uint32_t *counters, void *state, float *out2, void *stream) {
  LORA_AMD_CHECK(qsites_dev && blockmap_dev && u && v && sign && hist && counters && state && out2 && total_blocks >= 1,
                 LORA_AMD_EINVAL, "thin_select: null argument");
  LORA_AMD_CHECK(pass >= 0 && pass <= 2 && rank >= 1 && rank <= 16, LORA_AMD_EINVAL, "thin_select: pass in [0, 2], rank in [1, 16]");
  hipLaunchKernelGGL(thin_select_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const QSite *)qsites_dev, blockmap_dev, u, v, sign, (int)rank, (int)pass, hist, counters, (QState *)state,
                     out2);
  return check_launch("lora_amd_thin_select");
}

extern "C" int lora_amd_thin_clamp(const lora_amd_thin_qsite *qsites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                   float *u, const float *v, const float *sign, const float *hi, float *down, int32_t rank,
                                   void *stream) {
  LORA_AMD_CHECK(qsites_dev && blockmap_dev && u && v && sign && hi && down && total_blocks >= 1 && rank >= 1 && rank <= 16,
                 LORA_AMD_EINVAL, "thin_clamp: null argument");
  hipLaunchKernelGGL(thin_clamp_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const QSite *)qsites_dev, blockmap_dev, u, v, sign, hi, down, (int)rank);
  return check_launch("lora_amd_thin_clamp");
}

extern "C" int lora_amd_split16_residual(const lora_amd_resid_desc *descs_dev, int32_t n, int64_t tiles, int32_t in_dtype,
                                         int32_t plane_dtype, float *norm_part, void *stream) {
  LORA_AMD_CHECK(descs_dev && norm_part && n >= 1 && tiles >= 1, LORA_AMD_EINVAL, "split16_residual: null argument");
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16, LORA_AMD_EINVAL, "split16_residual: bf16 planes");
  LORA_AMD_CHECK(dtype_ok(in_dtype), LORA_AMD_EINVAL, "split16_residual: weights must be f32, f16 or bf16");
  const dim3 grid((unsigned)tiles), block(256);
  const RDesc *d = (const RDesc *)descs_dev;
  if (in_dtype == LORA_AMD_F32)
    hipLaunchKernelGGL((split16_residual_kernel<f32_t, bf16_t>), grid, block, 0, (hipStream_t)stream, d, (int)n, norm_part);
  else if (in_dtype == LORA_AMD_F16)
    hipLaunchKernelGGL((split16_residual_kernel<f16_t, bf16_t>), grid, block, 0, (hipStream_t)stream, d, (int)n, norm_part);
  else
    hipLaunchKernelGGL((split16_residual_kernel<bf16_t, bf16_t>), grid, block, 0, (hipStream_t)stream, d, (int)n, norm_part);
  return check_launch("lora_amd_split16_residual");
}

extern "C" int lora_amd_thin_pack(const lora_amd_thin_site *sites_dev, const int32_t *blockmap_dev, int64_t total_blocks,
                                  const float *src, void *dst, int32_t plane_dtype, void *stream) {
  LORA_AMD_CHECK(sites_dev && blockmap_dev && src && dst && total_blocks >= 1, LORA_AMD_EINVAL, "thin_pack: null argument");
  LORA_AMD_CHECK(plane_dtype == LORA_AMD_BF16 || plane_dtype == LORA_AMD_F16, LORA_AMD_EINVAL, "thin_pack: 16-bit fragments only");
  const dim3 grid((unsigned)total_blocks), block(256);
  if (plane_dtype == LORA_AMD_BF16)
    hipLaunchKernelGGL(thin_pack_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const Site *)sites_dev, blockmap_dev, src,
                       reinterpret_cast<__bf16 *>(dst));
  else
    hipLaunchKernelGGL(thin_pack_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, (const Site *)sites_dev, blockmap_dev, src,
                       reinterpret_cast<_Float16 *>(dst));
  return check_launch("lora_amd_thin_pack");
}
