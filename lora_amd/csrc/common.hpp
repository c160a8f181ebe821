// Shared device helpers for the gfx950 LoRA kernels (wave64, 16-byte lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lora_amd.h"

namespace lora_amd {

constexpr int kWave = 64;

void set_error(const char *fmt, ...);

#define LORA_AMD_CHECK(cond, code, ...)      \
  do {                                       \
    if (!(cond)) {                           \
      ::lora_amd::set_error(__VA_ARGS__);    \
      return (code);                         \
    }                                        \
  } while (0)

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LORA_AMD_ELAUNCH;
  }
  return LORA_AMD_OK;
}

inline int dtype_size(int dt) { return dt == LORA_AMD_F32 ? 4 : 2; }
inline bool dtype_ok(int dt) {
  return dt == LORA_AMD_F32 || dt == LORA_AMD_F16 || dt == LORA_AMD_BF16;
}

// ---- global address space ---------------------------------------------------
// A pointer read out of a descriptor table in memory (every batched / ragged kernel here) has no address space the
// compiler can see: it emits FLAT loads and stores, which count on lgkmcnt as well as vmcnt — every `s_waitcnt lgkmcnt(0)`
// that guards an LDS read then ALSO waits for all global loads in flight (measured round 4: the matrix-core factor pass spent
// one memory round trip per LDS wait; 640 flat_load / 0 global_load in its code object).  All device data of this library
// is global memory: the helpers below and the kernels' own accesses go through gl(p).
#define LORA_AMD_AS_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const T LORA_AMD_AS_GLOBAL *gl(const T *p) { return (const T LORA_AMD_AS_GLOBAL *)p; }
template <class T>
__device__ __forceinline__ T LORA_AMD_AS_GLOBAL *gl(T *p) { return (T LORA_AMD_AS_GLOBAL *)p; }
// 16 bytes of floats through a plain vector type (class types such as float4 cannot be copied through an
// address-space-qualified pointer)
typedef float gf32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gl_ld4(const float *p) {
  const gf32x4 v = *gl(reinterpret_cast<const gf32x4 *>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void gl_st4(float *p, float a, float b, float c, float d) {
  *gl(reinterpret_cast<gf32x4 *>(p)) = gf32x4{a, b, c, d};
}

// ---- element traits: T is the storage type -------------------------------
struct f32_t {
  using storage = float;
  static constexpr int kCode = LORA_AMD_F32;
  __device__ static float to_f(storage v) { return v; }
  __device__ static storage from_f(float v) { return v; }
};
struct f16_t {
  using storage = _Float16;
  static constexpr int kCode = LORA_AMD_F16;
  __device__ static float to_f(storage v) { return (float)v; }
  __device__ static storage from_f(float v) { return (_Float16)v; }  // RNE
};
struct bf16_t {
  using storage = __bf16;
  static constexpr int kCode = LORA_AMD_BF16;
  __device__ static float to_f(storage v) { return (float)v; }
  __device__ static storage from_f(float v) { return (__bf16)v; }  // v_cvt_pk_bf16_f32, RNE
};

// A 16-byte (or 32-byte for f32) chunk of 8 elements, the unit every lane moves.
template <class E>
struct alignas(sizeof(typename E::storage) * 8) Chunk8 {
  typename E::storage v[8];
};

template <class E>
__device__ inline void load8(const typename E::storage *p, float (&out)[8]) {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  union { V v; Chunk8<E> c; } u;
  u.v = *gl(reinterpret_cast<const V *>(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = E::to_f(u.c.v[i]);
}
template <class E>
__device__ inline void store8(typename E::storage *p, const float (&in)[8]) {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  union { V v; Chunk8<E> c; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.c.v[i] = E::from_f(in[i]);
  *gl(reinterpret_cast<V *>(p)) = u.v;
}

// Branch-free masked loads: the address is always a valid one (callers clamp it), the value is zeroed afterwards.
// A load under `if (ok)` becomes a branch with its own s_waitcnt, i.e. one HBM round trip PER load; selected loads
// all issue back to back.  The 16-byte vector type also carries the alignment the compiler otherwise gives up on for
// `plane + row * W` (it then splits the access into four 2-byte-aligned pieces).
template <class E>
__device__ inline void load8_sel(const typename E::storage *p, bool ok, float (&out)[8]) {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  union { V v; Chunk8<E> c; } u;
  u.v = *gl(reinterpret_cast<const V *>(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = ok ? E::to_f(u.c.v[i]) : 0.f;
}
// The same in two steps, for kernels that want every load of a thread issued before the first value is converted
// (hipcc otherwise tends to convert chunk 0 — s_waitcnt vmcnt(0) — before it issues the load of chunk 1).
template <class E>
struct Raw8 {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  V v;
};
template <class E>
__device__ inline Raw8<E> load8_raw(const typename E::storage *p) {
  Raw8<E> r;
  r.v = *gl(reinterpret_cast<const typename Raw8<E>::V *>(p));
  return r;
}
template <class E>
__device__ inline void unpack8_sel(const Raw8<E> &r, bool ok, float (&out)[8]) {
  union { typename Raw8<E>::V v; Chunk8<E> c; } u;
  u.v = r.v;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = ok ? E::to_f(u.c.v[i]) : 0.f;
}
__device__ inline void ld8f_sel(const float *p, bool ok, float (&v)[8]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 a = *gl(reinterpret_cast<const f4 *>(p)), b = *gl(reinterpret_cast<const f4 *>(p + 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = ok ? a[i] : 0.f; v[i + 4] = ok ? b[i] : 0.f; }
}

// Non-temporal variants for data that is streamed exactly once (the merge's W): the lines are not kept in L2/MALL
// ahead of data other kernels will re-use.
template <class E>
__device__ inline void load8_nt(const typename E::storage *p, float (&out)[8]) {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  union { V v; Chunk8<E> c; } u;
  u.v = __builtin_nontemporal_load(gl(reinterpret_cast<const V *>(p)));
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = E::to_f(u.c.v[i]);
}
template <class E>
__device__ inline void store8_nt(typename E::storage *p, const float (&in)[8]) {
  using V = unsigned int __attribute__((ext_vector_type(sizeof(typename E::storage) * 2)));
  union { V v; Chunk8<E> c; } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.c.v[i] = E::from_f(in[i]);
  __builtin_nontemporal_store(u.v, gl(reinterpret_cast<V *>(p)));
}

// Round an f32 value to E's precision and back (used to reproduce torch's
// per-op rounding in the merge kernel).
template <class E>
__device__ inline float round_to(float v) {
  return E::to_f(E::from_f(v));
}

// ---- Philox4x32-10: counter-based RNG for dropout (no mask tensor in HBM) --
struct Philox {
  uint32_t k0, k1;
  __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  // counter = (idx_lo, idx_hi, off_lo, off_hi) -> 4 x u32
  __device__ inline void operator()(uint64_t idx, uint64_t off, uint32_t (&out)[4]) const {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32);
    uint32_t c2 = (uint32_t)off, c3 = (uint32_t)(off >> 32);
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      uint32_t h0, l0, h1, l1;
      mulhilo(0xD2511F53u, c0, h0, l0);
      mulhilo(0xCD9E8D57u, c2, h1, l1);
      c0 = h1 ^ c1 ^ a;
      c1 = l1;
      c2 = h0 ^ c3 ^ b;
      c3 = l0;
      a += 0x9E3779B9u;
      b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

// Dropout multipliers for the 8 elements of chunk `chunk_idx` (one Philox call:
// 8 x 16-bit uniforms).  keep  <=>  u16 >= round(p * 65536);  kept values are
// scaled by 1/(1-p) as nn.Dropout does (lora.py:45).
__device__ inline void dropout_mult8(uint64_t seed, uint64_t offset, uint64_t chunk_idx,
                                     float p, float (&m)[8]) {
  uint32_t r[4];
  Philox ph(seed);
  ph(chunk_idx, offset, r);
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
  const float keep = 1.0f / (1.0f - p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[2 * i] = ((r[i] & 0xFFFFu) >= thr) ? keep : 0.0f;
    m[2 * i + 1] = ((r[i] >> 16) >= thr) ? keep : 0.0f;
  }
}

// Effective Philox offset of a launch: the scalar argument plus (optionally) a value read from device memory.  The
// device part is what makes dropout correct under hipGraph replay and activation checkpointing: the host draws it with
// torch's generator INSIDE the captured / recomputed region, so every replay sees a fresh value and a recompute sees
// the forward's value again, while the scalar arguments stay baked into the graph.
__device__ inline uint64_t dropout_offset(uint64_t offset, const uint64_t *offset_dev) {
  return offset_dev != nullptr ? offset + __builtin_nontemporal_load(gl(offset_dev)) : offset;
}

// ---- cross-lane moves on the DPP path (no LDS crossbar traffic) ------------
template <int CTRL>
__device__ inline float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float lane_from_prev(float v) { return dpp_mov<0x138>(v); }  // wave_shr:1  lane l <- l-1 (lane 0 <- 0)
__device__ inline float lane_from_next(float v) { return dpp_mov<0x130>(v); }  // wave_shl:1  lane l <- l+1 (lane 63 <- 0)

// Sums of FOUR per-lane values over groups of 2^logn consecutive lanes (logn in [2, 6]) with a butterfly instead of
// four independent xor-reductions: 3 quad permutes transpose-and-add the 4 values over the 2 low lane bits, then lanes
// with equal (lane & 3) are folded with rotate-by-4 / rotate-by-8 inside a 16-lane row (DPP) and xor-16 / xor-32
// across rows (LDS permutes).  5..7 cross-lane ops instead of 4*logn.
// Result: valid in the lanes whose position inside the group is < 4; lane l holds the group total of value idx4(l).
__device__ inline int idx4(int lane) { return ((lane & 1) ? 2 : 0) + ((lane & 2) ? 1 : 0); }
__device__ inline float group_sum4(float d0, float d1, float d2, float d3, int lane, int logn) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float k0 = b0 ? d2 : d0, s0 = b0 ? d0 : d2;
  float k1 = b0 ? d3 : d1, s1 = b0 ? d1 : d3;
  k0 += dpp_mov<0xB1>(s0);  // quad_perm [1,0,3,2] = xor 1
  k1 += dpp_mov<0xB1>(s1);
  float k = b1 ? k1 : k0, s = b1 ? k0 : k1;
  k += dpp_mov<0x4E>(s);    // quad_perm [2,3,0,1] = xor 2
  if (logn >= 3) k += dpp_mov<0x12C>(k);  // row_ror:12: lane i <- lane i+4 of its 16-lane row (positions 0..3 <- 4..7)
  if (logn >= 4) k += dpp_mov<0x128>(k);  // row_ror:8:  lane i <- lane i+8
  if (logn >= 5) k += __shfl_xor(k, 16, 64);
  if (logn >= 6) k += __shfl_xor(k, 32, 64);
  return k;
}
__device__ inline float wave_sum4(float d0, float d1, float d2, float d3, int lane) {
  return group_sum4(d0, d1, d2, d3, lane, 6);
}

// XCD-major remap of a 1-D grid: consecutive logical tiles land on the same XCD
// (block b is observed on XCD b % 8; speed only, never correctness).
__device__ inline int64_t xcd_remap(int64_t bid, int64_t nblk) {
  constexpr int kXcd = 8;
  int64_t q = nblk / kXcd, rem = nblk % kXcd;
  int64_t xcd = bid % kXcd, idx = bid / kXcd;
  int64_t base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  return base + idx;
}

}  // namespace lora_amd
