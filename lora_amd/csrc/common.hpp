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
  Chunk8<E> c = *reinterpret_cast<const Chunk8<E> *>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = E::to_f(c.v[i]);
}
template <class E>
__device__ inline void store8(typename E::storage *p, const float (&in)[8]) {
  Chunk8<E> c;
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = E::from_f(in[i]);
  *reinterpret_cast<Chunk8<E> *>(p) = c;
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
