// Shared pieces of the matrix-core kernels (factor_mfma.hip, rank16_mfma.hip): v_mfma_f32_16x16x32 on 16-bit operands.
//   A operand: lane l holds row (l & 15), k = 8 (l >> 4) + e, e = 0..7 (16 bytes)
//   B operand: lane l holds column (l & 15), k = 8 (l >> 4) + e
//   D / C:     lane l holds column (l & 15), rows 4 (l >> 4) + reg, reg = 0..3
#pragma once
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

// eight f32 values -> their 16-bit "hi" parts and the 16-bit residues "lo" (v ~ hi + lo to ~16 mantissa bits): the
// operands that are not data (factors, T, Gt) enter the matrix pipe as two fragments into the same accumulator
template <class E>
__device__ __forceinline__ void split_hi_lo(const float (&v)[8], mu32x4 &hi, mu32x4 &lo) {
  union { Chunk8<E> c; mu32x4 u; } h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h.c.v[e] = E::from_f(v[e]);
    l.c.v[e] = E::from_f(v[e] - E::to_f(h.c.v[e]));
  }
  hi = h.u;
  lo = l.u;
}

// keep-mask of the 8 elements of 16-byte chunk `chunk` under nn.Dropout(p) as the forward drew it (common.hpp's
// dropout_mult8: keep <=> u16 >= thr), as AND masks over the four dwords of a 16-bit chunk
__device__ __forceinline__ mu32x4 dropout_and8(uint64_t seed, uint64_t off, uint64_t chunk, uint32_t thr) {
  uint32_t rr[4];
  Philox ph(seed);
  ph(chunk, off, rr);
  mu32x4 m;
#pragma unroll
  for (int w = 0; w < 4; ++w) m[w] = ((rr[w] & 0xFFFFu) >= thr ? 0x0000FFFFu : 0u) | ((rr[w] >> 16) >= thr ? 0xFFFF0000u : 0u);
  return m;
}

}  // namespace lora_amd
