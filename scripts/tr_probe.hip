// Probe of ds_read_b64_tr_b16 on gfx950 (the LDS transpose read csrc/factor_mfma.hip builds its phase 2 on):
// LDS holds element id = byte offset / 2; every lane reads with the address pattern fm_colfrag uses on a [16 rows][pitch]
// tile and writes what it got.  Expected (the semantics scripts/fm_model.py assumes): lane (q, i) receives rows
// 4q .. 4q+3 of column i.    hipcc --offload-arch=gfx950 -O2 scripts/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
constexpr int kPitch = 96;  // bytes: 16 columns used, padded
__global__ void probe(short *out) {
  __shared__ short lds[16 * kPitch / 2];
  for (int i = threadIdx.x; i < 16 * kPitch / 2; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, q = l >> 4, i = l & 15;
  const char *p = (const char *)lds + (4 * q + (i >> 2)) * kPitch + (4 * (i & 3)) * 2;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)p);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short *d, h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const int q = l >> 4, i = l & 15;
      const int want = ((4 * q + e) * kPitch) / 2 + i;  // row 4q + e, column i
      if (h[l * 4 + e] != want) {
        if (bad < 16) printf("lane %d elem %d: got id %d (row %d col %d) want row %d col %d\n", l, e, h[l * 4 + e],
                             (h[l * 4 + e] * 2) / kPitch, ((h[l * 4 + e] * 2) % kPitch) / 2, 4 * q + e, i);
        ++bad;
      }
    }
  printf("tr_probe mismatches: %d\n", bad);
  return bad ? 1 : 0;
}
