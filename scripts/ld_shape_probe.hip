// How fast does gfx950 stream a row-major [M, C] bf16 matrix when every wave-level load has the SHAPE of an MFMA operand?
//
// The factor pass (lora_amd/csrc/factor_mfma.hip) loads its operands as v_mfma_f32_16x16x32 A fragments straight from the
// row-major tensor: lane (row = l & 15, q = l >> 4) reads 16 bytes at row * ld + (4 cg + q) * 16 — one instruction touches 16
// rows x 64 B, i.e. HALF of sixteen 128-byte lines; the other half of every line belongs to the neighbouring 32-column group,
// which the round-4 kernel gives to ANOTHER wave.  cdna_hip_programming.md measured +18..45 % for fragment-shaped against
// full-line loads in a GEMM; this probe measures it for pure streaming at the factor pass's occupancy:
//   mode 0  fragment-shaped, 32-column groups dealt round-robin to the 4 waves (the round-4..6 kernel)
//   mode 1  fragment-shaped, both halves of a line in the SAME wave, issued back to back
//   mode 2  full lines: lane (row = l >> 3, chunk = l & 7), 8 rows x 128 B per instruction (not an MFMA operand: the ceiling)
//   mode 3  the SAME 16 rows x 64 B per instruction as mode 0, but FOUR CONSECUTIVE LANES on one row's 64 bytes
//           (row = l >> 2, chunk = l & 3): 16 contiguous runs per instruction instead of 64 scattered 16-byte accesses
// A workgroup = 256 threads = 64 rows x C columns, every load issued before the first use (as the kernel's resident block).
// usage: ld_shape_probe [M] [C] [wrap]   wrap > 0: block b reads row block b % wrap — a footprint of wrap x 64 rows that stays in
//        the L2 (wrap = 64: 2.6 MB at C = 320) or the MALL (wrap = 512): what the LOAD PATH of a CU (address coalescer, L1 tag
//        pipeline) sustains for each shape when HBM is out of the picture (round 6, call c25)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define AS1 __attribute__((address_space(1)))

template <int MODE, int NG>   // NG = C / 32 column groups
__global__ __launch_bounds__(256, 2) void probe(const unsigned short *__restrict__ x, int64_t ld, unsigned *__restrict__ out,
                                                int lds_words, int wrap) {
  extern __shared__ unsigned pad[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)(wrap > 0 ? blockIdx.x % wrap : blockIdx.x) * 64;
  const unsigned short *base = x + m0 * ld;
  u32x4 acc = {0u, 0u, 0u, 0u};
  if (MODE == 2) {
    // wave w: rows 16 w .. 16 w + 15, 8 rows per instruction, NG / 2 lines per row
    constexpr int NL = NG / 2;
    u32x4 v[2 * NL];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int ln = 0; ln < NL; ++ln)
        v[h * NL + ln] = *(const u32x4 AS1 *)(base + (int64_t)(wave * 16 + h * 8 + (lane >> 3)) * ld + ln * 64 + (lane & 7) * 8);
#pragma unroll
    for (int i = 0; i < 2 * NL; ++i) acc ^= v[i];
  } else {
    const int jj = MODE == 3 ? lane >> 2 : lane & 15, q = MODE == 3 ? lane & 3 : lane >> 4;
    constexpr int GPW = (NG + 3) / 4;   // groups per wave (upper bound)
    u32x4 v[GPW * 4];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      int cg;
      if (MODE == 0 || MODE == 3) {
        cg = wave + 4 * i;
      } else {   // full rounds of 4 line pairs (8 groups), the remaining groups one per wave
        constexpr int NR = NG / 8;
        cg = i < 2 * NR ? 2 * (wave + 4 * (i >> 1)) + (i & 1) : 8 * NR + wave + 4 * (i - 2 * NR);
      }
      const bool ok = cg < NG;
#pragma unroll
      for (int u = 0; u < 4; ++u) {   // (row step, row half)
        const int row = (u >> 1) * 32 + (u & 1) * 16 + jj;
        v[i * 4 + u] = ok ? *(const u32x4 AS1 *)(base + (int64_t)row * ld + (cg * 4 + q) * 8) : (u32x4){0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < GPW * 4; ++i) acc ^= v[i];
  }
  if (lds_words > 0 && tid == 0) pad[lds_words - 1] = acc[0];
  const unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (r == 0x12345678u) out[blockIdx.x * 256 + tid] = r;   // never true for random data: keeps the loads alive
}

static int g_wrap = 0;
template <int NG>
static void run(int mode, const unsigned short *x, int64_t M, int64_t ld, unsigned *out, int lds_bytes, const char *tag) {
  const int blocks = (int)(M / 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL((probe<0, NG>), dim3(blocks), dim3(256), lds_bytes, 0, x, ld, out, lds_bytes / 4, g_wrap);
    else if (mode == 1) hipLaunchKernelGGL((probe<1, NG>), dim3(blocks), dim3(256), lds_bytes, 0, x, ld, out, lds_bytes / 4, g_wrap);
    else if (mode == 3) hipLaunchKernelGGL((probe<3, NG>), dim3(blocks), dim3(256), lds_bytes, 0, x, ld, out, lds_bytes / 4, g_wrap);
    else hipLaunchKernelGGL((probe<2, NG>), dim3(blocks), dim3(256), lds_bytes, 0, x, ld, out, lds_bytes / 4, g_wrap);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms / 10 < best ? ms / 10 : best;
  }
  const double bytes = (double)M * NG * 64;
  printf("%-44s C=%4d lds=%3d KB  %8.1f us  %6.2f TB/s\n", tag, NG * 32, lds_bytes >> 10, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 16384 * 96;   // ~1 GB at C = 320: far beyond L2 + MALL
  const int C = argc > 2 ? atoi(argv[2]) : 320;
  g_wrap = argc > 3 ? atoi(argv[3]) : 0;
  printf("# M %lld C %d wrap %d (footprint %.1f MB)\n", (long long)M, C, g_wrap, (g_wrap ? g_wrap * 64.0 : (double)M) * C * 2 / 1e6);
  const int64_t ld = C;
  std::vector<unsigned short> h((size_t)1 << 20);
  for (auto &v : h) v = (unsigned short)rand();
  unsigned short *x;
  unsigned *out;
  hipMalloc(&x, (size_t)M * ld * 2);
  hipMalloc(&out, (size_t)(M / 64) * 256 * 4);
  for (size_t off = 0; off < (size_t)M * ld * 2; off += h.size() * 2) {
    size_t n = h.size() * 2;
    if (off + n > (size_t)M * ld * 2) n = (size_t)M * ld * 2 - off;
    hipMemcpy((char *)x + off, h.data(), n, hipMemcpyHostToDevice);
  }
  const char *tags[4] = {"mode 0 fragment-shaped, groups round-robin", "mode 1 fragment-shaped, line mates same wave",
                         "mode 2 full 128-byte lines", "mode 3 16 rows x 64 B, 4 lanes per row"};
  for (int lds_kb : {72, 48, 36}) {   // 2, 3, 4 workgroups per CU by LDS (160 KB)
    for (int mode = 0; mode < 4; ++mode) {
      if (C == 320) run<10>(mode, x, M, ld, out, lds_kb << 10, tags[mode]);
      else if (C == 640) run<20>(mode, x, M, ld, out, lds_kb << 10, tags[mode]);
      else if (C == 2560) run<80>(mode, x, M, ld, out, lds_kb << 10, tags[mode]);
    }
  }
  return 0;
}
