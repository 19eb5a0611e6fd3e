// Developer tool (GPU box): what the cheaper parity mode (fp16 hi x hi + e4m3 cross terms) needs to know about
// v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950:
//   1. the A and the B operand use the SAME (lane, byte) -> k map (then any consistent byte order in the LDS rows works);
//   2. the E8M0 scale operands multiply the product exactly (2^-15 here), accumulating into the fp32 C;
//   3. v_cvt_pk_fp8_f32 is OCP e4m3fn with round-to-nearest-even (values printed; compare with torch.float8_e4m3fn);
//   4. sustained rate on random data: 64 x 16x16x32_f16 alone, 32 x 16x16x128_fp8 alone, and the mode's mix (64 + 32).
// hipcc -O3 --offload-arch=gfx950 tools/mfma_f8_probe.hip -o /tmp/f8p && /tmp/f8p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// slot = lane group g (0..3) x byte (0..31) of row / column 0: D[0][0] = 1 iff A's slot and B's slot are the same k
__global__ void map_kernel(unsigned char* same, float* raw) {
  const int lane = threadIdx.x;
  for (int sa = 0; sa < 128; ++sa)
    for (int sb = 0; sb < 128; ++sb) {
      v8i a, b;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = (lane == (sa >> 5) * 16 && e == ((sa & 31) >> 2)) ? (0x38 << (8 * (sa & 3))) : 0;
        b[e] = (lane == (sb >> 5) * 16 && e == ((sb & 31) >> 2)) ? (0x38 << (8 * (sb & 3))) : 0;
      }
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
      if (lane == 0) same[sa * 128 + sb] = c[0] == 1.0f ? 1 : (c[0] == 0.0f ? 0 : 2), raw[sa * 128 + sb] = c[0];
    }
}

__global__ void dump_kernel(float* out, int sa, int sb) {
  const int lane = threadIdx.x;
  v8i a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (lane == (sa >> 5) * 16 && e == ((sa & 31) >> 2)) ? (0x38 << (8 * (sa & 3))) : 0;
    b[e] = (lane == (sb >> 5) * 16 && e == ((sb & 31) >> 2)) ? (0x38 << (8 * (sb & 3))) : 0;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

__global__ void scale_kernel(float* out) {
  const int lane = threadIdx.x;
  v8i a, b;
  for (int e = 0; e < 8; ++e) a[e] = 0x38383838, b[e] = 0x40404040;  // 1.0 x 2.0 over K = 128
  f32x4 c = {3.f, 3.f, 3.f, 3.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 112, 0, 127);  // x 2^-15
  out[lane] = c[0];
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, d, 0, 0, 0, 127, 0, 120);  // x 2^-7 on B's side
  out[64 + lane] = d[0];
}

__global__ void cvt_kernel(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false) & 0xff;
}

template <int MODE>  // 0: f16 only, 1: fp8 only, 2: 64 f16 + 32 fp8
__global__ __launch_bounds__(256, 2) void rate_kernel(const unsigned* __restrict__ seed, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  v8i a8[2], b8[4];
  f16x8 ah[2], bh[4];
  for (int e = 0; e < 8; ++e) {
    for (int k = 0; k < 2; ++k) {
      unsigned r = seed[(tid * 8 + e + 31 * k) & 65535];
      a8[k][e] = r & 0x3f3f3f3f;  // |values| < 2: finite, random mantissas
      ah[k][e] = (_Float16)((float)(r & 1023) * (1.f / 1024.f) - 0.5f);
    }
    for (int k = 0; k < 4; ++k) {
      unsigned r = seed[(tid * 8 + e + 977 * k + 7) & 65535];
      b8[k][e] = r & 0x3f3f3f3f;
      bh[k][e] = (_Float16)((float)(r & 1023) * (1.f / 1024.f) - 0.5f);
    }
  }
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[(i + kk) & 1], acc[i][j], 0, 0, 0);
    }
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8[j], a8[i & 1], acc[i][j], 0, 0, 0, 112, 0, 127);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = s;
}

template <int MODE>
static void rate(const unsigned* seed, float* out, const char* name) {
  const int iters = 4000, grid = 512;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) rate_kernel<MODE><<<grid, 256>>>(seed, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // fp16-equivalent "units": one unit = 32 x 16x16x32 MFMAs' flops (a k32 sub-step); f16 part = 2 units, fp8 part = 4 units of K
    const double f16_flops = MODE != 1 ? 64.0 * 16 * 16 * 32 * 2 : 0.0;
    const double f8_flops = MODE != 0 ? 32.0 * 16 * 16 * 128 * 2 : 0.0;
    const double total = (f16_flops + f8_flops) * iters * 10.0 * grid * 4;
    printf("%-14s rep %d: %.3f ms  %.0f TFLOP/s (f16 part %.0f, fp8 part %.0f)  time per loop body per wave %.2f us\n", name, rep, ms,
           total / ms * 1e-9, f16_flops * iters * 10.0 * grid * 4 / ms * 1e-9, f8_flops * iters * 10.0 * grid * 4 / ms * 1e-9,
           ms * 1e3 / (iters * 10.0));
  }
}

int main() {
  unsigned char* d_same;
  hipMalloc(&d_same, 128 * 128);
  float* d_raw;
  hipMalloc(&d_raw, 128 * 128 * 4);
  map_kernel<<<1, 64>>>(d_same, d_raw);
  std::vector<float> raw(128 * 128);
  hipMemcpy(raw.data(), d_raw, raw.size() * 4, hipMemcpyDeviceToHost);
  for (int a : {0, 1, 4, 15, 16, 17, 31, 32, 33, 64, 127}) {
    printf("sa %3d:", a);
    int shown = 0;
    for (int b = 0; b < 128 && shown < 12; ++b)
      if (raw[a * 128 + b] != 0.f) printf(" [%d]=%g", b, raw[a * 128 + b]), ++shown;
    printf("\n");
  }
  for (int pr = 0; pr < 6; ++pr) {
    const int sas[] = {40, 40, 40, 72, 127, 33}, sbs[] = {40, 8, 72, 72, 127, 1};
    float* d_d;
    hipMalloc(&d_d, 256 * 4);
    dump_kernel<<<1, 64>>>(d_d, sas[pr], sbs[pr]);
    float hd[256];
    hipMemcpy(hd, d_d, sizeof(hd), hipMemcpyDeviceToHost);
    printf("dump sa %d sb %d:", sas[pr], sbs[pr]);
    for (int i = 0; i < 256; ++i) if (hd[i] != 0.f) printf(" lane %d reg %d = %g", i >> 2, i & 3, hd[i]);
    printf("\n");
  }
  std::vector<unsigned char> same(128 * 128);
  hipMemcpy(same.data(), d_same, same.size(), hipMemcpyDeviceToHost);
  int diag = 0, off = 0, odd = 0;
  for (int a = 0; a < 128; ++a)
    for (int b = 0; b < 128; ++b) {
      if (same[a * 128 + b] == 2) ++odd;
      else if (same[a * 128 + b] == 1) (a == b ? diag : off)++;
    }
  printf("A/B slot map: %d of 128 diagonal pairs multiply, %d off-diagonal pairs multiply, %d odd values  (want 128, 0, 0)\n", diag, off, odd);

  float* d_out;
  hipMalloc(&d_out, 128 * 4);
  scale_kernel<<<1, 64>>>(d_out);
  float h[128];
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf("scale: 3 + 128 * 1 * 2 * 2^-15 = %.10f, got %.10f; 256 * 2^-7 = 2, got %.10f\n", 3.0 + 256.0 / 32768.0, h[0], h[64]);

  const float vals[] = {1.0f, 1.0625f, 1.1875f, 0.3f, 447.f, 448.f, 449.f, 480.f, 1000.f, -1000.f, 0.001953125f, 0.0009765625f, 0.00146f, 0.003f, -0.7f, 17.3f, 1e-6f};
  const int nv = sizeof(vals) / 4;
  float* d_in;
  unsigned* d_u;
  hipMalloc(&d_in, sizeof(vals)), hipMalloc(&d_u, nv * 4);
  hipMemcpy(d_in, vals, sizeof(vals), hipMemcpyHostToDevice);
  cvt_kernel<<<1, 64>>>(d_in, d_u, nv);
  unsigned hu[64];
  hipMemcpy(hu, d_u, nv * 4, hipMemcpyDeviceToHost);
  printf("cvt_pk_fp8_f32:");
  for (int i = 0; i < nv; ++i) printf(" %g->0x%02x", vals[i], hu[i]);
  printf("\n");

  unsigned* d_seed;
  hipMalloc(&d_seed, 65536 * 4);
  std::vector<unsigned> seed(65536);
  srand(1);
  for (auto& v : seed) v = (unsigned)rand() * 2654435761u ^ (unsigned)rand();
  hipMemcpy(d_seed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice);
  float* d_big;
  hipMalloc(&d_big, 512 * 256 * 4);
  rate<0>(d_seed, d_big, "f16 16x16x32");
  rate<1>(d_seed, d_big, "fp8 16x16x128");
  rate<2>(d_seed, d_big, "mix 64 + 32");
  return 0;
}
