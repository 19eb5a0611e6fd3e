// Practical MFMA ceiling on this chip: a register-operand loop of v_mfma_f32_32x32x16_bf16 (and 16x16x32)
// on random bf16 data, every CU busy, 1 or 2 waves per SIMD.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// DESIGN.md quotes its result (after ~2 s of sustained load: 1.76 / 1.80 PFLOP/s for 32x32x16 at 1 / 2 waves per SIMD,
// 1.51 / 1.89 for 16x16x32, against 2.5 nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b,
                                             float* __restrict__ out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  bf16x8 av[4], bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    av[i] = a[(t * 4 + i) & 0xffff];
    bv[i] = b[(t * 4 + i) & 0xffff];
  }
  if constexpr (SHAPE == 32) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[i], acc[i], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[t] = s;
  } else {
    // separate named accumulators: with an array hipcc (ROCm 7.2) allocated overlapping AGPR ranges and shuffled them
    // with v_accvgpr_* every iteration, which halved the measured rate (the reason round 1 called this shape slower)
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], bv[2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[3], bv[3], c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], bv[2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], bv[3], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[3], bv[0], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], bv[1], c3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[t] = s;
  }
}

template <int SHAPE>
static void run(const bf16x8* a, const bf16x8* b, float* out, int blocks, const char* name) {
  // the chip lowers its clock under sustained MFMA load: warm up with ~2 s of back-to-back launches, then time 20
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 700; ++w) hipLaunchKernelGGL(probe<SHAPE>, dim3(blocks), dim3(256), 0, 0, a, b, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(probe<SHAPE>, dim3(blocks), dim3(256), 0, 0, a, b, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
  const double flops = 20.0 * blocks * 4 /*waves*/ * iters * (SHAPE == 32 ? 4 : 8) /*mfma per iter*/ * flop_per_mfma;
  printf("%s, %d workgroups of 4 waves: %.1f TFLOP/s\n", name, blocks, flops / (ms * 1e-3) / 1e12);
}

int main() {
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));  // bf16 in +-[0.5, 2)
  bf16x8 *a, *b;
  float* out;
  hipMalloc(&a, h.size() * 2);
  hipMalloc(&b, h.size() * 2);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (auto& v : h) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
  hipMemcpy(b, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int blocks : {256, 512}) {
    run<32>(a, b, out, blocks, "v_mfma_f32_32x32x16_bf16");
    run<16>(a, b, out, blocks, "v_mfma_f32_16x16x32_bf16");
  }
  return 0;
}
