// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs, and how exact is the three-product
// split  a*b ~= ah*bh + ah*bl + al*bh  (ah = rn16(a), al = rn16(a - ah)) against a double dot product?
// DESIGN.md section 4 quotes the result.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A[32][16] row-major fp32, B[32][16] row-major fp32 (B^T of the GEMM): D[i][j] = sum_k A[i][k] * B[j][k]
__global__ void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D3,
                      float* __restrict__ D1, float scale_a, float scale_b) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f16x8 ah, al, bh, bl;
  for (int e = 0; e < 8; ++e) {
    const float a = A[r * 16 + 8 * h + e] * scale_a, b = B[r * 16 + 8 * h + e] * scale_b;
    ah[e] = (_Float16)a;
    al[e] = (_Float16)(a - (float)ah[e]);
    bh[e] = (_Float16)b;
    bl[e] = (_Float16)(b - (float)bh[e]);
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  f32x16 c1 = {};
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
  // accumulator layout: element 4q + e of lane (r, h) = D[row 8q + 4h + e][col r]
  for (int q = 0; q < 4; ++q)
    for (int e = 0; e < 4; ++e) {
      D3[(8 * q + 4 * h + e) * 32 + r] = c[4 * q + e];
      D1[(8 * q + 4 * h + e) * 32 + r] = c1[4 * q + e];
    }
}

int main() {
  std::vector<float> A(32 * 16), B(32 * 16), D3(32 * 32), D1(32 * 32);
  srand(3);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *dA, *dB, *dD3, *dD1;
  hipMalloc(&dA, A.size() * 4);
  hipMalloc(&dB, B.size() * 4);
  hipMalloc(&dD3, D3.size() * 4);
  hipMalloc(&dD1, D1.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  // scale_a shrinks the weights: at 2^-6 the low halves are all fp16 subnormals (< 2^-14)
  for (float sa : {1.f, 1.f / 64, 1.f / 1024})
    for (float sb : {1.f, 1.f / 64}) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD3, dD1, sa, sb);
      hipMemcpy(D3.data(), dD3, D3.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(D1.data(), dD1, D1.size() * 4, hipMemcpyDeviceToHost);
      double e3 = 0, e1 = 0, ref_max = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int k = 0; k < 16; ++k) ref += (double)(A[i * 16 + k] * sa) * (double)(B[j * 16 + k] * sb);
          e3 = fmax(e3, fabs(D3[i * 32 + j] - ref));
          e1 = fmax(e1, fabs(D1[i * 32 + j] - ref));
          ref_max = fmax(ref_max, fabs(ref));
        }
      printf("scale_a %.6f scale_b %.6f: max|ref| %.3e  one product err %.3e (rel %.2e)  three products err %.3e (rel %.2e)\n", sa,
             sb, ref_max, e1, e1 / ref_max, e3, e3 / ref_max);
    }
  return 0;
}
