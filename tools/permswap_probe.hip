#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a': lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("b': lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
}
