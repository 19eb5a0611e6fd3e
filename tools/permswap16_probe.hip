// Developer tool (GPU box): lane semantics of v_permlane16_swap_b32 on gfx950 (the direct epilogue of halo16.h pairs
// 16-lane rows with it).  hipcc -O3 --offload-arch=gfx950 tools/permswap16_probe.hip -o /tmp/p16 && /tmp/p16
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = 100 + lane, b = 200 + lane;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a': "); for (int r = 0; r < 4; ++r) printf("row%d %u..%u  ", r, h[16 * r], h[16 * r + 15]); printf("\n");
  printf("b': "); for (int r = 0; r < 4; ++r) printf("row%d %u..%u  ", r, h[64 + 16 * r], h[64 + 16 * r + 15]); printf("\n");
  printf("expected if a'.row1 <- b.row0, a'.row3 <- b.row2, b'.row0 <- a.row1, b'.row2 <- a.row3:\n");
  printf("a': row0 100..115  row1 200..215  row2 132..147  row3 232..247\n");
  printf("b': row0 116..131  row1 216..231  row2 148..163  row3 248..263\n");
}
