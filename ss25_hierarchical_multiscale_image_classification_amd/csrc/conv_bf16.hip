// bf16 instantiation of the ResNet18 trunk (separate TU so the two precisions compile in parallel).
#include "conv_igemm.h"
namespace hipac {
int run_trunk_bf16(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                   hipStream_t s, int first, int last) {
  return run_trunk<__bf16>(net, p, ws, xin, n_early, img_off, n_late, s, first, last);
}
bool halo_pool_compiled() { return halo_pool_available<__bf16, false>(); }
}  // namespace hipac

#ifdef HIPAC_HALO_STAMPS
extern "C" int hipac_debug_halo_stamps(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hipac::g_halo_stamps), 64) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hipac::g_halo_stamps), z, 64) != hipSuccess) return 1;
  }
  return 0;
}
extern "C" int hipac_debug_blk_stamps(unsigned long long* out4, int reset) {
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(hipac::g_blk_stamps), 32) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[4] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hipac::g_blk_stamps), z, 32) != hipSuccess) return 1;
  }
  return 0;
}
#endif
