// fp16x3 instantiation of the ResNet18 trunk: (hi, lo) fp16 pairs, three MFMA products per term (conv_igemm.h,
// conv_glds_kernel's SPLIT note).  The precision mode that meets the reference's fp32 results to 1e-3.
#include "conv_igemm.h"
namespace hipac {
int run_trunk_f16x3(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                    hipStream_t s, int first, int last) {
  return run_trunk<_Float16, true>(net, p, ws, xin, n_early, img_off, n_late, s, first, last);
}

int launch_u8_to_nhwc4_f32(const unsigned char* x, const float* lut, float* out, int n, hipStream_t s) {
  const long long total = (long long)n * kPadH * kPadW;
  hipLaunchKernelGGL((u8_to_nhwc4_f32_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, lut, out, n);
  return (int)hipGetLastError();
}
}  // namespace hipac
