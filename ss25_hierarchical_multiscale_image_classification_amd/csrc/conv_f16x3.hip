// fp16x3 instantiation of the ResNet18 trunk: (hi, lo) fp16 pairs, three MFMA products per term -- on halo16x2.h's X3 form
// (round 4; HIPAC_X3_HALO16=0: round 3's SPLIT forms of the 32x32x16 kernels, conv_igemm.h).  The tighter of the two precision
// modes that meet the reference's fp32 results to 1e-3.
#include "conv_igemm.h"
namespace hipac {
int run_trunk_f16x3(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                    hipStream_t s, int first, int last) {
  return run_trunk<_Float16, true, HIPAC_X3_HALO16 ? 2 : 0>(net, p, ws, xin, n_early, img_off, n_late, s, first, last);
}

bool x3_on_halo16() { return HIPAC_X3_HALO16 != 0; }

int launch_u8_to_nhwc4_f32(const unsigned char* x, const float* lut, float* out, int n, hipStream_t s) {
  const long long total = (long long)n * kPadH * kPadW;
  hipLaunchKernelGGL((u8_to_nhwc4_f32_kernel<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, lut, out, n);
  return (int)hipGetLastError();
}
}  // namespace hipac
