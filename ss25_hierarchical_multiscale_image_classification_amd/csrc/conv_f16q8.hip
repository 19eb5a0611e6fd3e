// fp16q8 instantiation of the ResNet18 trunk: the fp16x3 pair layout with the two cross products of every term on the
// e4m3 MX MFMA (halo16x2.h).  The faster of the two precision modes that meet the reference's fp32 results to 1e-3.
#include "conv_igemm.h"
namespace hipac {
int run_trunk_f16q8(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                    hipStream_t s, int first, int last) {
  return run_trunk<_Float16, true, 1>(net, p, ws, xin, n_early, img_off, n_late, s, first, last);
}
}  // namespace hipac
