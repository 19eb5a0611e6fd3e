// fp32 parity-mode instantiation of the ResNet18 trunk (exact f32 MFMA; every conv on the v1 kernel).
#include "conv_igemm.h"
namespace hipac {
int run_trunk_f32(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                  hipStream_t s, int first, int last) {
  return run_trunk<float>(net, p, ws, xin, n_early, img_off, n_late, s, first, last);
}
}  // namespace hipac
