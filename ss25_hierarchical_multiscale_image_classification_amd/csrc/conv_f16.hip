// fp16 instantiation of the ResNet18 trunk.
#include "conv_igemm.h"
namespace hipac {
int run_trunk_f16(const Net& net, const Plan& p, char* ws, const void* xin, int bc, hipStream_t s) {
  return run_trunk<_Float16>(net, p, ws, xin, bc, s);
}
}  // namespace hipac
