// float -> OCP e4m3fn byte on the host (weight packing of precision fp16q8): round to nearest even, saturating at +-448
// (0x7e), subnormals of 2^-9; NaN -> 0x7f.  The device side is v_cvt_pk_fp8_f32 (halo16x2.h, cvt4_e4m3) -- the same values,
// tools/mfma_f8_probe.hip; tests/test_e4m3.py compares this function with torch.float8_e4m3fn.
#pragma once
#include <cmath>
#include <cstdint>

namespace hipac {

inline uint8_t f32_to_e4m3(float v) {
  if (v != v) return 0x7f;
  const uint8_t sign = std::signbit(v) ? 0x80 : 0;
  float a = std::fabs(v);
  if (a >= 448.f) return sign | 0x7e;
  if (a < 0.015625f) {  // below 2^-6: multiples of 2^-9 (nearbyint: ties to even in the default rounding mode)
    return sign | (uint8_t)std::nearbyint(a * 512.f);  // 8 -> 0x08 = 2^-6, the smallest normal
  }
  int e;
  (void)std::frexp(a, &e);          // a = m 2^e, m in [0.5, 1): a = (1 + f / 8) 2^(e - 1)
  int E = e - 1;
  int q = (int)std::nearbyint(std::ldexp(a, 3 - E));  // 8 .. 16
  if (q == 16) q = 8, E += 1;
  const int bits = ((E + 7) << 3) | (q - 8);
  return sign | (uint8_t)(bits > 0x7e ? 0x7e : bits);
}

}  // namespace hipac
