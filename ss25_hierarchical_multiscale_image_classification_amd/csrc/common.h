// Shared declarations for libhipac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/hipac.h"

namespace hipac {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Element traits: the MFMA operand types for the two supported precisions.
template <typename T> struct Elem;
template <> struct Elem<__bf16> {
  using frag = bf16x8;
  using vec4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<_Float16> {
  using frag = f16x8;
  using vec4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// Fixed ResNet18@224 geometry.
constexpr int kPatch = HIPAC_PATCH;
constexpr int kPadH = HIPAC_PAD_H;
constexpr int kPadW = HIPAC_PAD_W;

void set_error(const char* fmt, ...);

#define HIPAC_CHECK_HIP(expr)                                                     \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      ::hipac::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                         __FILE__, __LINE__);                                     \
      return (int)e__;                                                            \
    }                                                                             \
  } while (0)

#define HIPAC_REQUIRE(cond, code, ...)   \
  do {                                   \
    if (!(cond)) {                       \
      ::hipac::set_error(__VA_ARGS__);   \
      return (code);                     \
    }                                    \
  } while (0)

// ---- packed network description shared by the TUs ---------------------------------
struct ConvW {
  void* w;      // device, T[Cout][K] (K = kh*kw*Cin, or 7*32 for the stem)
  float* bias;  // device, float[Cout]
};

struct Net {
  int precision;
  int num_classes;
  ConvW stem;
  ConvW block[8][2];
  ConvW down[3];
  float* fc_w;  // device float[num_classes][512]
  float* fc_b;
};

// Workspace plan for one sub-batch (element type size = 2 bytes).
struct Plan {
  int bc;            // images per sub-batch
  size_t xin;        // T[bc,230,232,4]
  size_t stem;       // T[bc,112,112,64]
  size_t pool;       // T[bc,56,56,64]
  size_t tmp;        // T[bc,56,56,64]   conv1 output of the current block
  size_t ds;         // T[bc,28,28,128]  projection shortcut of the current block
  size_t blk[8];     // block outputs (blk[7] is float32)
  size_t total;
};
Plan make_plan(int batch);

// per-precision launchers (conv_bf16.hip / conv_f16.hip)
int run_trunk_bf16(const Net& net, const Plan& p, char* ws, const void* xin, int bc, hipStream_t s, int first,
                   int last);
int run_trunk_f16(const Net& net, const Plan& p, char* ws, const void* xin, int bc, hipStream_t s, int first,
                  int last);
constexpr int kNumOps = 21;

// elementwise.hip
int launch_nchw_to_nhwc4(const float* x, void* out, int n, int precision, hipStream_t s);
int launch_head(const float* last, int n, const float* fc_w, const float* fc_b, int num_classes,
                float* feats, float* logits, int64_t* labels, hipStream_t s);
int launch_tap_export(const void* src, int is_f32, int precision, int n, int C, int H, int W, float* dst,
                      hipStream_t s);

}  // namespace hipac
