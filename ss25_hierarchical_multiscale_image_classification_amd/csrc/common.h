// Shared declarations for libhipac_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <type_traits>

#include "../../include/hipac.h"

namespace hipac {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#ifndef HIPAC_NT_STORES
#define HIPAC_NT_STORES 0  // 1: the conv epilogues of the big early maps (layer1's fused block, layer2's convs) store their activations
                           // non-temporally (streamed past the L2's LRU).  Measured (bf16): the ops timed ALONE get faster (layer2 160 / 218 /
                           // 193 / 227 -> 146 / 205 / 187 / 202 ns per patch; layers 3-4 2-5 % slower, the fp16q8 kernels +-0), but the whole
                           // forward gets 0.5 % SLOWER (318.9 k vs 320.6 k patches/s, three alternating runs on one box): inside the
                           // pipeline the next kernel finds part of a 100-200 MB map still in the L2 / infinity cache, which a per-op
                           // loop that never reads its output cannot show.  Left off.
#endif
// 16-byte activation store of a conv epilogue
#ifndef HIPAC_WT_STORES
#define HIPAC_WT_STORES 0  // developer experiment: 1 = `sc1` (write-through: the line goes to memory at once and stays valid in the L2), 2 = `sc0 sc1`, 3 = `nt sc1`.
                           // Whole forward, alternating runs on one box: 312.9 k plain, 312.1 / 312.0 / 313.0 k -- no policy moves it
#endif
template <bool NT>
__device__ __forceinline__ void store16_out(void* p, u32x4 v) {
#if HIPAC_WT_STORES == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif HIPAC_WT_STORES == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif HIPAC_WT_STORES == 3
  asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(p), "v"(v) : "memory");
#else
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  else *reinterpret_cast<u32x4*>(p) = v;
#endif
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;

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

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// fp32 parity mode: exact f32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate).  A fragment is
// the same "8 consecutive k of one row" as for bf16; MFMA t of a k16 step consumes element t of both
// operands, i.e. the k pair {t, 8 + t} (lane halves), so the 8 instructions cover all 16 k.
typedef __attribute__((ext_vector_type(8))) float f32x8;
template <> struct Elem<float> {
  using frag = f32x8;
  using vec4 = f32x4;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
#pragma unroll
    for (int t = 0; t < 8; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], c, 0, 0, 0);
    return c;
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

// precision fp16q8 (halo16x2.h): the constant power-of-two scales of the e4m3 byte planes.  Activations: hi8 = e4m3(hi), lo8 = e4m3(lo * 2^11);
// weights (pack_conv_q8): whi8 = e4m3(whi * 2^4), wlo8 = e4m3(wlo * 2^15).  Both cross products carry 2^(11 + 4) = 2^(0 + 15).
constexpr int kQ8LoShift = 11, kQ8WhiShift = 4, kQ8WloShift = 15;
static_assert(kQ8LoShift + kQ8WhiShift == kQ8WloShift, "one scale for both cross products");

// ---- packed network description shared by the TUs ---------------------------------
struct ConvW {
  void* w;      // device, T[Cout][K] (K = kh*kw*Cin, or 7*32 for the stem)
  float* bias;  // device, float[Cout]
};

struct Net {
  int precision;
  int num_classes;
  ConvW stem;
  ConvW stem_u8;      // strip kernel (uint8 input): T[64][192] in its own K order, normalisation folded in;
                      // its "bias" is the table float[4 row classes][4 column classes][64]: bias + border correction
  ConvW block[8][2];
  ConvW down[3];
  float* fc_w;  // device float[num_classes][512]
  float* fc_b;
  char* zero_page;  // device, 256 zero bytes: DMA source for out-of-image conv taps
  unsigned short* lut_t;  // device, T[3][256]: (v/255 - mean)/std in fp32, rounded to T (uint8 input path)
  float* lut_f32;         // device, float[3][256]: the same table unrounded (fp16x3 mode, uint8 input)
  float* bias_c2p[3];     // device, float[Cout]: bias of block0.conv2 + bias of the projection, stages 2-4 (folded projection)
  int projk;              // 1: layers 3, 4 fold the projection shortcut into the block's second conv (HIPAC_PROJK, default 1)
};

// Workspace plan.  The trunk runs in two phases so every launch fills the chip:
//   early (stem, pool, layer1, layer2: big activations, many tiles) in sub-batches of `bc`
//   late  (layer3, layer4, head: small maps)                        in groups of `gc` >= bc
// Offsets are bytes into the caller's workspace; T = 2-byte element.
struct Plan {
  int bc, gc;
  int esz;        // bytes per activation element: 2 (bf16 / fp16) or 4 (fp32 parity mode; fp16x3: one (hi, lo) pair)
  int u8_input;   // 1: the stem reads raw uint8 HWC patches (normalise fused); needs fuse_stem
  int fuse_stem;  // 1: stem conv + max-pool in one kernel (default); 0: separate kernels (keeps the stem tap)
  int stem_strip; // uint8 input: 1 = strip kernel (default), 0 = tile kernel with the LDS table (first form)
  int l1_fused;   // 1 = a layer1 BasicBlock is one kernel (default), 0 = conv1 and conv2 as separate launches
  int pool_head;  // 1 = the last conv's epilogue reduces the 7x7 map per image (partial sums in `part`, head_pool_kernel
                  // finishes), 0 = it stores the fp32 map blk[7] and head_kernel reads it (HIPAC_POOL_HEAD=0; fp32 / fp16x3)
  size_t part;    // float[ceil(gc * 49 / 256)][2][kPoolSlots = 7][2][512] partial sums of the global average pool
  size_t q8;      // precision fp16q8: the e4m3 tensor [pixel][C / 64][lo8: 64 | hi8: 64] of the pair tensor at offset o is at q8 + o / 2
  // early, sized for bc images
  size_t xin;     // T[bc,230,232,4]
  size_t stem;    // T[bc,112,112,64]
  size_t pool;    // T[bc,56,56,64]
  size_t tmp_e;   // T[bc,56,56,64]   conv1 output of the current early block
  size_t ds_e;    // T[bc,28,28,128]  projection shortcut (layer2)
  // late, sized for gc images
  size_t tmp_l;   // T[gc,14,14,256]
  size_t ds_l;    // T[gc,14,14,256]
  // block outputs: blk[0..2] early (bc), blk[3..7] late (gc); blk[7] is float32
  size_t blk[8];
  size_t total;
};
Plan make_plan(int batch, int precision = 0);
constexpr int kNumOps = 21;
constexpr int kNumEarlyOps = 11;  // stem, pool, layer1 (4), layer2 (5)

// per-precision launchers (conv_bf16.hip / conv_f16.hip): ops first..last of the trunk.
// Early ops run on `n_early` images (one sub-batch, whose layer2 output lands at image
// offset `img_off` of the group buffer); late ops on `n_late` images (one group).
int run_trunk_bf16(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                   hipStream_t s, int first, int last);
int run_trunk_f16(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                  hipStream_t s, int first, int last);
int run_trunk_f32(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                  hipStream_t s, int first, int last);
int run_trunk_f16x3(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                    hipStream_t s, int first, int last);
int run_trunk_f16q8(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                    hipStream_t s, int first, int last);
int launch_u8_to_nhwc4_f32(const unsigned char* x, const float* lut, float* out, int n, hipStream_t s);

// train.hip: strided fp32 GEMM on the f32 MFMA, C[m][n] = sum_k a[m*sam + k*sak] * b[n*sbn + k*sbk]
int launch_gemm_f32(const float* a, long long sam, long long sak, const float* b, long long sbn, long long sbk, float* c, long long ldc,
                    int M, int N, int K, hipStream_t s);

// elementwise.hip
int launch_nchw_to_nhwc4(const float* x, void* out, int n, int precision, hipStream_t s);
int launch_head(const float* last, int n, const float* fc_w, const float* fc_b, int num_classes,
                float* feats, float* logits, int64_t* labels, hipStream_t s);
int launch_head_pool(const float* part, int n, const float* fc_w, const float* fc_b, int num_classes,
                     float* feats, float* logits, int64_t* labels, hipStream_t s);
bool x3_on_halo16();  // conv_f16x3.hip: does fp16x3 run on halo16x2.h (weight rows [whi | wlo] per chunk) or on the round-3 SPLIT kernels?
bool halo_pool_compiled();  // conv_bf16.hip: was the 16x16x32 halo kernel with the direct epilogue compiled in?
int launch_tap_export(const void* src, int is_f32, int precision, int n, int C, int H, int W, float* dst,
                      hipStream_t s);

}  // namespace hipac
