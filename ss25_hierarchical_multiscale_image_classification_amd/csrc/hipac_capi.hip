// C-ABI entry points of libhipac_hip.so: weight packing (BN fold + repack), the
// ResNet18 forward driver, and the small elementwise kernels around the trunk.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "e4m3.h"

namespace hipac {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- host-side rounding to the storage type (round-to-nearest-even) ----------------
static inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;  // host compiler: IEEE RNE conversion
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static inline uint16_t to_bits(float f, int precision) {
  return precision == HIPAC_PREC_BF16 ? f32_to_bf16_bits(f) : f32_to_f16_bits(f);
}
static inline bool pair_mode(int precision) { return precision == HIPAC_PREC_FP16X3 || precision == HIPAC_PREC_FP16Q8; }  // (hi, lo) fp16 pairs
static inline int elem_size(int precision) { return precision == HIPAC_PREC_FP32 || pair_mode(precision) ? 4 : 2; }
static inline bool wide_mode(int precision) { return precision == HIPAC_PREC_FP32 || pair_mode(precision); }

static int env_int(const char* name, int dflt, int lo, int hi) {
  if (const char* e = getenv(name)) {
    const int v = atoi(e);
    if (v >= lo && v <= hi) return v;
  }
  return dflt;
}

Plan make_plan(int batch, int precision) {
  Plan p;
  p.esz = elem_size(precision);
  // bc: early sub-batch -- 512 images give layer1/2 thousands of tiles (block-round
  // quantisation < 10 %).  gc: late group -- layer4 has only 49 pixels per image, so it
  // needs thousands of images (default group 4096) to fill 256 CUs x 2 workgroups for several rounds.
  // (tuning knobs; a whole run must use one setting)
  const int bc_cap = env_int("HIPAC_SUBBATCH", 512, 1, 1024);
  int gc_cap = env_int("HIPAC_GROUP", 4096, 1, 8192);
  // fp16q8: halo16x2.h addresses its pair tensors with 32-bit byte offsets (buffer descriptors): layer3's stride-2 entry conv sees
  // 4 x gc x 196 pixels x 128 channels x 4 bytes, which stays below 2^31 up to gc = 5 349
  if ((precision == HIPAC_PREC_FP16Q8 || (precision == HIPAC_PREC_FP16X3 && x3_on_halo16())) && gc_cap > 4096) gc_cap = 4096;
  p.fuse_stem = wide_mode(precision) ? 0 : env_int("HIPAC_FUSE_STEM", 1, 0, 1);
  p.u8_input = 0;
  p.stem_strip = env_int("HIPAC_STEM_STRIP", 1, 0, 1);
  p.l1_fused = wide_mode(precision) ? 0 : env_int("HIPAC_L1_FUSED", 1, 0, 1);
  const bool on_halo16x2 = precision == HIPAC_PREC_FP16Q8 || (precision == HIPAC_PREC_FP16X3 && x3_on_halo16());
  p.pool_head = ((wide_mode(precision) && !on_halo16x2) || !halo_pool_compiled()) ? 0 : env_int("HIPAC_POOL_HEAD", 1, 0, 1);
  if (batch < 1) batch = 1;
  p.bc = batch < bc_cap ? batch : bc_cap;
  p.gc = batch < gc_cap ? batch : gc_cap;
  if (p.gc < p.bc) p.gc = p.bc;
  p.gc = (p.gc + p.bc - 1) / p.bc * p.bc;  // whole sub-batches per group
  const size_t b = (size_t)p.bc, g = (size_t)p.gc;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t e = (size_t)p.esz;
  p.xin = take(b * kPadH * kPadW * 4 * e);
  p.stem = take(b * 112 * 112 * 64 * e);
  p.pool = take(b * 56 * 56 * 64 * e);
  p.tmp_e = take(b * 56 * 56 * 64 * e);
  p.ds_e = take(b * 28 * 28 * 128 * e);
  p.blk[0] = take(b * 56 * 56 * 64 * e);
  p.blk[1] = take(b * 56 * 56 * 64 * e);
  p.blk[2] = take(b * 28 * 28 * 128 * e);
  p.blk[3] = take(g * 28 * 28 * 128 * e);
  p.tmp_l = take(g * 14 * 14 * 256 * e);
  p.ds_l = take(g * 14 * 14 * 256 * e);
  p.blk[4] = take(g * 14 * 14 * 256 * e);
  p.blk[5] = take(g * 14 * 14 * 256 * e);
  p.blk[6] = take(g * 7 * 7 * 512 * e);
  p.blk[7] = take(g * 7 * 7 * 512 * 4);
  p.part = take(((g * 49 + 255) / 256) * 2 * 7 * 2 * 512 * 4);
  p.q8 = 0;
  if (precision == HIPAC_PREC_FP16Q8) {
    const size_t pairs_end = p.blk[7];  // every pair tensor lies below the fp32 map
    p.q8 = take(pairs_end / 2 + 256);
  }
  p.total = off;
  return p;
}

// ---- small kernels ------------------------------------------------------------------

// float32 NCHW [n,3,224,224] -> T NHWC4 zero-padded [n,230,232,4]
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, T* __restrict__ out,
                                                            int n) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * kPadH * kPadW;
  if (gid >= total) return;
  const int px = (int)(gid % kPadW);
  const long long t = gid / kPadW;
  const int py = (int)(t % kPadH);
  const int b = (int)(t / kPadH);
  const int y = py - 3, xx = px - 3;
  typename Elem<T>::vec4 v;
  v[0] = v[1] = v[2] = v[3] = (T)0.f;
  if ((unsigned)y < (unsigned)kPatch && (unsigned)xx < (unsigned)kPatch) {
    const size_t plane = (size_t)kPatch * kPatch;
    const float* src = x + (size_t)b * 3 * plane + (size_t)y * kPatch + xx;
    v[0] = (T)src[0];
    v[1] = (T)src[plane];
    v[2] = (T)src[2 * plane];
  }
  *reinterpret_cast<typename Elem<T>::vec4*>(out + (size_t)gid * 4) = v;
}

int launch_nchw_to_nhwc4(const float* x, void* out, int n, int precision, hipStream_t s) {
  const long long total = (long long)n * kPadH * kPadW;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (precision == HIPAC_PREC_BF16)
    hipLaunchKernelGGL((nchw_to_nhwc4_kernel<__bf16>), dim3(grid), dim3(256), 0, s, x, (__bf16*)out, n);
  else if (precision == HIPAC_PREC_FP16)
    hipLaunchKernelGGL((nchw_to_nhwc4_kernel<_Float16>), dim3(grid), dim3(256), 0, s, x, (_Float16*)out, n);
  else  // fp32 and fp16x3: the stem of both runs on fp32 input
    hipLaunchKernelGGL((nchw_to_nhwc4_kernel<float>), dim3(grid), dim3(256), 0, s, x, (float*)out, n);
  return (int)hipGetLastError();
}

// Global average pool over the 7x7 map of the last block (float32 NHWC
// [n,49,512]) -> feats[n,512]; optional fc -> logits[n,C]; optional argmax.
// One 256-thread workgroup per image, two channels per thread.
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ last, const float* __restrict__ fc_w,
                                                   const float* __restrict__ fc_b, int num_classes,
                                                   float* __restrict__ feats, float* __restrict__ logits,
                                                   long long* __restrict__ labels) {
  __shared__ float red[4][16];
  __shared__ float lg[16];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* src = last + (size_t)b * 49 * 512 + tid * 2;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll 7
  for (int p = 0; p < 49; ++p) {
    const float2 v = *reinterpret_cast<const float2*>(src + (size_t)p * 512);
    s0 += v.x;
    s1 += v.y;
  }
  const float f0 = s0 / 49.0f, f1 = s1 / 49.0f;
  if (feats) *reinterpret_cast<float2*>(feats + (size_t)b * 512 + tid * 2) = make_float2(f0, f1);
  if (num_classes <= 0 || (!logits && !labels)) return;
  for (int j = 0; j < num_classes; ++j) {
    const float2 w = *reinterpret_cast<const float2*>(fc_w + (size_t)j * 512 + tid * 2);
    float v = f0 * w.x + f1 * w.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) red[tid >> 6][j] = v;
  }
  __syncthreads();
  if (tid < num_classes) {
    const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + fc_b[tid];
    lg[tid] = v;
    if (logits) logits[(size_t)b * num_classes + tid] = v;
  }
  __syncthreads();
  if (tid == 0 && labels) {
    int best = 0;
    float bv = lg[0];
    for (int j = 1; j < num_classes; ++j)
      if (lg[j] > bv) {  // strict: first maximum wins, as torch.argmax
        bv = lg[j];
        best = j;
      }
    labels[b] = best;
  }
}

// The same head over the partial sums the last conv's pooled epilogue leaves (halo16.h, POOL): image b = pixels
// [49 b, 49 b + 48] of the flattened 7x7 maps meets at most two 256-pixel tiles mt and both 128-pixel wave halves wm of
// each; part[mt][wm][slot = b - (256 mt) / 49][2][512] = (sum of the pixel values rounded to the grid 2^-10, sum of the
// remainders on the grid 2^-29): both sums are EXACT in fp32 (halo16.h), so the features do not depend on how the image's
// pixels were spread over lanes, waves and tiles -- the same patch gives the same bits at any position of any batch.
// Only the (mt, wm) pairs that overlap the image are read (slots a wave never met are not written).
__global__ __launch_bounds__(256) void head_pool_kernel(const float* __restrict__ part, const float* __restrict__ fc_w,
                                                        const float* __restrict__ fc_b, int num_classes,
                                                        float* __restrict__ feats, float* __restrict__ logits,
                                                        long long* __restrict__ labels) {
  __shared__ float red[4][16];
  __shared__ float lg[16];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int p0 = 49 * b, p1 = p0 + 48;
  float h0 = 0.f, h1 = 0.f, l0 = 0.f, l1 = 0.f;  // exact sums (grid 2^-10 parts, grid 2^-29 remainders): any order gives these bits
  for (int mt = p0 >> 8; mt <= (p1 >> 8); ++mt)
    for (int wm = 0; wm < 2; ++wm) {
      const int w0 = mt * 256 + wm * 128;
      if (w0 + 127 < p0 || w0 > p1) continue;  // this wave half holds no pixel of the image
      const int slot = b - (mt * 256) / 49;
      const float* src = part + (((size_t)(mt * 2 + wm) * 7 + slot) * 2) * 512 + tid * 2;
      const float2 vh = *reinterpret_cast<const float2*>(src), vl = *reinterpret_cast<const float2*>(src + 512);
      h0 += vh.x, h1 += vh.y;
      l0 += vl.x, l1 += vl.y;
    }
  const float f0 = (h0 + l0) / 49.0f, f1 = (h1 + l1) / 49.0f;
  if (feats) *reinterpret_cast<float2*>(feats + (size_t)b * 512 + tid * 2) = make_float2(f0, f1);
  if (num_classes <= 0 || (!logits && !labels)) return;
  for (int j = 0; j < num_classes; ++j) {
    const float2 w = *reinterpret_cast<const float2*>(fc_w + (size_t)j * 512 + tid * 2);
    float v = f0 * w.x + f1 * w.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) red[tid >> 6][j] = v;
  }
  __syncthreads();
  if (tid < num_classes) {
    const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + fc_b[tid];
    lg[tid] = v;
    if (logits) logits[(size_t)b * num_classes + tid] = v;
  }
  __syncthreads();
  if (tid == 0 && labels) {
    int best = 0;
    float bv = lg[0];
    for (int j = 1; j < num_classes; ++j)
      if (lg[j] > bv) {  // strict: first maximum wins, as torch.argmax
        bv = lg[j];
        best = j;
      }
    labels[b] = best;
  }
}

int launch_head_pool(const float* part, int n, const float* fc_w, const float* fc_b, int num_classes, float* feats,
                     float* logits, int64_t* labels, hipStream_t s) {
  hipLaunchKernelGGL(head_pool_kernel, dim3(n), dim3(256), 0, s, part, fc_w, fc_b, num_classes, feats, logits,
                     (long long*)labels);
  return (int)hipGetLastError();
}

int launch_head(const float* last, int n, const float* fc_w, const float* fc_b, int num_classes, float* feats,
                float* logits, int64_t* labels, hipStream_t s) {
  hipLaunchKernelGGL(head_kernel, dim3(n), dim3(256), 0, s, last, fc_w, fc_b, num_classes, feats, logits,
                     (long long*)labels);
  return (int)hipGetLastError();
}

// NHWC (T or float) -> NCHW float32, test tap only.
template <typename T>
__global__ __launch_bounds__(256) void tap_export_kernel(const T* __restrict__ src, float* __restrict__ dst, int n,
                                                         int C, int H, int W) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * C * H * W;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long t = gid / C;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  dst[(((size_t)b * C + c) * H + h) * W + w] = (float)src[gid];
}

// fp16x3: NHWC pairs [pixel][hi: C | lo: C] -> NCHW float32 (hi + lo is exact in fp32)
__global__ __launch_bounds__(256) void tap_export_split_kernel(const _Float16* __restrict__ src, float* __restrict__ dst,
                                                               int n, int C, int H, int W) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * C * H * W;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long t = gid / C;
  const int w = (int)(t % W);
  t /= W;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  const _Float16* px = src + (gid / C) * (2 * C);
  dst[(((size_t)b * C + c) * H + h) * W + w] = (float)px[c] + (float)px[C + c];
}

int launch_tap_export(const void* src, int is_f32, int precision, int n, int C, int H, int W, float* dst,
                      hipStream_t s) {
  const long long total = (long long)n * C * H * W;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (!is_f32 && pair_mode(precision))
    hipLaunchKernelGGL(tap_export_split_kernel, dim3(grid), dim3(256), 0, s, (const _Float16*)src, dst, n, C, H, W);
  else if (is_f32 || precision == HIPAC_PREC_FP32)
    hipLaunchKernelGGL((tap_export_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)src, dst, n, C, H, W);
  else if (precision == HIPAC_PREC_BF16)
    hipLaunchKernelGGL((tap_export_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)src, dst, n, C, H,
                       W);
  else
    hipLaunchKernelGGL((tap_export_kernel<_Float16>), dim3(grid), dim3(256), 0, s, (const _Float16*)src, dst, n, C,
                       H, W);
  return (int)hipGetLastError();
}

// ---- packing ------------------------------------------------------------------------

static int upload(const void* host, size_t bytes, void** dev) {
  HIPAC_CHECK_HIP(hipMalloc(dev, bytes));
  HIPAC_CHECK_HIP(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
  return 0;
}

// Fold BN and repack one conv: src [Cout][Cin][ks][ks] -> dst [Cout][ks][ks][Cin].
static int pack_conv(const hipac_convbn_t& c, int cout, int cin, int ks, float eps, int precision, bool stem,
                     ConvW* out) {
  HIPAC_REQUIRE(c.conv_w && c.bn_gamma && c.bn_beta && c.bn_mean && c.bn_var, HIPAC_EINVAL,
                "pack: null tensor pointer (cout=%d cin=%d ks=%d)", cout, cin, ks);
  const int K = stem ? 7 * 32 : ks * ks * cin;
  const bool f32 = precision == HIPAC_PREC_FP32;
  std::vector<uint16_t> w(f32 ? 0 : (size_t)cout * K, 0);
  std::vector<float> w32(f32 ? (size_t)cout * K : 0, 0.f);
  std::vector<float> bias(cout);
  for (int o = 0; o < cout; ++o) {
    const double scale = (double)c.bn_gamma[o] / sqrt((double)c.bn_var[o] + (double)eps);
    bias[o] = (float)((double)c.bn_beta[o] - (double)c.bn_mean[o] * scale);
    for (int i = 0; i < cin; ++i)
      for (int kh = 0; kh < ks; ++kh)
        for (int kw = 0; kw < ks; ++kw) {
          const float v = (float)((double)c.conv_w[(((size_t)o * cin + i) * ks + kh) * ks + kw] * scale);
          const size_t k = stem ? (size_t)kh * 32 + kw * 4 + i : ((size_t)kh * ks + kw) * cin + i;
          if (f32) w32[(size_t)o * K + k] = v;
          else w[(size_t)o * K + k] = to_bits(v, precision);
        }
  }
  int rc = f32 ? upload(w32.data(), w32.size() * 4, &out->w) : upload(w.data(), w.size() * 2, &out->w);
  if (rc) return rc;
  return upload(bias.data(), bias.size() * 4, (void**)&out->bias);
}

// fp16x3: BN folded as in pack_conv, every weight split into hi = rn16(v), lo = rn16(v - hi); K order per tap =
// 64-channel triples (hi_c | lo_c | hi_c) matching the activation chunks (hi_c, hi_c, lo_c) of the kernels' K loop
// (conv_igemm.h, conv_glds_kernel's SPLIT note).
static int pack_conv_split(const hipac_convbn_t& c, int cout, int cin, int ks, float eps, ConvW* out) {
  HIPAC_REQUIRE(c.conv_w && c.bn_gamma && c.bn_beta && c.bn_mean && c.bn_var, HIPAC_EINVAL,
                "pack: null tensor pointer (cout=%d cin=%d ks=%d)", cout, cin, ks);
  HIPAC_REQUIRE(cin % 64 == 0, HIPAC_EINVAL, "pack: split layout needs cin %% 64 == 0 (%d)", cin);
  const int K = ks * ks * 3 * cin;
  std::vector<uint16_t> w((size_t)cout * K, 0);
  std::vector<float> bias(cout);
  for (int o = 0; o < cout; ++o) {
    const double scale = (double)c.bn_gamma[o] / sqrt((double)c.bn_var[o] + (double)eps);
    bias[o] = (float)((double)c.bn_beta[o] - (double)c.bn_mean[o] * scale);
    for (int i = 0; i < cin; ++i)
      for (int kh = 0; kh < ks; ++kh)
        for (int kw = 0; kw < ks; ++kw) {
          const float v = (float)((double)c.conv_w[(((size_t)o * cin + i) * ks + kh) * ks + kw] * scale);
          const uint16_t hb = f32_to_f16_bits(v);
          _Float16 hh;
          memcpy(&hh, &hb, 2);
          const uint16_t lb = f32_to_f16_bits(v - (float)hh);
          uint16_t* row = &w[(size_t)o * K + ((size_t)kh * ks + kw) * 3 * cin + (size_t)(i / 64) * 192 + (i % 64)];
          row[0] = hb;
          row[64] = lb;
          row[128] = hb;
        }
  }
  int rc = upload(w.data(), w.size() * 2, &out->w);
  if (rc) return rc;
  return upload(bias.data(), bias.size() * 4, (void**)&out->bias);
}

static int pack_conv_split3(const hipac_convbn_t& c, int cout, int cin, float eps, ConvW* out) { return pack_conv_split(c, cout, cin, 3, eps, out); }

// fp16q8, 3x3 / stride 1 convs (halo16x2.h): BN folded, every weight split into the fp16 pair (hi, lo); per output channel and tap,
// per 64-channel chunk 256 bytes: [hi: 64 fp16 | e4m3(hi * 2^4): 64 | e4m3(lo * 2^15): 64]
// (q8 = false: fp16x3 on the same kernel -- the second 128 bytes of a chunk are the 64 lo halves as fp16)
static int pack_conv_q8(const hipac_convbn_t& c, int cout, int cin, float eps, ConvW* out, int taps, bool q8 = true) {
  HIPAC_REQUIRE(c.conv_w && c.bn_gamma && c.bn_beta && c.bn_mean && c.bn_var, HIPAC_EINVAL,
                "pack: null tensor pointer (cout=%d cin=%d q8)", cout, cin);
  HIPAC_REQUIRE(cin % 64 == 0, HIPAC_EINVAL, "pack: q8 layout needs cin %% 64 == 0 (%d)", cin);
  const size_t KROW = (size_t)taps * (cin / 64) * 256;
  std::vector<uint8_t> w((size_t)cout * KROW, 0);
  std::vector<float> bias(cout);
  for (int o = 0; o < cout; ++o) {
    const double scale = (double)c.bn_gamma[o] / sqrt((double)c.bn_var[o] + (double)eps);
    bias[o] = (float)((double)c.bn_beta[o] - (double)c.bn_mean[o] * scale);
    for (int i = 0; i < cin; ++i)
      for (int tap = 0; tap < taps; ++tap) {
        const float v = (float)((double)c.conv_w[((size_t)o * cin + i) * taps + tap] * scale);
        const uint16_t hb = f32_to_f16_bits(v);
        _Float16 hh;
        memcpy(&hh, &hb, 2);
        const uint16_t lb = f32_to_f16_bits(v - (float)hh);
        _Float16 ll;
        memcpy(&ll, &lb, 2);
        uint8_t* row = &w[(size_t)o * KROW + ((size_t)tap * (cin / 64) + i / 64) * 256];
        memcpy(row + (i % 64) * 2, &hb, 2);
        if (q8) {
          row[128 + (i % 64)] = f32_to_e4m3(ldexpf((float)hh, kQ8WhiShift));
          row[192 + (i % 64)] = f32_to_e4m3(ldexpf((float)ll, kQ8WloShift));
        } else {
          memcpy(row + 128 + (i % 64) * 2, &lb, 2);
        }
      }
  }
  int rc = upload(w.data(), w.size(), &out->w);
  if (rc) return rc;
  return upload(bias.data(), bias.size() * 4, (void**)&out->bias);
}

static int pack_conv_q8_3x3(const hipac_convbn_t& c, int cout, int cin, float eps, ConvW* out) { return pack_conv_q8(c, cout, cin, eps, out, 9); }
static int pack_conv_x3rows_3x3(const hipac_convbn_t& c, int cout, int cin, float eps, ConvW* out) { return pack_conv_q8(c, cout, cin, eps, out, 9, false); }

// Stem weights for the strip kernel (uint8 input, conv_igemm.h: stem_pool_strip_kernel): BN folded as in
// pack_conv, ToTensor / Normalize (reference src/main.py:815-816) folded too -- the kernel feeds the centred byte
// value v - 128 (exact in bf16 and fp16; bytes outside the image arrive as 0, i.e. -128), so w'' = w * scale / (255 std_c)
// and the bias takes sum w'' (128 - mu''_c), mu''_c = 255 mean_c (the byte value of the normalised 0 the reference
// pads with), over the taps INSIDE the image and 128 sum w'' over the taps outside: one bias per (row class, column
// class) of the stem pixel, 16 x 64 floats.
// The fold uses the ROUNDED weights, so what is left of the weight rounding multiplies the centred value
// v - mu'', as in the unfolded form.
// K order: k = 16 s + 8 h + j, s = 4 c + rp, kh = 2 rp + (j & 1), kw = 4 h + (j >> 1); kh, kw = 7 are zero.
static float round_to(float v, int precision) {
  const uint16_t b = to_bits(v, precision);
  if (precision == HIPAC_PREC_BF16) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  }
  _Float16 hh;
  memcpy(&hh, &b, 2);
  return (float)hh;
}
// precision HIPAC_PREC_FP16X3: w = the hi halves [64][192] followed by the lo halves [64][192] (fp16 pairs)
static int pack_stem_u8(const hipac_convbn_t& c, float eps, int precision, ConvW* out) {
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
  const bool split = pair_mode(precision);
  std::vector<uint16_t> w((size_t)64 * 192 * (split ? 2 : 1), 0);
  std::vector<float> tab((size_t)16 * 64 + 1);  // + tab[1024]: the factor that undoes the split weights' power-of-two scale
  // fp16x3: the folded weights are ~1e-3 (w / (255 std)), whose lo halves would be fp16 subnormals (2^-24 quantum = only
  // 2^-15 of the weight): everything is scaled by 2^S (exact) so that the largest weight sits near 2^13, and the kernel
  // multiplies the pooled result by 2^-S
  double wscale = 1.0;
  if (split) {
    double wmax = 0.0;
    for (int o = 0; o < 64; ++o) {
      const double scale = (double)c.bn_gamma[o] / sqrt((double)c.bn_var[o] + (double)eps);
      for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < 49; ++k) wmax = fmax(wmax, fabs((double)c.conv_w[((size_t)o * 3 + ch) * 49 + k] * scale / (255.0 * stdv[ch])));
    }
    int S = wmax > 0.0 ? (int)floor(log2(8192.0 / wmax)) : 0;
    S = S < 0 ? 0 : (S > 24 ? 24 : S);
    wscale = ldexp(1.0, S);
  }
  tab[16 * 64] = (float)(1.0 / wscale);
  double mu[3];
  for (int ch = 0; ch < 3; ++ch) mu[ch] = 255.0 * mean[ch];
  // taps of a stem pixel that fall outside the image, by class: 0 none, 1: row / column 0 (taps 0-2), 2: row / column 1
  // (tap 0), 3: row / column 111 (taps 5, 6)
  auto tap_out = [](int cls, int k) { return cls == 1 ? k <= 2 : (cls == 2 ? k == 0 : (cls == 3 ? k >= 5 : false)); };
  for (int o = 0; o < 64; ++o) {
    const double scale = (double)c.bn_gamma[o] / sqrt((double)c.bn_var[o] + (double)eps);
    double rw[3][7][7];  // rounded weights, as the kernel multiplies them
    for (int ch = 0; ch < 3; ++ch)
      for (int kh = 0; kh < 7; ++kh)
        for (int kw = 0; kw < 7; ++kw) {
          const double v = (double)c.conv_w[(((size_t)o * 3 + ch) * 7 + kh) * 7 + kw] * scale / (255.0 * stdv[ch]) * wscale;
          const int s = 4 * ch + (kh >> 1), hq = kw >> 2, j = 2 * (kw & 3) + (kh & 1);
          if (split) {
            const float hi = round_to((float)v, HIPAC_PREC_FP16), lo = round_to((float)v - hi, HIPAC_PREC_FP16);
            w[(size_t)o * 192 + 16 * s + 8 * hq + j] = to_bits(hi, HIPAC_PREC_FP16);
            w[(size_t)(64 + o) * 192 + 16 * s + 8 * hq + j] = to_bits(lo, HIPAC_PREC_FP16);
            rw[ch][kh][kw] = (double)hi + (double)lo;
          } else {
            w[(size_t)o * 192 + 16 * s + 8 * hq + j] = to_bits((float)v, precision);
            rw[ch][kh][kw] = (double)round_to((float)v, precision);
          }
        }
    const double b0 = ((double)c.bn_beta[o] - (double)c.bn_mean[o] * scale) * wscale;
    for (int rc = 0; rc < 4; ++rc)
      for (int cc = 0; cc < 4; ++cc) {
        // the kernel feeds v - 128 inside the image and 0 - 128 outside; the reference's sum is w (v - mu) over the
        // taps inside: bias + sum_inside w (128 - mu) + sum_outside 128 w
        double b = b0;
        for (int ch = 0; ch < 3; ++ch)
          for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 7; ++kw)
              b += (!tap_out(rc, kh) && !tap_out(cc, kw)) ? rw[ch][kh][kw] * (128.0 - mu[ch]) : rw[ch][kh][kw] * 128.0;
        tab[((size_t)rc * 4 + cc) * 64 + o] = (float)b;
      }
  }
  int rc = upload(w.data(), w.size() * 2, &out->w);
  if (rc) return rc;
  return upload(tab.data(), tab.size() * 4, (void**)&out->bias);
}

static void free_convw(ConvW& c) {
  if (c.w) (void)hipFree(c.w);
  if (c.bias) (void)hipFree(c.bias);
  c.w = nullptr;
  c.bias = nullptr;
}

}  // namespace hipac

using namespace hipac;

struct hipac_weights {
  Net net;
  // Second launch lane of hipac_resnet18_forward (created at pack time, never blocking):
  // large batches are split in two halves that run concurrently, one on the caller's stream
  // and one here, so the tail of every launch (partly filled last round of workgroups) is
  // covered by the other lane's kernels.  Fork / join with the caller's stream by events.
  hipStream_t lane_stream = nullptr;       // lane 1
  hipStream_t lane_stream_x[2] = {nullptr, nullptr};  // lanes 2, 3 (HIPAC_LANES = 3 | 4)
  int device = 0;  // the device that was current at pack time: weights, lane stream and kernel attributes live there
};
constexpr int kMaxLanes = 4;

// Split of one forward call into lanes.  Each lane owns a whole workspace plan.
struct Lanes {
  int n;        // 1 .. kMaxLanes
  int chunk;    // images handled by every lane but the last (which takes the rest)
  Plan p;       // per-lane plan (sized for `chunk` images)
  size_t total; // workspace bytes
};
static Lanes make_lanes(int batch, int precision) {
  Lanes L;
  const Plan single = make_plan(batch, precision);
  int want = env_int("HIPAC_LANES", 2, 1, kMaxLanes);
  while (want > 1 && batch < 2 * want * single.bc) --want;  // every lane gets at least two sub-batches
  L.n = want;
  if (L.n == 1) {
    L.chunk = batch, L.p = single, L.total = single.total;
    return L;
  }
  L.chunk = ((batch + L.n - 1) / L.n + single.bc - 1) / single.bc * single.bc;  // whole sub-batches per lane
  while (L.n > 1 && (long long)(L.n - 1) * L.chunk >= batch) --L.n;                  // (rounding up may empty the last lanes)
  L.p = make_plan(L.chunk, precision);
  // run_ops / tap address the workspace with the single-lane plan of `batch`: keep room for it
  L.total = (size_t)L.n * L.p.total > single.total ? (size_t)L.n * L.p.total : single.total;
  return L;
}

extern "C" {

int hipac_abi_version(void) { return HIPAC_ABI_VERSION; }
const char* hipac_last_error(void) { return g_err; }

void hipac_weights_free(hipac_weights_t* w) {
  if (!w) return;
  free_convw(w->net.stem);
  free_convw(w->net.stem_u8);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 2; ++j) free_convw(w->net.block[i][j]);
  for (int i = 0; i < 3; ++i) free_convw(w->net.down[i]);
  if (w->net.fc_w) (void)hipFree(w->net.fc_w);
  if (w->net.fc_b) (void)hipFree(w->net.fc_b);
  if (w->net.zero_page) (void)hipFree(w->net.zero_page);
  if (w->net.lut_t) (void)hipFree(w->net.lut_t);
  if (w->net.lut_f32) (void)hipFree(w->net.lut_f32);
  for (int i = 0; i < 3; ++i)
    if (w->net.bias_c2p[i]) (void)hipFree(w->net.bias_c2p[i]);
  if (w->lane_stream) (void)hipStreamDestroy(w->lane_stream);
  for (int i = 0; i < 2; ++i)
    if (w->lane_stream_x[i]) (void)hipStreamDestroy(w->lane_stream_x[i]);
  delete w;
}

int hipac_resnet18_pack(const hipac_resnet18_params_t* params, int precision, hipac_weights_t** out) {
  HIPAC_REQUIRE(params && out, HIPAC_EINVAL, "pack: null argument");
  HIPAC_REQUIRE(precision == HIPAC_PREC_BF16 || precision == HIPAC_PREC_FP16 || precision == HIPAC_PREC_FP32 ||
                    precision == HIPAC_PREC_FP16X3 || precision == HIPAC_PREC_FP16Q8,
                HIPAC_EINVAL, "pack: unknown precision %d", precision);
  HIPAC_REQUIRE(params->num_classes >= 0 && params->num_classes <= 16, HIPAC_EINVAL,
                "pack: num_classes %d out of range", params->num_classes);
  HIPAC_REQUIRE((params->num_classes == 0) == (params->fc_w == nullptr), HIPAC_EINVAL,
                "pack: fc_w / num_classes mismatch");
  hipac_weights_t* w = new hipac_weights_t();
  memset(&w->net, 0, sizeof(Net));
  HIPAC_CHECK_HIP(hipGetDevice(&w->device));
  w->net.precision = precision;
  w->net.num_classes = params->num_classes;
  const float eps = params->bn_eps;
  const bool split = pair_mode(precision);
  const bool q8 = precision == HIPAC_PREC_FP16Q8;
  // fp16x3 / fp16q8: the stem runs on the exact f32 MFMA (fp32 weights); every other conv on split pairs
  // (fp16q8: the 3x3 / stride 1 convs in halo16x2.h's mixed rows; the entry convs and projections as in fp16x3)
  int rc = pack_conv(params->stem, 64, 3, 7, eps, split ? HIPAC_PREC_FP32 : precision, true, &w->net.stem);
  if (!rc && precision != HIPAC_PREC_FP32) rc = pack_stem_u8(params->stem, eps, precision, &w->net.stem_u8);
  const int ch[4] = {64, 128, 256, 512};
  for (int s = 0; s < 4 && !rc && split; ++s) {
    const int cin = s == 0 ? 64 : ch[s - 1];
    const bool rows = q8 || x3_on_halo16();  // halo16x2.h's weight rows
    auto pack3 = q8 ? pack_conv_q8_3x3 : rows ? pack_conv_x3rows_3x3 : pack_conv_split3;
    rc = pack3(params->block[2 * s][0], ch[s], cin, eps, &w->net.block[2 * s][0]);
    if (!rc) rc = pack3(params->block[2 * s][1], ch[s], ch[s], eps, &w->net.block[2 * s][1]);
    if (!rc) rc = pack3(params->block[2 * s + 1][0], ch[s], ch[s], eps, &w->net.block[2 * s + 1][0]);
    if (!rc) rc = pack3(params->block[2 * s + 1][1], ch[s], ch[s], eps, &w->net.block[2 * s + 1][1]);
    if (!rc && s > 0)
      rc = rows ? pack_conv_q8(params->down[s - 1], ch[s], cin, eps, &w->net.down[s - 1], 1, q8)  // (folded into conv2: halo16x2.h, PCIN)
                : pack_conv_split(params->down[s - 1], ch[s], cin, 1, eps, &w->net.down[s - 1]);
  }
  for (int s = 0; s < 4 && !rc && !split; ++s) {
    const int cin = s == 0 ? 64 : ch[s - 1];
    rc = pack_conv(params->block[2 * s][0], ch[s], cin, 3, eps, precision, false, &w->net.block[2 * s][0]);
    if (!rc) rc = pack_conv(params->block[2 * s][1], ch[s], ch[s], 3, eps, precision, false, &w->net.block[2 * s][1]);
    if (!rc) rc = pack_conv(params->block[2 * s + 1][0], ch[s], ch[s], 3, eps, precision, false, &w->net.block[2 * s + 1][0]);
    if (!rc) rc = pack_conv(params->block[2 * s + 1][1], ch[s], ch[s], 3, eps, precision, false, &w->net.block[2 * s + 1][1]);
    if (!rc && s > 0) rc = pack_conv(params->down[s - 1], ch[s], cin, 1, eps, precision, false, &w->net.down[s - 1]);
  }
  for (int st = 1; st < 4 && !rc && (!wide_mode(precision) || q8 || (split && x3_on_halo16())); ++st) {
    // block0.conv2's bias + the projection's, for the kernel that accumulates both into one accumulator
    const hipac_convbn_t& a = params->block[2 * st][1];
    const hipac_convbn_t& b = params->down[st - 1];
    std::vector<float> bs(ch[st]);
    for (int o = 0; o < ch[st]; ++o) {
      const double sa = (double)a.bn_gamma[o] / sqrt((double)a.bn_var[o] + (double)eps);
      const double sb = (double)b.bn_gamma[o] / sqrt((double)b.bn_var[o] + (double)eps);
      bs[o] = (float)((double)a.bn_beta[o] - (double)a.bn_mean[o] * sa) + (float)((double)b.bn_beta[o] - (double)b.bn_mean[o] * sb);
    }
    rc = upload(bs.data(), bs.size() * 4, (void**)&w->net.bias_c2p[st - 1]);
  }
  w->net.projk = env_int("HIPAC_PROJK", 1, 0, 1);
  if (!rc) {
    const char zeros[256] = {0};
    rc = upload(zeros, sizeof(zeros), (void**)&w->net.zero_page);
  }
  if (!rc) {
    // ToTensor + Normalize table in fp32 with torchvision's op order (v/255, -mean, /std;
    // reference src/main.py:815-816), then rounded to the network's storage type
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    std::vector<uint16_t> lut(3 * 256);
    std::vector<float> lutf(3 * 256);
    for (int c = 0; c < 3; ++c)
      for (int v = 0; v < 256; ++v) {
        const float t = (float)v / 255.0f;
        const float d = t - mean[c];
        lutf[c * 256 + v] = d / stdv[c];
        lut[c * 256 + v] = to_bits(d / stdv[c], wide_mode(precision) ? HIPAC_PREC_BF16 : precision);
      }
    rc = upload(lut.data(), lut.size() * 2, (void**)&w->net.lut_t);
    if (!rc) rc = upload(lutf.data(), lutf.size() * 4, (void**)&w->net.lut_f32);
  }
  if (!rc && params->num_classes > 0) {
    HIPAC_REQUIRE(params->fc_b != nullptr, HIPAC_EINVAL, "pack: fc_b is null");
    rc = upload(params->fc_w, (size_t)params->num_classes * 512 * 4, (void**)&w->net.fc_w);
    if (!rc) rc = upload(params->fc_b, (size_t)params->num_classes * 4, (void**)&w->net.fc_b);
  }
  if (rc) {
    hipac_weights_free(w);
    return rc;
  }
  // optional: without it forward simply runs single-lane
  if (hipStreamCreateWithFlags(&w->lane_stream, hipStreamNonBlocking) != hipSuccess) w->lane_stream = nullptr;
  for (int i = 0; i < 2; ++i)
    if (hipStreamCreateWithFlags(&w->lane_stream_x[i], hipStreamNonBlocking) != hipSuccess) w->lane_stream_x[i] = nullptr;
  *out = w;
  return 0;
}

int hipac_weights_precision(const hipac_weights_t* w) { return w ? w->net.precision : HIPAC_EINVAL; }
int hipac_weights_num_classes(const hipac_weights_t* w) { return w ? w->net.num_classes : HIPAC_EINVAL; }

size_t hipac_resnet18_workspace_bytes(int batch, int precision) {
  if (batch <= 0) return 0;
  return make_lanes(batch, precision).total;
}

int hipac_resnet18_forward(const hipac_weights_t* w, const void* x, int batch, int in_layout, float* feats,
                           float* logits, int64_t* labels, void* workspace, size_t workspace_bytes, void* stream) {
  HIPAC_REQUIRE(w && x && workspace, HIPAC_EINVAL, "forward: null argument");
  HIPAC_REQUIRE(batch > 0, HIPAC_EINVAL, "forward: batch %d", batch);
  HIPAC_REQUIRE(in_layout == HIPAC_IN_NCHW_F32 || in_layout == HIPAC_IN_NHWC4_PAD || in_layout == HIPAC_IN_U8_HWC,
                HIPAC_EINVAL, "forward: unknown in_layout %d", in_layout);
  HIPAC_REQUIRE(!(logits || labels) || w->net.num_classes > 0, HIPAC_EINVAL,
                "forward: logits/labels requested but the weights carry no fc (fc = Identity)");
  HIPAC_REQUIRE(((uintptr_t)workspace & 255) == 0, HIPAC_EINVAL, "forward: workspace must be 256-byte aligned");
  HIPAC_REQUIRE(((uintptr_t)x & 15) == 0, HIPAC_EINVAL, "forward: x must be 16-byte aligned");
  {
    int dev = -1;
    HIPAC_CHECK_HIP(hipGetDevice(&dev));
    HIPAC_REQUIRE(dev == w->device, HIPAC_EINVAL, "forward: handle was packed on device %d, current device is %d",
                  w->device, dev);
  }
  const Lanes L = make_lanes(batch, w->net.precision);
  Plan p = L.p;
  HIPAC_REQUIRE(workspace_bytes >= L.total, HIPAC_EWORKSPACE, "forward: workspace %zu < required %zu",
                workspace_bytes, L.total);
  const bool split = pair_mode(w->net.precision);
  HIPAC_REQUIRE(in_layout != HIPAC_IN_U8_HWC || p.fuse_stem || split, HIPAC_EUNSUPPORTED,
                "forward: uint8 input needs the fused stem (bf16 / fp16 weights, HIPAC_FUSE_STEM not 0) or fp16x3");
  p.u8_input = in_layout == HIPAC_IN_U8_HWC && (!split || p.stem_strip);  // fp16x3 without the strip kernel: converted to fp32 below
  const Net& net = w->net;
  const size_t in_img_bytes = (size_t)kPadH * kPadW * 4 * p.esz;
  auto trunk = net.precision == HIPAC_PREC_BF16 ? run_trunk_bf16
               : net.precision == HIPAC_PREC_FP16 ? run_trunk_f16
               : net.precision == HIPAC_PREC_FP16Q8 ? run_trunk_f16q8
               : split ? run_trunk_f16x3 : run_trunk_f32;
  // images [i0, i0 + n) on stream s with the lane's own workspace
  auto run_lane = [&](char* ws, int i0, int n, hipStream_t s) -> int {
    for (int g0 = i0; g0 < i0 + n; g0 += p.gc) {
      const int gn = i0 + n - g0 < p.gc ? i0 + n - g0 : p.gc;
      for (int b0 = 0; b0 < gn; b0 += p.bc) {
        const int bn = gn - b0 < p.bc ? gn - b0 : p.bc;
        const void* xin = ws + p.xin;
        if (in_layout == HIPAC_IN_NCHW_F32) {
          int rc = launch_nchw_to_nhwc4((const float*)x + (size_t)(g0 + b0) * 3 * kPatch * kPatch, ws + p.xin, bn,
                                        net.precision, s);
          HIPAC_REQUIRE(rc == 0, rc, "forward: input conversion launch failed (%d)", rc);
        } else if (in_layout == HIPAC_IN_U8_HWC && !p.u8_input) {
          int rc = launch_u8_to_nhwc4_f32((const unsigned char*)x + (size_t)(g0 + b0) * kPatch * kPatch * 3, net.lut_f32,
                                          (float*)(ws + p.xin), bn, s);
          HIPAC_REQUIRE(rc == 0, rc, "forward: input conversion launch failed (%d)", rc);
        } else if (in_layout == HIPAC_IN_U8_HWC) {
          xin = (const char*)x + (size_t)(g0 + b0) * kPatch * kPatch * 3;  // raw patches, normalise fused in the stem
        } else {
          xin = (const char*)x + (size_t)(g0 + b0) * in_img_bytes;  // native layout: stem reads the caller's buffer
        }
        int rc = trunk(net, p, ws, xin, bn, b0, 0, s, 0, kNumEarlyOps - 1);
        if (rc) return rc;
      }
      int rc = trunk(net, p, ws, nullptr, 0, 0, gn, s, kNumEarlyOps, kNumOps - 1);
      if (rc) return rc;
      rc = (p.pool_head ? launch_head_pool((const float*)(ws + p.part), gn, net.fc_w, net.fc_b, net.num_classes,
                                           feats ? feats + (size_t)g0 * 512 : nullptr,
                                           logits ? logits + (size_t)g0 * net.num_classes : nullptr, labels ? labels + g0 : nullptr, s)
                        : launch_head((const float*)(ws + p.blk[7]), gn, net.fc_w, net.fc_b, net.num_classes,
                                      feats ? feats + (size_t)g0 * 512 : nullptr,
                                      logits ? logits + (size_t)g0 * net.num_classes : nullptr, labels ? labels + g0 : nullptr, s));
      HIPAC_REQUIRE(rc == 0, rc, "forward: head launch failed (%d)", rc);
    }
    return 0;
  };
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  hipStream_t lane_s[kMaxLanes] = {s, w->lane_stream, w->lane_stream_x[0], w->lane_stream_x[1]};
  int n_lanes = L.n;
  for (int k = 1; k < n_lanes; ++k)
    if (!lane_s[k]) n_lanes = 1;  // a stream could not be created at pack time: single lane
  if (n_lanes == 1) {
    if (L.n == 1) return run_lane(ws, 0, batch, s);
    // the plan `p` is sized for one lane's chunk: walk the chunks one after another on the caller's stream
    for (int i0 = 0; i0 < batch; i0 += L.chunk) {
      int rc1 = run_lane(ws, i0, batch - i0 < L.chunk ? batch - i0 : L.chunk, s);
      if (rc1) return rc1;
    }
    return 0;
  }
  // fork: lanes 1.. (the handle's streams) start after everything already queued on s; join: s waits for all of them
  hipEvent_t fork = nullptr, join[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  HIPAC_CHECK_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  hipError_t e = hipSuccess;
  for (int k = 1; k < n_lanes && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&join[k], hipEventDisableTiming);
  int rc = 0;
  if (e == hipSuccess) e = hipEventRecord(fork, s);
  if (e == hipSuccess) {
    for (int k = n_lanes - 1; k >= 1 && e == hipSuccess; --k) {
      e = hipStreamWaitEvent(lane_s[k], fork, 0);
      if (e != hipSuccess) break;
      const int i0 = k * L.chunk, n = batch - i0 < L.chunk ? batch - i0 : L.chunk;
      if (rc == 0) rc = run_lane(ws + (size_t)k * L.p.total, i0, n, lane_s[k]);
      // join even after a failed launch so the caller's stream stays ordered behind every lane
      e = hipEventRecord(join[k], lane_s[k]);
    }
    if (rc == 0 && e == hipSuccess) rc = run_lane(ws, 0, L.chunk, s);
    for (int k = 1; k < n_lanes; ++k)
      if (join[k] && e == hipSuccess) e = hipStreamWaitEvent(s, join[k], 0);
  }
  (void)hipEventDestroy(fork);  // released by the runtime once the recorded work has completed
  for (int k = 1; k < n_lanes; ++k)
    if (join[k]) (void)hipEventDestroy(join[k]);
  HIPAC_CHECK_HIP(e);
  return rc;
}

int hipac_resnet18_run_ops(const hipac_weights_t* w, const void* x, int in_layout, void* workspace,
                           size_t workspace_bytes, int batch, int first_op, int last_op, void* stream) {
  HIPAC_REQUIRE(w && workspace, HIPAC_EINVAL, "run_ops: null argument");
  Plan p = make_plan(batch, w->net.precision);
  HIPAC_REQUIRE(batch > 0 && batch <= p.gc, HIPAC_EINVAL, "run_ops: batch %d exceeds one group (%d)", batch, p.gc);
  HIPAC_REQUIRE(workspace_bytes >= p.total, HIPAC_EWORKSPACE, "run_ops: workspace %zu < required %zu",
                workspace_bytes, p.total);
  HIPAC_REQUIRE(first_op >= 0 && first_op <= last_op && last_op < kNumOps, HIPAC_EINVAL, "run_ops: range %d..%d",
                first_op, last_op);
  HIPAC_REQUIRE(in_layout == HIPAC_IN_NHWC4_PAD || in_layout == HIPAC_IN_U8_HWC || in_layout == HIPAC_IN_NCHW_F32,
                HIPAC_EINVAL, "run_ops: in_layout %d", in_layout);
  HIPAC_REQUIRE(first_op > 0 || x != nullptr || in_layout == HIPAC_IN_NCHW_F32, HIPAC_EINVAL,
                "run_ops: op 0 needs the input batch");
  char* ws = (char*)workspace;
  const bool split = pair_mode(w->net.precision);
  auto trunk = w->net.precision == HIPAC_PREC_BF16 ? run_trunk_bf16
               : w->net.precision == HIPAC_PREC_FP16 ? run_trunk_f16
               : w->net.precision == HIPAC_PREC_FP16Q8 ? run_trunk_f16q8
               : split ? run_trunk_f16x3 : run_trunk_f32;
  // early ops act on the first sub-batch, late ops on the whole group; an NCHW input was
  // converted into the workspace by the preceding forward
  p.u8_input = in_layout == HIPAC_IN_U8_HWC && (split ? p.stem_strip : p.fuse_stem);
  // (fp16x3 with uint8 input and HIPAC_STEM_STRIP=0: converted into the workspace by the preceding forward, like NCHW)
  const void* xin = in_layout == HIPAC_IN_NCHW_F32 || (split && in_layout == HIPAC_IN_U8_HWC && !p.u8_input)
                        ? (const void*)(ws + p.xin) : x;
  const int ne = batch < p.bc ? batch : p.bc;
  return trunk(w->net, p, ws, xin, ne, 0, batch, (hipStream_t)stream, first_op, last_op);
}

int hipac_resnet18_tap(const hipac_weights_t* w, const void* workspace, int batch, int tap, float* dst,
                       void* stream) {
  HIPAC_REQUIRE(w && workspace && dst, HIPAC_EINVAL, "tap: null argument");
  const Plan p = make_plan(batch, w->net.precision);
  HIPAC_REQUIRE(batch > 0 && batch <= p.bc, HIPAC_EINVAL, "tap: batch %d exceeds one sub-batch (%d)", batch, p.bc);
  HIPAC_REQUIRE(tap >= 0 && tap <= 9, HIPAC_EINVAL, "tap: index %d", tap);
  const char* ws = (const char*)workspace;
  const void* src;
  int C, H, is_f32 = 0;
  if (tap == 0) {
    HIPAC_REQUIRE(!p.fuse_stem, HIPAC_EUNSUPPORTED,
                  "tap 0 (stem) does not exist when the stem is fused with the max-pool; set HIPAC_FUSE_STEM=0");
    src = ws + p.stem, C = 64, H = 112;
    is_f32 = pair_mode(w->net.precision);  // its stem map is fp32
  } else if (tap == 1) {
    src = ws + p.pool, C = 64, H = 56;
  } else {
    const int blk = tap - 2, st = blk / 2;
    const int ch[4] = {64, 128, 256, 512}, hw[4] = {56, 28, 14, 7};
    src = ws + p.blk[blk], C = ch[st], H = hw[st];
    is_f32 = blk == 7;
    if (blk == 7 && p.pool_head) {
      // the forward left the pooled partial sums, not the map: the last conv runs once more with its fp32-map epilogue
      // (same accumulators) on the activations still in the workspace
      Plan q = p;
      q.pool_head = 0;
      auto trunk = w->net.precision == HIPAC_PREC_BF16 ? run_trunk_bf16
                   : w->net.precision == HIPAC_PREC_FP16Q8 ? run_trunk_f16q8
                   : w->net.precision == HIPAC_PREC_FP16X3 ? run_trunk_f16x3 : run_trunk_f16;
      int rc_t = trunk(w->net, q, (char*)workspace, nullptr, 0, 0, batch, (hipStream_t)stream, kNumOps - 1, kNumOps - 1);
      HIPAC_REQUIRE(rc_t == 0, rc_t, "tap: re-running the last conv failed (%d)", rc_t);
    }
  }
  int rc = launch_tap_export(src, is_f32, w->net.precision, batch, C, H, H, dst, (hipStream_t)stream);
  HIPAC_REQUIRE(rc == 0, rc, "tap: launch failed (%d)", rc);
  return 0;
}

}  // extern "C"
