// Training-view augmentation on the device.  SimCLR (SURVEY a-12: src/models/simclr.py:57-66, get_simclr_transform):
//     RandomResizedCrop(224) -> RandomHorizontalFlip -> RandomApply([ColorJitter(.4, .4, .4, .1)], p = .8)
//     -> RandomGrayscale(p = .2) -> ToTensor -> Normalize
// and the classifier loops' transform of tumour patches (a-13: src/main.py:417-425; 224-pixel patches): flips, RandomRotation(90),
// ColorJitter(.2, .2, .2, .1), Resize (the identity there), ToTensor, Normalize --
// on patches that stay resident in HBM as uint8 [N][P][P][3] (a pool of decoded PNGs: 150 KB per 224-pixel patch, i.e.
// 1.9 M patches in 288 GB).  The reference runs these transforms per sample on DataLoader workers through Pillow
// (torchvision's PIL backend); at the native step's rate (14 k view pairs/s in fp16) that host path is 10x too slow, so the
// whole per-step input pipeline is two launches here.  The RANDOM DRAWS stay on the host (augment.py mirrors torchvision's
// distributions and draw order); the kernels are the deterministic image arithmetic, restated from Pillow's C and checked
// against Pillow itself bit for bit (tests/test_gpu_augment.py; the formulas were first checked on the CPU over the whole
// 2^24 colour cube):
//   * crop + resize: ImagingResample's two 8-bit passes (22-bit coefficients from hipac_resample_coeffs, tables for every
//     source size 1..P resident on the device), horizontal into a uint8 intermediate, then vertical; the flip is the
//     store address;
//   * brightness / contrast / saturation = ImagingBlend(degenerate, image, factor) in float: (int)a + f * ((int)b - (int)a),
//     one multiply and one add in fp32 (no fma), truncation for 0 <= f <= 1, clipping otherwise; degenerate = black / the
//     rounded mean of the L image / the L image; L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//   * hue: rgb2hsv / hsv2rgb of libImaging/Convert.c with their float / double mix, H shifted with uint8 wrap;
//   * ToTensor / Normalize through the float[3][256] table of (v / 255 - mean) / std evaluated in fp32.
// One workgroup per view keeps the 224 x 224 x 3 image in LDS (147 KB) through the colour operations (the contrast
// step needs a whole-image mean between two of them).  Bound: HBM (0.6 MB written per view); 2 x 1024 views take ~0.2 ms.
#include "common.h"

// Pillow's C is compiled without fused multiply-add: every product below is rounded before it is added
#pragma clang fp contract(off)

namespace hipac {

constexpr int kAugParams = 24;  // int32 per view, see include/hipac.h
constexpr int kOut = 224;

__device__ __forceinline__ int clip8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// horizontal pass of the crop: tmp[view][y][x][c], y < h, x < 224
__global__ __launch_bounds__(256) void aug_hpass_kernel(const unsigned char* __restrict__ pool, int P,
                                                        const int* __restrict__ params, const int* __restrict__ tb,
                                                        const int* __restrict__ tk, int KS, unsigned char* __restrict__ tmp) {
  const int v = blockIdx.y;
  const int* pr = params + v * kAugParams;
  const int src = pr[0], top = pr[1], left = pr[2], h = pr[3], w = pr[4];
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int y = t / kOut, x = t - y * kOut;
  if (y >= h) return;
  const int* b = tb + ((size_t)(w - 1) * kOut + x) * 2;
  const int* k = tk + ((size_t)(w - 1) * kOut + x) * KS;
  const int xmin = b[0], cnt = b[1];
  const unsigned char* row = pool + (((size_t)src * P + top + y) * P + left + xmin) * 3;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int i = 0; i < cnt; ++i) {
    const int c = k[i];
    a0 += row[3 * i + 0] * c, a1 += row[3 * i + 1] * c, a2 += row[3 * i + 2] * c;
  }
  unsigned char* o = tmp + (((size_t)v * P + y) * kOut + x) * 3;
  o[0] = (unsigned char)clip8i(a0 >> 22), o[1] = (unsigned char)clip8i(a1 >> 22), o[2] = (unsigned char)clip8i(a2 >> 22);
}

// vertical pass + horizontal flip: out[view][oy][ox'][c]
__global__ __launch_bounds__(256) void aug_vpass_kernel(const unsigned char* __restrict__ tmp, int P, const int* __restrict__ params,
                                                        const int* __restrict__ tb, const int* __restrict__ tk, int KS,
                                                        unsigned char* __restrict__ out) {
  const int v = blockIdx.y;
  const int* pr = params + v * kAugParams;
  const int h = pr[3], flip = pr[5];
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= kOut * kOut) return;
  const int oy = t / kOut, ox = t - oy * kOut;
  const int* b = tb + ((size_t)(h - 1) * kOut + oy) * 2;
  const int* k = tk + ((size_t)(h - 1) * kOut + oy) * KS;
  const int ymin = b[0], cnt = b[1];
  const unsigned char* col = tmp + (((size_t)v * P + ymin) * kOut + ox) * 3;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  for (int i = 0; i < cnt; ++i) {
    const int c = k[i];
    const unsigned char* p = col + (size_t)i * kOut * 3;
    a0 += p[0] * c, a1 += p[1] * c, a2 += p[2] * c;
  }
  unsigned char* o = out + (((size_t)v * kOut + oy) * kOut + (flip ? kOut - 1 - ox : ox)) * 3;
  o[0] = (unsigned char)clip8i(a0 >> 22), o[1] = (unsigned char)clip8i(a1 >> 22), o[2] = (unsigned char)clip8i(a2 >> 22);
}

// geometry 1 (the classifier loops' train_transform on 224-pixel patches, src/main.py:417-425): RandomHorizontalFlip,
// RandomVerticalFlip, then RandomRotation's Image.rotate(angle, NEAREST, fillcolor = 0) = libImaging/Geometry.c affine_fixed:
// 16.16 fixed-point source coordinates xx = a2 + x a0 + y a1, yy = a5 + x a3 + y a4 (the six integers are computed on the host
// exactly as Image.rotate / affine_fixed compute them), pixel (yy >> 16, xx >> 16) of the flipped image or black outside
__global__ __launch_bounds__(256) void aug_affine_kernel(const unsigned char* __restrict__ pool, const int* __restrict__ params,
                                                         unsigned char* __restrict__ out) {
  const int v = blockIdx.y;
  const int* pr = params + v * kAugParams;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= kOut * kOut) return;
  const int y = t / kOut, x = t - y * kOut;
  const int xin = (pr[19] + x * pr[17] + y * pr[18]) >> 16, yin = (pr[22] + x * pr[20] + y * pr[21]) >> 16;
  unsigned char* o = out + ((size_t)v * kOut * kOut + t) * 3;
  if (xin >= 0 && xin < kOut && yin >= 0 && yin < kOut) {
    const int xs = pr[5] ? kOut - 1 - xin : xin, ys = pr[16] ? kOut - 1 - yin : yin;
    const unsigned char* src = pool + (((size_t)pr[0] * kOut + ys) * kOut + xs) * 3;
    o[0] = src[0], o[1] = src[1], o[2] = src[2];
  } else {
    o[0] = o[1] = o[2] = 0;
  }
}

// ImagingBlend(in1, in2, alpha) for one byte (libImaging/Blend.c): float arithmetic, one multiply, one add
__device__ __forceinline__ int blend8(int in1, int in2, float alpha, int interp) {
  const float t = __fadd_rn((float)in1, __fmul_rn(alpha, (float)(in2 - in1)));
  if (interp) return (int)t & 255;  // 0 <= alpha <= 1: (UINT8) of a value inside [0, 255]
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
__device__ __forceinline__ int luma8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// libImaging/Convert.c rgb2hsv_row (follows colorsys.py; float with double constants)
__device__ __forceinline__ void rgb2hsv8(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) {
    uh = 0, us = 0;
    return;
  }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float hf;
  if (r == maxc) hf = __fsub_rn(bc, gc);
  else if (g == maxc) hf = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
  else hf = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
  // fmod(x, 1.0) for x = h / 6 + 1 in [1/6, 17/6): x - floor(x), exact in floating point
  const double hx = __dadd_rn(__ddiv_rn((double)hf, 6.0), 1.0);
  const double hd = hx - floor(hx);
  const float h = (float)hd;
  uh = clip8i((int)__dmul_rn((double)h, 255.0));
  us = clip8i((int)__dmul_rn((double)s, 255.0));
}
__device__ __forceinline__ int c_round(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5)); }
// libImaging/Convert.c hsv2rgb
__device__ __forceinline__ void hsv2rgb8(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) {
    r = g = b = v;
    return;
  }
  const double hh = __ddiv_rn(__dmul_rn((double)(float)h, 6.0), 255.0);
  const int i = (int)floor(hh);
  const float f = (float)__dsub_rn(hh, (double)(float)i);
  const float fs = (float)__ddiv_rn((double)(float)s, 255.0);
  const double vf = (double)(float)v;
  const int p = clip8i(c_round(__dmul_rn(vf, __dsub_rn(1.0, (double)fs))));
  const int q = clip8i(c_round(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn((double)fs, (double)f)))));
  const int t = clip8i(c_round(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn((double)fs, __dsub_rn(1.0, (double)f))))));
  switch (i % 6) {
    case 0: r = v, g = t, b = p; break;
    case 1: r = q, g = v, b = p; break;
    case 2: r = p, g = v, b = t; break;
    case 3: r = p, g = q, b = v; break;
    case 4: r = t, g = p, b = v; break;
    default: r = v, g = p, b = q; break;
  }
}

// colour operations of one view in LDS, then ToTensor / Normalize.  ops[4]: 0 brightness, 1 contrast, 2 saturation, 3 hue,
// -1 none, applied in this order; out: float [view][3][224][224]
constexpr int kColorThreads = 1024;  // one workgroup per CU (LDS): 16 waves hide the hue step's fp64 latency
__global__ __launch_bounds__(kColorThreads) void aug_color_kernel(const unsigned char* __restrict__ img, const int* __restrict__ params,
                                                        const float* __restrict__ lut, float* __restrict__ out,
                                                        unsigned char* __restrict__ out_u8) {
  extern __shared__ __attribute__((aligned(16))) unsigned char px[];  // [224*224][3] then int red[4]
  constexpr int NPX = kOut * kOut;
  int* red = reinterpret_cast<int*>(px + NPX * 3);  // [kColorThreads / 64]
  const int v = blockIdx.x, tid = threadIdx.x;
  const int* pr = params + v * kAugParams;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(img + (size_t)v * NPX * 3);  // 150528 bytes = 37632 dwords
    unsigned* dst = reinterpret_cast<unsigned*>(px);
    for (int i = tid; i < NPX * 3 / 4; i += kColorThreads) dst[i] = src[i];
  }
  __syncthreads();
  for (int o = 0; o < 4; ++o) {
    const int op = pr[6 + o];  // uniform
    if (op < 0) continue;
    if (op == 0 || op == 1 || op == 2) {
      const float f = __int_as_float(pr[11 + op]);
      if (f == 1.0f) continue;  // ImagingBlend returns a copy of the image
      int mean = 0;
      if (op == 1) {  // int(mean(L) + 0.5): the sum is exact, the tie-free rounding is integer arithmetic
        int s = 0;
        for (int p = tid; p < NPX; p += kColorThreads) s += luma8(px[3 * p], px[3 * p + 1], px[3 * p + 2]);
        for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        long long tot = 0;
        for (int wv = 0; wv < kColorThreads / 64; ++wv) tot += red[wv];
        mean = (int)((2 * tot + NPX) / (2 * NPX));
        __syncthreads();
      }
      const int interp = f >= 0.f && f <= 1.f;
      for (int p = tid; p < NPX; p += kColorThreads) {
        const int r = px[3 * p], g = px[3 * p + 1], b = px[3 * p + 2];
        int d0, d1, d2;
        if (op == 0) d0 = d1 = d2 = 0;
        else if (op == 1) d0 = d1 = d2 = mean;
        else d0 = d1 = d2 = luma8(r, g, b);
        if (f == 0.0f) {  // ImagingBlend returns a copy of the degenerate image
          px[3 * p] = (unsigned char)d0, px[3 * p + 1] = (unsigned char)d1, px[3 * p + 2] = (unsigned char)d2;
        } else {
          px[3 * p] = (unsigned char)blend8(d0, r, f, interp);
          px[3 * p + 1] = (unsigned char)blend8(d1, g, f, interp);
          px[3 * p + 2] = (unsigned char)blend8(d2, b, f, interp);
        }
      }
    } else {  // hue: H += shift (uint8 wrap) in Pillow's HSV
      const int shift = pr[14] & 255;
      for (int p = tid; p < NPX; p += kColorThreads) {
        int uh, us, uv, r, g, b;
        rgb2hsv8(px[3 * p], px[3 * p + 1], px[3 * p + 2], uh, us, uv);
        hsv2rgb8((uh + shift) & 255, us, uv, r, g, b);
        px[3 * p] = (unsigned char)r, px[3 * p + 1] = (unsigned char)g, px[3 * p + 2] = (unsigned char)b;
      }
    }
    __syncthreads();
  }
  if (pr[10]) {  // RandomGrayscale: L on all three channels
    for (int p = tid; p < NPX; p += kColorThreads) {
      const int l = luma8(px[3 * p], px[3 * p + 1], px[3 * p + 2]);
      px[3 * p] = px[3 * p + 1] = px[3 * p + 2] = (unsigned char)l;
    }
    __syncthreads();
  }
  if (out_u8) {  // test tap: the augmented uint8 image
    unsigned* dst = reinterpret_cast<unsigned*>(out_u8 + (size_t)v * NPX * 3);
    const unsigned* s4 = reinterpret_cast<const unsigned*>(px);
    for (int i = tid; i < NPX * 3 / 4; i += kColorThreads) dst[i] = s4[i];
  }
  if (out) {
    float* o = out + (size_t)v * 3 * NPX;
    for (int c = 0; c < 3; ++c)
      for (int p = tid; p < NPX; p += kColorThreads) o[(size_t)c * NPX + p] = lut[c * 256 + px[3 * p + c]];
  }
}

}  // namespace hipac

using namespace hipac;

extern "C" {

int hipac_augment_views(const uint8_t* pool, int64_t n_pool, int P, int geometry, const int32_t* params_host, int32_t* params_dev,
                        int n_views, const int32_t* tab_bounds, const int32_t* tab_kk, int ksize, const float* lut, uint8_t* tmp,
                        uint8_t* crops, float* out, uint8_t* out_u8, void* stream) {
  HIPAC_REQUIRE(pool && params_host && params_dev && lut && crops && (out || out_u8), HIPAC_EINVAL, "augment_views: null argument");
  HIPAC_REQUIRE(geometry == 0 || geometry == 1, HIPAC_EINVAL, "augment_views: geometry %d (0: resized crop, 1: flips + rotation)", geometry);
  HIPAC_REQUIRE(geometry == 1 || (tab_bounds && tab_kk && tmp && ksize >= 1), HIPAC_EINVAL, "augment_views: the resized crop needs its tables");
  HIPAC_REQUIRE(geometry == 0 || P == kOut, HIPAC_EINVAL, "augment_views: flips + rotation act on 224-pixel patches (P = %d)", P);
  HIPAC_REQUIRE(n_pool > 0 && P >= 1 && P <= 4096 && n_views > 0 && n_views <= 65535, HIPAC_EINVAL,
                "augment_views: bad size (P %d, views %d)", P, n_views);
  // the kernels index the pool with these numbers: check them where they are still host memory
  for (int v = 0; v < n_views; ++v) {
    const int32_t* pr = params_host + (size_t)v * kAugParams;
    const bool crop_ok = pr[0] >= 0 && pr[0] < n_pool &&
                         (geometry == 1 || (pr[3] >= 1 && pr[4] >= 1 && pr[1] >= 0 && pr[2] >= 0 && (int64_t)pr[1] + pr[3] <= P &&
                                            (int64_t)pr[2] + pr[4] <= P));
    bool ops_ok = true;
    for (int o = 0; o < 4; ++o) ops_ok = ops_ok && pr[6 + o] >= -1 && pr[6 + o] <= 3;
    // 16.16 coordinates of a 224-pixel image must stay inside int32 (Pillow's check_fixed allows +-32768 pixels; a rotation
    // about the centre stays within a few hundred)
    bool fix_ok = true;
    if (geometry == 1)
      for (int k = 17; k <= 22; ++k) fix_ok = fix_ok && pr[k] > -(1 << 26) && pr[k] < (1 << 26);
    HIPAC_REQUIRE(crop_ok && ops_ok && fix_ok, HIPAC_EINVAL,
                  "augment_views: view %d: source %d, crop (%d, %d, %d, %d) of a %d-pixel patch, ops %d %d %d %d", v, pr[0], pr[1],
                  pr[2], pr[3], pr[4], P, pr[6], pr[7], pr[8], pr[9]);
  }
  hipStream_t s = (hipStream_t)stream;
  HIPAC_CHECK_HIP(hipMemcpyAsync(params_dev, params_host, (size_t)n_views * kAugParams * sizeof(int32_t), hipMemcpyHostToDevice, s));
  const int32_t* params = params_dev;
  dim3 gv((kOut * kOut + 255) / 256, (unsigned)n_views);
  if (geometry == 0) {
    dim3 gh((unsigned)(((size_t)P * kOut + 255) / 256), (unsigned)n_views);
    hipLaunchKernelGGL(aug_hpass_kernel, gh, dim3(256), 0, s, pool, P, params, tab_bounds, tab_kk, ksize, tmp);
    hipLaunchKernelGGL(aug_vpass_kernel, gv, dim3(256), 0, s, (const unsigned char*)tmp, P, params, tab_bounds, tab_kk, ksize, crops);
  } else {
    hipLaunchKernelGGL(aug_affine_kernel, gv, dim3(256), 0, s, pool, params, crops);
  }
  constexpr int LDS = kOut * kOut * 3 + kColorThreads / 64 * 4;
  static bool attr_done[64] = {};
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !attr_done[dev]) {
      HIPAC_CHECK_HIP(hipFuncSetAttribute((const void*)aug_color_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      if (dev >= 0) attr_done[dev] = true;
    }
  }
  hipLaunchKernelGGL(aug_color_kernel, dim3((unsigned)n_views), dim3(kColorThreads), LDS, s, (const unsigned char*)crops, params, lut, out,
                     out_u8);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
