// Tile / crop / white-pad / whiteness / Pillow-exact resize / normalise kernels.
//
// Integer-exact restatement of what the reference does per window on the CPU
// (src/main.py:693-703, :718-720) followed by torchvision's
// Resize((224,224)) -> ToTensor -> Normalize (src/main.py:812-818), i.e.
// Pillow's two-pass 8bpc resampler (libImaging/Resample.c): horizontal pass,
// uint8 rounding, vertical pass, uint8 rounding, 22-bit fixed-point weights.
#include <math.h>

#include <vector>

#include "common.h"

namespace hipac {

constexpr int kPrecisionBits = 32 - 8 - 2;  // Pillow PRECISION_BITS
constexpr int kStripRows = 8;               // output rows per workgroup
constexpr int kStrips = kPatch / kStripRows;

__device__ __forceinline__ unsigned clip8(int acc) {
  int v = acc >> kPrecisionBits;
  v = v < 0 ? 0 : v;
  return (unsigned)(v > 255 ? 255 : v);
}

__device__ __forceinline__ unsigned block_sum_u32(unsigned v, unsigned* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__device__ __forceinline__ void store_nhwc4(void* out, size_t pix, float a, float b, float c) {
  typename Elem<T>::vec4 v;
  v[0] = (T)a;
  v[1] = (T)b;
  v[2] = (T)c;
  v[3] = (T)0.f;
  *reinterpret_cast<typename Elem<T>::vec4*>(reinterpret_cast<T*>(out) + pix * 4) = v;
}

// Write one resized pixel (r,g,b as uint8 values) of window `w` at (oy, ox).
__device__ __forceinline__ void emit_pixel(void* out, int fmt, const float* __restrict__ lut, int w, int oy,
                                           int ox, unsigned r, unsigned g, unsigned b) {
  if (fmt == HIPAC_OUT_U8_HWC) {
    uint8_t* o = reinterpret_cast<uint8_t*>(out) + (((size_t)w * kPatch + oy) * kPatch + ox) * 3;
    o[0] = (uint8_t)r;
    o[1] = (uint8_t)g;
    o[2] = (uint8_t)b;
    return;
  }
  const float fr = lut[r], fg = lut[256 + g], fb = lut[512 + b];
  if (fmt == HIPAC_OUT_NCHW_F32) {
    float* o = reinterpret_cast<float*>(out) + ((size_t)w * 3 * kPatch + oy) * kPatch + ox;
    o[0] = fr;
    o[(size_t)kPatch * kPatch] = fg;
    o[(size_t)2 * kPatch * kPatch] = fb;
  } else {
    const size_t pix = ((size_t)w * kPadH + oy + 3) * kPadW + ox + 3;
    if (fmt == HIPAC_OUT_NHWC4_PAD_BF16)
      store_nhwc4<__bf16>(out, pix, fr, fg, fb);
    else
      store_nhwc4<_Float16>(out, pix, fr, fg, fb);
  }
}

// ---------------------------------------------------------------------------------------
// General P (multiple of 224, > 224): one workgroup = one window x one strip of 8 output
// rows.  Each wave stages one source row of the window at a time into LDS (16-byte
// coalesced loads, white beyond the level's right/bottom edge), runs the horizontal
// pass for it into the strip's uint8 H buffer, then all threads run the vertical pass.
//   dynamic LDS: kk table [224*ksize] int32 | H [maxrows][672] u8 | raw [4][P*CH] u8
// ---------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void tile_resize_kernel(const uint8_t* __restrict__ level, int W, int H,
                                                          long long pitch, const int* __restrict__ xy, int P,
                                                          const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize,
                                                          const float* __restrict__ lut, void* __restrict__ out,
                                                          int fmt, unsigned* __restrict__ sums, int maxrows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ unsigned red[4];
  int* kk_s = reinterpret_cast<int*>(dsm);
  unsigned char* Hbuf = dsm + (size_t)kPatch * ksize * 4;
  const int rawpitch = P * CH;  // multiple of 16 (P multiple of 224)
  unsigned char* raw = Hbuf + (size_t)maxrows * (kPatch * 3);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int strip = blockIdx.x, w = blockIdx.y;
  const int x0 = xy[2 * w], y0 = xy[2 * w + 1];
  const int scale = P / kPatch;

  for (int i = tid; i < kPatch * ksize; i += 256) kk_s[i] = kk[i];

  const int oy0 = strip * kStripRows;
  const int rowbeg = bounds[2 * oy0];
  const int rowend = bounds[2 * (oy0 + kStripRows - 1)] + bounds[2 * (oy0 + kStripRows - 1) + 1];
  const int nrows = rowend - rowbeg;
  const int own_beg = oy0 * scale, own_end = (oy0 + kStripRows) * scale;
  __syncthreads();

  unsigned mysum = 0;
  unsigned char* myraw = raw + (size_t)wave * rawpitch;
  const long long rowbytes_valid = (long long)W * CH;  // bytes of real pixels in a level row
  for (int it = 0; it < nrows; it += 4) {
    const int rr = it + wave;  // row within the strip's staged range
    const bool active = rr < nrows;
    const int wy = rowbeg + rr;  // row within the window
    const long long gy = (long long)y0 + wy;
    const bool count = active && wy >= own_beg && wy < own_end && sums != nullptr;
    if (active) {
      for (int p = lane; p * 16 < rawpitch; p += 64) {
        const long long gbyte = (long long)x0 * CH + (long long)p * 16;  // byte within the level row
        u32x4 v = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (gy < H && gbyte < pitch) {
          v = *reinterpret_cast<const u32x4*>(level + gy * pitch + gbyte);
          const long long nv = rowbytes_valid - gbyte;  // valid bytes in this piece
          if (nv < 16) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const long long n = nv - 4 * d;
              const unsigned m = n >= 4 ? 0u : (n <= 0 ? 0xffffffffu : (0xffffffffu << (8 * (int)n)));
              v[d] |= m;
            }
          }
        }
        *reinterpret_cast<u32x4*>(myraw + p * 16) = v;
        if (count) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            if constexpr (CH == 3) {
              mysum = __builtin_amdgcn_sad_u8(v[d], 0u, mysum);
            } else {
              mysum = __builtin_amdgcn_sad_u8(v[d] & 0x00ffffffu, 0u, mysum);
            }
          }
        }
      }
    }
    __syncthreads();
    if (active) {
      unsigned char* hrow = Hbuf + (size_t)rr * (kPatch * 3);
      for (int j = lane; j < kPatch; j += 64) {
        const int xmin = bounds[2 * j], cnt = bounds[2 * j + 1];
        const int* k = kk_s + j * ksize;
        int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
        const unsigned char* src = myraw + xmin * CH;
        for (int t = 0; t < cnt; ++t) {
          const int kv = k[t];
          a0 += (int)src[t * CH + 0] * kv;
          a1 += (int)src[t * CH + 1] * kv;
          a2 += (int)src[t * CH + 2] * kv;
        }
        hrow[j * 3 + 0] = (unsigned char)clip8(a0);
        hrow[j * 3 + 1] = (unsigned char)clip8(a1);
        hrow[j * 3 + 2] = (unsigned char)clip8(a2);
      }
    }
    __syncthreads();
  }

  // vertical pass + ToTensor/Normalize + store
  for (int item = tid; item < kStripRows * kPatch; item += 256) {
    const int yy = item / kPatch, j = item - yy * kPatch;
    const int oy = oy0 + yy;
    const int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
    const int* k = kk_s + oy * ksize;
    int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
    const unsigned char* src = Hbuf + (size_t)(ymin - rowbeg) * (kPatch * 3) + j * 3;
    for (int t = 0; t < cnt; ++t) {
      const int kv = k[t];
      a0 += (int)src[(size_t)t * (kPatch * 3) + 0] * kv;
      a1 += (int)src[(size_t)t * (kPatch * 3) + 1] * kv;
      a2 += (int)src[(size_t)t * (kPatch * 3) + 2] * kv;
    }
    emit_pixel(out, fmt, lut, w, oy, j, clip8(a0), clip8(a1), clip8(a2));
  }

  if (sums != nullptr) {
    const unsigned tot = block_sum_u32(mysum, red);
    if (tid == 0) atomicAdd(&sums[w], tot);
  }
}

// P == 224: Resize is the identity (Pillow returns a copy).  One workgroup = one
// window x 8 rows.  `img_stride` != 0 selects "n separate 224x224 images" addressing
// (hipac_patches_normalize) instead of windows of one level image.
template <int CH>
__global__ __launch_bounds__(256) void tile_identity_kernel(const uint8_t* __restrict__ level, int W, int H,
                                                            long long pitch, long long img_stride,
                                                            const int* __restrict__ xy,
                                                            const float* __restrict__ lut, void* __restrict__ out,
                                                            int fmt, unsigned* __restrict__ sums) {
  __shared__ unsigned red[4];
  const int tid = threadIdx.x;
  const int strip = blockIdx.x, w = blockIdx.y;
  const int x0 = xy ? xy[2 * w] : 0, y0 = xy ? xy[2 * w + 1] : 0;
  const uint8_t* base = level + (long long)w * img_stride;
  unsigned mysum = 0;
  for (int item = tid; item < kStripRows * kPatch; item += 256) {
    const int yy = item / kPatch, j = item - yy * kPatch;
    const int oy = strip * kStripRows + yy;
    const long long gy = (long long)y0 + oy, gx = (long long)x0 + j;
    unsigned r = 255, g = 255, b = 255;
    if (gy < H && gx < W) {
      const uint8_t* s = base + gy * pitch + gx * CH;
      r = s[0];
      g = s[1];
      b = s[2];
    }
    mysum += r + g + b;
    emit_pixel(out, fmt, lut, w, oy, j, r, g, b);
  }
  if (sums != nullptr) {
    const unsigned tot = block_sum_u32(mysum, red);
    if (tid == 0) atomicAdd(&sums[w], tot);
  }
}

__global__ void keep_kernel(const unsigned* __restrict__ sums, unsigned char* __restrict__ keep, int n,
                            unsigned threshold) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keep[i] = sums[i] <= threshold ? 1 : 0;
}

// tumour label: any mask byte > 0 inside the window (outside the mask counts as 0).
__global__ __launch_bounds__(256) void window_labels_kernel(const uint8_t* __restrict__ mask, int W, int H,
                                                            long long pitch, const int* __restrict__ xy, int P,
                                                            unsigned char* __restrict__ labels) {
  __shared__ int any_s;
  const int w = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) any_s = 0;
  __syncthreads();
  const int x0 = xy[2 * w], y0 = xy[2 * w + 1];
  const int x1 = min(x0 + P, W), y1 = min(y0 + P, H);
  const int cols = x1 - x0, rows = y1 - y0;
  int found = 0;
  if (cols > 0 && rows > 0) {
    const long long total = (long long)rows * cols;
    for (long long i = tid; i < total && !found; i += 256) {
      const int ry = (int)(i / cols), rx = (int)(i - (long long)ry * cols);
      if (mask[(long long)(y0 + ry) * pitch + x0 + rx] > 0) found = 1;
    }
  }
  if (found) atomicOr(&any_s, 1);
  __syncthreads();
  if (tid == 0) labels[w] = (unsigned char)any_s;
}

static double triangle(double x) {
  x = fabs(x);
  return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace hipac

using namespace hipac;

constexpr int kMaxGridY = 65535;  // HIP limit of gridDim.y

extern "C" {

// Pillow: precompute_coeffs + normalize_coeffs_8bpc (libImaging/Resample.c) for the
// bilinear filter over the full box [0, in_size).  Built with -ffp-contract=off so no
// multiply-add is fused: every operation rounds to double exactly as in Pillow's build.
int hipac_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int kk_stride) {
  HIPAC_REQUIRE(in_size > 0 && out_size > 0, HIPAC_EINVAL, "resample_coeffs: sizes %d -> %d", in_size, out_size);
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (!bounds || !kk) return ksize;
  HIPAC_REQUIRE(kk_stride >= ksize, HIPAC_EINVAL, "resample_coeffs: kk_stride %d < ksize %d", kk_stride, ksize);
  std::vector<double> k(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double wgt = triangle((x + xmin - center + 0.5) * ss);
      k[x] = wgt;
      ww += wgt;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < kk_stride; ++x) {
      int32_t q = 0;
      if (x < xmax) {
        const double v = k[x];
        q = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << kPrecisionBits)) : (int32_t)(0.5 + v * (double)(1 << kPrecisionBits));
      }
      kk[(size_t)xx * kk_stride + x] = q;
    }
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  return ksize;
}

static int check_fmt(int fmt) {
  return fmt == HIPAC_OUT_NCHW_F32 || fmt == HIPAC_OUT_NHWC4_PAD_BF16 || fmt == HIPAC_OUT_NHWC4_PAD_FP16 ||
         fmt == HIPAC_OUT_U8_HWC;
}

static size_t out_bytes_per_patch(int fmt) {
  switch (fmt) {
    case HIPAC_OUT_NCHW_F32: return (size_t)3 * kPatch * kPatch * 4;
    case HIPAC_OUT_U8_HWC: return (size_t)3 * kPatch * kPatch;
    default: return (size_t)kPadH * kPadW * 4 * 2;
  }
}

int hipac_tile_preprocess(const uint8_t* level, int W, int H, int64_t pitch, int chans, const int32_t* xy, int n,
                          int P, const int32_t* coeff_bounds, const int32_t* coeff_kk, int ksize,
                          const float* lut, void* out, int out_format, uint32_t* sums, uint8_t* keep,
                          void* stream) {
  HIPAC_REQUIRE(level && xy && out, HIPAC_EINVAL, "tile_preprocess: null argument");
  HIPAC_REQUIRE(n >= 0, HIPAC_EINVAL, "tile_preprocess: n %d", n);
  HIPAC_REQUIRE(chans == 3 || chans == 4, HIPAC_EINVAL, "tile_preprocess: chans %d (3 or 4)", chans);
  HIPAC_REQUIRE(P >= kPatch && P % kPatch == 0 && P <= 8 * kPatch, HIPAC_EINVAL,
                "tile_preprocess: P %d must be 224*s, s in 1..8", P);
  HIPAC_REQUIRE(W > 0 && H > 0 && pitch >= (int64_t)W * chans, HIPAC_EINVAL, "tile_preprocess: bad geometry");
  HIPAC_REQUIRE(check_fmt(out_format), HIPAC_EINVAL, "tile_preprocess: out_format %d", out_format);
  HIPAC_REQUIRE(out_format == HIPAC_OUT_U8_HWC || lut, HIPAC_EINVAL, "tile_preprocess: lut is null");
  HIPAC_REQUIRE(!keep || sums, HIPAC_EINVAL, "tile_preprocess: keep needs sums");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (sums) HIPAC_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)n * 4, s));
  if (out_format == HIPAC_OUT_NHWC4_PAD_BF16 || out_format == HIPAC_OUT_NHWC4_PAD_FP16)
    HIPAC_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n * out_bytes_per_patch(out_format), s));
  // gridDim.y is limited to 65 535: windows go in chunks (pointers advanced per chunk, kernels index from 0)
  const size_t obytes = out_bytes_per_patch(out_format);
  const int maxrows = P == kPatch ? 0 : (kStripRows + 1) * (P / kPatch);
  const size_t lds = P == kPatch ? 0 : (size_t)kPatch * ksize * 4 + (size_t)maxrows * kPatch * 3 + (size_t)4 * P * chans;
  if (P != kPatch) {
    HIPAC_REQUIRE(coeff_bounds && coeff_kk, HIPAC_EINVAL, "tile_preprocess: coefficient tables are null");
    const int scale = P / kPatch;
    HIPAC_REQUIRE(ksize == 2 * scale + 1, HIPAC_EINVAL, "tile_preprocess: ksize %d != %d", ksize, 2 * scale + 1);
    HIPAC_REQUIRE(pitch % 16 == 0 && ((uintptr_t)level & 15) == 0, HIPAC_EINVAL,
                  "tile_preprocess: level base and pitch must be 16-byte aligned");
    HIPAC_REQUIRE(lds <= 160 * 1024, HIPAC_EUNSUPPORTED, "tile_preprocess: LDS %zu", lds);
    HIPAC_CHECK_HIP(hipFuncSetAttribute(chans == 3 ? (const void*)tile_resize_kernel<3> : (const void*)tile_resize_kernel<4>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  for (int w0 = 0; w0 < n; w0 += kMaxGridY) {
    const int nw = n - w0 < kMaxGridY ? n - w0 : kMaxGridY;
    dim3 grid(kStrips, nw);
    const int32_t* xyc = xy + 2 * (size_t)w0;
    void* outc = (char*)out + (size_t)w0 * obytes;
    uint32_t* sumc = sums ? sums + w0 : nullptr;
    if (P == kPatch) {
      if (chans == 3)
        hipLaunchKernelGGL((tile_identity_kernel<3>), grid, dim3(256), 0, s, level, W, H, (long long)pitch, 0LL, xyc,
                           lut, outc, out_format, sumc);
      else
        hipLaunchKernelGGL((tile_identity_kernel<4>), grid, dim3(256), 0, s, level, W, H, (long long)pitch, 0LL, xyc,
                           lut, outc, out_format, sumc);
    } else if (chans == 3) {
      hipLaunchKernelGGL((tile_resize_kernel<3>), grid, dim3(256), lds, s, level, W, H, (long long)pitch, xyc, P,
                         coeff_bounds, coeff_kk, ksize, lut, outc, out_format, sumc, maxrows);
    } else {
      hipLaunchKernelGGL((tile_resize_kernel<4>), grid, dim3(256), lds, s, level, W, H, (long long)pitch, xyc, P,
                         coeff_bounds, coeff_kk, ksize, lut, outc, out_format, sumc, maxrows);
    }
  }
  HIPAC_CHECK_HIP(hipGetLastError());
  if (keep) {
    const unsigned thr = 240u * 3u * (unsigned)P * (unsigned)P;  // <= 2.32e9 < 2^32 for P <= 1792
    hipLaunchKernelGGL(keep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sums, keep, n, thr);
    HIPAC_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

int hipac_window_labels(const uint8_t* mask, int W, int H, int64_t pitch, const int32_t* xy, int n, int P,
                        uint8_t* labels, void* stream) {
  HIPAC_REQUIRE(mask && xy && labels, HIPAC_EINVAL, "window_labels: null argument");
  HIPAC_REQUIRE(n >= 0 && P > 0 && W > 0 && H > 0 && pitch >= W, HIPAC_EINVAL, "window_labels: bad geometry");
  if (n == 0) return 0;
  hipLaunchKernelGGL(window_labels_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, mask, W, H, (long long)pitch,
                     xy, P, labels);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipac_patches_normalize(const uint8_t* patches, int n, const float* lut, void* out, int out_format,
                            void* stream) {
  HIPAC_REQUIRE(patches && lut && out, HIPAC_EINVAL, "patches_normalize: null argument");
  HIPAC_REQUIRE(check_fmt(out_format) && out_format != HIPAC_OUT_U8_HWC, HIPAC_EINVAL,
                "patches_normalize: out_format %d", out_format);
  if (n <= 0) return n == 0 ? 0 : HIPAC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (out_format != HIPAC_OUT_NCHW_F32)
    HIPAC_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)n * out_bytes_per_patch(out_format), s));
  const size_t obytes = out_bytes_per_patch(out_format);
  for (int w0 = 0; w0 < n; w0 += kMaxGridY) {
    const int nw = n - w0 < kMaxGridY ? n - w0 : kMaxGridY;
    hipLaunchKernelGGL((tile_identity_kernel<3>), dim3(kStrips, nw), dim3(256), 0, s,
                       patches + (size_t)w0 * kPatch * kPatch * 3, kPatch, kPatch, (long long)kPatch * 3,
                       (long long)kPatch * kPatch * 3, (const int*)nullptr, lut, (char*)out + (size_t)w0 * obytes,
                       out_format, (unsigned*)nullptr);
  }
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
