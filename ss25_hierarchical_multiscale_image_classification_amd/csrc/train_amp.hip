// Mixed-precision training step of the ResNet18 encoder (opt-in for a-12): what the reference's
// fine-tune loops compute under torch.cuda.amp.autocast() + GradScaler (src/main.py:499-508, :578-587) -- fp16
// operands on v_mfma_f32_32x32x16_f16 with fp32 accumulation, fp16 activations and activation gradients, fp32
// master parameters / gradients / Adam state (the flat buffers of train.hip, unchanged), a loss scale owned by the
// caller (train_native.GradScaler) -- as hand-written HIP:
//
//   forward   weights re-packed fp32 -> fp16 per step; convolutions on the inference kernels of conv_igemm.h (halo
//             kernel for 3x3 / stride 1, LDS-DMA kernel for stride 2 and 1x1, v1 kernel for the stem) with a zero
//             bias and no ReLU; batch statistics from the fp16 map in fp64, two-stage and DETERMINISTIC (per
//             workgroup partials, summed in a fixed order); normalise (+ residual) (+ ReLU) fp16 -> fp16.
//   backward  BN backward (same two-stage reductions), data gradient = the same convolution kernels on flipped /
//             transposed fp16 weights (stride 2: on the zero-interleaved gradient), weight gradient = wgrad_f16_kernel
//             below: an MFMA GEMM over the pixel axis whose operands (dY and X, both [pixel][channel] in memory) are
//             read TRANSPOSED from LDS by ds_read_b64_tr_b16, split-K partials in fp32 summed in a fixed order (no
//             atomics: a step run twice gives the same bits).
// The gradients that arrive (dfeats) and leave (grads) carry the caller's loss scale.
#include <stdlib.h>

#include "conv_igemm.h"
#include "train_common.h"

namespace hipac {

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

static size_t packed_w_halfs(int i) { return i == 0 ? (size_t)64 * 224 : conv_w_floats(i); }
static size_t wpack_offset_h(int i) {
  size_t o = 0;
  for (int k = 0; k < i; ++k) o += (packed_w_halfs(k) + 127) & ~(size_t)127;  // 256-byte aligned rows of the table
  return o;
}
constexpr int kRedBlocks = 512;  // workgroups of a BN reduction pass = rows of the partial-sum table

struct AmpPlan {
  size_t xin;               // h16[B,230,232,4]
  size_t pre[kNumConvs];    // conv output before BN, h16
  size_t post[kNumConvs];   // after BN (+ residual) (+ ReLU), h16
  size_t pool, pool_idx;    // h16[B,56,56,64], uint8 arg-max
  size_t mean_rstd;         // floats, packed by stat_offset
  size_t red;               // double[kRedBlocks][2 * 512] partial sums of one reduction pass
  size_t sums;              // double[2 * 512]
  size_t wpack, wpack_d;    // h16 packed forward weights (all convs) / data-gradient weights (largest conv)
  size_t wg_part;           // float split-K partials of one weight gradient
  size_t zero_bias, zero_page;
  size_t g[3], up;          // h16 gradient maps, zero-interleaved gradient
  size_t total;
};
constexpr size_t kWgPartBytes = (size_t)160 << 20;

static AmpPlan make_amp_plan(int B) {
  AmpPlan p;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t b = (size_t)B;
  p.xin = take(b * kPadH * kPadW * 4 * 2);
  size_t maxact = 0, maxw = 0;
  for (int i = 0; i < kNumConvs; ++i) {
    const size_t n = b * kConvs[i].hout * kConvs[i].hout * kConvs[i].cout;
    p.pre[i] = take(n * 2);
    p.post[i] = take(n * 2);
    if (n > maxact) maxact = n;
    if (packed_w_halfs(i) > maxw) maxw = packed_w_halfs(i);
  }
  p.pool = take(b * 56 * 56 * 64 * 2);
  p.pool_idx = take(b * 56 * 56 * 64);
  p.mean_rstd = take(stat_offset(kNumConvs) * 4);
  p.red = take((size_t)kRedBlocks * 1024 * 8);
  p.sums = take(1024 * 8);
  p.wpack = take(wpack_offset_h(kNumConvs) * 2);
  p.wpack_d = take(maxw * 2);
  p.wg_part = take(kWgPartBytes);
  p.zero_bias = take(512 * 4);
  p.zero_page = take(256);
  for (int k = 0; k < 3; ++k) p.g[k] = take(maxact * 2);
  p.up = take(b * 56 * 56 * 128 * 2);
  p.total = off;
  return p;
}

// ---------------------------------------------------------------------------------------------
// small kernels (fp16 storage, fp32 / fp64 arithmetic)
// ---------------------------------------------------------------------------------------------
// fp32 [co][ci][kh][kw] -> fp16: mode 0 forward [co][(kh,kw)][ci], 1 data gradient [ci][flipped (kh,kw)][co], 2 stem [co][kh*32+kw*4+ci],
// 3 data gradient of a 3x3 / stride 2 conv, four parity-class blocks
__global__ __launch_bounds__(256) void pack_w_h_kernel(const float* __restrict__ w, h16* __restrict__ dst, int cout, int cin,
                                                       int ks, int mode) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)cout * cin * ks * ks) return;
  const int kw = (int)(gid % ks);
  long long t = gid / ks;
  const int kh = (int)(t % ks);
  t /= ks;
  const int ci = (int)(t % cin), co = (int)(t / cin);
  const h16 v = (h16)w[gid];
  if (mode == 0) dst[(size_t)co * ks * ks * cin + (size_t)(kh * ks + kw) * cin + ci] = v;
  else if (mode == 1) dst[(size_t)ci * ks * ks * cout + (size_t)((ks - 1 - kh) * ks + ks - 1 - kw) * cout + co] = v;
  else if (mode == 3) {
    // data gradient of a 3x3 / stride 2 conv by parity class (launch_dgrad_s2, conv_igemm.h): class (py, px) = (kh != 1, kw != 1),
    // its taps (a, b) = ((2 - kh) / 2, (2 - kw) / 2) -- tap a = 0 is the coarse row of the output position itself (kh = 2), a = 1 the
    // row below (kh = 0); blocks of 1, 2, 2, 4 taps back to back, each [ci][tap][co]
    const int py = kh != 1, px = kw != 1, a = py ? (2 - kh) / 2 : 0, b = px ? (2 - kw) / 2 : 0;
    const int ntap = (py ? 2 : 1) * (px ? 2 : 1), tap = a * (px ? 2 : 1) + b;
    const size_t blk = (size_t)cin * cout, off = (py ? 3 : 0) * blk + (px ? (py ? 2 : 1) : 0) * blk;
    dst[off + (size_t)ci * ntap * cout + (size_t)tap * cout + co] = v;
  } else dst[(size_t)co * 224 + kh * 32 + kw * 4 + ci] = v;
}

__device__ __forceinline__ void ld8(const h16* p, float (&v)[8]) {
  const f16x8 t = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void st8(h16* p, const float (&v)[8]) {
  f16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = (h16)v[e];
  *reinterpret_cast<f16x8*>(p) = t;
}

// Two per-channel sums over the M rows of [M][C] maps, 8 channels per thread, fp64, NO atomics: workgroup b leaves its
// partial sums in part[b][0..C) and part[b][512..512+C).  MODE 0: (sum x, sum x^2).  MODE 1 (BN backward): (sum dy,
// sum dy * xhat) with dy masked by (ymask > 0) when given.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_h_kernel(const h16* __restrict__ a, const h16* __restrict__ x,
                                                          const h16* __restrict__ ymask, long long M, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          double* __restrict__ part) {
  const int c8 = C >> 3, rows_per_pass = 256 / c8, tid = threadIdx.x, g = tid % c8, rsub = tid / c8;
  double s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.0;
  if (rsub < rows_per_pass) {
    float mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) mu[k] = mean[8 * g + k], rs[k] = rstd[8 * g + k];
    }
    const long long step = (long long)gridDim.x * rows_per_pass;
    long long r = (long long)blockIdx.x * rows_per_pass + rsub;
    if (MODE == 0) {
      // four rows in flight per thread (the pass is a chain of dependent 16-byte loads otherwise: 3.2 TB/s), added in row order
      for (; r + 3 * step < M; r += 4 * step) {
        float v0[8], v1[8], v2[8], v3[8];
        ld8(a + r * C + 8 * g, v0), ld8(a + (r + step) * C + 8 * g, v1), ld8(a + (r + 2 * step) * C + 8 * g, v2), ld8(a + (r + 3 * step) * C + 8 * g, v3);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s[k] += v0[k], q[k] += (double)v0[k] * v0[k];
          s[k] += v1[k], q[k] += (double)v1[k] * v1[k];
          s[k] += v2[k], q[k] += (double)v2[k] * v2[k];
          s[k] += v3[k], q[k] += (double)v3[k] * v3[k];
        }
      }
    }
    if (MODE == 1) {
      // two rows (six loads) in flight per thread, added in row order
      for (; r + step < M; r += 2 * step) {
        float va[8], vb[8], xa[8], xb[8];
        ld8(a + r * C + 8 * g, va), ld8(a + (r + step) * C + 8 * g, vb);
        ld8(x + r * C + 8 * g, xa), ld8(x + (r + step) * C + 8 * g, xb);
        if (ymask) {
          float ma[8], mb[8];
          ld8(ymask + r * C + 8 * g, ma), ld8(ymask + (r + step) * C + 8 * g, mb);
#pragma unroll
          for (int k = 0; k < 8; ++k) va[k] = ma[k] > 0.f ? va[k] : 0.f, vb[k] = mb[k] > 0.f ? vb[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s[k] += va[k], q[k] += (double)va[k] * ((xa[k] - mu[k]) * rs[k]);
          s[k] += vb[k], q[k] += (double)vb[k] * ((xb[k] - mu[k]) * rs[k]);
        }
      }
    }
    for (; r < M; r += step) {
      float v[8];
      ld8(a + r * C + 8 * g, v);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k], q[k] += (double)v[k] * v[k];
      } else {
        float xv[8];
        ld8(x + r * C + 8 * g, xv);
        if (ymask) {
          float m[8];
          ld8(ymask + r * C + 8 * g, m);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = m[k] > 0.f ? v[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k], q[k] += (double)v[k] * ((xv[k] - mu[k]) * rs[k]);
      }
    }
  }
  __shared__ double red[2][256][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) red[0][tid][k] = s[k], red[1][tid][k] = q[k];
  __syncthreads();
  if (tid < c8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double sa = 0, sb = 0;
      for (int rr = 0; rr < rows_per_pass; ++rr) sa += red[0][rr * c8 + tid][k], sb += red[1][rr * c8 + tid][k];
      part[(size_t)blockIdx.x * 1024 + 8 * tid + k] = sa;
      part[(size_t)blockIdx.x * 1024 + 512 + 8 * tid + k] = sb;
    }
  }
}

// partial sums -> sums[c], sums[512 + c] in a FIXED order (lane l of the channel's 32 adds blocks l, l + 32, ... in turn,
// then a shuffle tree): 8 channels per workgroup; FINAL: also mean / rstd and the running statistics
template <bool FINAL>
__global__ __launch_bounds__(256) void bn_sum_parts_kernel(const double* __restrict__ part, int nblocks, int C,
                                                           double* __restrict__ sums, long long M, float eps, float momentum,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  double a = 0, b = 0;
  if (c < C)
    for (int k = l; k < nblocks; k += 32) a += part[(size_t)k * 1024 + c], b += part[(size_t)k * 1024 + 512 + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a += __shfl_down(a, o, 32), b += __shfl_down(b, o, 32);
  if (c >= C || l != 0) return;
  sums[c] = a, sums[512 + c] = b;
  if (FINAL) {
    const double mu = a / (double)M;
    double var = b / (double)M - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
      const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
      run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
      run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
  }
}

// y = (x - mean) * rstd * gamma + beta (+ resid) (ReLU), fp16 -> fp16.  The grid stride (a multiple of 2048 elements) is a
// multiple of C, so a thread meets the same 8 channels in every iteration: their constants are loaded once.
__global__ __launch_bounds__(256) void bn_apply_h_kernel(const h16* __restrict__ x, const h16* __restrict__ resid,
                                                         h16* __restrict__ y, long long n8, int C, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int relu) {
  const int c = (threadIdx.x * 8) % C;
  float mu[8], sc[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) mu[k] = mean[c + k], sc[k] = rstd[c + k], be[k] = beta[c + k];
  float ga[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ga[k] = gamma[c + k];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8], o[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (v[k] - mu[k]) * sc[k] * ga[k] + be[k];
    if (resid) {
      float r[8];
      ld8(resid + i * 8, r);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] += r[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    st8(y + i * 8, o);
  }
}

// BN backward, pass 2: dx = gamma * rstd * (dy - sum_dy / M - xhat * sum_dy_xhat / M); d gamma, d beta (fp32 gradients)
__global__ __launch_bounds__(256) void bn_bwd_apply_h_kernel(const h16* __restrict__ dy, const h16* __restrict__ x,
                                                             const h16* __restrict__ ymask, h16* __restrict__ dx, long long n8,
                                                             long long M, int C, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const double* __restrict__ sums, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int accumulate) {
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const float dg = (float)sums[512 + c], db = (float)sums[c];
      dgamma[c] = accumulate ? dgamma[c] + dg : dg;
      dbeta[c] = accumulate ? dbeta[c] + db : db;
    }
  }
  const double invM = 1.0 / (double)M;
  const int c = (threadIdx.x * 8) % C;  // the same 8 channels in every iteration (grid stride is a multiple of C)
  float mu[8], rs[8], gr[8], sb[8], sg[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    mu[k] = mean[c + k], rs[k] = rstd[c + k], gr[k] = gamma[c + k] * rstd[c + k];
    sb[k] = (float)(sums[c + k] * invM), sg[k] = (float)(sums[512 + c + k] * invM);
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float d[8], v[8], o[8];
    ld8(dy + i * 8, d);
    ld8(x + i * 8, v);
    if (ymask) {
      float m[8];
      ld8(ymask + i * 8, m);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = m[k] > 0.f ? d[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gr[k] * (d[k] - sb[k] - (v[k] - mu[k]) * rs[k] * sg[k]);
    st8(dx + i * 8, o);
  }
}

// out = (a + b) masked by (y > 0); b / y optional
__global__ __launch_bounds__(256) void add_mask_h_kernel(const h16* __restrict__ a, const h16* __restrict__ b,
                                                         const h16* __restrict__ y, h16* __restrict__ out, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    ld8(a + i * 8, v);
    if (b) {
      float w[8];
      ld8(b + i * 8, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += w[k];
    }
    if (y) {
      float m[8];
      ld8(y + i * 8, m);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = m[k] > 0.f ? v[k] : 0.f;
    }
    st8(out + i * 8, v);
  }
}

// zero-interleave: up[b][2y][2x][c] = g[b][y][x][c], every other position 0
__global__ __launch_bounds__(256) void upsample_zero_h_kernel(const h16* __restrict__ g, h16* __restrict__ up, long long n8, int H,
                                                              int C) {
  const int c8 = C >> 3, H2 = 2 * H;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const int cg = (int)(i % c8);
    long long t = i / c8;
    const int X = (int)(t % H2);
    t /= H2;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!(X & 1) && !(Y & 1)) v = *reinterpret_cast<const u32x4*>(g + (((b * H + (Y >> 1)) * H + (X >> 1)) * C) + 8 * cg);
    *reinterpret_cast<u32x4*>(up + i * 8) = v;
  }
}

// 3x3/2 max-pool, pad 1, with the arg-max kept (first maximum in (dy, dx) scan order, as torch); 8 channels per thread
__global__ __launch_bounds__(256) void maxpool_idx_h_kernel(const h16* __restrict__ in, h16* __restrict__ out,
                                                            unsigned char* __restrict__ idx, long long total8) {
  constexpr int HI = 112, HO = 56, C = 64;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total8) return;
  const int c8 = (int)(gid % (C / 8));
  long long p = gid / (C / 8);
  const int ow = (int)(p % HO);
  p /= HO;
  const int oh = (int)(p % HO);
  const long long b = p / HO;
  float best[8];
  int bi[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) best[k] = -INFINITY, bi[k] = 9;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)HI) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)HI) continue;
      float v[8];
      ld8(in + ((b * HI + ih) * HI + iw) * C + 8 * c8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (v[k] > best[k] || bi[k] == 9) best[k] = v[k], bi[k] = dy * 3 + dx;
    }
  }
  st8(out + gid * 8, best);
  unsigned lo = 0, hi = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) lo |= (unsigned)bi[k] << (8 * k), hi |= (unsigned)bi[4 + k] << (8 * k);
  *reinterpret_cast<u32x2*>(idx + gid * 8) = u32x2{lo, hi};
}

// max-pool backward (gather form): every input position sums the gradients of the <= 4 windows that chose it.  A thread owns a
// 2 x 2 quad of input positions (2y .. 2y+1, 2x .. 2x+1) x 8 channels: the quad only ever belongs to the four windows
// (y .. y+1) x (x .. x+1) -- row 2y to window row y alone (dy = 1), row 2y+1 to window rows y (dy = 2) and y+1 (dy = 0) -- so four
// window loads serve four outputs (one thread per position loaded nine for four)
__global__ __launch_bounds__(256) void maxpool_bwd_h_kernel(const h16* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                            h16* __restrict__ din, long long total8) {
  constexpr int HI = 112, HO = 56, C = 64;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;  // (b, y, x, c8) over the 56 x 56 quads
  if (gid >= total8) return;
  const int c8 = (int)(gid % (C / 8));
  long long p = gid / (C / 8);
  const int x = (int)(p % HO);
  p /= HO;
  const int y = (int)(p % HO);
  const long long b = p / HO;
  float acc[2][2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[i][j][k] = 0.f;
#pragma unroll
  for (int wy = 0; wy < 2; ++wy) {
    const int oh = y + wy;
    if (oh >= HO) continue;
#pragma unroll
    for (int wx = 0; wx < 2; ++wx) {
      const int ow = x + wx;
      if (ow >= HO) continue;
      const long long o = ((b * HO + oh) * HO + ow) * C + 8 * c8;
      const u32x2 ib = *reinterpret_cast<const u32x2*>(idx + o);
      float g[8];
      ld8(dout + o, g);
      // window (oh, ow) covers input rows 2 oh - 1 + dy: quad row i = 0 (input row 2y) is dy = 1 of wy = 0; quad row i = 1
      // (input row 2y + 1) is dy = 2 of wy = 0 and dy = 0 of wy = 1 -- the same in x
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int dy = wy == 0 ? 1 + i : (i == 1 ? 0 : -1);
        if (dy < 0) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int dx = wx == 0 ? 1 + j : (j == 1 ? 0 : -1);
          if (dx < 0) continue;
          const int code = dy * 3 + dx;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if ((int)(((k < 4 ? ib[0] : ib[1]) >> (8 * (k & 3))) & 0xffu) == code) acc[i][j][k] += g[k];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      st8(din + (((b * HI + 2 * y + i) * HI + 2 * x + j) * C + 8 * c8), acc[i][j]);
}

__global__ __launch_bounds__(256) void avgpool_h_kernel(const h16* __restrict__ last, float* __restrict__ feats, int n) {
  const int b = blockIdx.x, t = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  for (int p = 0; p < 49; ++p) {
    const f16x2 v = *reinterpret_cast<const f16x2*>(last + ((size_t)b * 49 + p) * 512 + 2 * t);
    s0 += (float)v[0], s1 += (float)v[1];
  }
  *reinterpret_cast<float2*>(feats + (size_t)b * 512 + 2 * t) = make_float2(s0 / 49.0f, s1 / 49.0f);
}

__global__ __launch_bounds__(256) void avgpool_bwd_h_kernel(const float* __restrict__ dfeats, const h16* __restrict__ last,
                                                            h16* __restrict__ dlast, long long total) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % 512);
  const long long b = gid / (49 * 512);
  dlast[gid] = (float)last[gid] > 0.f ? (h16)(dfeats[b * 512 + c] * (1.0f / 49.0f)) : (h16)0.f;
}

// ---------------------------------------------------------------------------------------------
// weight gradient on the fp16 MFMA:  part[slice][tap][co][ci] = sum over the slice's output pixels m of
//     dY[m][co] * X[pixel(m, tap)][ci]
// One workgroup = one (TC x TC) (co x ci) tile of NTAP filter taps over one slice of the pixel axis; 4 waves = 2 x 2
// wave tiles of (TC/2)^2 per tap.  The dY tile is staged ONCE per 32-pixel sub-chunk and multiplied with the NTAP
// shifted X tiles (one filter row = 3 taps for the 64-channel layers, 7 filter rows for the stem, 1 tap for the wide
// layers, whose 128 x 128 tiles are bound by the operand streams, not by the staging): NTAP MFMAs per dY fragment.  The contraction runs over PIXELS, but both operands lie
// [pixel][channel] in memory: they are staged as they come (16-byte pieces, rows of TC channels) and the MFMA
// fragments -- 8 consecutive pixels of one channel -- are read TRANSPOSED with ds_read_b64_tr_b16 (two per fragment).
// LDS row pitch = 2 TC + 64 bytes: the 4 rows x 2 channel blocks of a 32-lane half then fall on 64 distinct banks.
// The next sub-chunk's global loads are in flight behind the MFMAs; two barriers per sub-chunk.
// STEM: X is the padded NHWC4 input, a "tap" is filter row kh and the "ci" axis of its tile the 32 values (kw, c)
// (TC = 64: only the first 32 columns exist, waves wj = 1 take the odd filter rows instead of a second column half).
// ---------------------------------------------------------------------------------------------
template <int TC, int NTAP, bool STEM>
__global__ __launch_bounds__(256) void wgrad_f16_kernel(const h16* __restrict__ dY, const h16* __restrict__ X,
                                                        float* __restrict__ part, int Cout, int Cin, int KS, int stride, int HO,
                                                        int HI, long long M, int chunk) {
  constexpr int PITCH = 2 * TC + 64;             // bytes per staged pixel row
  constexpr int CH = TC / 8;                     // 16-byte pieces per row
  constexpr int PPT = 32 * CH / 256;             // pieces per thread and tile (1 for TC = 64, 2 for TC = 128)
  constexpr int WT = TC / 2, NF = WT / 32;       // wave tile, 32-wide fragments per wave and operand
  constexpr int TILE = 32 * PITCH;
  constexpr int MYT = STEM ? (NTAP + 1) / 2 : NTAP;  // taps a wave accumulates
  __shared__ __attribute__((aligned(16))) unsigned char smem[(1 + NTAP) * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ci_tiles = STEM ? 1 : Cin / TC, co_tiles = Cout / TC;
  int t = blockIdx.x;
  const int cit = t % ci_tiles;
  t /= ci_tiles;
  const int cot = t % co_tiles;
  const int tap0 = (t / co_tiles) * NTAP;
  const int pad = STEM ? 0 : KS / 2;
  const int wi = wave & 1, wj = wave >> 1;  // co half, ci half (STEM: filter-row parity)
  f32x16 acc[MYT][NF][NF];
#pragma unroll
  for (int tp = 0; tp < MYT; ++tp)
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tp][i][j][e] = 0.f;
  const long long m_begin = (long long)blockIdx.y * chunk;
  const long long m_end = m_begin + chunk < M ? m_begin + chunk : M;

  u32x4 ra[PPT], rb[NTAP][PPT];
  auto gload = [&](long long m0) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int piece = tid + 256 * k, spx = piece / CH, sc = piece % CH;
      const long long m = m0 + spx;
      ra[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int tp = 0; tp < NTAP; ++tp) rb[tp][k] = u32x4{0u, 0u, 0u, 0u};
      if (m < m_end) {
        ra[k] = *reinterpret_cast<const u32x4*>(dY + m * Cout + cot * TC + 8 * sc);
        const int ox = (int)(m % HO);
        const long long tt = m / HO;
        const int oy = (int)(tt % HO);
        const long long b = tt / HO;
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
          if constexpr (STEM) {
            if (sc < 4) rb[tp][k] = *reinterpret_cast<const u32x4*>(X + (((b * kPadH + 2 * oy + tap0 + tp) * kPadW) + 2 * ox) * 4 + 8 * sc);
          } else {
            const int tap = tap0 + tp, kh = tap / KS, kw = tap - kh * KS;
            const int iy = oy * stride + kh - pad, ix = ox * stride + kw - pad;
            if ((unsigned)iy < (unsigned)HI && (unsigned)ix < (unsigned)HI)
              rb[tp][k] = *reinterpret_cast<const u32x4*>(X + ((b * HI + iy) * HI + ix) * (long long)Cin + cit * TC + 8 * sc);
          }
        }
      }
    }
  };
  // transposed fragment address of this lane inside a tile: block row q = (lane & 15) >> 2, column quad p = lane & 3
  const int tr_off = ((lane & 15) >> 2) * PITCH + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2 + 8 * h * PITCH;
  auto frag_tr = [&](const unsigned char* img, int col0, int kk) -> f16x8 {
    const unsigned char* p = img + tr_off + (16 * kk) * PITCH + col0 * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
    return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  gload(m_begin);
  for (long long m0 = m_begin; m0 < m_end; m0 += 32) {
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int piece = tid + 256 * k, spx = piece / CH, sc = piece % CH;
      *reinterpret_cast<u32x4*>(smem + spx * PITCH + 16 * sc) = ra[k];
#pragma unroll
      for (int tp = 0; tp < NTAP; ++tp) *reinterpret_cast<u32x4*>(smem + (1 + tp) * TILE + spx * PITCH + 16 * sc) = rb[tp][k];
    }
    __syncthreads();
    if (m0 + 32 < m_end) gload(m0 + 32);  // in flight behind the MFMAs (and the other workgroups of the CU)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 af[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) af[i] = frag_tr(smem, wi * WT + 32 * i, kk);
#pragma unroll
      for (int tp = 0; tp < MYT; ++tp) {
        const int tile = STEM ? 2 * tp + wj : tp;
        if (STEM && tile >= NTAP) continue;
        f16x8 bf[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) bf[j] = frag_tr(smem + (1 + tile) * TILE, (STEM ? 0 : wj * WT) + 32 * j, kk);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j) acc[tp][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[tp][i][j], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave has read the tiles: the next sub-chunk may overwrite them
  }
  // D[co][ci]: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
  const int row_len = STEM ? 32 : Cin;
  const int ntaps = STEM ? 7 : KS * KS;
#pragma unroll
  for (int tp = 0; tp < MYT; ++tp) {
    const int tap = tap0 + (STEM ? 2 * tp + wj : tp);
    if (STEM && tap >= NTAP) continue;
    float* base = part + ((size_t)blockIdx.y * ntaps + tap) * (size_t)Cout * row_len;
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int col = (STEM ? 0 : cit * TC + wj * WT) + 32 * j + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = cot * TC + wi * WT + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
          base[(size_t)row * row_len + col] = acc[tp][i][j][e];
        }
      }
  }
}

// split-K partials -> the PyTorch-layout gradient (accumulate or overwrite).  32 weights per workgroup x 8 slice groups:
// group g adds slices g, g + 8, ... in turn, then the 8 group sums are added in order -- a fixed order, hence deterministic
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int slices, float* __restrict__ dw,
                                                           int cout, int cin, int ks, int stem, int accumulate) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long gid = (long long)blockIdx.x * 32 + e;
  const long long total = (long long)cout * cin * ks * ks;
  float s = 0.f;
  if (gid < total) {
    const int kw = (int)(gid % ks);
    long long t = gid / ks;
    const int kh = (int)(t % ks);
    t /= ks;
    const int ci = (int)(t % cin), co = (int)(t / cin);
    const size_t per_slice = stem ? (size_t)7 * cout * 32 : (size_t)total;
    const size_t o = stem ? ((size_t)kh * cout + co) * 32 + kw * 4 + ci : ((size_t)(kh * ks + kw) * cout + co) * cin + ci;
    for (int k = g; k < slices; k += 8) s += part[(size_t)k * per_slice + o];
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && gid < total) {
    float v = red[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += red[k][e];
    dw[gid] = accumulate ? dw[gid] + v : v;
  }
}

// grads *= inv_scale; flag[0] = 1 when a gradient is not finite (GradScaler.unscale_)
__global__ __launch_bounds__(256) void unscale_check_kernel(float* __restrict__ g, long long n, float inv_scale,
                                                            int* __restrict__ flag) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = g[i] * inv_scale;
    g[i] = v;
    bad |= !(fabsf(v) <= 3.0e38f);
  }
  if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int HI, int KS, int STRIDE>
static int conv_h(const h16* in, const h16* wp, const float* zero_bias, h16* out, int n, hipStream_t s, const char* zero_page) {
  ConvW w{const_cast<h16*>(wp), const_cast<float*>(zero_bias)};
  return launch_conv<h16, CIN, COUT, HI, HI, KS, STRIDE, false, false, false>(in, w, nullptr, out, n, s, zero_page);
}
static int conv_forward_h(int i, const h16* in, const h16* wp, const float* zb, h16* out, int n, hipStream_t s, const char* zp) {
  const ConvDesc& d = kConvs[i];
  if (i == 0) {
    ConvW w{const_cast<h16*>(wp), const_cast<float*>(zb)};
    return launch_conv<h16, 4, 64, 224, 224, 7, 2, false, false, false, true>(in, w, nullptr, out, n, s);
  }
  if (d.ks == 3 && d.stride == 1) {
    switch (d.cout) {
      case 64: return conv_h<64, 64, 56, 3, 1>(in, wp, zb, out, n, s, zp);
      case 128: return conv_h<128, 128, 28, 3, 1>(in, wp, zb, out, n, s, zp);
      case 256: return conv_h<256, 256, 14, 3, 1>(in, wp, zb, out, n, s, zp);
      default: return conv_h<512, 512, 7, 3, 1>(in, wp, zb, out, n, s, zp);
    }
  }
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return conv_h<64, 128, 56, 3, 2>(in, wp, zb, out, n, s, zp);
      case 256: return conv_h<128, 256, 28, 3, 2>(in, wp, zb, out, n, s, zp);
      default: return conv_h<256, 512, 14, 3, 2>(in, wp, zb, out, n, s, zp);
    }
  }
  switch (d.cout) {
    case 128: return conv_h<64, 128, 56, 1, 2>(in, wp, zb, out, n, s, zp);
    case 256: return conv_h<128, 256, 28, 1, 2>(in, wp, zb, out, n, s, zp);
    default: return conv_h<256, 512, 14, 1, 2>(in, wp, zb, out, n, s, zp);
  }
}
#ifndef HIPAC_AMP_DGRAD_CLASSES
#define HIPAC_AMP_DGRAD_CLASSES 1  // stride-2 data gradients by parity class (1/4 of the MFMAs, no zero-interleaved map); 0: round-3 first form
#endif
// data gradient of a STRIDE-2 conv i by parity classes: g = gradient wrt the conv output on the coarse grid, weights in mode 3
// (3x3) or mode 1 (1x1; `out` zeroed by the caller)
static int conv_dgrad_s2_h(int i, const h16* g, const h16* wd, const float* zb, h16* out, int n, hipStream_t s, const char* zp) {
  const ConvDesc& d = kConvs[i];
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return launch_dgrad_s2<h16, 128, 64, 28, true>(g, wd, zb, out, n, s, zp);
      case 256: return launch_dgrad_s2<h16, 256, 128, 14, true>(g, wd, zb, out, n, s, zp);
      default: return launch_dgrad_s2<h16, 512, 256, 7, true>(g, wd, zb, out, n, s, zp);
    }
  }
  switch (d.cout) {
    case 128: return launch_dgrad_s2<h16, 128, 64, 28, false>(g, wd, zb, out, n, s, zp);
    case 256: return launch_dgrad_s2<h16, 256, 128, 14, false>(g, wd, zb, out, n, s, zp);
    default: return launch_dgrad_s2<h16, 512, 256, 7, false>(g, wd, zb, out, n, s, zp);
  }
}
// data gradient of conv i: g = gradient wrt the conv output (stride 2: already zero-interleaved to hin x hin), weights in mode 1
static int conv_dgrad_h(int i, const h16* g, const h16* wd, const float* zb, h16* out, int n, hipStream_t s, const char* zp) {
  const ConvDesc& d = kConvs[i];
  if (d.ks == 3 && d.stride == 1) {
    switch (d.cout) {
      case 64: return conv_h<64, 64, 56, 3, 1>(g, wd, zb, out, n, s, zp);
      case 128: return conv_h<128, 128, 28, 3, 1>(g, wd, zb, out, n, s, zp);
      case 256: return conv_h<256, 256, 14, 3, 1>(g, wd, zb, out, n, s, zp);
      default: return conv_h<512, 512, 7, 3, 1>(g, wd, zb, out, n, s, zp);
    }
  }
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return conv_h<128, 64, 56, 3, 1>(g, wd, zb, out, n, s, zp);
      case 256: return conv_h<256, 128, 28, 3, 1>(g, wd, zb, out, n, s, zp);
      default: return conv_h<512, 256, 14, 3, 1>(g, wd, zb, out, n, s, zp);
    }
  }
  switch (d.cout) {
    case 128: return conv_h<128, 64, 56, 1, 1>(g, wd, zb, out, n, s, zp);
    case 256: return conv_h<256, 128, 28, 1, 1>(g, wd, zb, out, n, s, zp);
    default: return conv_h<512, 256, 14, 1, 1>(g, wd, zb, out, n, s, zp);
  }
}

static int pack_weights_h(const float* w, h16* dst, int i, int mode, hipStream_t s) {
  const ConvDesc& d = kConvs[i];
  const long long total = (long long)conv_w_floats(i);
  if (mode == 2) HIPAC_CHECK_HIP(hipMemsetAsync(dst, 0, packed_w_halfs(0) * 2, s));
  hipLaunchKernelGGL(pack_w_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, d.cout, d.cin, d.ks, mode);
  return (int)hipGetLastError();
}

struct AmpCtx {
  const float* params;
  float* stats;
  char* ws;
  const AmpPlan* p;
  float eps, momentum;
  hipStream_t s;
};

static int red_blocks(long long M, int C) {
  const int rows_per_pass = 256 / (C / 8);
  long long gs = (M + rows_per_pass - 1) / rows_per_pass;
  return (int)(gs > kRedBlocks ? kRedBlocks : gs);
}

static int bn_forward_h(const AmpCtx& c, int i, int n, const h16* resid, int relu) {
  const ConvDesc& d = kConvs[i];
  const long long M = (long long)n * d.hout * d.hout;
  const h16* x = (const h16*)(c.ws + c.p->pre[i]);
  h16* y = (h16*)(c.ws + c.p->post[i]);
  double* part = (double*)(c.ws + c.p->red);
  double* sums = (double*)(c.ws + c.p->sums);
  float* mean = (float*)(c.ws + c.p->mean_rstd) + stat_offset(i);
  float* rstd = mean + d.cout;
  const float* gamma = c.params + param_offset(i) + conv_w_floats(i);
  const int gs = red_blocks(M, d.cout);
  hipLaunchKernelGGL((bn_reduce_h_kernel<0>), dim3(gs), dim3(256), 0, c.s, x, (const h16*)nullptr, (const h16*)nullptr, M, d.cout,
                     (const float*)nullptr, (const float*)nullptr, part);
  float* rm = c.stats ? c.stats + stat_offset(i) : nullptr;
  hipLaunchKernelGGL((bn_sum_parts_kernel<true>), dim3((d.cout + 7) / 8), dim3(256), 0, c.s, (const double*)part, gs, d.cout,
                     sums, M, c.eps, c.momentum, mean, rstd, rm, rm ? rm + d.cout : nullptr);
  const long long n8 = M * d.cout / 8;
  hipLaunchKernelGGL(bn_apply_h_kernel, dim3(grid_for(n8)), dim3(256), 0, c.s, x, resid, y, n8, d.cout, (const float*)mean,
                     (const float*)rstd, gamma, gamma + d.cout, relu);
  return (int)hipGetLastError();
}

static int bn_backward_h(const AmpCtx& c, int i, int n, const h16* dy, const h16* ymask, h16* dx, float* grads, int accumulate) {
  const ConvDesc& d = kConvs[i];
  const long long M = (long long)n * d.hout * d.hout;
  const h16* x = (const h16*)(c.ws + c.p->pre[i]);
  double* part = (double*)(c.ws + c.p->red);
  double* sums = (double*)(c.ws + c.p->sums);
  const float* mean = (const float*)(c.ws + c.p->mean_rstd) + stat_offset(i);
  const float* rstd = mean + d.cout;
  const float* gamma = c.params + param_offset(i) + conv_w_floats(i);
  float* dgamma = grads + param_offset(i) + conv_w_floats(i);
  const int gs = red_blocks(M, d.cout);
  hipLaunchKernelGGL((bn_reduce_h_kernel<1>), dim3(gs), dim3(256), 0, c.s, dy, x, ymask, M, d.cout, mean, rstd, part);
  hipLaunchKernelGGL((bn_sum_parts_kernel<false>), dim3((d.cout + 7) / 8), dim3(256), 0, c.s, (const double*)part, gs, d.cout,
                     sums, M, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  const long long n8 = M * d.cout / 8;
  hipLaunchKernelGGL(bn_bwd_apply_h_kernel, dim3(grid_for(n8)), dim3(256), 0, c.s, dy, x, ymask, dx, n8, M, d.cout, mean, rstd,
                     gamma, (const double*)sums, dgamma, dgamma + d.cout, accumulate);
  return (int)hipGetLastError();
}

// weight gradient of conv i: X = the conv's input map, dY = gradient wrt its output -> grads (PyTorch layout)
static int conv_wgrad_h(const AmpCtx& c, int i, int n, const h16* X, const h16* dY, float* grads, int accumulate) {
  const ConvDesc& d = kConvs[i];
  float* part = (float*)(c.ws + c.p->wg_part);
  const long long M = (long long)n * d.hout * d.hout;
  const bool stem = i == 0;
  const size_t pf = stem ? (size_t)7 * 64 * 32 : conv_w_floats(i);
  const bool big = !stem && d.cout >= 128 && d.cin >= 128;
  const int TC = big ? 128 : 64;
  static const int knob_big = getenv("HIPAC_WG_NTAP_BIG") ? atoi(getenv("HIPAC_WG_NTAP_BIG")) : 1;      // developer knobs; measured (2 x 1024 views, img/s):
  static const int knob_small = getenv("HIPAC_WG_NTAP_SMALL") ? atoi(getenv("HIPAC_WG_NTAP_SMALL")) : 3;  // (big, small) = (1, 3) 27.9 k, (1, 9) 27.4 k, (3, 3) 26.2 k, (3, 9) 25.8 k, (1, 1) 26.6 k
  static const int knob_wgs = getenv("HIPAC_WG_TARGET") ? atoi(getenv("HIPAC_WG_TARGET")) : 768;  // workgroups per launch: 256 23.0 k, 512 27.4 k, 768 28.5 k, 1536 27.9 k, 3072 27.1 k
  const int ntap_wg = stem ? 7 : (d.ks == 1 ? 1 : (big ? knob_big : knob_small));  // taps per workgroup (they share the dY tile)
  const int tiles = stem ? 1 : (d.ks * d.ks / ntap_wg) * (d.cout / TC) * (d.cin / TC);
  // split the pixel axis so that the launch has ~768 workgroups (3 per CU); slices bounded by the partials buffer
  long long slices = (knob_wgs + tiles - 1) / tiles;
  const long long cap = (long long)(kWgPartBytes / (pf * 4));
  if (slices > cap) slices = cap;
  if (slices < 1) slices = 1;
  long long chunk = (M + slices - 1) / slices;
  chunk = (chunk + 31) / 32 * 32;
  if (chunk < 128) chunk = 128;
  slices = (M + chunk - 1) / chunk;
  dim3 grid(tiles, (unsigned)slices);
#define HIPAC_WG(TC_, NT_, ST_)                                                                                          \
  hipLaunchKernelGGL((wgrad_f16_kernel<TC_, NT_, ST_>), grid, dim3(256), 0, c.s, dY, X, part, d.cout, d.cin, d.ks, d.stride, \
                     d.hout, d.hin, M, (int)chunk)
  if (stem) HIPAC_WG(64, 7, true);
  else if (big && ntap_wg == 3) HIPAC_WG(128, 3, false);
  else if (big) HIPAC_WG(128, 1, false);
  else if (ntap_wg == 9) HIPAC_WG(64, 9, false);
  else if (ntap_wg == 3) HIPAC_WG(64, 3, false);
  else HIPAC_WG(64, 1, false);
#undef HIPAC_WG
  const long long total = (long long)conv_w_floats(i);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, c.s, (const float*)part, (int)slices,
                     grads + param_offset(i), d.cout, d.cin, d.ks, stem ? 1 : 0, accumulate);
  return (int)hipGetLastError();
}

}  // namespace hipac

using namespace hipac;

extern "C" {

size_t hipac_train_amp_workspace_bytes(int batch) { return batch > 0 ? make_amp_plan(batch).total : 0; }

// Test tap, as hipac_train_debug_offset but for the fp16 workspace (maps are fp16 NHWC)
int64_t hipac_train_amp_debug_offset(int batch, int kind, int conv) {
  if (batch <= 0 || conv < 0 || conv >= kNumConvs) return -1;
  const AmpPlan p = make_amp_plan(batch);
  switch (kind) {
    case 0: return (int64_t)p.pre[conv];
    case 1: return (int64_t)p.post[conv];
    case 2: return (int64_t)p.pool;
    case 3: return (int64_t)(p.mean_rstd + stat_offset(conv) * 4);
    case 4: return (int64_t)p.pool_idx;
    default: return -1;
  }
}

#define TRY(e)                                                                          \
  do {                                                                                  \
    int rc__ = (e);                                                                     \
    HIPAC_REQUIRE(rc__ == 0, rc__, "train_amp: launch failed (%d) at line %d", rc__, __LINE__); \
  } while (0)

int hipac_train_amp_encoder_forward(const float* params, float* stats, const float* x, int batch, float momentum, float eps,
                                    float* feats, void* workspace, size_t workspace_bytes, void* stream) {
  HIPAC_REQUIRE(params && x && feats && workspace, HIPAC_EINVAL, "train_amp_forward: null argument");
  HIPAC_REQUIRE(batch > 0 && batch <= 2048, HIPAC_EINVAL, "train_amp_forward: batch %d (1 .. 2048: 32-bit byte offsets)", batch);
  const AmpPlan p = make_amp_plan(batch);
  HIPAC_REQUIRE(workspace_bytes >= p.total, HIPAC_EWORKSPACE, "train_amp_forward: workspace %zu < required %zu", workspace_bytes,
                p.total);
  HIPAC_REQUIRE(((uintptr_t)workspace & 255) == 0, HIPAC_EINVAL, "train_amp_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int n = batch;
  float* zb = (float*)(ws + p.zero_bias);
  const char* zp = ws + p.zero_page;
  HIPAC_CHECK_HIP(hipMemsetAsync(zb, 0, 512 * 4 + 256, s));  // zero_bias and zero_page are adjacent
  h16* wpack = (h16*)(ws + p.wpack);
  for (int i = 0; i < kNumConvs; ++i) TRY(pack_weights_h(params + param_offset(i), wpack + wpack_offset_h(i), i, i == 0 ? 2 : 0, s));
  TRY(launch_nchw_to_nhwc4(x, ws + p.xin, n, HIPAC_PREC_FP16, s));
  AmpCtx c{params, stats, ws, &p, eps, momentum, s};
  auto pre = [&](int i) { return (h16*)(ws + p.pre[i]); };
  auto post = [&](int i) { return (h16*)(ws + p.post[i]); };
  TRY(conv_forward_h(0, (const h16*)(ws + p.xin), wpack, zb, pre(0), n, s, zp));
  TRY(bn_forward_h(c, 0, n, nullptr, 1));
  {
    const long long total = (long long)n * 56 * 56 * 8;  // 8 channels per thread
    hipLaunchKernelGGL(maxpool_idx_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16*)post(0),
                       (h16*)(ws + p.pool), (unsigned char*)(ws + p.pool_idx), total);
    TRY((int)hipGetLastError());
  }
  const h16* cur = (const h16*)(ws + p.pool);
  int i = 1;
  for (int stage = 0; stage < 4; ++stage) {
    for (int blk = 0; blk < 2; ++blk) {
      const bool down = stage > 0 && blk == 0;
      const int c1 = i, c2 = i + 1, ds = down ? i + 2 : -1;
      TRY(conv_forward_h(c1, cur, wpack + wpack_offset_h(c1), zb, pre(c1), n, s, zp));
      TRY(bn_forward_h(c, c1, n, nullptr, 1));
      const h16* idt = cur;
      if (down) {
        TRY(conv_forward_h(ds, cur, wpack + wpack_offset_h(ds), zb, pre(ds), n, s, zp));
        TRY(bn_forward_h(c, ds, n, nullptr, 0));
        idt = post(ds);
      }
      TRY(conv_forward_h(c2, post(c1), wpack + wpack_offset_h(c2), zb, pre(c2), n, s, zp));
      TRY(bn_forward_h(c, c2, n, idt, 1));
      cur = post(c2);
      i += down ? 3 : 2;
    }
  }
  hipLaunchKernelGGL(avgpool_h_kernel, dim3(n), dim3(256), 0, s, cur, feats, n);
  TRY((int)hipGetLastError());
  return 0;
}

int hipac_train_amp_encoder_backward(const float* params, const float* dfeats, int batch, float* grads, int accumulate,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  HIPAC_REQUIRE(params && dfeats && grads && workspace, HIPAC_EINVAL, "train_amp_backward: null argument");
  HIPAC_REQUIRE(batch > 0 && batch <= 2048, HIPAC_EINVAL, "train_amp_backward: batch %d", batch);
  const AmpPlan p = make_amp_plan(batch);
  HIPAC_REQUIRE(workspace_bytes >= p.total, HIPAC_EWORKSPACE, "train_amp_backward: workspace %zu < required %zu", workspace_bytes,
                p.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int n = batch;
  const float* zb = (const float*)(ws + p.zero_bias);
  const char* zp = ws + p.zero_page;
  h16* wd = (h16*)(ws + p.wpack_d);
  AmpCtx c{params, nullptr, ws, &p, 0.f, 0.f, s};
  auto post = [&](int i) { return (h16*)(ws + p.post[i]); };
  h16* gA = (h16*)(ws + p.g[0]);
  h16* gB = (h16*)(ws + p.g[1]);
  h16* gC = (h16*)(ws + p.g[2]);
  h16* up = (h16*)(ws + p.up);
  {
    const long long total = (long long)n * 49 * 512;
    hipLaunchKernelGGL(avgpool_bwd_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dfeats, (const h16*)post(19), gA,
                       total);
    TRY((int)hipGetLastError());
  }
  static const int kFirst[4][2] = {{1, 3}, {5, 8}, {10, 13}, {15, 18}};
  for (int stage = 3; stage >= 0; --stage) {
    for (int blk = 1; blk >= 0; --blk) {
      const bool down = stage > 0 && blk == 0;
      const int c1 = kFirst[stage][blk], c2 = c1 + 1, ds = down ? c1 + 2 : -1;
      const h16* xin_blk;
      const h16* prev_post;
      if (stage == 0 && blk == 0) xin_blk = (const h16*)(ws + p.pool), prev_post = nullptr;
      else {
        const int pc2 = (blk == 1 ? kFirst[stage][0] : kFirst[stage - 1][1]) + 1;
        xin_blk = post(pc2), prev_post = xin_blk;
      }
      const ConvDesc& d1 = kConvs[c1];
      const long long n_in8 = (long long)n * d1.hin * d1.hin * d1.cin / 8;
      TRY(bn_backward_h(c, c2, n, gA, nullptr, gB, grads, accumulate));
      TRY(conv_wgrad_h(c, c2, n, post(c1), gB, grads, accumulate));
      TRY(pack_weights_h(params + param_offset(c2), wd, c2, 1, s));
      TRY(conv_dgrad_h(c2, gB, wd, zb, gC, n, s, zp));
      TRY(bn_backward_h(c, c1, n, gC, post(c1), gC, grads, accumulate));
      TRY(conv_wgrad_h(c, c1, n, xin_blk, gC, grads, accumulate));
      if (d1.stride == 2 && HIPAC_AMP_DGRAD_CLASSES) {
        TRY(pack_weights_h(params + param_offset(c1), wd, c1, 3, s));
        TRY(conv_dgrad_s2_h(c1, gC, wd, zb, gB, n, s, zp));
      } else {
        TRY(pack_weights_h(params + param_offset(c1), wd, c1, 1, s));
        const h16* g1 = gC;
        if (d1.stride == 2) {
          const long long nu8 = (long long)n * d1.hin * d1.hin * d1.cout / 8;
          hipLaunchKernelGGL(upsample_zero_h_kernel, dim3(grid_for(nu8)), dim3(256), 0, s, (const h16*)gC, up, nu8, d1.hout, d1.cout);
          TRY((int)hipGetLastError());
          g1 = up;
        }
        TRY(conv_dgrad_h(c1, g1, wd, zb, gB, n, s, zp));
      }
      if (down) {
        TRY(bn_backward_h(c, ds, n, gA, nullptr, gC, grads, accumulate));
        TRY(conv_wgrad_h(c, ds, n, xin_blk, gC, grads, accumulate));
        TRY(pack_weights_h(params + param_offset(ds), wd, ds, 1, s));
        if (HIPAC_AMP_DGRAD_CLASSES) {
          // 1x1 / stride 2: only the even positions of the fine grid receive a gradient; `up` takes it (gC holds the input)
          HIPAC_CHECK_HIP(hipMemsetAsync(up, 0, (size_t)n * d1.hin * d1.hin * kConvs[ds].cin * 2, s));
          TRY(conv_dgrad_s2_h(ds, gC, wd, zb, up, n, s, zp));
          hipLaunchKernelGGL(add_mask_h_kernel, dim3(grid_for(n_in8)), dim3(256), 0, s, (const h16*)gB, (const h16*)up, prev_post, gA,
                             n_in8);
          TRY((int)hipGetLastError());
          continue;
        }
        const long long nu8 = (long long)n * d1.hin * d1.hin * kConvs[ds].cout / 8;
        hipLaunchKernelGGL(upsample_zero_h_kernel, dim3(grid_for(nu8)), dim3(256), 0, s, (const h16*)gC, up, nu8, kConvs[ds].hout,
                           kConvs[ds].cout);
        TRY((int)hipGetLastError());
        TRY(conv_dgrad_h(ds, up, wd, zb, gC, n, s, zp));
        hipLaunchKernelGGL(add_mask_h_kernel, dim3(grid_for(n_in8)), dim3(256), 0, s, (const h16*)gB, (const h16*)gC, prev_post, gA,
                           n_in8);
      } else {
        hipLaunchKernelGGL(add_mask_h_kernel, dim3(grid_for(n_in8)), dim3(256), 0, s, (const h16*)gB, (const h16*)gA, prev_post, gA,
                           n_in8);
      }
      TRY((int)hipGetLastError());
    }
  }
  {
    const long long total = (long long)n * 56 * 56 * 8;  // a 2 x 2 quad of positions x 8 channels per thread
    hipLaunchKernelGGL(maxpool_bwd_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const h16*)gA,
                       (const unsigned char*)(ws + p.pool_idx), gB, total);
    TRY((int)hipGetLastError());
  }
  TRY(bn_backward_h(c, 0, n, gB, post(0), gB, grads, accumulate));
  TRY(conv_wgrad_h(c, 0, n, (const h16*)(ws + p.xin), gB, grads, accumulate));
  return 0;
}

// GradScaler.unscale_: grads *= inv_scale in place; found_inf[0] (device int32, zeroed by the caller) is set when a
// gradient is inf / nan (src/main.py:506-508 scaler.step / scaler.update skip the optimizer step then)
int hipac_grads_unscale_check(float* grads, int64_t n, float inv_scale, int32_t* found_inf, void* stream) {
  HIPAC_REQUIRE(grads && found_inf && n > 0, HIPAC_EINVAL, "grads_unscale_check: bad argument");
  hipLaunchKernelGGL(unscale_check_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grads, (long long)n, inv_scale,
                     (int*)found_inf);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
