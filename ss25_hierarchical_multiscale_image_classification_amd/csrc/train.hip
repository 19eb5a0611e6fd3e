// Native training step of the ResNet18 encoder: train-mode forward and
// backward as hand-written HIP, fp32 throughout (the reference's pretrain_simclr runs fp32 without
// autocast, src/models/simclr.py:85-96), convolutions on the exact f32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   forward   conv (implicit GEMM, the v1 kernel of conv_igemm.h on re-packed weights) -> batch statistics
//             (fp64 atomics) -> normalise (+ residual) (+ ReLU); 3x3/2 max-pool with saved arg-max;
//             global average pool.  Pre-BN and post-activation maps of every conv are kept for the backward.
//   backward  BN backward (two passes: d gamma / d beta, then dx), conv weight gradient (MFMA GEMM over the
//             pixel axis, split-K with fp32 atomics), conv data gradient = the forward kernel on flipped /
//             transposed weights (stride-2 layers: on the zero-interleaved gradient), max-pool / average-pool
//             backward, ReLU masks fused into the consumers.
//   plus      linear layers (projector, fc) on a strided fp32 MFMA GEMM, weighted cross-entropy, Adam.
//
// Activations are NHWC fp32.  Parameters live in ONE flat fp32 buffer in a fixed order (per conv: weight in
// the PyTorch layout [Cout][Cin][kh][kw], then BN gamma, beta); running statistics in a second flat buffer
// (per conv: running_mean, running_var); gradients in a buffer shaped like the parameters.
#include "conv_igemm.h"
#include "train_common.h"

namespace hipac {

// ---------------------------------------------------------------------------------------------
// workspace of one forward (everything the backward needs) + scratch shared by forward / backward
// ---------------------------------------------------------------------------------------------
struct TrainPlan {
  size_t xin;               // float[B,230,232,4]
  size_t pre[kNumConvs];    // conv output before BN
  size_t post[kNumConvs];   // after BN (+ residual) (+ ReLU)
  size_t pool, pool_idx;    // float[B,56,56,64], uint8 arg-max (0..8, 9 = none)
  size_t mean_rstd;         // per conv: mean[cout], rstd[cout] (floats), packed by stat_offset
  size_t sums;              // double[2 * 512] scratch of the statistics / BN-backward reductions
  size_t red;               // double[kRedBlocks32][2 * 512]: per-workgroup partial sums of one reduction pass (no atomics)
  size_t wpack;             // packed forward weights of all convs
  size_t wpack_d;           // packed data-gradient weights (largest conv)
  size_t wgrad_p;           // packed weight-gradient accumulator (largest conv)
  size_t zero_bias;         // float[512] zeros
  size_t g[3];              // gradient maps (largest activation each)
  size_t up;                // zero-interleaved gradient of a stride-2 layer
  size_t total;
};

constexpr int kRedBlocks32 = 512;  // workgroups of a BN reduction pass = rows of the partial-sum table

// split K of conv i's weight gradient over `slices` chunks of `chunk` pixels so that the launch has ~2048 workgroups
static void wgrad_split(int i, long long M, long long& slices, long long& chunk) {
  const ConvDesc& d = kConvs[i];
  const int tiles = i == 0 ? 7 : d.ks * d.ks * (d.cout / 64) * (d.cin / 64);
  slices = (2048 + tiles - 1) / tiles;
  chunk = (M + slices - 1) / slices;
  chunk = (chunk + 31) / 32 * 32;
  if (chunk < 256) chunk = 256;
  slices = (M + chunk - 1) / chunk;
}

static TrainPlan make_train_plan(int B) {
  TrainPlan p;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t b = (size_t)B;
  p.xin = take(b * kPadH * kPadW * 4 * 4);
  size_t maxact = 0, maxw = 0;
  for (int i = 0; i < kNumConvs; ++i) {
    const size_t n = b * kConvs[i].hout * kConvs[i].hout * kConvs[i].cout;
    p.pre[i] = take(n * 4);
    p.post[i] = take(n * 4);
    if (n > maxact) maxact = n;
    if (packed_w_floats(i) > maxw) maxw = packed_w_floats(i);
  }
  p.pool = take(b * 56 * 56 * 64 * 4);
  p.pool_idx = take(b * 56 * 56 * 64);
  p.mean_rstd = take(stat_offset(kNumConvs) * 4);
  p.sums = take(2 * 512 * 8);
  p.red = take((size_t)kRedBlocks32 * 1024 * 8);
  size_t wtot = 0;
  for (int i = 0; i < kNumConvs; ++i) wtot += packed_w_floats(i);
  p.wpack = take(wtot * 4);
  p.wpack_d = take(maxw * 4);
  size_t maxpart = 0;  // split-K partials of a weight gradient: [slices][packed weights], summed in slice order afterwards
  for (int i = 0; i < kNumConvs; ++i) {
    long long sl, ch;
    wgrad_split(i, (long long)b * kConvs[i].hout * kConvs[i].hout, sl, ch);
    const size_t pf = i == 0 ? (size_t)7 * 64 * 32 : conv_w_floats(i);
    if ((size_t)sl * (i == 0 ? 2 : 1) * pf > maxpart) maxpart = (size_t)sl * (i == 0 ? 2 : 1) * pf;
  }
  p.wgrad_p = take(maxpart * 4);
  p.zero_bias = take(512 * 4);
  for (int k = 0; k < 3; ++k) p.g[k] = take(maxact * 4);
  p.up = take(b * 56 * 56 * 128 * 4);  // largest: layer2 entry (128 ch at 56 x 56)
  p.total = off;
  return p;
}
static size_t wpack_offset(int i) {
  size_t o = 0;
  for (int k = 0; k < i; ++k) o += packed_w_floats(k);
  return o;
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// mode 0: forward pack  dst[co][(kh*ks+kw)*cin + ci]        = w[co][ci][kh][kw]
// mode 1: data-gradient dst[ci][((ks-1-kh)*ks + ks-1-kw)*cout + co] = w[co][ci][kh][kw]
// mode 2: stem          dst[co][kh*32 + kw*4 + ci] (row of 224, rest zero: the caller clears dst)
// mode 3: data gradient of a 3x3 / stride 2 conv, four parity-class blocks (see below)
__global__ __launch_bounds__(256) void pack_w_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout,
                                                     int cin, int ks, int mode) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)cout * cin * ks * ks;
  if (gid >= total) return;
  const int kw = (int)(gid % ks);
  long long t = gid / ks;
  const int kh = (int)(t % ks);
  t /= ks;
  const int ci = (int)(t % cin), co = (int)(t / cin);
  const float v = w[gid];
  if (mode == 0) dst[(size_t)co * ks * ks * cin + (size_t)(kh * ks + kw) * cin + ci] = v;
  else if (mode == 1) dst[(size_t)ci * ks * ks * cout + (size_t)((ks - 1 - kh) * ks + ks - 1 - kw) * cout + co] = v;
  else if (mode == 3) {
    // data gradient of a 3x3 / stride 2 conv by parity class (launch_dgrad_s2_v1, conv_igemm.h): class (py, px) = (kh != 1, kw != 1),
    // taps (a, b) = ((2 - kh) / 2, (2 - kw) / 2); blocks of 1, 2, 2, 4 taps back to back, each [ci][tap][co]
    const int py = kh != 1, px = kw != 1, a = py ? (2 - kh) / 2 : 0, b = px ? (2 - kw) / 2 : 0;
    const int ntap = (py ? 2 : 1) * (px ? 2 : 1), tap = a * (px ? 2 : 1) + b;
    const size_t blk = (size_t)cin * cout, off = (py ? 3 : 0) * blk + (px ? (py ? 2 : 1) : 0) * blk;
    dst[off + (size_t)ci * ntap * cout + (size_t)tap * cout + co] = v;
  } else dst[(size_t)co * 224 + kh * 32 + kw * 4 + ci] = v;
}

// packed weight gradient -> PyTorch layout (accumulate or overwrite).  generic: src[tap][co][ci];
// stem: src[kh][co][kw*4 + ci] (32 per row)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dw,
                                                           int cout, int cin, int ks, int stem, int accumulate) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)cout * cin * ks * ks;
  if (gid >= total) return;
  const int kw = (int)(gid % ks);
  long long t = gid / ks;
  const int kh = (int)(t % ks);
  t /= ks;
  const int ci = (int)(t % cin), co = (int)(t / cin);
  const float v = stem ? src[((size_t)kh * cout + co) * 32 + kw * 4 + ci]
                       : src[((size_t)(kh * ks + kw) * cout + co) * cin + ci];
  dw[gid] = accumulate ? dw[gid] + v : v;
}

// per-channel sum and sum of squares over M rows of an [M][C] map (C % 4 == 0): workgroup b leaves its fp64 partial sums in
// part[b][0..C) and part[b][512..512+C); bn_sum_parts32_kernel adds the rows in a fixed order (no atomics: the step is reproducible)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long M, int C,
                                                       double* __restrict__ sums) {
  const int c4 = C >> 2;               // float4 groups per row
  const int rows_per_pass = 256 / c4;  // C <= 512 -> c4 <= 128
  const int tid = threadIdx.x;
  const int g = tid % c4, rsub = tid / c4;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rsub < rows_per_pass) {
    auto add = [&](const float4& v) {
      s[0] += v.x, s[1] += v.y, s[2] += v.z, s[3] += v.w;
      q[0] += (double)v.x * v.x, q[1] += (double)v.y * v.y, q[2] += (double)v.z * v.z, q[3] += (double)v.w * v.w;
    };
    const long long step = (long long)gridDim.x * rows_per_pass;
    long long r = (long long)blockIdx.x * rows_per_pass + rsub;
    for (; r + 3 * step < M; r += 4 * step) {  // four rows in flight per thread: the pass is a pure HBM stream
      const float4 v0 = *reinterpret_cast<const float4*>(x + r * C + 4 * g);
      const float4 v1 = *reinterpret_cast<const float4*>(x + (r + step) * C + 4 * g);
      const float4 v2 = *reinterpret_cast<const float4*>(x + (r + 2 * step) * C + 4 * g);
      const float4 v3 = *reinterpret_cast<const float4*>(x + (r + 3 * step) * C + 4 * g);
      add(v0), add(v1), add(v2), add(v3);
    }
    for (; r < M; r += step) add(*reinterpret_cast<const float4*>(x + r * C + 4 * g));
  }
  __shared__ double red[2][256][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[0][tid][k] = s[k], red[1][tid][k] = q[k];
  __syncthreads();
  if (tid < c4) {
    for (int k = 0; k < 4; ++k) {
      double a = 0, b = 0;
      for (int rr = 0; rr < rows_per_pass; ++rr) a += red[0][rr * c4 + tid][k], b += red[1][rr * c4 + tid][k];
      sums[(size_t)blockIdx.x * 1024 + 4 * tid + k] = a;
      sums[(size_t)blockIdx.x * 1024 + 512 + 4 * tid + k] = b;
    }
  }
}

// partial sums -> sums[c], sums[512 + c] in a FIXED order: lane l of the channel's 32 adds blocks l, l + 32, ... in turn, then a
// shuffle tree; 8 channels per workgroup
__global__ __launch_bounds__(256) void bn_sum_parts32_kernel(const double* __restrict__ part, int nblocks, int C,
                                                             double* __restrict__ sums) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  double a = 0, b = 0;
  if (c < C)
    for (int k = l; k < nblocks; k += 32) a += part[(size_t)k * 1024 + c], b += part[(size_t)k * 1024 + 512 + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a += __shfl_down(a, o, 32), b += __shfl_down(b, o, 32);
  if (c < C && l == 0) sums[c] = a, sums[512 + c] = b;
}

// sums -> mean, rstd (biased variance, as the normalisation uses); running statistics updated with the unbiased one
__global__ void bn_finalize_kernel(const double* __restrict__ sums, long long M, int C, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = sums[c] / (double)M;
  double var = sums[512 + c] / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
    run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
  }
}

// y = (x - mean) * rstd * gamma + beta (+ resid) (ReLU)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ resid,
                                                       float* __restrict__ y, long long n4, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)((i * 4) % C);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 m = *reinterpret_cast<const float4*>(mean + c), r = *reinterpret_cast<const float4*>(rstd + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v.x - m.x) * r.x * g.x + b.x, o.y = (v.y - m.y) * r.y * g.y + b.y;
    o.z = (v.z - m.z) * r.z * g.z + b.z, o.w = (v.w - m.w) * r.w * g.w + b.w;
    if (resid) {
      const float4 rs = *reinterpret_cast<const float4*>(resid + i * 4);
      o.x += rs.x, o.y += rs.y, o.z += rs.z, o.w += rs.w;
    }
    if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    *reinterpret_cast<float4*>(y + i * 4) = o;
  }
}

// BN backward, pass 1: sums[c] = sum dy, sums[512 + c] = sum dy * xhat, with dy masked by (ymask > 0) when given
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ ymask, long long M, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ sums) {
  const int c4 = C >> 2, rows_per_pass = 256 / c4, tid = threadIdx.x, g = tid % c4, rsub = tid / c4;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rsub < rows_per_pass) {
    const float4 m = *reinterpret_cast<const float4*>(mean + 4 * g), r = *reinterpret_cast<const float4*>(rstd + 4 * g);
    auto add = [&](float4 d, const float4& v, const float4& k) {
      if (ymask) d.x = k.x > 0.f ? d.x : 0.f, d.y = k.y > 0.f ? d.y : 0.f, d.z = k.z > 0.f ? d.z : 0.f, d.w = k.w > 0.f ? d.w : 0.f;
      s[0] += d.x, s[1] += d.y, s[2] += d.z, s[3] += d.w;
      q[0] += (double)d.x * ((v.x - m.x) * r.x), q[1] += (double)d.y * ((v.y - m.y) * r.y);
      q[2] += (double)d.z * ((v.z - m.z) * r.z), q[3] += (double)d.w * ((v.w - m.w) * r.w);
    };
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
    const long long step = (long long)gridDim.x * rows_per_pass;
    long long row = (long long)blockIdx.x * rows_per_pass + rsub;
    for (; row + step < M; row += 2 * step) {  // two rows (4-6 loads) in flight per thread
      const long long o0 = row * C + 4 * g, o1 = (row + step) * C + 4 * g;
      const float4 d0 = *reinterpret_cast<const float4*>(dy + o0), d1 = *reinterpret_cast<const float4*>(dy + o1);
      const float4 v0 = *reinterpret_cast<const float4*>(x + o0), v1 = *reinterpret_cast<const float4*>(x + o1);
      const float4 k0 = ymask ? *reinterpret_cast<const float4*>(ymask + o0) : one;
      const float4 k1 = ymask ? *reinterpret_cast<const float4*>(ymask + o1) : one;
      add(d0, v0, k0), add(d1, v1, k1);
    }
    for (; row < M; row += step) {
      const long long o0 = row * C + 4 * g;
      add(*reinterpret_cast<const float4*>(dy + o0), *reinterpret_cast<const float4*>(x + o0),
          ymask ? *reinterpret_cast<const float4*>(ymask + o0) : one);
    }
  }
  __shared__ double red[2][256][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[0][tid][k] = s[k], red[1][tid][k] = q[k];
  __syncthreads();
  if (tid < c4) {
    for (int k = 0; k < 4; ++k) {
      double a = 0, b = 0;
      for (int rr = 0; rr < rows_per_pass; ++rr) a += red[0][rr * c4 + tid][k], b += red[1][rr * c4 + tid][k];
      sums[(size_t)blockIdx.x * 1024 + 4 * tid + k] = a;
      sums[(size_t)blockIdx.x * 1024 + 512 + 4 * tid + k] = b;
    }
  }
}

// BN backward, pass 2: dx = gamma * rstd * (dy - sum_dy / M - xhat * sum_dy_xhat / M); also d gamma, d beta
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ ymask, float* __restrict__ dx,
                                                           long long n4, long long M, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int accumulate) {
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const float dg = (float)sums[512 + c], db = (float)sums[c];
      dgamma[c] = accumulate ? dgamma[c] + dg : dg;
      dbeta[c] = accumulate ? dbeta[c] + db : db;
    }
  }
  const double invM = 1.0 / (double)M;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = (int)((i * 4) % C);
    float4 d = *reinterpret_cast<const float4*>(dy + i * 4);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    if (ymask) {
      const float4 k = *reinterpret_cast<const float4*>(ymask + i * 4);
      d.x = k.x > 0.f ? d.x : 0.f, d.y = k.y > 0.f ? d.y : 0.f, d.z = k.z > 0.f ? d.z : 0.f, d.w = k.w > 0.f ? d.w : 0.f;
    }
    float o[4];
    const float dd[4] = {d.x, d.y, d.z, d.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (vv[k] - mean[c + k]) * rstd[c + k];
      const float sb = (float)(sums[c + k] * invM), sg = (float)(sums[512 + c + k] * invM);
      o[k] = gamma[c + k] * rstd[c + k] * (dd[k] - sb - xh * sg);
    }
    *reinterpret_cast<float4*>(dx + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// out = (a + b) masked by (y > 0); b / y optional
__global__ __launch_bounds__(256) void add_mask_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ y, float* __restrict__ out, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = *reinterpret_cast<const float4*>(a + i * 4);
    if (b) {
      const float4 w = *reinterpret_cast<const float4*>(b + i * 4);
      v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    if (y) {
      const float4 k = *reinterpret_cast<const float4*>(y + i * 4);
      v.x = k.x > 0.f ? v.x : 0.f, v.y = k.y > 0.f ? v.y : 0.f, v.z = k.z > 0.f ? v.z : 0.f, v.w = k.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4*>(out + i * 4) = v;
  }
}

// zero-interleave: up[b][2y][2x][c] = g[b][y][x][c], every other position 0 (up is H2 x H2, H2 = 2 * H)
__global__ __launch_bounds__(256) void upsample_zero_kernel(const float* __restrict__ g, float* __restrict__ up,
                                                            long long n4, int H, int C) {
  const int c4 = C >> 2, H2 = 2 * H;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int cg = (int)(i % c4);
    long long t = i / c4;
    const int X = (int)(t % H2);
    t /= H2;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(X & 1) && !(Y & 1)) v = *reinterpret_cast<const float4*>(g + (((b * H + (Y >> 1)) * H + (X >> 1)) * C) + 4 * cg);
    *reinterpret_cast<float4*>(up + i * 4) = v;
  }
}

// 3x3/2 max-pool, pad 1, with the arg-max kept (first maximum in (dy, dx) scan order, as torch)
__global__ __launch_bounds__(256) void maxpool_idx_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          unsigned char* __restrict__ idx, long long total) {
  constexpr int HI = 112, HO = 56, C = 64;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long p = gid / C;
  const int ow = (int)(p % HO);
  p /= HO;
  const int oh = (int)(p % HO);
  const long long b = p / HO;
  float best = -INFINITY;
  int bi = 9;
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)HI) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)HI) continue;
      const float v = in[((b * HI + ih) * HI + iw) * C + c];
      if (v > best || bi == 9) best = v, bi = dy * 3 + dx;
    }
  }
  out[gid] = best;
  idx[gid] = (unsigned char)bi;
}

// max-pool backward (gather form): every input position sums the gradients of the <= 4 windows that chose it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx,
                                                          float* __restrict__ din, long long total) {
  constexpr int HI = 112, HO = 56, C = 64;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long long p = gid / C;
  const int iw = (int)(p % HI);
  p /= HI;
  const int ih = (int)(p % HI);
  const long long b = p / HI;
  float acc = 0.f;
  // windows (oh, ow) with ih = 2 oh - 1 + dy, iw = 2 ow - 1 + dx
  for (int dy = 0; dy < 3; ++dy) {
    const int t = ih + 1 - dy;
    if (t < 0 || (t & 1)) continue;
    const int oh = t >> 1;
    if (oh >= HO) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int u = iw + 1 - dx;
      if (u < 0 || (u & 1)) continue;
      const int ow = u >> 1;
      if (ow >= HO) continue;
      const long long o = ((b * HO + oh) * HO + ow) * C + c;
      if (idx[o] == dy * 3 + dx) acc += dout[o];
    }
  }
  din[gid] = acc;
}

// feats[b][c] = mean over the 49 pixels of last[b][49][512]
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ last, float* __restrict__ feats, int n) {
  const int b = blockIdx.x, t = threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  for (int p = 0; p < 49; ++p) {
    const float2 v = *reinterpret_cast<const float2*>(last + ((size_t)b * 49 + p) * 512 + 2 * t);
    s0 += v.x, s1 += v.y;
  }
  *reinterpret_cast<float2*>(feats + (size_t)b * 512 + 2 * t) = make_float2(s0 / 49.0f, s1 / 49.0f);
}

// d last[b][p][c] = dfeats[b][c] / 49 where last > 0 (the final ReLU)
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dfeats, const float* __restrict__ last,
                                                          float* __restrict__ dlast, long long total) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % 512);
  const long long b = gid / (49 * 512);
  dlast[gid] = last[gid] > 0.f ? dfeats[b * 512 + c] * (1.0f / 49.0f) : 0.f;
}

// ---------------------------------------------------------------------------------------------
// weight gradient: dWp[tap][co][ci] += sum_m dY[m][co] * X[pixel(m, tap)][ci]
// One workgroup = one 64 x 64 (co x ci) tile of one filter tap over a contiguous chunk of output pixels
// (split-K over the grid's y dimension); 4 waves = 2 x 2 MFMA tiles of 32 x 32 on v_mfma_f32_32x32x2_f32
// (A = dY[pixel][co], B = X[pixel][ci]: both operands are read along the channel axis, coalesced, no
// transpose); operands staged through LDS 32 pixels at a time; every slice stores its tile into its OWN copy of the packed
// gradient (dWp + slice * per_slice) and wgrad_reduce32_kernel adds the slices in a fixed order: no atomics.
// STEM form: X is the padded NHWC4 input, the "ci" axis of a tile is the 32 floats (kw, c) of filter row kh.
// ---------------------------------------------------------------------------------------------
template <bool STEM>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                    float* __restrict__ dWp, int Cout, int Cin, int KS, int stride,
                                                    int HO, int HI, long long M, int chunk) {
  constexpr int LDP = 68;  // LDS row: 64 floats + 4 (keeps float4 stores aligned, spreads banks)
  __shared__ __attribute__((aligned(16))) float As[32 * LDP], Bs[32 * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ci_tiles = STEM ? 1 : Cin / 64, co_tiles = Cout / 64;
  int t = blockIdx.x;
  const int cit = t % ci_tiles;
  t /= ci_tiles;
  const int cot = t % co_tiles;
  const int tap = t / co_tiles;
  const int kh = STEM ? tap : tap / KS, kw = STEM ? 0 : tap % KS;
  const int pad = STEM ? 0 : KS / 2;
  const int wi = wave & 1, wj = wave >> 1;  // co half, ci half (STEM: ci half is the pixel half instead)
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const long long m_begin = (long long)blockIdx.y * chunk;
  const long long m_end = m_begin + chunk < M ? m_begin + chunk : M;
  const int spx = tid >> 3, sc = tid & 7;  // staging: pixel of the sub-chunk, float4 column (and + 8)
  for (long long m0 = m_begin; m0 < m_end; m0 += 32) {
    const long long m = m0 + spx;
    const bool ok = m < m_end;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, b0 = a0, b1 = a0;
    if (ok) {
      const float* ap = dY + m * Cout + cot * 64 + 4 * sc;
      a0 = *reinterpret_cast<const float4*>(ap);
      a1 = *reinterpret_cast<const float4*>(ap + 32);
      const int ox = (int)(m % HO);
      const long long tt = m / HO;
      const int oy = (int)(tt % HO);
      const long long b = tt / HO;
      if constexpr (STEM) {
        const float* bp = X + (((b * kPadH + 2 * oy + kh) * kPadW) + 2 * ox) * 4 + 4 * sc;  // 32 floats = 8 float4
        b0 = *reinterpret_cast<const float4*>(bp);
      } else {
        const int iy = oy * stride + kh - pad, ix = ox * stride + kw - pad;
        if ((unsigned)iy < (unsigned)HI && (unsigned)ix < (unsigned)HI) {
          const float* bp = X + ((b * HI + iy) * HI + ix) * (long long)Cin + cit * 64 + 4 * sc;
          b0 = *reinterpret_cast<const float4*>(bp);
          b1 = *reinterpret_cast<const float4*>(bp + 32);
        }
      }
    }
    __syncthreads();  // the previous sub-chunk's fragments have been read
    *reinterpret_cast<float4*>(As + spx * LDP + 4 * sc) = a0;
    *reinterpret_cast<float4*>(As + spx * LDP + 32 + 4 * sc) = a1;
    *reinterpret_cast<float4*>(Bs + spx * LDP + 4 * sc) = b0;
    if constexpr (!STEM) *reinterpret_cast<float4*>(Bs + spx * LDP + 32 + 4 * sc) = b1;
    __syncthreads();
    if constexpr (STEM) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // this wave's 16 pixels of the sub-chunk
        const int px = wj * 16 + 2 * k + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[px * LDP + wi * 32 + r], Bs[px * LDP + r], acc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int px = 2 * k + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[px * LDP + wi * 32 + r], Bs[px * LDP + wj * 32 + r], acc, 0, 0, 0);
      }
    }
  }
  // D[co][j]: col j = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h
  const int row_len = STEM ? 32 : Cin;
  const size_t per_slice = STEM ? (size_t)7 * 64 * 32 : (size_t)KS * KS * Cout * Cin;
  // STEM: the two waves of a channel half split the PIXELS of a sub-chunk, i.e. both hold a partial sum of the same tile:
  // each gets its own copy (slot 2 * slice + wj)
  const size_t slot = STEM ? (size_t)blockIdx.y * 2 + wj : (size_t)blockIdx.y;
  float* base = dWp + slot * per_slice + ((size_t)tap * Cout + cot * 64 + wi * 32) * row_len +
                (STEM ? 0 : cit * 64 + wj * 32) + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    base[(size_t)row * row_len] = acc[e];
  }
}

// split-K partials -> the PyTorch-layout gradient (accumulate or overwrite).  32 weights per workgroup x 8 slice groups: group g
// adds slices g, g + 8, ... in turn, then the 8 group sums are added in order -- a fixed order, hence reproducible
__global__ __launch_bounds__(256) void wgrad_reduce32_kernel(const float* __restrict__ part, int slices, float* __restrict__ dw,
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

// ---------------------------------------------------------------------------------------------
// strided fp32 GEMM on the f32 MFMA: C[m][n] (+)= sum_k A(m,k) * B(n,k) (+ bias[n]) (ReLU), A(m,k) = a[m*sam + k*sak],
// B(n,k) = b[n*sbn + k*sbk].  64 x 64 tile per workgroup, K stepped by 32 through LDS; every access is
// bounds-checked, so M, N, K are arbitrary (the projector / fc layers are tiny next to the encoder).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, long long sam, long long sak,
                                                       const float* __restrict__ b, long long sbn, long long sbk,
                                                       float* __restrict__ c, long long scm, int M, int N, int K,
                                                       const float* __restrict__ bias, int relu, int accumulate) {
  constexpr int LDP = 33;
  __shared__ float As[64 * LDP], Bs[64 * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wi = wave & 1, wj = wave >> 1;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // this thread's 8 + 8 elements of a K tile: rows row0 + 8 j, column kk; the NEXT tile is fetched into registers behind the MFMAs
  const int kk_t = tid & 31, row0 = tid >> 5;
  float ra[8], rb[8];
  auto fetch = [&](int k0) {
    const int k = k0 + kk_t;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + row0 + 8 * j, n = n0 + row0 + 8 * j;
      ra[j] = (m < M && k < K) ? a[m * sam + k * sak] : 0.f;
      rb[j] = (n < N && k < K) ? b[n * sbn + k * sbk] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) As[(row0 + 8 * j) * LDP + kk_t] = ra[j], Bs[(row0 + 8 * j) * LDP + kk_t] = rb[j];
    __syncthreads();
    if (k0 + 32 < K) fetch(k0 + 32);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(wi * 32 + r) * LDP + 2 * kk + h], Bs[(wj * 32 + r) * LDP + 2 * kk + h],
                                                 acc, 0, 0, 0);
  }
  const int n = n0 + wj * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int m = m0 + wi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (m < M && n < N) {
      float v = acc[e] + (bias ? bias[n] : 0.f);
      if (accumulate) v += c[m * scm + n];
      c[m * scm + n] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

// column sums of dy[M][N] (bias gradient), masked by (ymask > 0) when given; also writes the masked dy.  A workgroup owns 16
// columns; its 16 row groups (rows m = rg mod 16) run side by side and are added in a fixed order: one thread per column
// walking all M rows was a chain of M dependent loads on two workgroups (377 us for 2048 x 512).
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                        float* __restrict__ dym, int M, int N, float* __restrict__ db,
                                                        int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + c;
  float s = 0.f;
  if (n < N) {
    for (int m = rg; m < M; m += 16) {
      float v = dy[(size_t)m * N + n];
      if (ymask && !(ymask[(size_t)m * N + n] > 0.f)) v = 0.f;
      if (dym) dym[(size_t)m * N + n] = v;
      s += v;
    }
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && n < N && db) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    db[n] = accumulate ? db[n] + t : t;
  }
}

// weighted cross-entropy (mean reduction as torch: sum w[y] * nll / sum w[y]) and its gradient
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                 const float* __restrict__ cw, int M, int C, float* __restrict__ loss,
                                                 float* __restrict__ dlogits, float* __restrict__ scratch /* [2] */,
                                                 int phase) {
  // phase 0: per-wave (sum w nll, sum w) into scratch[2 ..]; phase 2: their sums in order -> scratch[0], scratch[1];
  // phase 1: gradient (needs scratch[1]) and the loss value
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (phase == 0) {
    float nll = 0.f, w = 0.f;
    if (m < M) {
      const float* l = logits + (size_t)m * C;
      float mx = l[0];
      for (int j = 1; j < C; ++j) mx = fmaxf(mx, l[j]);
      float se = 0.f;
      for (int j = 0; j < C; ++j) se += expf(l[j] - mx);
      const long long yl = labels[m];
      if (yl < 0 || yl >= C) {  // torch raises here; no out-of-bounds read, and the loss comes out NaN
        nll = __builtin_nanf(""), w = 0.f;
      } else {
        const int y = (int)yl;
        w = cw ? cw[y] : 1.f;
        nll = w * (logf(se) + mx - l[y]);
      }
    }
    for (int o = 32; o > 0; o >>= 1) nll += __shfl_down(nll, o, 64), w += __shfl_down(w, o, 64);
    // every workgroup's four wave sums, then the workgroups in order (phase 2 below): no atomics, a reproducible loss
    if ((threadIdx.x & 63) == 0) scratch[2 + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = nll, scratch[3 + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = w;
  } else if (phase == 2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      float a = 0.f, b = 0.f;
      const int nw = 4 * ((M + 255) / 256);
      for (int k = 0; k < nw; ++k) a += scratch[2 + 2 * k], b += scratch[3 + 2 * k];
      scratch[0] = a, scratch[1] = b;
    }
  } else if (m < M) {
    const float* l = logits + (size_t)m * C;
    float mx = l[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, l[j]);
    float se = 0.f;
    for (int j = 0; j < C; ++j) se += expf(l[j] - mx);
    const long long yl = labels[m];
    const int y = (yl < 0 || yl >= C) ? -1 : (int)yl;
    const float w = y < 0 ? __builtin_nanf("") : (cw ? cw[y] : 1.f) / scratch[1];
    for (int j = 0; j < C; ++j) dlogits[(size_t)m * C + j] = w * (expf(l[j] - mx) / se - (j == y ? 1.f : 0.f));
    if (m == 0) loss[0] = scratch[0] / scratch[1];
  }
}

// torch.optim.Adam (no weight decay, no amsgrad): in place on p, m, v
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------

// forward-kernel dispatch on the fixed layer geometries (v1 implicit-GEMM kernel, fp32, no epilogue extras)
template <int CIN, int COUT, int HI, int KS, int STRIDE>
static int conv_f32(const float* in, const float* wp, const float* zero_bias, float* out, int n, hipStream_t s) {
  ConvW w{const_cast<float*>(wp), const_cast<float*>(zero_bias)};
  return launch_conv<float, CIN, COUT, HI, HI, KS, STRIDE, false, false, false>(in, w, nullptr, out, n, s);
}
// conv i of the table on input `in`
static int conv_forward(int i, const float* in, const float* wp, const float* zb, float* out, int n, hipStream_t s) {
  const ConvDesc& d = kConvs[i];
  if (i == 0) {
    ConvW w{const_cast<float*>(wp), const_cast<float*>(zb)};
    return launch_conv<float, 4, 64, 224, 224, 7, 2, false, false, false, true>(in, w, nullptr, out, n, s);
  }
  if (d.ks == 3 && d.stride == 1) {
    switch (d.cout) {
      case 64: return conv_f32<64, 64, 56, 3, 1>(in, wp, zb, out, n, s);
      case 128: return conv_f32<128, 128, 28, 3, 1>(in, wp, zb, out, n, s);
      case 256: return conv_f32<256, 256, 14, 3, 1>(in, wp, zb, out, n, s);
      default: return conv_f32<512, 512, 7, 3, 1>(in, wp, zb, out, n, s);
    }
  }
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return conv_f32<64, 128, 56, 3, 2>(in, wp, zb, out, n, s);
      case 256: return conv_f32<128, 256, 28, 3, 2>(in, wp, zb, out, n, s);
      default: return conv_f32<256, 512, 14, 3, 2>(in, wp, zb, out, n, s);
    }
  }
  switch (d.cout) {
    case 128: return conv_f32<64, 128, 56, 1, 2>(in, wp, zb, out, n, s);
    case 256: return conv_f32<128, 256, 28, 1, 2>(in, wp, zb, out, n, s);
    default: return conv_f32<256, 512, 14, 1, 2>(in, wp, zb, out, n, s);
  }
}
// data gradient of conv i: g (gradient wrt the conv output; stride-2 layers: already zero-interleaved to hin x hin)
// -> gradient wrt the conv input, with weights packed in mode 1
static int conv_dgrad(int i, const float* g, const float* wd, const float* zb, float* out, int n, hipStream_t s) {
  const ConvDesc& d = kConvs[i];
  if (d.ks == 3 && d.stride == 1) {
    switch (d.cout) {
      case 64: return conv_f32<64, 64, 56, 3, 1>(g, wd, zb, out, n, s);
      case 128: return conv_f32<128, 128, 28, 3, 1>(g, wd, zb, out, n, s);
      case 256: return conv_f32<256, 256, 14, 3, 1>(g, wd, zb, out, n, s);
      default: return conv_f32<512, 512, 7, 3, 1>(g, wd, zb, out, n, s);
    }
  }
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return conv_f32<128, 64, 56, 3, 1>(g, wd, zb, out, n, s);
      case 256: return conv_f32<256, 128, 28, 3, 1>(g, wd, zb, out, n, s);
      default: return conv_f32<512, 256, 14, 3, 1>(g, wd, zb, out, n, s);
    }
  }
  switch (d.cout) {
    case 128: return conv_f32<128, 64, 56, 1, 1>(g, wd, zb, out, n, s);
    case 256: return conv_f32<256, 128, 28, 1, 1>(g, wd, zb, out, n, s);
    default: return conv_f32<512, 256, 14, 1, 1>(g, wd, zb, out, n, s);
  }
}

#ifndef HIPAC_F32_DGRAD_CLASSES
#define HIPAC_F32_DGRAD_CLASSES 1  // stride-2 data gradients by parity class (a quarter of the MFMAs); 0: the zero-interleaved form
#endif
// data gradient of a STRIDE-2 conv i by parity classes: g on the coarse grid, weights in mode 3 (3x3) or 1 (1x1; `out` zeroed)
static int conv_dgrad_s2(int i, const float* g, const float* wd, const float* zb, float* out, int n, hipStream_t s) {
  const ConvDesc& d = kConvs[i];
  if (d.ks == 3) {
    switch (d.cout) {
      case 128: return launch_dgrad_s2_v1<float, 128, 64, 28, true>(g, wd, zb, out, n, s);
      case 256: return launch_dgrad_s2_v1<float, 256, 128, 14, true>(g, wd, zb, out, n, s);
      default: return launch_dgrad_s2_v1<float, 512, 256, 7, true>(g, wd, zb, out, n, s);
    }
  }
  switch (d.cout) {
    case 128: return launch_dgrad_s2_v1<float, 128, 64, 28, false>(g, wd, zb, out, n, s);
    case 256: return launch_dgrad_s2_v1<float, 256, 128, 14, false>(g, wd, zb, out, n, s);
    default: return launch_dgrad_s2_v1<float, 512, 256, 7, false>(g, wd, zb, out, n, s);
  }
}

static int pack_weights(const float* w, float* dst, int i, int mode, hipStream_t s) {
  const ConvDesc& d = kConvs[i];
  const long long total = (long long)conv_w_floats(i);
  if (mode == 2) HIPAC_CHECK_HIP(hipMemsetAsync(dst, 0, packed_w_floats(0) * 4, s));
  hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, dst, d.cout, d.cin, d.ks, mode);
  return (int)hipGetLastError();
}

struct BnCtx {
  const float* params;  // flat parameter buffer
  float* stats;         // running statistics (may be null: not updated)
  char* ws;
  const TrainPlan* p;
  float eps, momentum;
  hipStream_t s;
};

// batch-norm (training statistics) of conv i's output, optional residual and ReLU
static int bn_forward(const BnCtx& c, int i, int n, const float* resid, int relu) {
  const ConvDesc& d = kConvs[i];
  const long long M = (long long)n * d.hout * d.hout;
  const float* x = (const float*)(c.ws + c.p->pre[i]);
  float* y = (float*)(c.ws + c.p->post[i]);
  double* sums = (double*)(c.ws + c.p->sums);
  float* mean = (float*)(c.ws + c.p->mean_rstd) + stat_offset(i);
  float* rstd = mean + d.cout;
  const float* gamma = c.params + param_offset(i) + conv_w_floats(i);
  double* part = (double*)(c.ws + c.p->red);
  const int rows_per_pass = 256 / (d.cout / 4);
  long long gs = (M + rows_per_pass - 1) / rows_per_pass;
  if (gs > kRedBlocks32) gs = kRedBlocks32;  // 2 workgroups per CU
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)gs), dim3(256), 0, c.s, x, M, d.cout, part);
  hipLaunchKernelGGL(bn_sum_parts32_kernel, dim3((d.cout + 7) / 8), dim3(256), 0, c.s, (const double*)part, (int)gs, d.cout, sums);
  float* rm = c.stats ? c.stats + stat_offset(i) : nullptr;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((d.cout + 255) / 256), dim3(256), 0, c.s, (const double*)sums, M, d.cout, c.eps,
                     c.momentum, mean, rstd, rm, rm ? rm + d.cout : nullptr);
  const long long n4 = M * d.cout / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(n4)), dim3(256), 0, c.s, x, resid, y, n4, d.cout, (const float*)mean,
                     (const float*)rstd, gamma, gamma + d.cout, relu);
  return (int)hipGetLastError();
}

// BN backward of conv i: dy (masked by ymask > 0 if given) -> dx (may alias dy), d gamma / d beta into grads
static int bn_backward(const BnCtx& c, int i, int n, const float* dy, const float* ymask, float* dx, float* grads,
                       int accumulate) {
  const ConvDesc& d = kConvs[i];
  const long long M = (long long)n * d.hout * d.hout;
  const float* x = (const float*)(c.ws + c.p->pre[i]);
  double* sums = (double*)(c.ws + c.p->sums);
  const float* mean = (const float*)(c.ws + c.p->mean_rstd) + stat_offset(i);
  const float* rstd = mean + d.cout;
  const float* gamma = c.params + param_offset(i) + conv_w_floats(i);
  float* dgamma = grads + param_offset(i) + conv_w_floats(i);
  double* part = (double*)(c.ws + c.p->red);
  const int rows_per_pass = 256 / (d.cout / 4);
  long long gs = (M + rows_per_pass - 1) / rows_per_pass;
  if (gs > kRedBlocks32) gs = kRedBlocks32;  // 2 workgroups per CU
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)gs), dim3(256), 0, c.s, dy, x, ymask, M, d.cout, mean, rstd, part);
  hipLaunchKernelGGL(bn_sum_parts32_kernel, dim3((d.cout + 7) / 8), dim3(256), 0, c.s, (const double*)part, (int)gs, d.cout, sums);
  const long long n4 = M * d.cout / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(n4)), dim3(256), 0, c.s, dy, x, ymask, dx, n4, M, d.cout, mean, rstd,
                     gamma, (const double*)sums, dgamma, dgamma + d.cout, accumulate);
  return (int)hipGetLastError();
}

// weight gradient of conv i: X = the conv's input map, dY = gradient wrt its output -> grads (PyTorch layout)
static int conv_wgrad(const BnCtx& c, int i, int n, const float* X, const float* dY, float* grads, int accumulate) {
  const ConvDesc& d = kConvs[i];
  float* dwp = (float*)(c.ws + c.p->wgrad_p);
  const long long M = (long long)n * d.hout * d.hout;
  const bool stem = i == 0;
  const int tiles = stem ? 7 : d.ks * d.ks * (d.cout / 64) * (d.cin / 64);
  long long slices, chunk;
  wgrad_split(i, M, slices, chunk);  // the workspace plan sized dwp for exactly this split
  if (stem)
    hipLaunchKernelGGL((wgrad_kernel<true>), dim3(tiles, (unsigned)slices), dim3(256), 0, c.s, dY, X, dwp, 64, 3, 7, 2, 112, 224,
                       M, (int)chunk);
  else
    hipLaunchKernelGGL((wgrad_kernel<false>), dim3(tiles, (unsigned)slices), dim3(256), 0, c.s, dY, X, dwp, d.cout, d.cin, d.ks,
                       d.stride, d.hout, d.hin, M, (int)chunk);
  const long long total = (long long)conv_w_floats(i);
  hipLaunchKernelGGL(wgrad_reduce32_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, c.s, (const float*)dwp,
                     (int)(stem ? 2 * slices : slices),
                     grads + param_offset(i), d.cout, d.cin, d.ks, stem ? 1 : 0, accumulate);
  return (int)hipGetLastError();
}

}  // namespace hipac

using namespace hipac;

extern "C" {

int hipac_train_num_convs(void) { return kNumConvs; }

int hipac_train_conv_desc(int i, int* cout, int* cin, int* ks, int* stride, int64_t* param_off, int64_t* stat_off) {
  HIPAC_REQUIRE(i >= 0 && i < kNumConvs, HIPAC_EINVAL, "train_conv_desc: index %d", i);
  if (cout) *cout = kConvs[i].cout;
  if (cin) *cin = kConvs[i].cin;
  if (ks) *ks = kConvs[i].ks;
  if (stride) *stride = kConvs[i].stride;
  if (param_off) *param_off = (int64_t)param_offset(i);
  if (stat_off) *stat_off = (int64_t)stat_offset(i);
  return 0;
}
size_t hipac_train_param_floats(void) { return param_offset(kNumConvs); }
size_t hipac_train_stat_floats(void) { return stat_offset(kNumConvs); }
size_t hipac_train_workspace_bytes(int batch) { return batch > 0 ? make_train_plan(batch).total : 0; }

// Test tap: byte offset inside the workspace of a map the forward keeps (kind 0: conv output before BN, 1: after BN
// (+ residual) (+ ReLU), both NHWC float32 [batch][H][W][Cout]; 2: the pooled stem map [batch][56][56][64];
// 3: batch mean[Cout] then rstd[Cout] of conv `conv`; 4: the pool's arg-max bytes [batch][56][56][64], 0..8 = dy * 3 + dx).
// Returns -1 on a bad argument.
int64_t hipac_train_debug_offset(int batch, int kind, int conv) {
  if (batch <= 0 || conv < 0 || conv >= kNumConvs) return -1;
  const TrainPlan p = make_train_plan(batch);
  switch (kind) {
    case 0: return (int64_t)p.pre[conv];
    case 1: return (int64_t)p.post[conv];
    case 2: return (int64_t)p.pool;
    case 3: return (int64_t)(p.mean_rstd + stat_offset(conv) * 4);
    case 4: return (int64_t)p.pool_idx;
    default: return -1;
  }
}

int hipac_train_encoder_forward(const float* params, float* stats, const float* x, int batch, float momentum, float eps,
                                float* feats, void* workspace, size_t workspace_bytes, void* stream) {
  HIPAC_REQUIRE(params && x && feats && workspace, HIPAC_EINVAL, "train_forward: null argument");
  HIPAC_REQUIRE(batch > 0 && batch <= 4096, HIPAC_EINVAL, "train_forward: batch %d (1 .. 4096: 32-bit pixel offsets)", batch);
  const TrainPlan p = make_train_plan(batch);
  HIPAC_REQUIRE(workspace_bytes >= p.total, HIPAC_EWORKSPACE, "train_forward: workspace %zu < required %zu", workspace_bytes,
                p.total);
  HIPAC_REQUIRE(((uintptr_t)workspace & 255) == 0, HIPAC_EINVAL, "train_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int n = batch;
  float* zb = (float*)(ws + p.zero_bias);
  HIPAC_CHECK_HIP(hipMemsetAsync(zb, 0, 512 * 4, s));
  float* wpack = (float*)(ws + p.wpack);
  for (int i = 0; i < kNumConvs; ++i) {
    int rc = pack_weights(params + param_offset(i), wpack + wpack_offset(i), i, i == 0 ? 2 : 0, s);
    HIPAC_REQUIRE(rc == 0, rc, "train_forward: weight pack launch failed (%d)", rc);
  }
  int rc = launch_nchw_to_nhwc4(x, ws + p.xin, n, HIPAC_PREC_FP32, s);
  HIPAC_REQUIRE(rc == 0, rc, "train_forward: input conversion failed (%d)", rc);
  BnCtx c{params, stats, ws, &p, eps, momentum, s};
  auto pre = [&](int i) { return (float*)(ws + p.pre[i]); };
  auto post = [&](int i) { return (float*)(ws + p.post[i]); };
#define TRY(e)                                                                      \
  do {                                                                              \
    int rc__ = (e);                                                                 \
    HIPAC_REQUIRE(rc__ == 0, rc__, "train: launch failed (%d) at line %d", rc__, __LINE__); \
  } while (0)
  // stem
  TRY(conv_forward(0, (const float*)(ws + p.xin), wpack, zb, pre(0), n, s));
  TRY(bn_forward(c, 0, n, nullptr, 1));
  {
    const long long total = (long long)n * 56 * 56 * 64;
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)post(0),
                       (float*)(ws + p.pool), (unsigned char*)(ws + p.pool_idx), total);
    TRY((int)hipGetLastError());
  }
  const float* cur = (const float*)(ws + p.pool);
  int i = 1;
  for (int stage = 0; stage < 4; ++stage) {
    for (int blk = 0; blk < 2; ++blk) {
      const bool down = stage > 0 && blk == 0;
      const int c1 = i, c2 = i + 1, ds = down ? i + 2 : -1;
      TRY(conv_forward(c1, cur, wpack + wpack_offset(c1), zb, pre(c1), n, s));
      TRY(bn_forward(c, c1, n, nullptr, 1));
      const float* idt = cur;
      if (down) {
        TRY(conv_forward(ds, cur, wpack + wpack_offset(ds), zb, pre(ds), n, s));
        TRY(bn_forward(c, ds, n, nullptr, 0));
        idt = post(ds);
      }
      TRY(conv_forward(c2, post(c1), wpack + wpack_offset(c2), zb, pre(c2), n, s));
      TRY(bn_forward(c, c2, n, idt, 1));
      cur = post(c2);
      i += down ? 3 : 2;
    }
  }
  hipLaunchKernelGGL(avgpool_kernel, dim3(n), dim3(256), 0, s, cur, feats, n);
  TRY((int)hipGetLastError());
  return 0;
}

int hipac_train_encoder_backward(const float* params, const float* dfeats, int batch, float* grads, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  HIPAC_REQUIRE(params && dfeats && grads && workspace, HIPAC_EINVAL, "train_backward: null argument");
  HIPAC_REQUIRE(batch > 0 && batch <= 4096, HIPAC_EINVAL, "train_backward: batch %d", batch);
  const TrainPlan p = make_train_plan(batch);
  HIPAC_REQUIRE(workspace_bytes >= p.total, HIPAC_EWORKSPACE, "train_backward: workspace %zu < required %zu", workspace_bytes,
                p.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int n = batch;
  const float* zb = (const float*)(ws + p.zero_bias);
  float* wd = (float*)(ws + p.wpack_d);
  BnCtx c{params, nullptr, ws, &p, 0.f, 0.f, s};
  auto post = [&](int i) { return (float*)(ws + p.post[i]); };
  float* gA = (float*)(ws + p.g[0]);  // gradient wrt the current block's output (after its ReLU mask)
  float* gB = (float*)(ws + p.g[1]);
  float* gC = (float*)(ws + p.g[2]);
  float* up = (float*)(ws + p.up);
  // global average pool + the last block's ReLU
  {
    const long long total = (long long)n * 49 * 512;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dfeats, (const float*)post(19),
                       gA, total);
    TRY((int)hipGetLastError());
  }
  // blocks in reverse.  conv indices of block (stage, blk): see kConvs
  static const int kFirst[4][2] = {{1, 3}, {5, 8}, {10, 13}, {15, 18}};
  for (int stage = 3; stage >= 0; --stage) {
    for (int blk = 1; blk >= 0; --blk) {
      const bool down = stage > 0 && blk == 0;
      const int c1 = kFirst[stage][blk], c2 = c1 + 1, ds = down ? c1 + 2 : -1;
      // input of the block = output of the previous block (the pooled map for the very first); that map is
      // also the ReLU mask of the gradient handed to the previous block (nothing to mask after the pool)
      const float* xin_blk;
      const float* prev_post;
      if (stage == 0 && blk == 0) xin_blk = (const float*)(ws + p.pool), prev_post = nullptr;
      else {
        const int pc2 = (blk == 1 ? kFirst[stage][0] : kFirst[stage - 1][1]) + 1;  // conv2 of the previous block
        xin_blk = post(pc2), prev_post = xin_blk;
      }
      const ConvDesc& d1 = kConvs[c1];
      const long long n_in4 = (long long)n * d1.hin * d1.hin * d1.cin / 4;
      // --- main path: bn2 -> conv2 -> (ReLU) bn1 -> conv1
      TRY(bn_backward(c, c2, n, gA, nullptr, gB, grads, accumulate));                 // gB = d pre(c2)
      TRY(conv_wgrad(c, c2, n, post(c1), gB, grads, accumulate));
      TRY(pack_weights(params + param_offset(c2), wd, c2, 1, s));
      TRY(conv_dgrad(c2, gB, wd, zb, gC, n, s));                                       // gC = d post(c1) (before its ReLU mask)
      TRY(bn_backward(c, c1, n, gC, post(c1), gC, grads, accumulate));                 // gC = d pre(c1)
      TRY(conv_wgrad(c, c1, n, xin_blk, gC, grads, accumulate));
      if (d1.stride == 2 && HIPAC_F32_DGRAD_CLASSES) {
        TRY(pack_weights(params + param_offset(c1), wd, c1, 3, s));
        TRY(conv_dgrad_s2(c1, gC, wd, zb, gB, n, s));                                  // gB = d block input via the main path
      } else {
        TRY(pack_weights(params + param_offset(c1), wd, c1, 1, s));
        const float* g1 = gC;
        if (d1.stride == 2) {
          const long long nu4 = (long long)n * d1.hin * d1.hin * d1.cout / 4;
          hipLaunchKernelGGL(upsample_zero_kernel, dim3(grid_for(nu4)), dim3(256), 0, s, (const float*)gC, up, nu4, d1.hout, d1.cout);
          TRY((int)hipGetLastError());
          g1 = up;
        }
        TRY(conv_dgrad(c1, g1, wd, zb, gB, n, s));                                     // gB = d block input via the main path
      }
      // --- identity path
      if (down) {
        TRY(bn_backward(c, ds, n, gA, nullptr, gC, grads, accumulate));                // gC = d pre(ds)
        TRY(conv_wgrad(c, ds, n, xin_blk, gC, grads, accumulate));
        TRY(pack_weights(params + param_offset(ds), wd, ds, 1, s));
        if (HIPAC_F32_DGRAD_CLASSES) {
          // 1x1 / stride 2: only the even positions of the fine grid receive a gradient; `up` takes it (gC holds the input)
          HIPAC_CHECK_HIP(hipMemsetAsync(up, 0, (size_t)n * d1.hin * d1.hin * kConvs[ds].cin * 4, s));
          TRY(conv_dgrad_s2(ds, gC, wd, zb, up, n, s));
          hipLaunchKernelGGL(add_mask_kernel, dim3(grid_for(n_in4)), dim3(256), 0, s, (const float*)gB, (const float*)up, prev_post,
                             gA, n_in4);
          TRY((int)hipGetLastError());
          continue;
        }
        const long long nu4 = (long long)n * d1.hin * d1.hin * kConvs[ds].cout / 4;
        hipLaunchKernelGGL(upsample_zero_kernel, dim3(grid_for(nu4)), dim3(256), 0, s, (const float*)gC, up, nu4, kConvs[ds].hout,
                           kConvs[ds].cout);
        TRY((int)hipGetLastError());
        TRY(conv_dgrad(ds, up, wd, zb, gC, n, s));                                     // gC = d block input via the projection
        hipLaunchKernelGGL(add_mask_kernel, dim3(grid_for(n_in4)), dim3(256), 0, s, (const float*)gB, (const float*)gC, prev_post,
                           gA, n_in4);
      } else {
        hipLaunchKernelGGL(add_mask_kernel, dim3(grid_for(n_in4)), dim3(256), 0, s, (const float*)gB, (const float*)gA, prev_post,
                           gA, n_in4);
      }
      TRY((int)hipGetLastError());
    }
  }
  // max-pool, stem BN (+ ReLU mask), stem weight gradient
  {
    const long long total = (long long)n * 112 * 112 * 64;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)gA,
                       (const unsigned char*)(ws + p.pool_idx), gB, total);
    TRY((int)hipGetLastError());
  }
  TRY(bn_backward(c, 0, n, gB, post(0), gB, grads, accumulate));
  TRY(conv_wgrad(c, 0, n, (const float*)(ws + p.xin), gB, grads, accumulate));
  return 0;
}

}  // extern "C"
namespace hipac {
// the strided f32-MFMA GEMM for the other translation units (ntxent.hip): C[m][n] = sum_k A(m,k) B(n,k)
int launch_gemm_f32(const float* a, long long sam, long long sak, const float* b, long long sbn, long long sbk, float* c, long long ldc,
                    int M, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((M + 63) / 64, (N + 63) / 64), dim3(256), 0, s, a, sam, sak, b, sbn, sbk, c, ldc, M, N, K,
                     (const float*)nullptr, 0, 0);
  return (int)hipGetLastError();
}
}  // namespace hipac
extern "C" {

// y[M][N] = x[M][K] w[N][K]^T + b (ReLU)    -- nn.Linear forward (src/models/simclr.py:20-24 projector, resnet.py:66 fc)
int hipac_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, void* stream) {
  HIPAC_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, HIPAC_EINVAL, "linear_forward: bad argument");
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((M + 63) / 64, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, (long long)K, 1LL,
                     w, (long long)K, 1LL, y, (long long)N, M, N, K, b, relu, 0);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

// backward of y = relu?(x w^T + b): dy masked by (y > 0) when y is given (dym: scratch [M][N], required then);
// dx[M][K] = dy w (or NULL), dw[N][K] (+)= dy^T x, db[N] (+)= column sums
int hipac_linear_backward(const float* x, const float* w, const float* dy, const float* y, float* dym, float* dx, float* dw,
                          float* db, int M, int N, int K, int accumulate, void* stream) {
  HIPAC_REQUIRE(x && w && dy && dw && M > 0 && N > 0 && K > 0, HIPAC_EINVAL, "linear_backward: bad argument");
  HIPAC_REQUIRE(!y || dym, HIPAC_EINVAL, "linear_backward: a ReLU mask needs the dym scratch");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bias_grad_kernel, dim3((N + 15) / 16), dim3(256), 0, s, dy, y, y ? dym : nullptr, M, N, db, accumulate);
  const float* g = y ? dym : dy;
  if (dx)  // dx[m][k] = sum_n g[m][n] w[n][k]:  A(m, n) = g, B(k, n) = w[n][k]
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((M + 63) / 64, (K + 63) / 64), dim3(256), 0, s, g, (long long)N, 1LL, w, 1LL,
                       (long long)K, dx, (long long)K, M, K, N, (const float*)nullptr, 0, 0);
  // dw[n][k] = sum_m g[m][n] x[m][k]:  A(n, m) = g[m][n], B(k, m) = x[m][k]
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 63) / 64, (K + 63) / 64), dim3(256), 0, s, g, 1LL, (long long)N, x, 1LL,
                     (long long)K, dw, (long long)K, N, K, M, (const float*)nullptr, 0, accumulate);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

// nn.CrossEntropyLoss(weight=class_w) value and gradient (src/main.py:490, :552-566); scratch: float[2 + 8 * ceil(M / 256)]
int hipac_cross_entropy_fwd_bwd(const float* logits, const int64_t* labels, const float* class_w, int M, int C, float* loss,
                                float* dlogits, float* scratch, void* stream) {
  HIPAC_REQUIRE(logits && labels && loss && dlogits && scratch && M > 0 && C > 0 && C <= 64, HIPAC_EINVAL,
                "cross_entropy: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_kernel, dim3((M + 255) / 256), dim3(256), 0, s, logits, (const long long*)labels, class_w, M, C, loss,
                     dlogits, scratch, 0);
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(64), 0, s, logits, (const long long*)labels, class_w, M, C, loss, dlogits, scratch, 2);
  hipLaunchKernelGGL(ce_kernel, dim3((M + 255) / 256), dim3(256), 0, s, logits, (const long long*)labels, class_w, M, C, loss,
                     dlogits, scratch, 1);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

// torch.optim.Adam step t (1-based) on a flat buffer (src/main.py:492, src/models/simclr.py:79)
int hipac_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, int step, void* stream) {
  HIPAC_REQUIRE(params && grads && m && v && n > 0 && step >= 1, HIPAC_EINVAL, "adam: bad argument");
  // bias corrections in double, as torch computes them with Python floats (1 - 0.999f in fp32 is 1.3e-5 off at step 1)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, (long long)n, lr,
                     beta1, beta2, eps, bc1, bc2);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
