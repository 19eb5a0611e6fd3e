// Implicit-GEMM convolution on MFMA for the ResNet18 trunk (gfx950).
//
// Data layout: activations NHWC in T (bf16 | fp16), weights [Cout][kh][kw][Cin]
// with BatchNorm folded in, fp32 bias per Cout.  The GEMM is
//     D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],   k = (kh, kw, cin)
// with the WEIGHTS as the MFMA "A" operand and the ACTIVATIONS as "B", so that
// in the 32x32 accumulator tile each lane owns one output pixel (lane & 31) and
// runs of 4 consecutive output channels in its registers: the NHWC store is then
// 8 bytes of consecutive channels per lane instead of 2-byte scalars.
//
// Tile: 128 output pixels x BN output channels per 256-thread workgroup
// (4 waves as 2 pixel-halves x 2 channel-halves), K stepped in tiles of one
// filter tap x 64 input channels (one 128-byte pixel run per row), staged through
// LDS with 16-byte pads (row stride 144 B => ds_read_b128 conflict-free) and
// register prefetch of the next K tile behind the MFMAs of the current one.
//
// The 7x7/2 stem reads the pre-padded NHWC4 image: for a fixed kh the 7 taps x 4
// channels of one output pixel are 56 contiguous bytes, so K = 7 tiles of 32
// (28 real + 4 zero-weight) elements and no bounds checks are needed.
#pragma once
#include <type_traits>
#include "common.h"

namespace hipac {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

template <typename T, int CIN, int COUT, int HI, int WI, int KS, int STRIDE, int BN, bool RELU,
          bool RESID, bool OUTF32, bool STEM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const T* __restrict__ in,
                                                         const T* __restrict__ wgt,
                                                         const float* __restrict__ bias,
                                                         const T* __restrict__ resid,
                                                         void* __restrict__ outp, int M) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int PAD = STEM ? 0 : KS / 2;
  constexpr int HO = STEM ? 112 : (HI + 2 * PAD - KS) / STRIDE + 1;
  constexpr int WO = STEM ? 112 : (WI + 2 * PAD - KS) / STRIDE + 1;
  constexpr int BK = STEM ? 32 : 64;
  constexpr int KT = STEM ? 7 : KS * KS * (CIN / 64);
  constexpr int KTOT = KT * BK;
  constexpr int LDA = BK + 8;  // padded LDS row, elements
  constexpr int CH = BK / 8;   // 16-byte chunks per row
  constexpr int BM = 128;
  constexpr int RPP = 256 / CH;       // rows covered per pass of the 256 threads
  constexpr int APT = BM / RPP;       // A pieces per thread
  constexpr int WPT = BN / RPP;       // W pieces per thread
  constexpr int NT = BN / 64;         // 32-wide cout tiles per wave
  constexpr int CC = STEM ? 1 : CIN / 64;
  static_assert(BN % 64 == 0 && COUT % BN == 0, "BN");
  static_assert(WPT >= 1, "WPT");

  __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LDA];
  T* As = smem;
  T* Ws = smem + BM * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread staging rows -------------------------------------------------
  const int chunk = tid % CH;
  const int row0 = tid / CH;
  int a_base[APT];   // element offset of tap (0,0) for this row (+chunk*8)
  int a_ih0[APT], a_iw0[APT];
  bool a_ok[APT];
#pragma unroll
  for (int i = 0; i < APT; ++i) {
    const int m = m0 + row0 + i * RPP;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    const int b = mm / (HO * WO);
    const int rem = mm - b * (HO * WO);
    const int oh = rem / WO;
    const int ow = rem - oh * WO;
    if constexpr (STEM) {
      a_ih0[i] = 0;
      a_iw0[i] = 0;
      a_base[i] = ((b * kPadH + 2 * oh) * kPadW + 2 * ow) * 4 + chunk * 8;
    } else {
      a_ih0[i] = oh * STRIDE - PAD;
      a_iw0[i] = ow * STRIDE - PAD;
      a_base[i] = ((b * HI + a_ih0[i]) * WI + a_iw0[i]) * CIN + chunk * 8;
    }
  }
  const T* wsrc = wgt + (size_t)(n0 + row0) * KTOT + chunk * 8;

  u32x4 areg[APT], wreg[WPT];  // native vectors: HIP's uint4 struct copies lower to memcpy and land in scratch
  // Branch-free staging: out-of-image taps load from offset 0 and are zeroed by
  // a select; the prefetch past the last K tile wraps to tile 0 (loaded, never used).
  auto gload = [&](int kh_, int kw_, int cc_, int t_) {
    static_for<APT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      bool ok = a_ok[i];
      int off;
      if constexpr (STEM) {
        off = a_base[i] + kh_ * (kPadW * 4);
      } else {
        ok = ok && (unsigned)(a_ih0[i] + kh_) < (unsigned)HI && (unsigned)(a_iw0[i] + kw_) < (unsigned)WI;
        off = a_base[i] + (kh_ * WI + kw_) * CIN + cc_ * 64;
      }
      const u32x4 v = *reinterpret_cast<const u32x4*>(in + (ok ? off : 0));
      areg[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    });
    static_for<WPT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      wreg[i] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)i * RPP * KTOT + t_ * BK);
    });
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const T* a_rd = As + (wm * 64 + r) * LDA + 8 * h;
  const T* w_rd = Ws + (wn * (BN / 2) + r) * LDA + 8 * h;

  int kh = 0, kw = 0, cc = 0;
  gload(0, 0, 0, 0);
  for (int t = 0; t < KT; ++t) {
    static_for<APT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      *reinterpret_cast<u32x4*>(As + (row0 + i * RPP) * LDA + chunk * 8) = areg[i];
    });
    static_for<WPT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      *reinterpret_cast<u32x4*>(Ws + (row0 + i * RPP) * LDA + chunk * 8) = wreg[i];
    });
    __syncthreads();
    // advance (kh, kw, cc) to tile t+1 and prefetch it behind the MFMAs
    if (++cc == CC) {
      cc = 0;
      if constexpr (STEM) {
        ++kh;
      } else if (++kw == KS) {
        kw = 0;
        ++kh;
      }
    }
    {
      const bool more = t + 1 < KT;
      const int kh_n = more ? kh : 0, kw_n = more ? kw : 0, cc_n = more ? cc : 0, t_n = more ? t + 1 : 0;
      gload(kh_n, kw_n, cc_n, t_n);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      frag af[2], wf[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const frag*>(a_rd + i * 32 * LDA + kk * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const frag*>(w_rd + j * 32 * LDA + kk * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[j], af[i], acc[i][j]);
    }
    __syncthreads();
  }

  // ---- epilogue: +bias (+residual) (ReLU) -> NHWC store ---------------------------
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + r;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = n0 + wn * (BN / 2) + j * 32 + 8 * q + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
        float v0 = acc[i][j][4 * q + 0] + bv.x;
        float v1 = acc[i][j][4 * q + 1] + bv.y;
        float v2 = acc[i][j][4 * q + 2] + bv.z;
        float v3 = acc[i][j][4 * q + 3] + bv.w;
        const size_t o = (size_t)m * COUT + c0;
        if constexpr (RESID) {
          const typename E::vec4 rv = *reinterpret_cast<const typename E::vec4*>(resid + o);
          v0 += (float)rv[0];
          v1 += (float)rv[1];
          v2 += (float)rv[2];
          v3 += (float)rv[3];
        }
        if constexpr (RELU) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
          v2 = fmaxf(v2, 0.f);
          v3 = fmaxf(v3, 0.f);
        }
        if constexpr (OUTF32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + o) = make_float4(v0, v1, v2, v3);
        } else {
          typename E::vec4 ov;
          ov[0] = (T)v0;
          ov[1] = (T)v1;
          ov[2] = (T)v2;
          ov[3] = (T)v3;
          *reinterpret_cast<typename E::vec4*>(reinterpret_cast<T*>(outp) + o) = ov;
        }
      }
    }
  }
}

// 3x3/2 max-pool, pad 1, NHWC, 8 channels (16 B) per thread.  Inputs are
// post-ReLU (>= 0) so the implicit -inf padding never wins; out-of-range taps
// are simply skipped.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                           int n) {
  constexpr int HI = 112, WI = 112, HO = 56, WO = 56, C = 64;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * HO * WO * (C / 8);
  if (gid >= total) return;
  const int c8 = (int)(gid % (C / 8));
  long long p = gid / (C / 8);
  const int ow = (int)(p % WO);
  p /= WO;
  const int oh = (int)(p % HO);
  const int b = (int)(p / HO);
  using frag = typename Elem<T>::frag;
  float best[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = -3.0e38f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if ((unsigned)ih >= (unsigned)HI) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if ((unsigned)iw >= (unsigned)WI) continue;
      const frag v = *reinterpret_cast<const frag*>(in + (((size_t)b * HI + ih) * WI + iw) * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], (float)v[e]);
    }
  }
  frag o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (T)best[e];
  *reinterpret_cast<frag*>(out + (((size_t)b * HO + oh) * WO + ow) * C + c8 * 8) = o;
}

template <typename T, int CIN, int COUT, int HI, int WI, int KS, int STRIDE, bool RELU, bool RESID,
          bool OUTF32, bool STEM = false>
static int launch_conv(const void* in, const ConvW& w, const void* resid, void* out, int n, hipStream_t s) {
  constexpr int BN = COUT >= 128 ? 128 : 64;
  constexpr int PAD = STEM ? 0 : KS / 2;
  constexpr int HO = STEM ? 112 : (HI + 2 * PAD - KS) / STRIDE + 1;
  constexpr int WO = STEM ? 112 : (WI + 2 * PAD - KS) / STRIDE + 1;
  const int M = n * HO * WO;
  dim3 grid((M + 127) / 128, COUT / BN);
  hipLaunchKernelGGL((conv_igemm_kernel<T, CIN, COUT, HI, WI, KS, STRIDE, BN, RELU, RESID, OUTF32, STEM>),
                     grid, dim3(256), 0, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out, M);
  return (int)hipGetLastError();
}

#define HIPAC_TRY(expr)          \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != 0) {             \
      ::hipac::set_error("kernel launch failed (%d) at %s:%d", rc__, __FILE__, __LINE__); \
      return rc__;               \
    }                            \
  } while (0)

// The trunk is a fixed sequence of 21 launches ("ops"): 0 stem, 1 max-pool, then per
// stage conv1(b0) [proj] conv2(b0) conv1(b1) conv2(b1).  `first..last` selects a
// sub-range (whole trunk by default) so single layers can be timed / profiled.
struct OpRange {
  int first, last, next;
  bool take() {
    const int i = next++;
    return i >= first && i <= last;
  }
};

// One ResNet stage = two BasicBlocks.  CI/HI: input channels / spatial size,
// CO/HO: output.  STRIDE 2 stages carry the 1x1/2 projection shortcut.
template <typename T, int CI, int CO, int HI, int STRIDE, bool LAST>
static int run_stage(const Net& net, int stage, const void* x, char* ws, const Plan& p, int bc, hipStream_t s,
                     OpRange& ops) {
  constexpr int HO = HI / STRIDE;
  void* tmp = ws + p.tmp;
  void* ds = ws + p.ds;
  void* o0 = ws + p.blk[2 * stage];
  void* o1 = ws + p.blk[2 * stage + 1];
  const ConvW(&bw)[2] = net.block[2 * stage];
  const ConvW(&bw1)[2] = net.block[2 * stage + 1];
  // block 0
  if (ops.take())
    HIPAC_TRY((launch_conv<T, CI, CO, HI, HI, 3, STRIDE, true, false, false>(x, bw[0], nullptr, tmp, bc, s)));
  const void* idt = x;
  if constexpr (STRIDE != 1 || CI != CO) {
    if (ops.take())
      HIPAC_TRY((launch_conv<T, CI, CO, HI, HI, 1, STRIDE, false, false, false>(x, net.down[stage - 1], nullptr,
                                                                               ds, bc, s)));
    idt = ds;
  }
  if (ops.take())
    HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, true, false>(tmp, bw[1], idt, o0, bc, s)));
  // block 1
  if (ops.take())
    HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, false, false>(o0, bw1[0], nullptr, tmp, bc, s)));
  if (ops.take())
    HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, true, LAST>(tmp, bw1[1], o0, o1, bc, s)));
  return 0;
}

template <typename T>
static int run_trunk(const Net& net, const Plan& p, char* ws, const void* xin, int bc, hipStream_t s,
                     int first, int last) {
  OpRange ops{first, last, 0};
  if (ops.take())
    HIPAC_TRY((launch_conv<T, 4, 64, 224, 224, 7, 2, true, false, false, true>(xin, net.stem, nullptr,
                                                                               ws + p.stem, bc, s)));
  if (ops.take()) {
    const long long total = (long long)bc * 56 * 56 * 8;
    hipLaunchKernelGGL((maxpool3x3s2_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const T*)(ws + p.stem), (T*)(ws + p.pool), bc);
    HIPAC_TRY((int)hipGetLastError());
  }
  HIPAC_TRY((run_stage<T, 64, 64, 56, 1, false>(net, 0, ws + p.pool, ws, p, bc, s, ops)));
  HIPAC_TRY((run_stage<T, 64, 128, 56, 2, false>(net, 1, ws + p.blk[1], ws, p, bc, s, ops)));
  HIPAC_TRY((run_stage<T, 128, 256, 28, 2, false>(net, 2, ws + p.blk[3], ws, p, bc, s, ops)));
  HIPAC_TRY((run_stage<T, 256, 512, 14, 2, true>(net, 3, ws + p.blk[5], ws, p, bc, s, ops)));
  return 0;
}

}  // namespace hipac
