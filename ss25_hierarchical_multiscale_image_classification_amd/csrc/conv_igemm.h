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

// TKH / TKW / UPS: the parity-class data gradient of a stride-2 convolution (training), see conv_glds_kernel's note: the input is the
// gradient on the coarse grid, the window TKH x TKW taps starting at the output pixel, no padding, output pixel (y, x) stored at
// (2y + PY, 2x + PX) of the fine grid, UPS = 4 | PY << 1 | PX.
template <typename T, int CIN, int COUT, int HI, int WI, int KS, int STRIDE, int BN, bool RELU,
          bool RESID, bool OUTF32, bool STEM, int TKH = 0, int TKW = 0, int UPS = 0>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const T* __restrict__ in,
                                                         const T* __restrict__ wgt,
                                                         const float* __restrict__ bias,
                                                         const T* __restrict__ resid,
                                                         void* __restrict__ outp, int M) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int KH = UPS ? TKH : KS, KW = UPS ? TKW : KS;
  constexpr int PAD = (STEM || UPS) ? 0 : KS / 2;
  constexpr int HO = STEM ? 112 : (UPS ? HI : (HI + 2 * PAD - KS) / STRIDE + 1);
  constexpr int WO = STEM ? 112 : (UPS ? WI : (WI + 2 * PAD - KS) / STRIDE + 1);
  static_assert(!UPS || (!STEM && STRIDE == 1 && TKH >= 1 && TKW >= 1 && !RESID), "parity-class data gradient");
  constexpr int BK = STEM ? 32 : 64;
  constexpr int KT = STEM ? 7 : KH * KW * (CIN / 64);
  constexpr int KTOT = KT * BK;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk: 8 (bf16/fp16) or 4 (fp32)
  constexpr int LDA = BK + EPC;             // LDS row padded by one chunk, elements
  constexpr int CH = BK / EPC;              // 16-byte chunks per row
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
      a_base[i] = ((b * kPadH + 2 * oh) * kPadW + 2 * ow) * 4 + chunk * EPC;
    } else {
      a_ih0[i] = oh * STRIDE - PAD;
      a_iw0[i] = ow * STRIDE - PAD;
      a_base[i] = ((b * HI + a_ih0[i]) * WI + a_iw0[i]) * CIN + chunk * EPC;
    }
  }
  const T* wsrc = wgt + (size_t)(n0 + row0) * KTOT + chunk * EPC;

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
      *reinterpret_cast<u32x4*>(As + (row0 + i * RPP) * LDA + chunk * EPC) = areg[i];
    });
    static_for<WPT>([&](auto I) {
      constexpr int i = decltype(I)::value;
      *reinterpret_cast<u32x4*>(Ws + (row0 + i * RPP) * LDA + chunk * EPC) = wreg[i];
    });
    __syncthreads();
    // advance (kh, kw, cc) to tile t+1 and prefetch it behind the MFMAs
    if (++cc == CC) {
      cc = 0;
      if constexpr (STEM) {
        ++kh;
      } else if (++kw == KW) {
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
        size_t o = (size_t)m * COUT + c0;
        if constexpr (UPS != 0) {  // coarse pixel m = (b, y, x) -> fine position (2y + PY, 2x + PX)
          const int ub = m / (HO * WO), urem = m - ub * (HO * WO), uy = urem / WO, ux = urem - uy * WO;
          o = ((size_t)(ub * 2 * HO + 2 * uy + ((UPS >> 1) & 1)) * (2 * WO) + 2 * ux + (UPS & 1)) * COUT + c0;
        }
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

// ---------------------------------------------------------------------------------------
// v2: LDS-DMA ring.  Same GEMM, operand roles and epilogue as above, but the K tiles
// (one filter tap x 64 input channels: 128-byte rows) are brought in by
// global_load_lds_dwordx4 (1 KiB = 8 rows per wave-instruction) into an NSTAGE-deep LDS
// ring that stays NSTAGE-1 tiles ahead of the MFMAs: counted s_waitcnt vmcnt(N), ONE raw
// s_barrier per K tile, no VGPRs spent on staging.  The LDS image is lane-linear (as the
// DMA requires), so the bank-conflict swizzle lives on the SOURCE address and on the read:
// 16-byte chunk c of row R sits at chunk c ^ ((R >> 1) & 7)  (ds_read_b128 conflict-free:
// within a 16-lane group (R & 1, (R >> 1) & 7) is unique).  Out-of-image taps read a
// 128-byte zero page.  Workgroup = (BM/64) x (BN/64) waves, each wave 64 pixels x 64
// channels (2 x 2 MFMA tiles of 32x32, 64 accumulator registers).
// Grid is 1-D and XCD-aware: the channel tiles of one pixel tile get consecutive slots
// on ONE XCD (ids congruent mod 8 share an XCD), so the shared A rows hit that XCD's L2.
// ---------------------------------------------------------------------------------------
// Buffer-descriptor LDS-DMA: 16 bytes per lane from base + voffset (VGPR, range-checked against the
// descriptor's size: out of range reads as zeros) + soffset (SGPR, not range-checked) into
// lds_base + lane * 16.  The builtins exist only in the device pass; the host pass needs the kernel
// templates to parse so that their launch stubs are emitted.
#if defined(__HIP_DEVICE_COMPILE__)
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buffer_load_lds16(rsrc_t rs, void* lds, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, 0, 0);
}
#else
struct rsrc_t {};
__device__ inline rsrc_t make_rsrc(const void*, int) { return {}; }
__device__ inline void buffer_load_lds16(rsrc_t, void*, int, int) {}
#endif

typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <typename T> struct PackPair;  // two exactly representable floats -> one dword of two T (lo, hi)
template <> struct PackPair<_Float16> {
  static __device__ __forceinline__ unsigned pack(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));  // exact inputs: the rounding mode is moot
  }
  static __device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    // round-to-nearest-even, as every other store of T in this library; one packed conversion
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, f16x2));
  }
};
template <> struct PackPair<__bf16> {
  static __device__ __forceinline__ unsigned pack(float lo, float hi) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
  }
  static __device__ __forceinline__ unsigned pack_rn(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
  }
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {  // a.upper <-> b.lower (32-lane rows)
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#else
__device__ inline void permlane32_swap(unsigned&, unsigned&) {}
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PROJ (3x3 / stride 2 layers only): the BasicBlock's 1x1 / stride 2 projection shortcut reads
// exactly the centre tap of this convolution, so it rides along as CC extra K tiles (centre-tap
// activation tile x projection weights) into a second accumulator set and leaves through a
// second epilogue (bias only, no ReLU) into `outp_p`: one launch, one pass over the input.
//
// SPLIT (precision fp16x3, the mode that meets the reference's fp32 results to 1e-3): every value is the PAIR
// hi = rn16(v), lo = rn16(v - hi) of fp16 numbers.  Activations are stored [pixel][hi: CIN | lo: CIN] (2 CIN
// elements per pixel), weights [Cout][tap][3 CIN] as per-64-channel triples (hi_c | lo_c | hi_c), and the GEMM
// runs over 3 CIN "virtual" channels: virtual chunk v = 3c + j reads activation chunk c of the hi plane (j = 0, 1)
// or of the lo plane (j = 2): D = Whi Xhi + Wlo Xhi + Whi Xlo on v_mfma_f32_32x32x16_f16 with fp32 accumulation
// (the dropped lo x lo term is 2^-22 relative).  The epilogue splits its fp32 result into a pair again.
template <bool SPLIT, int CC>
__device__ __forceinline__ constexpr int split_achunk(int v) {  // activation chunk (in units of 64 channels) of virtual chunk v
  return SPLIT ? ((v % 3 == 2) ? CC + v / 3 : v / 3) : v;
}
// fp32 -> (hi, lo) pair of fp16 fragments
__device__ __forceinline__ void split_pair8(const float* v, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)(v[e] - (float)hi[e]);
  }
}
__device__ __forceinline__ void split_pair4(const float* v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)(v[e] - (float)hi[e]);
  }
}

template <typename T, int CIN, int COUT, int HI, int WI, int KS, int STRIDE, int BM, int BN, int NSTAGE,
          bool RELU, bool RESID, bool OUTF32, bool PROJ = false, bool SPLIT = false, int WTM = 64, int TKH = 0, int TKW = 0, int UPS = 0>
__global__ __launch_bounds__((BM / WTM) * (BN / 64) * 64, 2) void conv_glds_kernel(
    const T* __restrict__ in, const T* __restrict__ wgt, const float* __restrict__ bias,
    const T* __restrict__ resid, void* __restrict__ outp, int M, int n_mtiles, const char* __restrict__ zero_page,
    const T* __restrict__ wgt_p = nullptr, const float* __restrict__ bias_p = nullptr,
    void* __restrict__ outp_p = nullptr) {
  using E = Elem<T>;
  using frag = typename E::frag;
  // UPS != 0 (training: data gradient of a stride-2 conv, one PARITY CLASS of the fine grid per launch): the input is the
  // gradient on the coarse grid, the window is TKH x TKW taps starting AT the output pixel (no padding; taps beyond the
  // bottom / right edge read zeros), the output grid equals the input grid and output pixel (y, x) is stored at fine
  // position (2y + PY, 2x + PX), UPS = 4 | PY << 1 | PX.  See launch_dgrad_s2.
  constexpr int KH = UPS ? TKH : KS, KW = UPS ? TKW : KS;
  constexpr int PAD = UPS ? 0 : KS / 2;
  constexpr int HO = UPS ? HI : (HI + 2 * PAD - KS) / STRIDE + 1;
  constexpr int WO = UPS ? WI : (WI + 2 * PAD - KS) / STRIDE + 1;
  static_assert(!UPS || (STRIDE == 1 && TKH >= 1 && TKW >= 1 && !PROJ && !RESID && !OUTF32 && !SPLIT), "parity-class data gradient");
  constexpr int RC = CIN / 64;                      // real 64-channel chunks
  constexpr int CC = SPLIT ? 3 * RC : RC;           // (virtual) chunks of the K loop
  constexpr int PIXC = SPLIT ? 2 * CIN : CIN;       // activation elements per input pixel
  constexpr int OPIX = SPLIT ? 2 * COUT : COUT;     // elements per output pixel (T outputs)
  constexpr int KT = KH * KW * CC;
  constexpr int KTOT = KT * 64;
  constexpr int KTP = PROJ ? KT + CC : KT;          // + the projection's K tiles
  static_assert(!PROJ || (KS == 3 && STRIDE == 2 && !RESID && !OUTF32), "projection rides on 3x3/2 only");
  static_assert(!SPLIT || (std::is_same<T, _Float16>::value && !RESID), "split pairs are fp16");
  static_assert(WTM == 64 || WTM == 128, "wave tile: 64 or 128 pixels x 64 channels");
  constexpr int MT = WTM / 32;                     // 32-pixel sub-tiles per wave (4: 0.75 LDS fragment reads per MFMA instead of 1)
  constexpr int WM = BM / WTM, WN = BN / 64, NWAVES = WM * WN;
  constexpr int APW = BM / 8 / NWAVES;  // 1-KiB A pieces per wave per K tile
  constexpr int WPW = BN / 8 / NWAVES;  // 1-KiB W pieces per wave per K tile
  constexpr int PPW = APW + WPW;
  constexpr int STAGE = (BM + BN) * 128;
  constexpr int NTILES_N = COUT / BN;
  static_assert(BM % WTM == 0 && BN % 64 == 0 && COUT % BN == 0 && CIN % 64 == 0, "tile shape");
  static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "piece split");
  static_assert(NSTAGE >= 2 && NSTAGE * STAGE <= 160 * 1024, "LDS ring");
  static_assert((NSTAGE - 1) * PPW < 64, "vmcnt range");

  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware decode of the 1-D grid
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / NTILES_N) * 8 + xcd;
  const int nt = slot % NTILES_N;
  if (mt >= n_mtiles) return;  // uniform per block; grid is padded to a multiple of 8 m-tiles
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- DMA source setup: lane -> (row within piece, destination chunk) -----------------
  const int prow = lane >> 3;     // row inside the 8-row piece
  const int dchunk = lane & 7;    // destination 16-byte chunk (lane-linear)
  int a_off[APW];                 // byte offset of tap (0,0), incl. the swizzled source chunk
  unsigned a_mask[APW];           // bit (kh*KW+kw): tap inside the image
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int row = (wave + NWAVES * i) * 8 + prow;  // row inside the BM tile
    const int schunk = dchunk ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int b = mm / (HO * WO);
    const int rem = mm - b * (HO * WO);
    const int oh = rem / WO, ow = rem - oh * WO;
    const int ih0 = oh * STRIDE - PAD, iw0 = ow * STRIDE - PAD;
    a_off[i] = (((b * HI + ih0) * WI + iw0) * PIXC + schunk * 8) * 2;
    unsigned mask = 0;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
        if (ok && (unsigned)(ih0 + kh) < (unsigned)HI && (unsigned)(iw0 + kw) < (unsigned)WI)
          mask |= 1u << (kh * KW + kw);
    a_mask[i] = mask;
  }
  int w_off[WPW], wp_off[PROJ ? WPW : 1];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (wave + NWAVES * i) * 8 + prow;  // row inside the BN tile
    const int schunk = dchunk ^ ((row >> 1) & 7);
    w_off[i] = ((n0 + row) * KTOT + schunk * 8) * 2;
    if constexpr (PROJ) wp_off[i] = ((n0 + row) * (CC * 64) + schunk * 8) * 2;
  }
  const char* in_b = reinterpret_cast<const char*>(in);
  const char* w_b = reinterpret_cast<const char*>(wgt);
  const char* wp_b = reinterpret_cast<const char*>(wgt_p);
  const char* zsrc = zero_page + dchunk * 16;

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  // LDS-DMA through buffer descriptors: a tap that leaves the image (or a row beyond M) gets an offset
  // past the descriptor's range and reads as zeros -- no zero-page select, no 64-bit address arithmetic
  const rsrc_t in_rsrc = make_rsrc(in_b, (M / (HO * WO)) * (HI * WI * PIXC * 2));
  const rsrc_t w_rsrc = make_rsrc(w_b, COUT * KTOT * 2);
  const rsrc_t wp_rsrc = make_rsrc(PROJ ? wp_b : w_b, COUT * CC * 64 * 2);
  auto issue = [&](int tap, int tapoff_bytes, int kofs_bytes, int stage, bool proj) {
    unsigned char* sbase = ring + stage * STAGE;
    static_for<APW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      const bool ok = (a_mask[i] >> tap) & 1u;
      buffer_load_lds16(in_rsrc, sbase + (wave + NWAVES * i) * 1024, ok ? a_off[i] + tapoff_bytes : (int)0x80000000, 0);
    });
    static_for<WPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (PROJ && proj) buffer_load_lds16(wp_rsrc, sbase + BM * 128 + (wave + NWAVES * i) * 1024, wp_off[i], kofs_bytes);
      else buffer_load_lds16(w_rsrc, sbase + BM * 128 + (wave + NWAVES * i) * 1024, w_off[i], kofs_bytes);
    });
  };

  // ---- fragment read offsets (bytes inside a stage) -------------------------------------
  const int sw = (r >> 1) & 7;
  int rd[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) rd[kk] = r * 128 + (((2 * kk + h) ^ sw) << 4);
  const int a_rd0 = wm * WTM * 128;
  const int w_rd0 = BM * 128 + wn * 64 * 128;

  f32x16 acc[MT][2], accp[PROJ ? MT : 1][PROJ ? 2 : 1];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (PROJ) accp[i][j][e] = 0.f;
      }

  // issue-side tile counters (tile index ti = (kh*KS + kw)*CC + cc; then the projection's cc tiles)
  int i_kh = 0, i_kw = 0, i_cc = 0, i_t = 0;
  auto issue_next = [&]() __attribute__((always_inline)) {
    if (PROJ && i_t >= KT) {  // centre tap (1,1), channel chunk i_t - KT, projection weights
      const int pc = i_t - KT;
      issue(4, ((WI + 1) * PIXC + split_achunk<SPLIT, RC>(pc) * 64) * 2, pc * 128, i_t % NSTAGE, true);
    } else {
      const int tap = i_kh * KW + i_kw;
      issue(tap, ((i_kh * WI + i_kw) * PIXC + split_achunk<SPLIT, RC>(i_cc) * 64) * 2, i_t * 128, i_t % NSTAGE, false);
      if (++i_cc == CC) {
        i_cc = 0;
        if (++i_kw == KW) {
          i_kw = 0;
          ++i_kh;
        }
      }
    }
    ++i_t;
  };
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (p < KTP) issue_next();

  for (int t = 0; t < KTP; ++t) {
    // tile t must have landed: tiles t+1 .. min(t+NSTAGE-2, KTP-1) may stay in flight
    const int ahead = (KTP - 1 - t) < (NSTAGE - 2) ? (KTP - 1 - t) : (NSTAGE - 2);
    if constexpr (NSTAGE >= 4) {
      if (ahead >= 2) wait_vmcnt<2 * PPW>();
      else if (ahead == 1) wait_vmcnt<PPW>();
      else wait_vmcnt<0>();
    } else if constexpr (NSTAGE == 3) {
      if (ahead >= 1) wait_vmcnt<PPW>();
      else wait_vmcnt<0>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile t are in; stage (t-1)%NSTAGE is free
    if (t + NSTAGE - 1 < KTP) issue_next();
    const unsigned char* st = ring + (t % NSTAGE) * STAGE;
    // fragment reads run one k16 step ahead of the MFMAs that consume them
    frag af[2][MT], wf[2][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const frag*>(st + a_rd0 + i * 4096 + rd[0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) wf[0][j] = *reinterpret_cast<const frag*>(st + w_rd0 + j * 4096 + rd[0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[(kk + 1) & 1][i] = *reinterpret_cast<const frag*>(st + a_rd0 + i * 4096 + rd[kk + 1]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[(kk + 1) & 1][j] = *reinterpret_cast<const frag*>(st + w_rd0 + j * 4096 + rd[kk + 1]);
      }
      if (PROJ && t >= KT) {  // uniform: the last CC tiles feed the projection's accumulators
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            accp[PROJ ? i : 0][PROJ ? j : 0] =
                E::mfma(wf[kk & 1][j], af[kk & 1][i], accp[PROJ ? i : 0][PROJ ? j : 0]);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = E::mfma(wf[kk & 1][j], af[kk & 1][i], acc[i][j]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // the reads of this stage must have retired before any wave passes the next barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- projection epilogue: + bias -> NHWC store (no ReLU, no residual) -------------------
  if constexpr (PROJ) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WTM + i * 32 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c0 = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
          const float4 bv = *reinterpret_cast<const float4*>(bias_p + c0);
          const float pv[4] = {accp[i][j][4 * q + 0] + bv.x, accp[i][j][4 * q + 1] + bv.y, accp[i][j][4 * q + 2] + bv.z,
                               accp[i][j][4 * q + 3] + bv.w};
          if constexpr (SPLIT) {
            f16x4 oh, ol;
            split_pair4(pv, oh, ol);
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp_p) + (size_t)m * OPIX + c0) = oh;
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp_p) + (size_t)m * OPIX + COUT + c0) = ol;
          } else {
            typename E::vec4 ov;
            ov[0] = (T)pv[0];
            ov[1] = (T)pv[1];
            ov[2] = (T)pv[2];
            ov[3] = (T)pv[3];
            *reinterpret_cast<typename E::vec4*>(reinterpret_cast<T*>(outp_p) + (size_t)m * COUT + c0) = ov;
          }
        }
    }
  }

  // ---- epilogue: +bias (+residual) (ReLU) -> NHWC store ---------------------------------
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * WTM + i * 32 + r;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = n0 + wn * 64 + j * 32 + 8 * q + 4 * h;
        const float4 bv = *reinterpret_cast<const float4*>(bias + c0);
        float v0 = acc[i][j][4 * q + 0] + bv.x;
        float v1 = acc[i][j][4 * q + 1] + bv.y;
        float v2 = acc[i][j][4 * q + 2] + bv.z;
        float v3 = acc[i][j][4 * q + 3] + bv.w;
        size_t o = (size_t)m * COUT + c0;
        if constexpr (UPS != 0) {  // coarse pixel m = (b, y, x) -> fine position (2y + PY, 2x + PX)
          const int ub = m / (HO * WO), urem = m - ub * (HO * WO), uy = urem / WO, ux = urem - uy * WO;
          o = ((size_t)(ub * 2 * HO + 2 * uy + ((UPS >> 1) & 1)) * (2 * WO) + 2 * ux + (UPS & 1)) * COUT + c0;
        }
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
        } else if constexpr (SPLIT) {
          const float sv[4] = {v0, v1, v2, v3};
          f16x4 oh, ol;
          split_pair4(sv, oh, ol);
          *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp) + (size_t)m * OPIX + c0) = oh;
          *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(outp) + (size_t)m * OPIX + COUT + c0) = ol;
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

// ---------------------------------------------------------------------------------------
// v3: halo direct convolution for the 3x3 / stride 1 / pad 1 layers (13 of the 20 convs,
// 83 % of the FLOPs).  The v2 kernel re-stages the activation tile once per filter tap
// (9x the input through L2 -> LDS, which is what bounds it at ~11 TB/s); here a workgroup
// brings the input rows its BM output pixels need -- a band of zero-padded rows, 64
// channels deep -- into LDS ONCE per 64-channel chunk and all 9 taps read it at shifted
// pixel offsets.  Only the weight tile (BN x 64 channels per tap) still streams, through a
// 2-slot LDS-DMA ring one tap ahead of the MFMAs.
//
// Geometry: NHWC activations flattened over (image, row, col) are one pixel array m; tap
// (kh,kw) of output pixel m reads pixel m + (kh-1)*W + (kw-1) unless that falls outside the
// image (x or y edge), where it reads zeros.  So the band a workgroup needs is simply the
// CONTIGUOUS pixel range [m0 - W - 1, mlast + W + 1] (it may run into neighbouring images;
// those pixels are never selected because the edge flags redirect such taps).  In LDS:
// slots 0 and 1 = pixels of zeros, slot q >= 2 = pixel m0 - W - 3 + q, [slot][64 ch] with the
// chunk swizzle of v2 (c ^ ((q >> 1) & 7)).  Consecutive output pixels sit in consecutive slots,
// so a ds_read_b128 lane group covers all 16 bank groups of the 256-byte bank row; an edge tap
// reads the zero slot of its own parity at its own swizzled chunk, which keeps that property
// (measured before this: 21-29 % of LDS cycles lost to bank conflicts from a single zero slot).
// No per-piece div/mod, no pad rows: BM + 2W + 2 pixels per band.
// ---------------------------------------------------------------------------------------
#ifdef HIPAC_HALO_STAMPS
// developer build: per-phase cycle totals of the halo kernel (s_memtime), summed over workgroups
static __device__ unsigned long long g_halo_stamps[8];
#define HALO_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define HALO_STAMP(var)
#endif
#ifndef HIPAC_C64_PF
#define HIPAC_C64_PF 2  // layer1 kernel: LDS fragment reads run this many k16 steps ahead of their MFMAs
#endif
// order of the 36 (tap, k16) steps of a 3x3 x 64-channel accumulation, shared by conv3x3_c64_kernel and the fused block
// (their results are compared bit for bit): 0 = (kh, kw, k16 step), the default; 1 = (kh, k16 step, kw), what the fragment-sharing
// experiment of block_c64.h (HIPAC_BLK_SHARE) needs
#ifndef HIPAC_C64_ORDER
#define HIPAC_C64_ORDER 0
#endif
constexpr int c64_step_kh(int st) { return st / 12; }
constexpr int c64_step_kw(int st) { return HIPAC_C64_ORDER ? st % 3 : (st % 12) / 4; }
constexpr int c64_step_kk(int st) { return HIPAC_C64_ORDER ? (st % 12) / 3 : st % 4; }

#ifndef HIPAC_HALO_TAP_UNROLL
#define HIPAC_HALO_TAP_UNROLL 3  // taps per unrolled group: 3 makes kw a constant (9 is slower: 2x, code size)
#endif
#ifndef HIPAC_HALO_W_ISSUE_KK
#define HIPAC_HALO_W_ISSUE_KK 1  // k16 sub-step after whose MFMAs the next weight tile is requested (-1: step start)
#endif
#ifndef HIPAC_HALO_GRID
#define HIPAC_HALO_GRID 512  // persistent halo workgroups: 2 per CU x 256 CUs
#endif
#ifndef HIPAC_HALO_DIRECT_EPI
#define HIPAC_HALO_DIRECT_EPI 0  // 1: epilogue straight from the accumulators (v_permlane32_swap pairs the lane halves), no LDS
                                 // staging.  Bit-identical; measured NOT faster (trunk 3.65 vs 3.59 us per patch): the epilogue
                                 // shrinks 13 k -> 8.6 k cycles without a residual, but 32-byte runs per pixel make the residual
                                 // reads and the stores slower than the staged form's full 128-byte lines
#endif
constexpr int halo_band_pieces(int W, int BM) { return (BM + 2 * W + 2 + 2 + 7) / 8; }  // 8-pixel (1 KB) pieces

// NSW = depth of the weight ring (2, or 3 where LDS leaves room for two workgroups per CU).
// Epilogue: every wave sends its 32-pixel sub-tiles through a private fp32 staging area in LDS so
// that global traffic is 16-byte items of contiguous channel runs (residual loads and stores)
// with no workgroup barrier; workgroups are persistent and the next tile's band is prefetched
// behind the epilogue.
// SPLIT: fp16 (hi, lo) pairs, see conv_glds_kernel; virtual chunk 3c + 1 (Wlo x Xhi) reuses the band of 3c.
// WM_ x WN_ = the 4 waves as pixel parts x channel parts; MINW = waves per SIMD the register budget is set for (1: one
// 512-register wave per SIMD with a 128 x 128 tile -- 0.5 LDS fragment reads per MFMA instead of 0.75).
// PCIN > 0 (second conv of a down-sampling BasicBlock): the block's 1x1 / stride 2 projection shortcut is folded in as
// PCIN / 64 extra K steps -- the "band" of such a step is a GATHER of the block input's pixels (2y, 2x), 64 channels
// each, placed where the centre tap reads, the weight tile comes from the projection's [COUT][PCIN] matrix, and `bias`
// is the sum of both biases: conv2 + projection accumulate in ONE fp32 accumulator, the shortcut map (one HBM round
// trip) and the projection launch disappear.  `resid` is then the block input [n][2H][2W][PCIN], `wgt_p` the projection.
// DBLW (SPLIT only, where the LDS budget allows a weight ring of two 2-tile slots): per real 64-channel chunk the K loop runs
// nine DOUBLE steps -- the band of the hi plane against the weight tiles [Whi | Wlo] of a tap, two MFMAs per activation
// fragment -- and nine single steps (lo plane x Whi): 18 barriers and (MTW + 2 NT) / (2 MTW NT) fragment reads per MFMA on two
// thirds of the work instead of 27 barriers and (MTW + NT) / (MTW NT).
template <typename T, int CIN, int COUT, int H, int W, int BM, int BN, int NSW, bool RELU, bool RESID, bool OUTF32,
          bool SPLIT = false, int WM_ = 2, int WN_ = 2, int MINW = 2, int PCIN = 0, bool DBLW = false>
__global__ __launch_bounds__(256, MINW) void conv3x3_halo_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                              const float* __restrict__ bias,
                                                              const T* __restrict__ resid, void* __restrict__ outp,
                                                              int M, int n_img, int n_mtiles,
                                                              const char* __restrict__ zero_page,
                                                              const T* __restrict__ wgt_p = nullptr) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int RC = CIN / 64;                      // real 64-channel chunks
  constexpr int CC = SPLIT ? 3 * RC : RC;           // (virtual) chunks of the K loop
  constexpr int VCIN = CC * 64;                     // K elements per tap
  constexpr int PIXC = SPLIT ? 2 * CIN : CIN;       // activation elements per input pixel
  constexpr int OPIX = SPLIT ? 2 * COUT : COUT;     // elements per pixel of T-typed outputs and of the residual
  constexpr int KTOT = 9 * VCIN;
  static_assert(!SPLIT || std::is_same<T, _Float16>::value, "split pairs are fp16");
  constexpr int WM = WM_, WN = WN_;                 // 4 waves: pixel parts x channel parts
  static_assert(WM * WN == 4, "four waves");
  constexpr int MTW = BM / (WM * 32);               // 32-pixel sub-tiles per wave (2 or 4)
  constexpr int WTN = BN / WN, NT = WTN / 32;       // channels per wave, 32-wide tiles per wave
  constexpr int A_PIECES = halo_band_pieces(W, BM);
  constexpr int A_BYTES = A_PIECES * 1024;
  constexpr int W_BYTES = BN * 128;
  constexpr int WPW = BN / 8 / 4;                    // W pieces per wave per tap
  constexpr int NTILES_N = COUT / BN;
  constexpr int PCC = PCIN / 64;                    // projection K steps (0: no folded projection)
  constexpr int NSTEP = 9 * CC + PCC;
  static_assert(PCIN % 64 == 0 && (PCIN == 0 || (!RESID && !SPLIT)), "folded projection replaces the residual input");
  static_assert((BM == 128 || BM == 256 || BM == 512) && WTN % 32 == 0 && COUT % BN == 0 && CIN % 64 == 0, "tile shape");
  static_assert((BN / 8) % 4 == 0, "W piece split");
  static_assert(NSW == 2 || NSW == 3, "weight ring depth");
  constexpr int STG_BYTES = 4 * 32 * (WTN * 4 + 16);  // 4 waves x [32 px][WTN fp32 + pad] epilogue staging
  static_assert(!DBLW || (SPLIT && NSW == 2 && PCIN == 0), "double weight steps: split pairs, two ring slots");
  constexpr int SLOT_BYTES = DBLW ? 2 * W_BYTES : W_BYTES;  // one ring slot
  constexpr int S_BYTES = NSW * SLOT_BYTES > STG_BYTES ? NSW * SLOT_BYTES : STG_BYTES;  // ring, aliased by the staging
  static_assert(A_BYTES + S_BYTES <= 160 * 1024 / MINW, "LDS: MINW workgroups per CU");

  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  unsigned char* const Abuf = ring;
  unsigned char* const Wbuf = ring + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 31, h = lane >> 5;

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const char* in_b = reinterpret_cast<const char*>(in);
  const char* w_b = reinterpret_cast<const char*>(wgt);
  const int prow = lane >> 3, dchunk = lane & 7;

  // band of the tile starting at pixel m0_: the contiguous pixel range [m0_ - W - 1, mlast_ + W + 1]
  auto issue_band_of = [&](int m0_, int cc) {
    const int mlast_ = (m0_ + BM <= M ? m0_ + BM : M) - 1;
    const int mstart_ = m0_ - W - 1;
    const int npx_ = mlast_ - m0_ + 1 + 2 * W + 2;  // band pixels; slots 2..npx+1 (slots 0, 1 = zeros)
    const int npieces_ = (npx_ + 2 + 7) >> 3;
    for (int p = wave; p < npieces_; p += 4) {
      const int q = p * 8 + prow;                // slot
      const int mm = mstart_ + q - 2;            // flattened pixel held by this slot
      const bool ok = q >= 2 && q <= npx_ + 1 && mm >= 0 && mm < M;
      const int schunk = dchunk ^ ((q >> 1) & 7);
      const char* src = ok ? in_b + ((size_t)mm * PIXC + split_achunk<SPLIT, RC>(cc) * 64 + schunk * 8) * 2
                           : zero_page + dchunk * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Abuf + p * 1024), 16, 0, 0);
    }
  };

  // Persistent workgroups: virtual block id vb = blockIdx.x + i * gridDim.x (gridDim.x % 8 == 0, so a
  // workgroup stays on its XCD's slice of the tile order).  The band of the NEXT tile is brought
  // in during the epilogue of the current one (the epilogue stages through the weight ring only).
  for (int vb = blockIdx.x, first_tile = 1;; vb += gridDim.x, first_tile = 0) {
  const int xcd = vb & 7, slot = vb >> 3;
  const int mt = (slot / NTILES_N) * 8 + xcd;
  const int nt = slot % NTILES_N;
  if (mt >= n_mtiles) break;  // mt grows with vb on a fixed XCD: nothing valid follows
  const int m0 = mt * BM, n0 = nt * BN;
  const int mlast = (m0 + BM <= M ? m0 + BM : M) - 1;
  const int mstart = m0 - W - 1;
  auto issue_band = [&](int cc) { issue_band_of(m0, cc); };
  if (!first_tile) __builtin_amdgcn_s_barrier();  // the previous tile's staging reads are done: ring is free
  int w_off[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (wave + 4 * i) * 8 + prow;
    w_off[i] = ((n0 + row) * KTOT + (dchunk ^ ((row >> 1) & 7)) * 8) * 2;
  }
  // weight DMA through a buffer descriptor: per-lane 32-bit row offset in a VGPR (computed once per
  // tile), the tap / chunk offset in an SGPR -- no per-piece address arithmetic in the tap loop
  const rsrc_t w_rsrc = make_rsrc(w_b, COUT * KTOT * 2);
  int wp_off[PCC > 0 ? WPW : 1];
  if constexpr (PCC > 0) {
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
      const int row = (wave + 4 * i) * 8 + prow;
      wp_off[i] = ((n0 + row) * PCIN + (dchunk ^ ((row >> 1) & 7)) * 8) * 2;
    }
  }
  const rsrc_t wp_rsrc = make_rsrc(PCC > 0 ? reinterpret_cast<const char*>(wgt_p) : w_b, COUT * (PCC > 0 ? PCIN : KTOT) * 2);
  auto issue_w = [&](int step, int slot_) {  // weights of step = cc*9 + tap: K offset (tap*CIN + cc*64)
    if (PCC > 0 && step >= 9 * CC) {  // uniform: a projection step, 64 input channels of the 1x1 matrix
      const int kofs_bytes = (step - 9 * CC) * 128;
      static_for<WPW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        buffer_load_lds16(wp_rsrc, Wbuf + slot_ * W_BYTES + (wave + 4 * i) * 1024, wp_off[PCC > 0 ? i : 0], kofs_bytes);
      });
      // (an empty statement hipcc cannot merge: without it the two paths' DMA calls are sunk into one block whose descriptor and
      // offsets are SELECTED -- the offset arrays then live in scratch and every DMA sits in a waterfall loop behind a vmcnt(0))
      asm volatile("" ::: "memory");
      return;
    }
    const int cc = step / 9, tap = step - cc * 9;
    const int kofs_bytes = (tap * VCIN + cc * 64) * 2;
    static_for<WPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      buffer_load_lds16(w_rsrc, Wbuf + slot_ * W_BYTES + (wave + 4 * i) * 1024, w_off[i], kofs_bytes);
    });
  };
  // folded projection: the block input's pixel (2y, 2x) of every output pixel of the tile, 64 channels of chunk pc, at the
  // slot the centre tap reads for that output pixel (q0 = m - mstart + 2)
  auto issue_gather = [&](int pc) {
    if constexpr (PCC > 0) {
      const int npx_ = mlast - m0 + 1;
      const int first = W + 3, last = first + npx_ - 1;  // slots that hold pixels
      for (int p = wave + (first >> 3); p <= (last >> 3); p += 4) {
        const int q = p * 8 + prow;
        const int mm = m0 + q - first;
        const bool ok = q >= first && q <= last;
        const int b = mm / (H * W), rem = mm - b * (H * W), y = rem / W, x = rem - y * W;
        const int schunk = dchunk ^ ((q >> 1) & 7);
        const char* src = ok ? reinterpret_cast<const char*>(resid) +
                                   ((((size_t)b * (2 * H) + 2 * y) * (2 * W) + 2 * x) * PCIN + pc * 64 + schunk * 8) * 2
                             : zero_page + dchunk * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Abuf + p * 1024), 16, 0, 0);
      }
    }
  };

  // ---- consumer side: slot of this lane's two output pixels for tap (0, centre column) ----
  int q0[MTW];
  // bit 0: x == 0, bit 1: x == W-1, bit 2: y == 0, bit 3: y == H-1.  One register per sub-tile and one AND
  // per tap: four bool arrays tested against the runtime (kh, kw) cost 7-13 % of the kernel (compare /
  // mask-combine chains in the tap loop)
  int eflags[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    int m = m0 + wm * (MTW * 32) + i * 32 + r;
    m = m <= mlast ? m : mlast;  // tail lanes read a valid pixel; their results are not stored
    const int b = m / (H * W);
    const int rem = m - b * (H * W);
    const int y = rem / W, x = rem - y * W;
    q0[i] = m - mstart + 2;                       // tap (kh,kw) -> q0 + (kh-1)*W + kw - 1
    eflags[i] = (x == 0 ? 1 : 0) | (x == W - 1 ? 2 : 0) | (y == 0 ? 4 : 0) | (y == H - 1 ? 8 : 0);
  }
  const int sw_w = (r >> 1) & 7;
  int rdw[4], ck[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    ck[kk] = (2 * kk + h) << 4;
    rdw[kk] = (wn * WTN + r) * 128 + (((2 * kk + h) ^ sw_w) << 4);
  }

  f32x16 acc[MTW][NT];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Epilogue geometry (per WAVE, no workgroup barriers): the wave owns MTW sub-tiles of 32 pixels x
  // WTN channels.  Sub-tile i goes through the wave's private fp32 staging [32 px][WTN] and leaves
  // as 16-byte items (8 channels): item = lane + 64k -> pixel item / CPW, channel group lane % CPW.
  constexpr int CPW = WTN / 8;                      // 8-channel items per pixel (wave's channel half)
  constexpr int IPT = 32 * CPW / 64;                // items per lane and sub-tile
  constexpr int SROWW = WTN * 4 + 16;               // staging row: WTN fp32 + pad
  static_assert(64 % CPW == 0 && (32 * CPW) % 64 == 0, "epilogue items");
  constexpr bool DIRECT = HIPAC_HALO_DIRECT_EPI && !SPLIT && sizeof(T) == 2 && IPT == 2 * NT;
  static_assert(4 * 32 * SROWW <= S_BYTES, "per-wave staging fits the ring region");
  const int e_c0 = n0 + wn * WTN + (lane % CPW) * 8;  // first of this lane's 8 output channels
  const int e_px = lane / CPW;                        // pixel of item k: e_px + k * (64 / CPW)
  // residual, prefetched into registers: sub-tile 0 behind the MFMAs of the last K step,
  // sub-tile i+1 behind the staging of sub-tile i (two register sets)
  // (SPLIT: the two register sets hold the hi and the lo fragments of ONE sub-tile, fetched at the top of that sub-tile)
  frag rv[2][RESID ? IPT : 1];
  auto load_resid = [&](auto SUB) {
    constexpr int i = decltype(SUB)::value;
    if constexpr (RESID) {
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        int m = m0 + wm * (MTW * 32) + i * 32 + e_px + k * (64 / CPW);
        m = m < M ? m : M - 1;  // unconditional load from a valid row (tail rows are never stored)
        if constexpr (SPLIT) {
          rv[0][k] = *reinterpret_cast<const frag*>(resid + (size_t)m * OPIX + e_c0);
          rv[1][k] = *reinterpret_cast<const frag*>(resid + (size_t)m * OPIX + COUT + e_c0);
        } else if constexpr (DIRECT) {
          // item k = (channel tile j, pair qp): this lane's pixel r, the 8 channels it will also store (16 qp + 8 h)
          int md = m0 + wm * (MTW * 32) + i * 32 + r;
          md = md < M ? md : M - 1;
          rv[i & 1][k] = *reinterpret_cast<const frag*>(resid + (size_t)md * COUT + n0 + wn * WTN + (k >> 1) * 32 + 16 * (k & 1) + 8 * h);
        } else {
          rv[i & 1][k] = *reinterpret_cast<const frag*>(resid + (size_t)m * COUT + e_c0);
        }
      }
    }
  };

  HALO_STAMP(t_start);
#ifdef HIPAC_HALO_STAMPS
  unsigned long long t_first = 0;
#endif
  int s = 0;  // K step counter
  if constexpr (DBLW) {
    // ---- double weight steps (see the template note): step = (chunk c, phase, tap), phase 0 = hi plane x [Whi | Wlo]
    constexpr int NSTEP2 = 18 * RC;
    auto issue_w2 = [&](int step, int slot_) {
      const int c = step / 18, rr = step - c * 18, ph = rr >= 9 ? 1 : 0, tap = rr - 9 * ph;
      const int kofs_bytes = (tap * VCIN + 3 * c * 64) * 2;  // the hi_c block of this tap; lo_c follows 128 bytes on
      static_for<WPW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        buffer_load_lds16(w_rsrc, Wbuf + slot_ * SLOT_BYTES + (wave + 4 * i) * 1024, w_off[i], kofs_bytes);
      });
      if (ph == 0)
        static_for<WPW>([&](auto I) {
          constexpr int i = decltype(I)::value;
          buffer_load_lds16(w_rsrc, Wbuf + slot_ * SLOT_BYTES + W_BYTES + (wave + 4 * i) * 1024, w_off[i], kofs_bytes + 128);
        });
    };
    if (first_tile) issue_band(0);
    issue_w2(0, 0);
    for (int c2 = 0; c2 < 2 * RC; ++c2) {
      const int ph = c2 & 1;
      if (c2 > 0) {
        __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous band
        issue_band(3 * (c2 >> 1) + 2 * ph);  // virtual chunk 3c = hi plane, 3c + 2 = lo plane
      }
#pragma unroll HIPAC_HALO_TAP_UNROLL
      for (int tap = 0; tap < 9; ++tap, ++s) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const int kh = tap / 3, kw = tap - kh * 3;
        const int toff = (kh - 1) * W + kw - 1;
        const unsigned char* wst = Wbuf + (s & 1) * SLOT_BYTES;
        const int tapmask = (kw == 0 ? 1 : 0) | (kw == 2 ? 2 : 0) | (kh == 0 ? 4 : 0) | (kh == 2 ? 8 : 0);
        int abase[MTW], asw[MTW];
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const bool off_img = (eflags[i] & tapmask) != 0;
          const int qt = q0[i] + toff;
          const int q = off_img ? (qt & 1) : qt;
          abase[i] = q << 7;
          asw[i] = ((qt >> 1) & 7) << 4;
        }
        frag af[2][MTW], wf[2][2 * NT];
#pragma unroll
        for (int i = 0; i < MTW; ++i) af[0][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[0] ^ asw[i]));
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[0]);
        if (ph == 0) {
#pragma unroll
          for (int j = 0; j < NT; ++j) wf[0][NT + j] = *reinterpret_cast<const frag*>(wst + W_BYTES + j * 4096 + rdw[0]);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk + 1 < 4) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
              af[(kk + 1) & 1][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[kk + 1] ^ asw[i]));
#pragma unroll
            for (int j = 0; j < NT; ++j)
              wf[(kk + 1) & 1][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[kk + 1]);
            if (ph == 0) {
#pragma unroll
              for (int j = 0; j < NT; ++j)
                wf[(kk + 1) & 1][NT + j] = *reinterpret_cast<const frag*>(wst + W_BYTES + j * 4096 + rdw[kk + 1]);
            }
          }
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[kk & 1][j], af[kk & 1][i], acc[i][j]);
          if (ph == 0) {
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[kk & 1][NT + j], af[kk & 1][i], acc[i][j]);
          }
          if (kk == 1 && s + 1 < NSTEP2) issue_w2(s + 1, (s + 1) & 1);  // its slot was freed by this step's barrier
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  } else {
  if (first_tile) issue_band(0);
#pragma unroll
  for (int pstep = 0; pstep < NSW - 1; ++pstep)
    if (pstep < NSTEP) issue_w(pstep, pstep);
  for (int cc = 0; cc < CC; ++cc) {
    if (cc > 0 && (!SPLIT || cc % 3 != 1)) {  // (SPLIT: chunk 3c + 1 multiplies the band of 3c by the low weight halves)
      __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous chunk's band
      issue_band(cc);
    }
#pragma unroll HIPAC_HALO_TAP_UNROLL
    for (int tap = 0; tap < 9; ++tap, ++s) {
      // W(s) must have landed; the band too at tap 0 (it was issued AFTER W(s+1..), so drain everything)
      if (NSW == 3 && tap != 0 && s + 1 < NSTEP) wait_vmcnt<WPW>();
      else wait_vmcnt<0>();
#ifndef HIPAC_ABL_NO_BARRIER
      __builtin_amdgcn_s_barrier();
#endif
#ifdef HIPAC_HALO_STAMPS
      if (s == 0) {
        t_first = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
#endif
#if !defined(HIPAC_ABL_NO_W_DMA) && HIPAC_HALO_W_ISSUE_KK < 0
      if (s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
#endif
      if (RESID && !SPLIT && s == NSTEP - 1) load_resid(std::integral_constant<int, 0>{});
      const int kh = tap / 3, kw = tap - kh * 3;
      const int toff = (kh - 1) * W + kw - 1;
      const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
      const int tapmask = (kw == 0 ? 1 : 0) | (kw == 2 ? 2 : 0) | (kh == 0 ? 4 : 0) | (kh == 2 ? 8 : 0);  // uniform
      int abase[MTW], asw[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const bool off_img = (eflags[i] & tapmask) != 0;
        // out-of-image taps read a zero pixel: slot 0 or 1 by the parity of the slot the lane would
        // have read, at the chunk position its swizzle selects -- i.e. the SAME 16-byte bank group
        // as the in-image address, so redirected lanes never collide with their neighbours
        const int qt = q0[i] + toff;
        const int q = off_img ? (qt & 1) : qt;
        abase[i] = q << 7;
        asw[i] = ((qt >> 1) & 7) << 4;
      }
      frag af[2][MTW], wf[2][NT];
#ifdef HIPAC_ABL_NO_LDSREAD
      {
        const frag c0 = __builtin_bit_cast(frag, u32x4{(unsigned)abase[0], (unsigned)asw[0], (unsigned)toff, 1u});
#pragma unroll
        for (int i = 0; i < MTW; ++i) af[0][i] = af[1][i] = c0;
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[0][j] = wf[1][j] = c0;
        (void)wst;
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[kk & 1][j], af[kk & 1][i], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      continue;
#endif
#pragma unroll
      for (int i = 0; i < MTW; ++i) af[0][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[0] ^ asw[i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[0]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
#pragma unroll
          for (int i = 0; i < MTW; ++i)
            af[(kk + 1) & 1][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[kk + 1] ^ asw[i]));
#pragma unroll
          for (int j = 0; j < NT; ++j)
            wf[(kk + 1) & 1][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[kk + 1]);
        }
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[kk & 1][j], af[kk & 1][i], acc[i][j]);
#if !defined(HIPAC_ABL_NO_W_DMA) && HIPAC_HALO_W_ISSUE_KK >= 0
        // the next step's weight DMA is issued from inside the MFMA stream (its slot was freed by this
        // step's barrier), where its issue cost hides behind queued MFMAs instead of delaying the
        // step's first LDS reads
        if (kk == HIPAC_HALO_W_ISSUE_KK && s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
#endif
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  }  // !DBLW
  if constexpr (PCC > 0) {
    // ---- the folded projection: PCC more K steps, centre tap only (no image-edge cases: pixel (2y, 2x) always exists)
    for (int pc = 0; pc < PCC; ++pc, ++s) {
      __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous band
      issue_gather(pc);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
      const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
      int abase[MTW], asw[MTW];
#pragma unroll
      for (int i = 0; i < MTW; ++i) abase[i] = q0[i] << 7, asw[i] = ((q0[i] >> 1) & 7) << 4;
      frag af[2][MTW], wf[2][NT];
#pragma unroll
      for (int i = 0; i < MTW; ++i) af[0][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[0] ^ asw[i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[0]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
#pragma unroll
          for (int i = 0; i < MTW; ++i)
            af[(kk + 1) & 1][i] = *reinterpret_cast<const frag*>(Abuf + abase[i] + (ck[kk + 1] ^ asw[i]));
#pragma unroll
          for (int j = 0; j < NT; ++j)
            wf[(kk + 1) & 1][j] = *reinterpret_cast<const frag*>(wst + j * 4096 + rdw[kk + 1]);
        }
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = E::mfma(wf[kk & 1][j], af[kk & 1][i], acc[i][j]);
      }
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  HALO_STAMP(t_loop);
  // ---- epilogue -------------------------------------------------------------------------------
  const float4 b_lo = *reinterpret_cast<const float4*>(bias + e_c0);
  const float4 b_hi = *reinterpret_cast<const float4*>(bias + e_c0 + 4);
  __builtin_amdgcn_s_barrier();  // every wave has left the K loop: band and ring are free
  {
    // prefetch the next tile's first band chunk; it lands behind this epilogue
    const int vn = vb + gridDim.x;
    const int mtn = ((vn >> 3) / NTILES_N) * 8 + (vn & 7);
    if (mtn < n_mtiles) issue_band_of(mtn * BM, 0);
  }
  unsigned char* const Sl = Wbuf + wave * (32 * SROWW);  // this wave's private staging
  if constexpr (DIRECT) {
    // Straight from the accumulators: lane (r, h) holds pixel r of the sub-tile, channels 8q + 4h .. +3 of every 32-wide
    // tile.  Bias, residual, ReLU and the rounding happen there; one v_permlane32_swap per packed dword then pairs the
    // lane halves so that every lane stores 16 contiguous bytes (channels 16 qp + 8 h .. +7) -- no LDS round trip, no
    // waits between sub-tiles.  The residual arrives in the stored layout and is un-paired by the same swap.  Same
    // additions in the same order as the staged form: bit-identical results.
    float4 bv[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const float4*>(bias + n0 + wn * WTN + j * 32 + 8 * q + 4 * h);
    static_for<MTW>([&](auto SUB) {
      constexpr int i = decltype(SUB)::value;
      if constexpr (i + 1 < MTW) load_resid(std::integral_constant<int, i + 1>{});
      const int m = m0 + wm * (MTW * 32) + i * 32 + r;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        unsigned P[4][2];
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          unsigned rd[4] = {0u, 0u, 0u, 0u};
          if constexpr (RESID) {
            const u32x4 t = __builtin_bit_cast(u32x4, rv[i & 1][2 * j + qp]);
            rd[0] = t[0], rd[1] = t[1], rd[2] = t[2], rd[3] = t[3];
            permlane32_swap(rd[0], rd[2]);  // -> (rd[0], rd[1]) = channels of q = 2 qp, (rd[2], rd[3]) = those of q = 2 qp + 1
            permlane32_swap(rd[1], rd[3]);
          }
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * qp + qq;
            float v0 = acc[i][j][4 * q + 0] + bv[j][q].x, v1 = acc[i][j][4 * q + 1] + bv[j][q].y;
            float v2 = acc[i][j][4 * q + 2] + bv[j][q].z, v3 = acc[i][j][4 * q + 3] + bv[j][q].w;
            if constexpr (RESID) {
              const typename E::vec4 rr = __builtin_bit_cast(typename E::vec4, u32x2{rd[2 * qq], rd[2 * qq + 1]});
              v0 += (float)rr[0], v1 += (float)rr[1], v2 += (float)rr[2], v3 += (float)rr[3];
            }
            if constexpr (RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f), v2 = fmaxf(v2, 0.f), v3 = fmaxf(v3, 0.f);
            if constexpr (OUTF32) {
              if (m < M)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * COUT + n0 + wn * WTN + j * 32 + 8 * q + 4 * h) =
                    make_float4(v0, v1, v2, v3);
            } else {
              P[q][0] = PackPair<T>::pack_rn(v0, v1);
              P[q][1] = PackPair<T>::pack_rn(v2, v3);
            }
          }
          if constexpr (!OUTF32) {
            permlane32_swap(P[2 * qp][0], P[2 * qp + 1][0]);
            permlane32_swap(P[2 * qp][1], P[2 * qp + 1][1]);
            if (m < M)
              *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(outp) + (size_t)m * COUT + n0 + wn * WTN + j * 32 + 16 * qp + 8 * h) =
                  u32x4{P[2 * qp][0], P[2 * qp][1], P[2 * qp + 1][0], P[2 * qp + 1][1]};
          }
        }
      }
    });
  } else
  static_for<MTW>([&](auto SUB) {
    constexpr int i = decltype(SUB)::value;
    if constexpr (SPLIT) {
      load_resid(std::integral_constant<int, i>{});  // lands behind the staging round trip below
    } else if constexpr (i + 1 < MTW) {
      load_resid(std::integral_constant<int, i + 1>{});
    }
    // accumulators -> fp32 rows (LDS operations of one wave complete in order: no barrier)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
        v[0] = acc[i][j][4 * q + 0];
        v[1] = acc[i][j][4 * q + 1];
        v[2] = acc[i][j][4 * q + 2];
        v[3] = acc[i][j][4 * q + 3];
        *reinterpret_cast<f32x4*>(Sl + r * SROWW + (j * 32 + 8 * q + 4 * h) * 4) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const int px = e_px + k * (64 / CPW);
      const int m = m0 + wm * (MTW * 32) + i * 32 + px;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Sl + px * SROWW + (lane % CPW) * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(Sl + px * SROWW + (lane % CPW) * 32 + 16);
      if (m < M) {
        float v[8] = {lo[0] + b_lo.x, lo[1] + b_lo.y, lo[2] + b_lo.z, lo[3] + b_lo.w,
                      hi[0] + b_hi.x, hi[1] + b_hi.y, hi[2] + b_hi.z, hi[3] + b_hi.w};
        const size_t o = (size_t)m * COUT + e_c0;
        if constexpr (RESID && SPLIT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rv[0][k][e] + (float)rv[1][k][e];  // hi + lo is exact in fp32
        } else if constexpr (RESID) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rv[i & 1][k][e];
        }
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (OUTF32) {
          float* op = reinterpret_cast<float*>(outp) + o;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else if constexpr (SPLIT) {
          f16x8 oh, ol;
          split_pair8(v, oh, ol);
          _Float16* op = reinterpret_cast<_Float16*>(outp) + (size_t)m * OPIX + e_c0;
          *reinterpret_cast<f16x8*>(op) = oh;
          *reinterpret_cast<f16x8*>(op + COUT) = ol;
        } else {
          frag ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (T)v[e];
          *reinterpret_cast<frag*>(reinterpret_cast<T*>(outp) + o) = ov;
        }
      }
    }
    // the next sub-tile overwrites the staging rows: this wave's reads above must have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
#ifdef HIPAC_HALO_STAMPS
  HALO_STAMP(t_end);
  if (tid == 0) {
    atomicAdd(&g_halo_stamps[0], t_first - t_start);  // prologue: band + first weight tile in flight
    atomicAdd(&g_halo_stamps[1], t_loop - t_first);   // K loop
    atomicAdd(&g_halo_stamps[2], t_end - t_loop);     // epilogue
    atomicAdd(&g_halo_stamps[3], 1ull);
  }
#endif
  }  // persistent tile loop
}

}  // namespace hipac
#include "halo16.h"
#include "halo16x2.h"
#include "band16.h"
namespace hipac {

// ---------------------------------------------------------------------------------------
// Fused stem: 7x7/2 conv (+BN+ReLU) and the 3x3/2 max-pool in one kernel, so the
// 112x112x64 stem activation (1.6 MB per patch, the largest tensor of the network) never
// reaches HBM.  A workgroup (4 waves) produces an 8x7 tile of POOLED pixels: it needs the
// 17x15 = 255 stem pixels around it (exactly 8 MFMA sub-tiles of 32), which need a 39x36
// pixel patch of the padded NHWC4 input (11 KB).  Workgroups are persistent (grid-stride
// over tiles) and every lane keeps ITS slice of the whole stem weight matrix in registers
// (2 channel tiles x 7 kh x 2 k16 fragments = 112 VGPRs): weights are fetched once per
// workgroup and never re-read, so LDS serves only the activation fragments (1 read per 2
// MFMAs).  The input patch is double-buffered with a register prefetch of the next tile
// behind the MFMAs; the stem tile goes to LDS (bias, ReLU, rounded to T exactly as the
// unfused path stores it) and each thread reduces pooled pixels x 8 channels.
// Stem pixels outside the image (row/col -1) are stored as 0: inputs are post-ReLU (>= 0),
// so they can never win the max.
// ---------------------------------------------------------------------------------------
// U8IN: the input is the raw uint8 HWC patch batch [n,224,224,3]; ToTensor/Normalize is
// applied while the patch is staged (a 3 x 256 table of T values in LDS == the fp32 LUT
// rounded to T, i.e. exactly what hipac_patches_normalize would have written), so the padded
// NHWC4 tensor (427 KB per patch written and read back) never exists.
#ifndef HIPAC_STEM_TILE16
#define HIPAC_STEM_TILE16 0  // measured: 660 vs 605 ns per patch (+12 % MFMA work outweighs conflict-free reads)
#endif
constexpr int kStemTilesPerImage = HIPAC_STEM_TILE16 ? 64 : 56;

template <typename T, bool U8IN>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const void* __restrict__ xin_, const T* __restrict__ wgt,
                                                        const float* __restrict__ bias, T* __restrict__ out,
                                                        int n_tiles, const unsigned short* __restrict__ lut_t,
                                                        long long in_bytes) {
  const T* xin = reinterpret_cast<const T*>(xin_);
  using E = Elem<T>;
  using frag = typename E::frag;
#if HIPAC_STEM_TILE16
  // 16 x 16 stem pixels in eight 8 x 4 lane blocks over a 320-byte patch pitch: every ds_read_b128 lane
  // group of the MFMA loop covers all 16 bank groups (conflict-free); the 7 x 7 pooled tile uses 15 x 15
  // of them (12 % more MFMA work than the 17 x 15 tile, whose 15-pixel rows cannot avoid 2-way conflicts)
  constexpr int PTH = 7, PTW = 7;                                       // pooled tile
  constexpr int STW = 16, STH = 16, NPX = STW * STH;                    // 256 stem pixels
  constexpr int PROWS = 2 * STH + 5, PCOLS = 40;                        // 37 x 40 input pixels (8 B each)
#else
  constexpr int PTH = 8, PTW = 7;                                       // pooled tile
  constexpr int STW = 2 * PTW + 1, STH = 2 * PTH + 1, NPX = STW * STH;  // 15 x 17 = 255 stem pixels
  constexpr int PROWS = 2 * STH + 5, PCOLS = 36;                        // 39 x 36 input pixels (8 B each)
#endif
  static_assert(56 / PTW * (56 / PTH) == kStemTilesPerImage, "tile count");
  constexpr int PPR = PCOLS / 2;                                        // 16-byte pieces per patch row
  constexpr int NPIECE = PROWS * PPR;                                   // 702
  constexpr int PF = (NPIECE + 255) / 256;                              // pieces per thread (3)
  constexpr int TILES_X = 56 / PTW, TILES_Y = 56 / PTH, TPI = TILES_X * TILES_Y;  // 8 x 7 = 56 per image
  constexpr int SPX = 144;  // stem-tile pixel stride in LDS: 128 B of channels + 16 B pad (bank spread)
  constexpr int P_BYTES = PROWS * PCOLS * 8, S_BYTES = 256 * SPX;
  constexpr int RAWROW = (PCOLS * 3 + 3 + 3) / 4 * 4, RAWDW = RAWROW / 4;  // raw uint8 window per patch row
  constexpr int RAW_BYTES = U8IN ? PROWS * RAWROW + 3 * 256 * 2 : 0;    // + the T-typed normalise table
  constexpr int NRAW = PROWS * RAWDW, PFR = (NRAW + 255) / 256;          // 1092 dwords, 5 per thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P_BYTES + S_BYTES + RAW_BYTES];
  unsigned char* const Sl = smem + 2 * P_BYTES;
  unsigned char* const Rl = smem + 2 * P_BYTES + S_BYTES;
  unsigned short* const Ll = reinterpret_cast<unsigned short*>(Rl + PROWS * RAWROW);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  if constexpr (U8IN) {
    for (int i = threadIdx.x; i < 3 * 256; i += 256) Ll[i] = lut_t[i];
  }
  // this lane's rows of the weight matrix, all of K, in registers for the kernel's lifetime
  frag wreg[2][7][2];
  {
    const char* wb = reinterpret_cast<const char*>(wgt) + r * 448 + 16 * h;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          wreg[j][kh][kk] = *reinterpret_cast<const frag*>(wb + j * 32 * 448 + kh * 64 + kk * 32);
  }
  float4 bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const float4*>(bias + j * 32 + 8 * q + 4 * h);

  // the two stem pixels of this lane (sub-tiles 2*wave, 2*wave+1); pixel 255 does not exist
  int P[2], a_rd[2], ly[2], lx[2], sidx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    P[i] = (2 * wave + i) * 32 + r;
#if HIPAC_STEM_TILE16
    const int sub = 2 * wave + i;                   // block (sub & 1, sub >> 1) of 8 x 4 pixels
    lx[i] = (sub & 1) * 8 + (r & 7);
    ly[i] = (sub >> 1) * 4 + (r >> 3);
#else
    const int Pc = P[i] < NPX ? P[i] : NPX - 1;
    ly[i] = Pc / STW;
    lx[i] = Pc - ly[i] * STW;
#endif
    // row-major index in the LDS stem tile; lane 255 of the 17 x 15 tile (a pixel that does not exist)
    // keeps its own dummy row 255 -- it must not share a row with pixel 254
    sidx[i] = P[i] < NPX ? ly[i] * STW + lx[i] : P[i];
    a_rd[i] = ((2 * ly[i]) * PCOLS + 2 * lx[i]) * 8 + 16 * h;
  }

  auto tile_origin = [&](int tile, int& b, int& py0, int& px0) {
    b = tile / TPI;
    const int t = tile - b * TPI;
    const int ty = t / TILES_X;
    py0 = ty * PTH;
    px0 = (t - ty * TILES_X) * PTW;
  };
  u32x4 pre[U8IN ? 1 : PF];
  unsigned praw[U8IN ? PFR : 1];
  auto fetch = [&](int tile) {  // global -> registers: the input patch of `tile`
    int b, py0, px0;
    tile_origin(tile, b, py0, px0);
    const int R0 = 2 * (2 * py0 - 1), C0 = 2 * (2 * px0 - 1);
    if constexpr (U8IN) {
      // raw bytes of image rows R0-3 .. , columns C0-3 .. C0+32, as aligned dwords.  The byte offset of
      // window row `row` is tile_base + row * 672 with ONE 64-bit scalar per tile; its misalignment
      // sh = offset & 3 is the same for every row and image (a row is 672 = 0 mod 4 bytes), so the
      // per-lane part is 32-bit arithmetic only
      const unsigned char* src = reinterpret_cast<const unsigned char*>(xin_);
      const int sh = ((C0 - 3) * 3) & 3;
      const long long tile_base = (((long long)b * kPatch + (R0 - 3)) * kPatch + (C0 - 3)) * 3 - sh;
      const long long lo = -tile_base, hi = in_bytes - 4 - tile_base;  // valid range of the per-lane offset
      const int lo32 = lo > 0 ? (lo < 0x7fffffff ? (int)lo : 0x7fffffff) : 0;
      const int hi32 = hi < 0 ? -1 : (hi < 0x7fffffff ? (int)hi : 0x7fffffff);
      static_for<PFR>([&](auto I) {
        constexpr int k = decltype(I)::value;
        const int i = tid + 256 * k;
        const int row = i / RAWDW, j = i - row * RAWDW;
        const int y = R0 + row - 3;
        const int voff = row * (kPatch * 3) + 4 * j;
        const bool ok = i < NRAW && (unsigned)y < (unsigned)kPatch && voff >= lo32 && voff <= hi32;
        const unsigned v = *reinterpret_cast<const unsigned*>(src + tile_base + (ok ? voff : lo32));
        praw[k] = ok ? v : 0u;
      });
    } else {
      const char* img = reinterpret_cast<const char*>(xin) + (size_t)b * kPadH * kPadW * 8;
      static_for<PF>([&](auto I) {
        constexpr int k = decltype(I)::value;
        const int i = tid + 256 * k;
        const int row = i / PPR, cp = i - row * PPR;
        const int R = R0 + row, C = C0 + 2 * cp;
        const bool ok = i < NPIECE && R >= 0 && C >= 0 && R < kPadH && C + 1 < kPadW;
        const u32x4 v = *reinterpret_cast<const u32x4*>(img + (ok ? ((size_t)R * kPadW + C) * 8 : 0));
        pre[k] = ok ? v : u32x4{0u, 0u, 0u, 0u};
      });
    }
  };
  auto stash = [&](int buf) {  // registers -> LDS (patch buffer, or the raw window when U8IN)
    if constexpr (U8IN) {
      static_for<PFR>([&](auto I) {
        constexpr int k = decltype(I)::value;
        const int i = tid + 256 * k;
        if (i < NRAW) *reinterpret_cast<unsigned*>(Rl + i * 4) = praw[k];
      });
    } else {
      static_for<PF>([&](auto I) {
        constexpr int k = decltype(I)::value;
        const int i = tid + 256 * k;
        if (i < NPIECE) *reinterpret_cast<u32x4*>(smem + buf * P_BYTES + i * 16) = pre[k];
      });
    }
  };
  // U8IN only: raw window -> normalised T NHWC4 patch (pairs of pixels = 16-byte pieces)
  auto convert = [&](int tile, int buf) {
    int b, py0, px0;
    tile_origin(tile, b, py0, px0);
    const int R0 = 2 * (2 * py0 - 1), C0 = 2 * (2 * px0 - 1);
    const int sh = ((C0 - 3) * 3) & 3;  // misalignment of the window's first byte: the same for every row
    for (int i = tid; i < NPIECE; i += 256) {
      const int row = i / PPR, cp = i - row * PPR;
      const int y = R0 + row - 3;
      const unsigned char* rr = Rl + row * RAWROW + sh + 6 * cp;  // byte of channel 0 of the first pixel
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)y < (unsigned)kPatch) {
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          const int x = C0 + 2 * cp + px - 3;
          if ((unsigned)x < (unsigned)kPatch) {
            const unsigned c0 = Ll[rr[3 * px + 0]], c1 = Ll[256 + rr[3 * px + 1]], c2 = Ll[512 + rr[3 * px + 2]];
            v[2 * px] = c0 | (c1 << 16);
            v[2 * px + 1] = c2;
          }
        }
      }
      *reinterpret_cast<u32x4*>(smem + buf * P_BYTES + i * 16) = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < n_tiles) {
    fetch(tile);
    stash(0);
    if constexpr (U8IN) {
      __syncthreads();
      convert(tile, 0);
    }
  }
  __syncthreads();
  for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
    const int buf = it & 1;
    const bool more = tile + (int)gridDim.x < n_tiles;
    if (more) fetch(tile + gridDim.x);  // in flight behind the MFMAs
    int b, py0, px0;
    tile_origin(tile, b, py0, px0);
    const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
    const unsigned char* Pl = smem + buf * P_BYTES;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#ifdef HIPAC_ABL_STEM_NO_MFMA
    if (n_tiles < 0)
#endif
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const frag*>(Pl + a_rd[i] + kh * (PCOLS * 8) + kk * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = E::mfma(wreg[j][kh][kk], af[i], acc[i][j]);
      }
    }
    // bias + ReLU -> LDS stem tile [pixel][64 ch]; out-of-image stem pixels become 0
#ifdef HIPAC_ABL_STEM_NO_EPI
    if (n_tiles < 0)
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool inside = (sy0 + ly[i]) >= 0 && (sx0 + lx[i]) >= 0 && P[i] < NPX;
      const unsigned inside_mask = inside ? 0xffffffffu : 0u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          typename E::vec4 ov;
          ov[0] = (T)fmaxf(acc[i][j][4 * q + 0] + bv[j][q].x, 0.f);
          ov[1] = (T)fmaxf(acc[i][j][4 * q + 1] + bv[j][q].y, 0.f);
          ov[2] = (T)fmaxf(acc[i][j][4 * q + 2] + bv[j][q].z, 0.f);
          ov[3] = (T)fmaxf(acc[i][j][4 * q + 3] + bv[j][q].w, 0.f);
          // out-of-image stem pixels become +0: one AND per packed dword instead of a select per value
          u32x2 pk = __builtin_bit_cast(u32x2, ov);
          pk[0] &= inside_mask;
          pk[1] &= inside_mask;
          *reinterpret_cast<u32x2*>(Sl + sidx[i] * SPX + (j * 32 + 8 * q + 4 * h) * 2) = pk;
        }
    }
    __syncthreads();  // stem tile complete; every wave is past its reads of patch[buf ^ 1]
    if (more) stash(buf ^ 1);  // patch[buf^1] (raw window) was last read in the previous iteration
    // 3x3/2 max-pool of the tile: pooled pixel x 8 channels per thread item
#ifdef HIPAC_ABL_STEM_NO_POOL
    if (n_tiles < 0)
#endif
    for (int item = tid; item < PTH * PTW * 8; item += 256) {
      const int pp = item >> 3, c8 = item & 7;
      const int py = pp / PTW, px = pp - py * PTW;
      // values are post-ReLU (sign bit clear, or -0.0): their 16-bit patterns order like
      // signed integers, so the max is a packed integer max -- no conversions
      s16x8 best = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          best = __builtin_elementwise_max(
              best, *reinterpret_cast<const s16x8*>(Sl + ((2 * py + dy) * STW + 2 * px + dx) * SPX + c8 * 16));
      const frag o = __builtin_bit_cast(frag, best);
      *reinterpret_cast<frag*>(out + (((size_t)b * 56 + py0 + py) * 56 + px0 + px) * 64 + c8 * 8) = o;
    }
    __syncthreads();  // pooling reads done (stem tile free) and patch[buf ^ 1] / raw window visible
    if constexpr (U8IN) {
#ifndef HIPAC_ABL_STEM_NO_CONVERT
      if (more) convert(tile + gridDim.x, buf ^ 1);
#endif
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fused stem, uint8 input, second form ("strip" kernel, stem_pool_strip2_kernel below): the same 7x7/2 conv +
// BN + ReLU + 3x3/2 max-pool on raw uint8 HWC patches, restructured around the MFMA loop:
//   * ToTensor / Normalize are folded into the weights and the bias at pack time: the kernel
//     multiplies the centred byte value v - 128 (an exact integer in bf16 and fp16) by w'' = w / (255 std_c)
//     and a bias table carries sum(w'' (128 - mu''_c)), mu''_c = 255 mean_c, over the taps inside the image and
//     128 sum(w'') over the taps outside (bytes there arrive as 0, i.e. -128, where the reference pads with the
//     normalised 0).  No per-pixel rounding of the input.
//   * K is packed as (channel plane c, row pair rp, column quad cq) = 3 x 4 x 2 fragments of 8
//     = 192 (147 real), 12 k16 steps instead of 14.  In LDS a plane holds, per column, the two rows
//     of a row pair in one dword, so the fragment of stem column sx (input columns 2sx-3+4cq ..+3,
//     rows 2sy-3+2rp, +1) is 16 contiguous bytes at 8 * sx + const: two conflict-free ds_read_b64,
//     no replication of the patch, offsets are immediates.
//   * lane = stem COLUMN, MFMA sub-tile = stem ROW: the 3x3/2 max-pool is a v_max3 over three
//     accumulator sets (rows) of the same lane and two lane shifts; the stem tile never goes to
//     LDS.  A workgroup walks DOWN a strip of 28 pooled columns (half the image width; wave =
//     (channel half, 14-column half)), 4 pooled rows = 8 stem rows per step, and carries the last stem
//     row in registers into the next step, so no stem row is computed twice in y
//     (MFMA efficiency = 28/32 columns x 147/192 of K).
//   * input rows arrive by LDS-DMA (buffer_load ... lds, dword pieces, range-checked) one step
//     ahead; the conversion reads them as aligned dwords and writes 16-byte pieces.
// ---------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void buffer_load_lds4(rsrc_t rs, void* lds, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 4, voffset, soffset, 0, 0);
}
#else
__device__ inline void buffer_load_lds4(rsrc_t, void*, int, int) {}
#endif
constexpr int kStripSteps = 14;  // 56 pooled rows / 4 per step

// ---------------------------------------------------------------------------------------
// Schedule: ONE 8-wave workgroup per CU holds two TEAMS of four waves (team = wave >> 2, i.e. the two
// waves that share a SIMD belong to different teams).  A team works on its own strips with its own LDS;
// its step is cut into two halves that alternate behind ONE workgroup barrier per half:
//     H1(n): request the input rows of step n+1 (LDS-DMA) . the 96-MFMA loop of step n
//     H2(n): epilogue of step n (pooling in registers, store) . conversion of the rows of step n+1
// and team B runs one half behind team A: whenever one wave of a SIMD is in its MFMA loop its partner is
// in the VALU / LDS / store half -- matrix beside vector work, never matrix beside matrix.  Every wave
// converts exactly the raw rows it requested itself (its own counted vmcnt orders them), so no barrier
// is needed inside a half.  The bias is the initial accumulator; ReLU is one packed integer max after
// the x max; the pooled rows leave through per-wave LDS staging as 64 contiguous bytes per pixel.
// ---------------------------------------------------------------------------------------
#ifndef HIPAC_STRIP_PRIO
#define HIPAC_STRIP_PRIO 2  // 0: no s_setprio, 1: around the MFMA loop, 2: on the vector half
#endif
// SPLIT (fp16x3 mode): the byte values are exact in fp16, so only the weights are pairs: `wgt` holds the hi halves
// [64][192] followed by the lo halves [64][192]; hi stays in registers, lo is fetched from LDS per channel plane, and
// every fragment feeds two MFMAs per (row, row pair).  The pooling stays in fp32 (v_max3 in y, two DPP shifts in x,
// ReLU) and the pooled rows leave as (hi, lo) pairs [pixel][hi: 64 | lo: 64], hi then lo through the same staging.
#ifndef HIPAC_NT_STEM
#define HIPAC_NT_STEM 0  // 1: the strip stem's pooled-map stores are non-temporal (see HIPAC_NT_STORES)
#endif
// Q8 (precision fp16q8): the pooled map's q8 tensor [pixel][lo8: 64 | hi8: 64] (halo16x2.h) is written too, from the same staging.
template <typename T, bool SPLIT = false, bool Q8 = false>
__global__ __launch_bounds__(512, 2) void stem_pool_strip2_kernel(const unsigned char* __restrict__ x,
                                                                  const T* __restrict__ wgt,
                                                                  const float* __restrict__ btab, T* __restrict__ out,
                                                                  int n_strips, int in_bytes, unsigned char* __restrict__ out_q = nullptr) {
  using E = Elem<T>;
  using frag = typename E::frag;
  static_assert(!SPLIT || std::is_same<T, _Float16>::value, "split pairs are fp16");
  static_assert(!Q8 || SPLIT, "the q8 tensor belongs to the pair layout");
  constexpr int OPIX = SPLIT ? 128 : 64;              // elements per output pixel
  constexpr int WLO_BYTES = SPLIT ? 2 * 12 * 1024 : 0;  // low weight halves in fragment order: [channel half][k16 step][lane] x 16 B
  constexpr int NRP = 11, PXW = 128;
  constexpr int PLANE = NRP * PXW * 4;
  constexpr int PATCH_BYTES = 3 * PLANE;              // 16 896
  constexpr int RAW_PITCH = 512, RAW_ROWS = 22;        // one 16-byte DMA instruction brings two rows (lanes 0-24, 32-56)
  constexpr int RAW_BYTES = RAW_ROWS * RAW_PITCH;     // 11 616
  constexpr int TEAM_BYTES = PATCH_BYTES + RAW_BYTES;
  constexpr int CARRY_BYTES = 512 * 64;               // per lane 16 floats: the raw last stem row of the previous step
  constexpr int STG_BYTES = 4 * 14 * 64;              // per wave: 4 pooled rows x 14 pixels x 32 channels of T
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TEAM_BYTES + CARRY_BYTES + 8 * STG_BYTES + 4096 + WLO_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3;          // team, wave inside the team
  const int jt = tw & 1, st = tw >> 1;                // channel half, 14-column half of the strip
  const int r = lane & 31, h = lane >> 5;
  unsigned char* const Pl = smem + team * TEAM_BYTES;
  unsigned char* const Rl = Pl + PATCH_BYTES;
  unsigned char* const Cl = smem + 2 * TEAM_BYTES + tid * 16;  // chunk k of this lane at + k * 8192 (conflict-free)
  // output staging of this wave (private: no barrier): [pooled row q][pixel k][64 B]; written by the lanes that
  // hold a pooled pixel (8 bytes each), read back as 224 linear 16-byte chunks and stored 64 contiguous
  // bytes per pixel -- per-lane 8-byte stores to 28 different lines cost 2 700 cycles per step
  unsigned char* const Sl = smem + 2 * TEAM_BYTES + CARRY_BYTES + wave * STG_BYTES;
  // Initial accumulators, in LDS (a global load inside the step loop would wait -- vmcnt retires in order -- for the
  // rows just requested): table [row class][column class][64 channels] = folded bias + the border correction.
  // Every byte outside the image arrives as 0 (rows and whole dwords of columns are zero-filled by the DMA's range
  // check: nothing to mask in the conversion) and is fed as 0 - 128, while the reference pads with the normalised 0,
  // i.e. the byte value mu_c = 255 mean_c: the difference depends only on which taps are outside -- stem row
  // 0 / 1 / 111 / other x stem column 0 / 1 / 111 / other -- and is part of the table.
  float* const Bl = reinterpret_cast<float*>(smem + 2 * TEAM_BYTES + CARRY_BYTES + 8 * STG_BYTES);
  for (int i = tid; i < 16 * 64; i += 512) Bl[i] = btab[i];  // visible after the first phase barrier (first read: H1(0))
  int s_off[4];  // element offset of chunk lane + 64 m from the step's first pixel
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = lane + 64 * m, pix = c >> 2;
    const int q = (pix * 147) >> 11, k = pix - 14 * q;
    // SPLIT: the staging holds two pooled rows at a time, hi rows then lo rows: staged row q = (half, row & 1)
    s_off[m] = SPLIT ? ((q & 1) * 56 + k) * 128 + (q >> 1) * 64 + (c & 3) * 8 : (q * 56 + k) * 64 + (c & 3) * 8;
  }
  // Q8: staged items c < 112 are the hi halves of (pooled row c / 56, pixel, 8 channels), item c + 112 the lo halves of the same
  int q_off[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int c = lane + 64 * m, pix = c >> 2;
    const int q = pix >= 14 ? 1 : 0, k = pix - 14 * q;
    q_off[m] = (q * 56 + k) * 128 + (c & 3) * 8;
  }

  const float unscale = SPLIT ? btab[16 * 64] : 1.f;  // 2^-S of the split weights' scale (pack_stem_u8)
  frag wreg[12];
  unsigned char* const Wl = smem + 2 * TEAM_BYTES + CARRY_BYTES + 8 * STG_BYTES + 4096 + jt * (12 * 1024) + lane * 16;
  {
    const char* wb = reinterpret_cast<const char*>(wgt) + (size_t)(jt * 32 + r) * (192 * 2) + 16 * h;
    if constexpr (SPLIT) {
      if (team == 0 && st == 0) {  // one wave per channel half parks the lo fragments (visible after the first phase barrier)
#pragma unroll
        for (int s = 0; s < 12; ++s)
          *reinterpret_cast<frag*>(Wl + s * 1024) = *reinterpret_cast<const frag*>(wb + 64 * 192 * 2 + s * 32);
      }
    }
#pragma unroll
    for (int s = 0; s < 12; ++s) wreg[s] = *reinterpret_cast<const frag*>(wb + s * 32);
#pragma unroll
    for (int s = 0; s < 12; ++s) asm volatile("" ::"v"(wreg[s]));
  }
  const rsrc_t in_rsrc = make_rsrc(x, in_bytes);

  // rows owned by this wave: window rows 6 tw .. 6 tw + 5 (row pairs 3 tw .. 3 tw + 2); wave 3 owns rows
  // 18 .. 20 (row 21 does not exist)
  // (wave 3: rows 18 .. 20; its fourth request covers row 20 and the nonexistent row 21 -> zeros)
  auto issue_dma = [&](int strip, int ys) {
    const int b = strip >> 1, side = strip & 1;
    const int base = b * (kPatch * kPatch * 3) + (16 * ys - 3) * (kPatch * 3) + 3 * (112 * side - 5) - 1;  // multiple of 16
    // lane -> (row of the pair, 16-byte chunk); chunks outside the image columns read as zeros: side 0: chunk 0
    // (bytes 0..15 = columns -5..-1), side 1: chunks 22.. (columns 224..)
    const int sub = lane >> 5, ch16 = lane & 31;
    const bool col_ok = ch16 < 25 && (side == 0 ? ch16 >= 1 : ch16 < 22);
    static_for<3>([&](auto K) {
      constexpr int k = decltype(K)::value;
      if (tw < 3 || k < 2) {
        const int row = 6 * tw + 2 * k + sub;
        const int iy = 16 * ys - 3 + row;
        const bool ok = col_ok && (unsigned)iy < (unsigned)kPatch && row < 21;
        buffer_load_lds16(in_rsrc, Rl + (6 * tw + 2 * k) * RAW_PITCH, ok ? base + row * (kPatch * 3) + ch16 * 16 : (int)0x80000000, 0);
      }
    });
  };

  // conversion task of this lane: row pair 3 tw + (lane >> 4) (lanes 48..63 idle; wave 3: lanes 32..63), 8 columns
  const int cRp = 3 * tw + (lane >> 4), cxg = lane & 15;
  const bool ctask = (lane >> 4) < (tw < 3 ? 3 : 2);
  const unsigned char* const craw = Rl + (2 * cRp) * RAW_PITCH + cxg * 24;
  unsigned char* const cdst = Pl + cRp * (PXW * 4) + cxg * 32;
  auto convert = [&](int strip, int ys) {
    (void)strip, (void)ys;
    if (!ctask) return;
    unsigned da[7], db[7];
    {
      const u32x2 a0 = *reinterpret_cast<const u32x2*>(craw), a1 = *reinterpret_cast<const u32x2*>(craw + 8),
                  a2 = *reinterpret_cast<const u32x2*>(craw + 16);
      const u32x2 b0 = *reinterpret_cast<const u32x2*>(craw + RAW_PITCH),
                  b1 = *reinterpret_cast<const u32x2*>(craw + RAW_PITCH + 8),
                  b2 = *reinterpret_cast<const u32x2*>(craw + RAW_PITCH + 16);
      da[0] = a0[0], da[1] = a0[1], da[2] = a1[0], da[3] = a1[1], da[4] = a2[0], da[5] = a2[1];
      db[0] = b0[0], db[1] = b0[1], db[2] = b1[0], db[3] = b1[1], db[4] = b2[0], db[5] = b2[1];
      da[6] = *reinterpret_cast<const unsigned*>(craw + 24);
      db[6] = *reinterpret_cast<const unsigned*>(craw + RAW_PITCH + 24);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      unsigned o[8];
#pragma unroll
      for (int px = 0; px < 8; ++px) {
        const int w = 1 + 3 * px + c;  // byte inside the dword run
        // v - 128: still an exact integer in bf16 / fp16, and the fp32 accumulation no longer carries the DC term
        // 128 sum(w) (bytes outside the image arrive as 0 -> -128: the bias table accounts for them)
        o[px] = PackPair<T>::pack((float)((da[w >> 2] >> (8 * (w & 3))) & 0xffu) - 128.f,
                                  (float)((db[w >> 2] >> (8 * (w & 3))) & 0xffu) - 128.f);
      }
      *reinterpret_cast<u32x4*>(cdst + c * PLANE) = u32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<u32x4*>(cdst + c * PLANE + 16) = u32x4{o[4], o[5], o[6], o[7]};
    }
  };

  const unsigned char* const fbase = Pl + (2 * r + 56 * st) * 4 + 16 * h;
  const int tg = 2 * blockIdx.x + team, tstride = 2 * gridDim.x;   // this team's strips: tg, tg + tstride, ...
  const int my_strips = tg < n_strips ? (n_strips - tg + tstride - 1) / tstride : 0;
  const int n_steps = my_strips * kStripSteps;
  // workgroup-uniform phase count: 2 halves per step + the prologue half, team B one phase behind
  const int max_strips = (n_strips - 2 * (int)blockIdx.x + tstride - 1) / tstride;  // team A's count >= team B's
  const int n_phases = 2 * max_strips * kStripSteps + 2;

#ifdef HIPAC_HALO_STAMPS
  unsigned long long z_sum[6] = {0, 0, 0, 0, 0, 0};
#endif
  f32x16 acc[8];
  const bool first_col_wave = st == 0;  // with side == 0: lane r == 0 is stem column -1
  for (int p = 0; p < n_phases; ++p) {
    const int hs = p - team - 1;            // half index of this team: -1 = prologue, 2n = H1(n), 2n+1 = H2(n)
    HALO_STAMP(z_t0);
    if (hs >= -1 && hs < 2 * n_steps) {
      const int n = hs >> 1;                // step (floor: -1 for the prologue)
      if (hs & 1) {
        // ---------------- H2(n): epilogue of step n, conversion of step n + 1 ----------------
#if HIPAC_STRIP_PRIO == 2
        __builtin_amdgcn_s_setprio(1);  // the vector half goes first: its partner needs one issue slot per 32 cycles
#endif
        const int strip = tg + ((n < 0 ? 0 : n) / kStripSteps) * tstride;
        const int ys = (n < 0 ? 0 : n) % kStripSteps;
#ifdef HIPAC_ABL_STRIP_NO_EPI
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i]));
        if (n_strips < 0)
#endif
        if (n >= 0) {
          const int b = strip >> 1, side = strip & 1;
          f32x16 carry;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(Cl + k * 8192);
#pragma unroll
            for (int t = 0; t < 4; ++t) carry[4 * k + t] = ys == 0 ? -3.0e38f : cv[t];  // stem row -1 lies outside the image
          }
          const bool writer = (r & 1) == 0 && r <= 26;
          const bool col_m1 = side == 0 && first_col_wave && r == 0;  // this lane holds stem column -1
#ifdef HIPAC_ABL_STRIP_STORE_LOCAL
          T* const dst0 = out + (size_t)blockIdx.x * 16384 + jt * 32 + (b & 0);
#else
          T* const dst0 = out + ((((size_t)b * 56 + 4 * ys) * 56 + 28 * side + 14 * st) * OPIX + jt * 32);
#endif
          typedef __attribute__((ext_vector_type(2))) short s16x2;
          // SPLIT: rows go out two at a time (hi halves in staging rows 0, 1, lo halves in rows 2, 3)
          auto flush_pair = [&](int g) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's staging writes
#pragma unroll
            for (int m = 0; m < 4; ++m)
              if (m < 3 || lane < 32)
                store16_out<HIPAC_NT_STEM != 0>(dst0 + g * (2 * 56 * 128) + s_off[m], *reinterpret_cast<const u32x4*>(Sl + (lane + 64 * m) * 16));
            if constexpr (Q8) {
              unsigned char* const qdst0 = out_q + ((((size_t)b * 56 + 4 * ys) * 56 + 28 * side + 14 * st) * 128 + jt * 32) + g * (2 * 56 * 128);
#pragma unroll
              for (int m = 0; m < 2; ++m)
                if (m < 1 || lane < 48) {
                  const f16x8 hv = *reinterpret_cast<const f16x8*>(Sl + (lane + 64 * m) * 16);
                  const f16x8 lv = *reinterpret_cast<const f16x8*>(Sl + (lane + 64 * m + 112) * 16);
                  u32x2 h8, l8;
#pragma unroll
                  for (int k = 0; k < 2; ++k) {
                    h8[k] = cvt4_e4m3((float)hv[4 * k], (float)hv[4 * k + 1], (float)hv[4 * k + 2], (float)hv[4 * k + 3]);
                    l8[k] = cvt4_e4m3((float)lv[4 * k] * kQ8LoScale, (float)lv[4 * k + 1] * kQ8LoScale, (float)lv[4 * k + 2] * kQ8LoScale,
                                      (float)lv[4 * k + 3] * kQ8LoScale);
                  }
                  *reinterpret_cast<u32x2*>(qdst0 + q_off[m]) = l8;
                  *reinterpret_cast<u32x2*>(qdst0 + q_off[m] + 64) = h8;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the rows are overwritten
          };
          // one pooled row at a time: y max (fp32, v_max3), round to T, then the x max of lanes r, r+1, r+2 by two
          // DPP wave shifts and ReLU, both on the 16-bit patterns as SIGNED integers -- among non-negative
          // floats that is the float order, every negative float is below every non-negative one, and the
          // final max with +0 removes whatever negative value is left
          static_for<4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            unsigned pk[8];
            unsigned lk_prev = 0;
            (void)lk_prev;
            if constexpr (SPLIT) {
              // fp32 all the way: y max, x max of lanes r, r+1, r+2 (two wave shifts), ReLU; then the (hi, lo) split
#pragma unroll
              for (int d = 0; d < 8; ++d) {
                float pv[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                  const int e = 2 * d + t;
                  const float top = q == 0 ? carry[e] : acc[q == 0 ? 0 : 2 * q - 1][e];
                  const float a0 = col_m1 ? -3.0e38f : fmaxf(fmaxf(top, acc[2 * q][e]), acc[2 * q + 1][e]);
                  const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x130, 0xf, 0xf, false));
                  const float t1 = fmaxf(a0, a1);
                  const float u2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t1), 0x130, 0xf, 0xf, false));
                  pv[t] = fmaxf(fmaxf(t1, u2), 0.f) * unscale;
                }
                const f16x2 h2 = __builtin_convertvector(f32x2{pv[0], pv[1]}, f16x2);
                const f16x2 l2 = __builtin_convertvector(f32x2{pv[0] - (float)h2[0], pv[1] - (float)h2[1]}, f16x2);
                pk[d] = __builtin_bit_cast(unsigned, h2);
                if (writer && (d & 1))  // (d - 1, d) = one 8-byte item of channel quad d / 2
                  *reinterpret_cast<u32x2*>(Sl + ((2 + (q & 1)) * 14 + (r >> 1)) * 64 + (d >> 1) * 16 + h * 8) =
                      u32x2{lk_prev, __builtin_bit_cast(unsigned, l2)};
                lk_prev = __builtin_bit_cast(unsigned, l2);
              }
            } else {
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
              float v[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int e = 4 * cq + t;
                const float top = q == 0 ? carry[e] : acc[q == 0 ? 0 : 2 * q - 1][e];
                v[t] = fmaxf(fmaxf(top, acc[2 * q][e]), acc[2 * q + 1][e]);
              }
              pk[2 * cq] = PackPair<T>::pack_rn(v[0], v[1]);
              pk[2 * cq + 1] = PackPair<T>::pack_rn(v[2], v[3]);
            }
            }
            if constexpr (!SPLIT)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
              const unsigned a0 = col_m1 ? 0x80008000u : pk[d];  // -0.0: below every value as int16, never wins
              const unsigned a1 = __builtin_amdgcn_update_dpp(0u, a0, 0x130, 0xf, 0xf, false);  // wave_shl:1
              const s16x2 t1 = __builtin_elementwise_max(__builtin_bit_cast(s16x2, a0), __builtin_bit_cast(s16x2, a1));
              const unsigned u2 = __builtin_amdgcn_update_dpp(0u, __builtin_bit_cast(unsigned, t1), 0x130, 0xf, 0xf, false);
              const s16x2 t2 = __builtin_elementwise_max(t1, __builtin_bit_cast(s16x2, u2));
              pk[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(t2, s16x2{0, 0}));
            }
            if (writer) {
#pragma unroll
              for (int cq = 0; cq < 4; ++cq)
                *reinterpret_cast<u32x2*>(Sl + ((SPLIT ? (q & 1) : q) * 14 + (r >> 1)) * 64 + cq * 16 + h * 8) = u32x2{pk[2 * cq], pk[2 * cq + 1]};
            }
            if constexpr (SPLIT && q == 1) flush_pair(0);
          });
#pragma unroll
          for (int k = 0; k < 4; ++k)
            *reinterpret_cast<f32x4*>(Cl + k * 8192) = f32x4{acc[7][4 * k], acc[7][4 * k + 1], acc[7][4 * k + 2], acc[7][4 * k + 3]};
          if constexpr (SPLIT) {
            flush_pair(1);
          } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's staging writes (LDS operations of a wave complete in order)
#ifdef HIPAC_ABL_STRIP_NO_STORE
          if (n_strips < 0)
#endif
#pragma unroll
          for (int m = 0; m < 4; ++m)
            if (m < 3 || lane < 32)
              store16_out<HIPAC_NT_STEM != 0>(dst0 + s_off[m], *reinterpret_cast<const u32x4*>(Sl + (lane + 64 * m) * 16));
          }
        }
        HALO_STAMP(z_te);
#ifdef HIPAC_HALO_STAMPS
        if (n >= 0) z_sum[1] += z_te - z_t0, z_sum[4] += 1;
#endif
        if (n + 1 < n_steps) {
          const int gn = n + 1;
          if (n < 0) {
            issue_dma(tg, 0);  // prologue: nothing was requested yet
            wait_vmcnt<0>();
          } else {
            wait_vmcnt<Q8 ? 16 : SPLIT ? 8 : 4>();  // this wave's raw rows of step n+1 (requested at the end of H2(n-1)) are older than its 4 (8; Q8: 16) stores
          }
          HALO_STAMP(z_tw);
#ifdef HIPAC_ABL_STRIP_NO_CONVERT
          if (n_strips < 0)
#endif
          convert(tg + (gn / kStripSteps) * tstride, gn % kStripSteps);
#ifdef HIPAC_HALO_STAMPS
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          HALO_STAMP(z_tc);
          if (n >= 0) z_sum[2] += z_tw - z_te, z_sum[3] += z_tc - z_tw;
#endif
          // the raw rows are consumed (this wave converts only rows it requested itself): request those of step n + 2
          // now, a whole MFMA half ahead of their use -- issuing them at the start of H1 sat in front of the MFMAs
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef HIPAC_ABL_STRIP_NO_DMA
          if (n_strips < 0)
#endif
          if (gn + 1 < n_steps) issue_dma(tg + ((gn + 1) / kStripSteps) * tstride, (gn + 1) % kStripSteps);
        }
      } else {
        // ---------------- H1(n): request the rows of step n + 1, MFMA loop of step n ----------------
        f32x16 binit;
        {
          const int strip1 = tg + (n / kStripSteps) * tstride, ys1 = n % kStripSteps, side1 = strip1 & 1;
          // column class of this lane's stem column (0 interior, 1: column 0, 2: column 1, 3: column 111)
          const int cc = (side1 == 0 && st == 0) ? (r == 1 ? 1 : (r == 2 ? 2 : 0)) : ((side1 == 1 && st == 1 && r == 28) ? 3 : 0);
          const float* bl = Bl + cc * 64 + jt * 32 + 4 * h;
          // rows 2..6 never have a special class: their accumulators start as the C operand of their first MFMA
          // (binit, 16 registers) instead of 80 v_mov; rows 0, 1, 7 are set here (from the row-class table when needed)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(bl + 8 * q);
            binit[4 * q + 0] = v.x, binit[4 * q + 1] = v.y, binit[4 * q + 2] = v.z, binit[4 * q + 3] = v.w;
          }
          acc[0] = binit, acc[1] = binit, acc[7] = binit;
          if (ys1 == 0 || ys1 == kStripSteps - 1) {  // uniform: stem rows 0, 1 (first step) / 111 (last step) have taps above / below the image
            static_for<3>([&](auto RC) {
              constexpr int rc = decltype(RC)::value + 1;
              constexpr int i = rc == 1 ? 0 : (rc == 2 ? 1 : 7);
              if ((rc == 3) == (ys1 != 0)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float4 v = *reinterpret_cast<const float4*>(bl + rc * 256 + 8 * q);
                  acc[i][4 * q + 0] = v.x, acc[i][4 * q + 1] = v.y, acc[i][4 * q + 2] = v.z, acc[i][4 * q + 3] = v.w;
                }
              }
            });
          }
        }
#if HIPAC_STRIP_PRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
        // Fragment (plane c, row pair j) feeds the up to four MFMAs (i, rp) with i + rp = j, so the loop runs over
        // the 33 fragments, each read ONCE, two fragments ahead of its MFMAs (3-slot register ring, order pinned:
        // left to itself hipcc issues a read right in front of the MFMA that needs it and exposes the LDS latency
        // ~25 times per step).  acc[i] is touched at most once per group of 4 MFMAs: no dependent-MFMA stalls.
#ifdef HIPAC_ABL_STRIP_NO_MFMA
        if (n_strips < 0)
#endif
        {
          frag ring3[3];
          auto rd = [&](auto F) {
            constexpr int f = decltype(F)::value;      // f = c * 11 + j
            const unsigned char* pp = fbase + f * (PXW * 4);
            const u32x2 lo = *reinterpret_cast<const u32x2*>(pp), hi = *reinterpret_cast<const u32x2*>(pp + 8);
            ring3[f % 3] = __builtin_bit_cast(frag, u32x4{lo[0], lo[1], hi[0], hi[1]});
          };
          frag wlo[SPLIT ? 4 : 1];  // SPLIT: the lo weight fragments of the current channel plane
          auto rdw = [&](int c_) {
            if constexpr (SPLIT) {
#pragma unroll
              for (int rp = 0; rp < 4; ++rp) wlo[rp] = *reinterpret_cast<const frag*>(Wl + (c_ * 4 + rp) * 1024);
            }
          };
          rdw(0);
          rd(std::integral_constant<int, 0>{});
          rd(std::integral_constant<int, 1>{});
          static_for<33>([&](auto F) {
            constexpr int f = decltype(F)::value, c = f / NRP, j = f % NRP;
            if constexpr (f + 2 < 33) rd(std::integral_constant<int, f + 2>{});
            static_for<4>([&](auto RP) {
              constexpr int rp = decltype(RP)::value, i = j - rp;
              if constexpr (i >= 0 && i < 8) {
                if constexpr (c == 0 && rp == 0 && i >= 2 && i <= 6) acc[i] = E::mfma(wreg[0], ring3[f % 3], binit);
                else acc[i] = E::mfma(wreg[c * 4 + rp], ring3[f % 3], acc[i]);
              }
            });
            if constexpr (SPLIT) {
              static_for<4>([&](auto RP) {
                constexpr int rp = decltype(RP)::value, i = j - rp;
                if constexpr (i >= 0 && i < 8) acc[i] = E::mfma(wlo[rp], ring3[f % 3], acc[i]);
              });
              if constexpr (j == NRP - 1 && c < 2) rdw(c + 1);  // the next plane's lo fragments, one fragment ahead
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        }
#if HIPAC_STRIP_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef HIPAC_HALO_STAMPS
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(acc[i]));
        HALO_STAMP(z_tm);
        z_sum[0] += z_tm - z_t0;
#endif
      }
    }
    HALO_STAMP(z_tb0);
#if HIPAC_STRIP_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // half boundary: patch written <-> patch read, for both teams
#ifdef HIPAC_HALO_STAMPS
    HALO_STAMP(z_tb1);
    z_sum[5] += z_tb1 - z_tb0;
#endif
  }
#ifdef HIPAC_HALO_STAMPS
  if (lane == 0) {
    atomicAdd(&g_halo_stamps[0], z_sum[0]);  // H1: DMA issue + bias + MFMA loop
    atomicAdd(&g_halo_stamps[1], z_sum[1]);  // H2: epilogue
    atomicAdd(&g_halo_stamps[2], z_sum[2]);  // H2: wait for the raw rows
    atomicAdd(&g_halo_stamps[4], z_sum[3]);  // H2: conversion
    atomicAdd(&g_halo_stamps[5], z_sum[5]);  // barrier waits
    atomicAdd(&g_halo_stamps[3], z_sum[4]);  // wave-steps
  }
#endif
}

// ---------------------------------------------------------------------------------------
// layer1 kernel: 3x3 / stride 1 / 64 -> 64 channels on the 56x56 map (4 of the 20 convs,
// 25 % of the FLOPs, and the largest activations after the stem).
//   * persistent workgroups (grid-stride over work units), 4 waves
//   * a unit = two 8x8 output tiles; each tile's 10x10 input halo (64 ch = 128 B per pixel)
//     is brought into LDS by LDS-DMA: 200 pixels per 128 outputs = 1.56x re-read
//   * the halos are DOUBLE-BUFFERED ACROSS UNITS: the DMA of unit u+1 is issued at the top of
//     unit u and has the whole unit (72 MFMAs + epilogue) to land -- one workgroup barrier per unit
//   * K = 576 is short enough for every lane to keep ITS weight fragments for all 9 taps in
//     registers (36 fragments = 144 VGPRs; wave = one tile x 32 channels), so weights are
//     fetched once per workgroup and LDS serves only activation fragments, with NO barrier
//     inside a unit's 72-MFMA loop
//   * LDS placement of halo pixel (hy,hx): slot hy*10+hx, 16-byte chunk c stored at
//     c ^ (((hx>>1)&1) | ((hy&3)<<1)).  For every tap, a ds_read_b128 lane group (4 runs of 4
//     consecutive x on 4 consecutive rows) then hits 16 distinct 16-byte slots: conflict-free
//   * epilogue per WAVE (no barrier): each 32-pixel x 32-channel sub-tile goes through the
//     wave's private fp32 staging rows and leaves as 16-byte items, 64 contiguous bytes per
//     pixel; the residual items are prefetched into registers one sub-tile ahead
// ---------------------------------------------------------------------------------------
template <typename T, bool RESID, bool RELU = true>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                             const float* __restrict__ bias,
                                                             const T* __restrict__ resid, T* __restrict__ out,
                                                             int n_tiles, const char* __restrict__ zero_page) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int H = 56, W = 56, C = 64, TPI = 49;  // 7 x 7 tiles of 8 x 8 per image
  constexpr int HALO = 10, HPIECES = 13;           // 100 halo pixels -> 13 pieces of 8
  constexpr int H_BYTES = HPIECES * 1024;
  constexpr int U_BYTES = 2 * H_BYTES;             // the two halos of a unit
  constexpr int SROW = 144;                        // staging row: 32 fp32 + 16 B pad
  constexpr int SPX = RESID ? 16 : 32;             // pixels staged at a time (RESID: LDS also holds the residual)
  constexpr int SW_BYTES = SPX * SROW;             // one wave's staging
  constexpr int R_BYTES = RESID ? 4 * 4096 : 0;    // residual of the unit: per wave 2 sub-tiles x 32 px x 64 B
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * U_BYTES + R_BYTES + 4 * SW_BYTES + 64 * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wt = wave & 1, wn = wave >> 1;  // tile of the pair, channel half
  const int r = lane & 31, h = lane >> 5;
  unsigned char* const Rl = smem + 2 * U_BYTES + wave * 4096;
  unsigned char* const Sl = smem + 2 * U_BYTES + R_BYTES + wave * SW_BYTES;
  float* const Bl = reinterpret_cast<float*>(smem + 2 * U_BYTES + R_BYTES + 4 * SW_BYTES);  // bias
  if (tid < 64) Bl[tid] = bias[tid];  // visible after the first unit's barrier

  // weights of channels wn*32 + r, all 9 taps x 64 input channels, in registers
  frag wreg[9][4];
  {
    const char* wb = reinterpret_cast<const char*>(wgt) + (size_t)(wn * 32 + r) * (9 * C * 2) + 16 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wreg[tap][kk] = *reinterpret_cast<const frag*>(wb + tap * 128 + kk * 32);
  }
  // epilogue item k of a sub-tile: pixel e_px + 16k (0..31), channels wn*32 + e_c8*8 .. +7
  const int e_c8 = lane & 3, e_px = lane >> 2;

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const int prow = lane >> 3, dchunk = lane & 7;
  const char* in_b = reinterpret_cast<const char*>(in);

  // LDS-DMA of the two halos of unit u (tiles 2u, 2u+1) into buffer `buf`; 26 pieces over 4 waves.
  // Everything about a piece that does not depend on the tile is computed once: the byte offset of
  // this lane's 16 bytes relative to the tile's first pixel, with 4 edge bits in its low nibble
  // (halo row 0 / row 9 / column 0 / column 9), so a unit costs a handful of VALU per piece.
  // Slots 100..103 of a halo do not exist: they re-fetch pixel 99 (never read).
  constexpr int NPW = (2 * HPIECES + 3) / 4;  // pieces per wave (7; waves 2, 3 have 6)
  int piece_pk[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int p = wave + 4 * k;
    const int pp = p >= HPIECES ? p - HPIECES : p;
    int q = pp * 8 + prow;
    q = q < HALO * HALO ? q : HALO * HALO - 1;
    const int hy = q / HALO, hx = q - hy * HALO;
    const int sw = ((hx >> 1) & 1) | ((hy & 3) << 1);
    const int rel = ((hy - 1) * W + (hx - 1)) * (C * 2) + (dchunk ^ sw) * 16;
    piece_pk[k] = rel | (hy == 0 ? 1 : 0) | (hy == HALO - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == HALO - 1 ? 8 : 0);
  }
  auto issue_unit = [&](int u, int buf) {
    const char* tbase[2];
    int tmask[2];
#pragma unroll
    for (int tsel = 0; tsel < 2; ++tsel) {  // wave-uniform tile scalars
      const int tile = 2 * u + tsel;
      const int b = tile / TPI, t = tile - b * TPI;
      const int ty = t / 7, tx = t - ty * 7;
      tbase[tsel] = in_b + (((size_t)b * H + ty * 8) * W + tx * 8) * (C * 2);
      // bit 4: the tile lies beyond the batch -> every piece reads zeros
      tmask[tsel] = tile < n_tiles ? ((ty == 0 ? 1 : 0) | (ty == 6 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == 6 ? 8 : 0)) : 16;
    }
    static_for<NPW>([&](auto K) {
      constexpr int k = decltype(K)::value;
      const int p = wave + 4 * k;
      if (p < 2 * HPIECES) {
        const int tsel = p >= HPIECES ? 1 : 0;
        const int pp = p - tsel * HPIECES;
        const int tm = tmask[tsel];
        const bool ok = tm != 16 && (piece_pk[k] & tm & 15) == 0;
        const char* src = ok ? tbase[tsel] + (piece_pk[k] & ~15) : zero_page + dchunk * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * U_BYTES + tsel * H_BYTES + pp * 1024),
                                         16, 0, 0);
      }
    });
  };

  // this lane's two output pixels inside its tile: sub-tile i = rows 4i..4i+3; (ly,lx) = (4i + r/8, r%8)
  int lx = r & 7, ly0 = r >> 3;

  const int n_units = (n_tiles + 1) >> 1;
  int u = blockIdx.x;
  // the weight / bias loads above must retire BEFORE the unit loop: otherwise the compiler drains
  // vmcnt -- and with it the prefetched DMA of the next unit -- in front of the first MFMA of
  // every unit (seen in the RESID variant: s_waitcnt vmcnt(0) right after s_setprio 1)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wreg[tap][kk]));  // a use: forces the wait here
  if (u < n_units) issue_unit(u, 0);
#ifdef HIPAC_HALO_STAMPS
  unsigned long long c_sum[5] = {0, 0, 0, 0, 0};
#endif
  bool stored = false;  // wave-uniform: the previous unit's epilogue issued its 4 stores
  for (int it = 0; u < n_units; u += gridDim.x, ++it) {
    const int buf = it & 1;
    HALO_STAMP(c_t0);
    // this unit's halo DMAs are OLDER than the 4 stores of the previous epilogue and vmcnt retires in
    // issue order: leave the stores in flight instead of paying their acknowledgement latency here
    if (stored) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    HALO_STAMP(c_t0b);
    __builtin_amdgcn_s_barrier();  // halos of this unit landed; every wave is past its reads of the other buffer
    HALO_STAMP(c_t1);

    // this wave's tile and the element offset of its epilogue items (pixel e_px + 16k of sub-tile i)
    const bool tile_ok = 2 * u + wt < n_tiles;
    stored = tile_ok;
    const int tile = tile_ok ? 2 * u + wt : n_tiles - 1;  // addresses stay inside the tensors; stores are guarded
    const int tb = tile / TPI, tt = tile - tb * TPI;
    const int ty = tt / 7, tx = tt - ty * 7;
    // pixel (4i + (e_px + 16k) / 8, (e_px + 16k) % 8) of the tile -> NHWC element offset
    auto item_off = [&](int i, int k) -> size_t {
      const int px = e_px + 16 * k;
      const int y = ty * 8 + 4 * i + (px >> 3), x = tx * 8 + (px & 7);
      return (((size_t)tb * H + y) * W + x) * C + wn * 32 + e_c8 * 8;
    };
    // residual of this unit by LDS-DMA: piece j = (sub-tile j/2, item j%2) -- every lane fetches exactly
    // the 16 bytes it adds in the epilogue (item-linear destination, no cross-lane dependency).
    // Issued ahead of the next unit's halos: vmcnt retires in issue order.
    if constexpr (RESID) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(resid + item_off(j >> 1, j & 1)), (lptr_t)(Rl + j * 1024), 16, 0, 0);
    }
    const bool more = u + (int)gridDim.x < n_units;
    if (more) issue_unit(u + gridDim.x, buf ^ 1);  // lands behind this whole unit

    // keep the 18 tap address bases from being hoisted out of the unit loop (they would
    // cost 18 VGPRs next to 144 of weights): make their inputs opaque per iteration
    asm volatile("" : "+v"(lx), "+v"(ly0));
    const unsigned char* const Hl = smem + buf * U_BYTES + wt * H_BYTES;
    // the bias is the initial accumulator (register group q = channels wn*32 + 8q + 4h .. +3), as in block_c64.h
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(Bl + wn * 32 + 8 * q + 4 * h);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = bq[e];
    }
    __builtin_amdgcn_s_setprio(1);
    // 36 k16 steps (tap-major); the two activation fragments of step s + PF are requested before the
    // MFMAs of step s, so an LDS read has PF MFMA pairs (PF x 64 cycles) to return
    constexpr int PF = HIPAC_C64_PF;
    frag ring[PF + 1][2];
    // address of fragment (tap, kk, i) = lane base + compile-time slot offset + swizzled chunk, with the
    // chunk term factored per axis: bit 4 = ((hx >> 1) & 1) ^ h depends on kw only, bits 5-6 =
    // (hy & 3) ^ kk on kh only (hy & 3 does not depend on i) -- 2 VALU per k16 step instead of ~5 per read
    int ax[3], by[3];
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
      ax[k3] = ((((lx + k3) >> 1) & 1) ^ h) << 4;
      by[k3] = ((ly0 + k3) & 3) << 5;
    }
    const unsigned char* const Hb = Hl + ((ly0 * HALO + lx) << 7);
    auto rd_step = [&](auto S) {
      constexpr int st = decltype(S)::value;
      constexpr int kh = c64_step_kh(st), kw = c64_step_kw(st), kk = c64_step_kk(st);
      const unsigned char* const ptr = Hb + (ax[kw] | (by[kh] ^ (kk << 5)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
        ring[st % (PF + 1)][i] = *reinterpret_cast<const frag*>(ptr + (((4 * i + kh) * HALO + kw) << 7));
    };
    static_for<PF>([&](auto S) { rd_step(S); });
    static_for<36>([&](auto S) {
      constexpr int st = decltype(S)::value;
      if constexpr (st + PF < 36) rd_step(std::integral_constant<int, st + PF>{});
#pragma unroll
      for (int i = 0; i < 2; ++i)
        acc[i] = E::mfma(wreg[3 * c64_step_kh(st) + c64_step_kw(st)][c64_step_kk(st)], ring[st % (PF + 1)][i], acc[i]);
      __builtin_amdgcn_sched_barrier(0);  // pin the read-ahead: the scheduler otherwise folds it back to one step
    });
    __builtin_amdgcn_s_setprio(0);
    HALO_STAMP(c_t2);

    // epilogue, per wave: SPX pixels of sub-tile i -> private fp32 rows -> + bias (+ residual) ReLU -> T
    if constexpr (RESID) {
      // this lane's residual pieces have landed once only the next unit's halo DMAs (issued later:
      // 7 per wave for waves 0-1, 6 for waves 2-3) are still outstanding
      if (!more) wait_vmcnt<0>();
      else if (wave < 2) wait_vmcnt<7>();
      else wait_vmcnt<6>();
    }
    static_for<2 * (32 / SPX)>([&](auto PH) {
      constexpr int i = decltype(PH)::value / (32 / SPX), hf = decltype(PH)::value % (32 / SPX);
      if (SPX == 32 || (r >> 4) == hf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v[0] = acc[i][4 * q + 0];
          v[1] = acc[i][4 * q + 1];
          v[2] = acc[i][4 * q + 2];
          v[3] = acc[i][4 * q + 3];
          *reinterpret_cast<f32x4*>(Sl + (r & (SPX - 1)) * SROW + (8 * q + 4 * h) * 4) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave's LDS operations complete in order
#pragma unroll
      for (int k = 0; k < SPX / 16; ++k) {
        const int px = e_px + 16 * k;            // pixel inside the staged rows
        constexpr int kk = SPX == 32 ? 0 : hf;   // item index inside the sub-tile = k (SPX 32) or hf (SPX 16)
        const int ki = SPX == 32 ? k : kk;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Sl + px * SROW + e_c8 * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Sl + px * SROW + e_c8 * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if constexpr (RESID) {
          const frag rvv = *reinterpret_cast<const frag*>(Rl + (2 * i + ki) * 1024 + lane * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rvv[e];
        }
        frag ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (T)(RELU ? fmaxf(v[e], 0.f) : v[e]);  // (training's convolutions carry no ReLU: BN follows)
        if (tile_ok) *reinterpret_cast<frag*>(out + item_off(i, ki)) = ov;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads returned before it is overwritten
    });
#ifdef HIPAC_HALO_STAMPS
    HALO_STAMP(c_t3);
    c_sum[0] += c_t0b - c_t0;  // vmcnt drain (DMA of this unit + own stores)
    c_sum[1] += c_t1 - c_t0b;  // barrier
    c_sum[2] += c_t2 - c_t1;   // DMA issue + 72 MFMAs
    c_sum[3] += c_t3 - c_t2;   // epilogue
    c_sum[4] += 1;
#endif
  }
#ifdef HIPAC_HALO_STAMPS
  if (lane == 0) {
    atomicAdd(&g_halo_stamps[4], c_sum[0]);
    atomicAdd(&g_halo_stamps[5], c_sum[1]);
    atomicAdd(&g_halo_stamps[6], c_sum[2]);
    atomicAdd(&g_halo_stamps[7], c_sum[3]);
    atomicAdd(&g_halo_stamps[3], c_sum[4]);
  }
#endif
}

// ---------------------------------------------------------------------------------------
// layer2 entry kernel: 3x3 / stride 2 / 64 -> 128 channels, 56x56 -> 28x28 (+BN+ReLU) AND the
// block's 1x1 / stride 2 projection shortcut (+BN), one launch, one pass over the input.
// Same skeleton as the layer1 kernel: persistent workgroups, every lane keeps ITS weight
// fragments in registers (wave w = output channels 32w .. 32w+31: 36 fragments of the 3x3 conv
// + 4 of the projection = 160 VGPRs), LDS serves only activation fragments, halos double-buffered
// across units by LDS-DMA, one workgroup barrier per unit, per-wave barrier-free epilogues.
//   * unit = one tile of 7 x 4 output pixels (28 of the 32 MFMA columns; 28 tiles per image);
//     all four waves read the same activation fragments (different weights)
//   * its 15 x 9 input halo is stored as [row hy][p] with the EVEN columns first
//     (p = (hx & 1) * 8 + hx / 2, 16 slots per row): a tap (kh, kw) then reads consecutive
//     slots p = (kw & 1) * 8 + x + kw / 2 for consecutive output x -- the stride disappears.
//     16-byte chunk c of slot (hy, p) sits at c ^ (((p >> 1) & 1) | (((hy >> 1) & 3) << 1)):
//     every ds_read_b128 lane group covers all 16 bank groups (simulated: 4.0 LDS cycles per read)
//   * the projection reads exactly the centre tap's fragments: 4 extra MFMAs, no extra LDS reads
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3s2_c64_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                               const float* __restrict__ bias,
                                                               const T* __restrict__ wgt_p,
                                                               const float* __restrict__ bias_p, T* __restrict__ out,
                                                               T* __restrict__ out_p, int n_tiles,
                                                               const char* __restrict__ zero_page) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int HI = 56, WI = 56, C = 64, HO = 28, WO = 28, CO = 128;
  constexpr int TW = 7, TH = 4, TPI = (WO / TW) * (HO / TH);  // 4 x 7 = 28 tiles per image
  constexpr int HPIECES = 18;                                // 9 rows x 16 slots = 144 slots
  constexpr int H_BYTES = HPIECES * 1024;
  constexpr int SROW = 144;                                  // staging row: 32 fp32 + 16 B pad
  constexpr int SW_BYTES = 32 * SROW;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * H_BYTES + 4 * SW_BYTES + 2 * CO * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  unsigned char* const Sl = smem + 2 * H_BYTES + wave * SW_BYTES;
  float* const Bl = reinterpret_cast<float*>(smem + 2 * H_BYTES + 4 * SW_BYTES);  // bias[128], bias_p[128]
  if (tid < CO) {
    Bl[tid] = bias[tid];
    Bl[CO + tid] = bias_p[tid];
  }

  // weights of output channel 32*wave + r: 9 taps x 64 input channels, and the projection's 64
  frag wreg[9][4], wpr[4];
  {
    const char* wb = reinterpret_cast<const char*>(wgt) + (size_t)(wave * 32 + r) * (9 * C * 2) + 16 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wreg[tap][kk] = *reinterpret_cast<const frag*>(wb + tap * 128 + kk * 32);
    const char* pb = reinterpret_cast<const char*>(wgt_p) + (size_t)(wave * 32 + r) * (C * 2) + 16 * h;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wpr[kk] = *reinterpret_cast<const frag*>(pb + kk * 32);
  }
  const int e_c8 = lane & 3, e_px = lane >> 2;  // epilogue item k: pixel e_px + 16k, channels 32*wave + 8*e_c8 ..

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const int prow = lane >> 3, dchunk = lane & 7;
  const char* in_b = reinterpret_cast<const char*>(in);

  // halo DMA: tile-independent part of every piece computed once (byte offset from the tile's first
  // input pixel, edge bits in the low nibble: bit 0 = halo row 0, bit 1 = halo column 0).
  constexpr int NPW = (HPIECES + 3) / 4;  // pieces per wave (5; waves 2, 3 have 4)
  int piece_pk[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int q = (wave + 4 * k) * 8 + prow;  // slot = hy*16 + p
    const int hy = (q >> 4) < 9 ? (q >> 4) : 8;
    int pcol = q & 15;
    pcol = pcol < 15 ? pcol : 14;             // slot 15 of a row does not exist: re-fetch slot 14 (never read)
    const int hx = pcol < 8 ? 2 * pcol : 2 * (pcol - 8) + 1;
    const int key = ((pcol >> 1) & 1) | (((hy >> 1) & 3) << 1);
    const int rel = ((hy - 1) * WI + (hx - 1)) * (C * 2) + (dchunk ^ key) * 16;
    piece_pk[k] = rel | (hy == 0 ? 1 : 0) | (hx == 0 ? 2 : 0);
  }
  auto issue_unit = [&](int tile, int buf) {
    const int b = tile / TPI, t = tile - b * TPI;
    const int ty = t / (WO / TW), tx = t - ty * (WO / TW);
    const char* tbase = in_b + (((size_t)b * HI + ty * (2 * TH)) * WI + tx * (2 * TW)) * (C * 2);
    const int tmask = (ty == 0 ? 1 : 0) | (tx == 0 ? 2 : 0);
    static_for<NPW>([&](auto K) {
      constexpr int k = decltype(K)::value;
      const int pc = wave + 4 * k;
      if (pc < HPIECES) {
        const bool ok = (piece_pk[k] & tmask) == 0;
        const char* src = ok ? tbase + (piece_pk[k] & ~15) : zero_page + dchunk * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * H_BYTES + pc * 1024), 16, 0, 0);
      }
    });
  };

  // this lane's output pixel inside the tile: (y, x) = (r / 8, r % 8); column 7 does not exist and
  // re-reads column 6 (same address: an LDS broadcast), its results are never stored
  int lx = (r & 7) < TW ? (r & 7) : TW - 1, ly = r >> 3;

#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wreg[tap][kk]));  // weight loads retire before the loop
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wpr[kk]));

  int tile = blockIdx.x;
  if (tile < n_tiles) issue_unit(tile, 0);
  for (int it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
    const int buf = it & 1;
    // the halo DMAs are older than the previous epilogue's 4 stores (vmcnt retires in issue order)
    if (it) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();  // this tile's halo landed; every wave is past its reads of the other buffer
    if (tile + (int)gridDim.x < n_tiles) issue_unit(tile + gridDim.x, buf ^ 1);  // lands behind this whole unit

    const int tb = tile / TPI, tt = tile - tb * TPI;
    const int ty = tt / (WO / TW), tx = tt - ty * (WO / TW);
    asm volatile("" : "+v"(lx), "+v"(ly));  // keep the tap address bases from being hoisted (VGPRs)
    const unsigned char* const Hl = smem + buf * H_BYTES;
    f32x16 acc, accp;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = accp[e] = 0.f;
    __builtin_amdgcn_s_setprio(1);
    constexpr int PF = 2;
    frag ring[PF + 1];
    // fragment address = lane base + compile-time slot offset + swizzled chunk, the chunk term factored per
    // axis: bit 4 = ((pcol >> 1) & 1) ^ h depends on kw only, bits 5-6 = ((hy >> 1) & 3) ^ kk on kh only
    int ax[3], by[3];
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
      const int pcol = (k3 & 1) * 8 + lx + (k3 >> 1);
      ax[k3] = (((pcol >> 1) & 1) ^ h) << 4;
      by[k3] = (((2 * ly + k3) >> 1) & 3) << 5;
    }
    const unsigned char* const Hb = Hl + (((2 * ly) << 4) + lx) * 128;
    auto rd_step = [&](auto S) {
      constexpr int st = decltype(S)::value;
      constexpr int tap = st / 4, kk = st % 4, kh = tap / 3, kw = tap % 3;
      ring[st % (PF + 1)] = *reinterpret_cast<const frag*>(Hb + (ax[kw] | (by[kh] ^ (kk << 5))) +
                                                           (((kh << 4) + (kw & 1) * 8 + (kw >> 1)) << 7));
    };
    static_for<PF>([&](auto S) { rd_step(S); });
    static_for<36>([&](auto S) {
      constexpr int st = decltype(S)::value;
      if constexpr (st + PF < 36) rd_step(std::integral_constant<int, st + PF>{});
      acc = E::mfma(wreg[st / 4][st % 4], ring[st % (PF + 1)], acc);
      if constexpr (st / 4 == 4) accp = E::mfma(wpr[st % 4], ring[st % (PF + 1)], accp);  // centre tap = 1x1/2 input
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);

    // epilogue, per wave: [32 px][32 ch] fp32 through private staging rows -> 16-byte items
    auto flush = [&](const f32x16& a, const float* bl, bool relu, T* dst) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
        v[0] = a[4 * q + 0];
        v[1] = a[4 * q + 1];
        v[2] = a[4 * q + 2];
        v[3] = a[4 * q + 3];
        *reinterpret_cast<f32x4*>(Sl + r * SROW + (8 * q + 4 * h) * 4) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave's LDS operations complete in order
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int px = e_px + 16 * k;
        const int y = px >> 3, x = px & 7;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Sl + px * SROW + e_c8 * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Sl + px * SROW + e_c8 * 32 + 16);
        const f32x4 b_lo = *reinterpret_cast<const f32x4*>(bl + wave * 32 + e_c8 * 8);
        const f32x4 b_hi = *reinterpret_cast<const f32x4*>(bl + wave * 32 + e_c8 * 8 + 4);
        float v[8] = {lo[0] + b_lo[0], lo[1] + b_lo[1], lo[2] + b_lo[2], lo[3] + b_lo[3],
                      hi[0] + b_hi[0], hi[1] + b_hi[1], hi[2] + b_hi[2], hi[3] + b_hi[3]};
        frag ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (T)(relu ? fmaxf(v[e], 0.f) : v[e]);
        if (x < TW)
          *reinterpret_cast<frag*>(dst + ((((size_t)tb * HO + ty * TH + y) * WO + tx * TW + x) * CO + wave * 32 +
                                         e_c8 * 8)) = ov;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staging reads returned before it is overwritten
    };
    flush(acc, Bl, true, out);
    flush(accp, Bl + CO, false, out_p);
  }
}

}  // namespace hipac
#include "block_c64.h"
#include "block16_c64.h"
namespace hipac {

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

// fp16x3 mode: the same max-pool over the fp32 stem map (exact f32 MFMA), written as (hi, lo) fp16 pairs
// [n,56,56, hi: 64 | lo: 64]
template <typename TO>  // (a template only so that the header can be included from several translation units)
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const float* __restrict__ in, TO* __restrict__ out,
                                                                 int n) {
  static_assert(std::is_same<TO, _Float16>::value, "pairs are fp16");
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
      const float* src = in + (((size_t)b * HI + ih) * WI + iw) * C + c8 * 8;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], v0[e]), best[4 + e] = fmaxf(best[4 + e], v1[e]);
    }
  }
  f16x8 oh8, ol8;
  split_pair8(best, oh8, ol8);
  _Float16* dst = out + (((size_t)b * HO + oh) * WO + ow) * (2 * C) + c8 * 8;
  *reinterpret_cast<f16x8*>(dst) = oh8;
  *reinterpret_cast<f16x8*>(dst + C) = ol8;
}

// fp16x3 mode, uint8 input: raw HWC patches -> the zero-padded fp32 NHWC4 tensor [n,230,232,4] through the fp32
// ToTensor / Normalize table (exactly the reference's (v/255 - mean)/std per byte value, src/main.py:815-816)
template <typename TO>
__global__ __launch_bounds__(256) void u8_to_nhwc4_f32_kernel(const unsigned char* __restrict__ x,
                                                              const float* __restrict__ lut, TO* __restrict__ out,
                                                              int n) {
  static_assert(std::is_same<TO, float>::value, "fp32 stem input");
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)n * kPadH * kPadW;
  if (gid >= total) return;
  const int px = (int)(gid % kPadW);
  const long long t = gid / kPadW;
  const int py = (int)(t % kPadH);
  const int b = (int)(t / kPadH);
  const int y = py - 3, xx = px - 3;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if ((unsigned)y < (unsigned)kPatch && (unsigned)xx < (unsigned)kPatch) {
    const unsigned char* src = x + (((size_t)b * kPatch + y) * kPatch + xx) * 3;
    v[0] = lut[src[0]];
    v[1] = lut[256 + src[1]];
    v[2] = lut[512 + src[2]];
  }
  *reinterpret_cast<f32x4*>(out + (size_t)gid * 4) = v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel, device)
constexpr int kMaxDevices = 64;
static inline int ensure_dynamic_lds(const void* kern, int lds, bool* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
  if (dev >= 0 && done[dev]) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0) done[dev] = true;
  return 0;
}

#ifndef HIPAC_HALO_BM256
#define HIPAC_HALO_BM256 1
#endif
#ifndef HIPAC_SPLIT_DBLW
#define HIPAC_SPLIT_DBLW 1  // fp16x3 layer1: double weight steps (see conv3x3_halo_kernel's DBLW note)
#endif
#ifndef HIPAC_HALO_BIG
#define HIPAC_HALO_BIG 0  // 1: layers 2-4 on one 512-register wave per SIMD (128 x 128 per wave)
#endif
#ifndef HIPAC_USE_C64
#define HIPAC_USE_C64 1
#endif
#ifndef HIPAC_USE_HALO
#define HIPAC_USE_HALO 1
#endif
#ifndef HIPAC_BM_A
#define HIPAC_BM_A 128
#endif
#ifndef HIPAC_NSTAGE_A
#define HIPAC_NSTAGE_A 2
#endif
#ifndef HIPAC_NSTAGE_B
#define HIPAC_NSTAGE_B 2
#endif
#ifndef HIPAC_BM_64
#define HIPAC_BM_64 256
#endif
// Tile shape per layer: COUT = 64 -> 256 pixels x 64 channels (4 waves);
// otherwise 128 x 128 (4 waves, 4-stage ring).
#ifndef HIPAC_BN_A
#define HIPAC_BN_A 128
#endif
#ifndef HIPAC_S2_WIDE
#define HIPAC_S2_WIDE 1  // layers 3-4 entry convs on 256 x 256 tiles of 128 x 64 wave tiles (see launch_conv)
#endif
#ifndef HIPAC_S2_BM
#define HIPAC_S2_BM 256
#endif
#ifndef HIPAC_S2_BN
#define HIPAC_S2_BN 256
#endif
#ifndef HIPAC_S2_WTM
#define HIPAC_S2_WTM 128
#endif
#ifndef HIPAC_S2_NSTAGE
#define HIPAC_S2_NSTAGE 2
#endif
#ifndef HIPAC_HALO_MF16
#define HIPAC_HALO_MF16 1  // layers 2-4 stride-1 convs on v_mfma_f32_16x16x32 (halo16.h); 0: the 32x32x16 form
#endif
// the halo kernel of a layer: the 16x16x32 form for the plain 16-bit precisions on 256 x 128 tiles, else the 32x32x16 form
template <typename T, int CIN, int COUT, int H, int W, int BM, int BN, int NSW, bool RELU, bool RESID, bool OUTF32, bool SPLIT, int PCIN,
          bool DBLW, bool POOL = false>
static auto halo_kernel_of() {
  if constexpr (HIPAC_HALO_MF16 && !SPLIT && sizeof(T) == 2 && BM == 256 && BN == 128)
    return conv3x3_halo16_kernel<T, CIN, COUT, H, W, BM, BN, NSW, RELU, RESID, OUTF32, PCIN, POOL>;
  else {
    static_assert(!POOL, "the pooled epilogue exists in the 16x16x32 halo kernel only");
    return conv3x3_halo_kernel<T, CIN, COUT, H, W, BM, BN, NSW, RELU, RESID, OUTF32, SPLIT, 2, 2, 2, PCIN, DBLW>;
  }
}
// does the last conv of the network (512 -> 512, 7 x 7) have the pooled epilogue in this build and precision?
template <typename T, bool SPLIT>
constexpr bool halo_pool_available() { return HIPAC_HALO_MF16 && HIPAC_H16_DIRECT && HIPAC_USE_HALO && !HIPAC_HALO_BIG && !SPLIT && sizeof(T) == 2; }
template <int COUT> struct TileCfg { static constexpr int BM = HIPAC_BM_A, BN = (COUT % HIPAC_BN_A == 0 ? HIPAC_BN_A : 128), NSTAGE = HIPAC_NSTAGE_A; };
template <> struct TileCfg<64> { static constexpr int BM = HIPAC_BM_64, BN = 64, NSTAGE = HIPAC_NSTAGE_B; };

#ifndef HIPAC_S2_HALO16
#define HIPAC_S2_HALO16 1  // the 3x3 / stride 2 entry convs on halo16's stride-2 form (whole-pixel plane bands); 0: band16.h
#endif
#ifndef HIPAC_S2_HALO16_MAXCIN
#define HIPAC_S2_HALO16_MAXCIN 512  // ... up to this many input channels.  Above it: band16.h's double-buffered half-chunk bands --
                                    // measured EQUAL for layers 3-4 (114 / 102 vs 116 / 103 ns per patch: what the hidden band round trips
                                    // gain, the 64-byte rows and the doubled per-tap address work cost), slower for layer2 (165 vs 183)
#endif
#ifndef HIPAC_BLK16
#define HIPAC_BLK16 1  // the fused layer1 block on v_mfma_f32_16x16x32 (block16_c64.h); 0: the 32x32x16 form (block_c64.h)
#endif
#ifndef HIPAC_BAND16
#define HIPAC_BAND16 1  // the 3x3 / stride 2 entry convs of layers 2-4 on the half-chunk band kernel (band16.h); layer2's projection
                        // shortcut then folds into its block's second conv as layers 3-4's do
#endif
#ifndef HIPAC_BAND16_S1
#define HIPAC_BAND16_S1 0  // ... the stride-1 ones too (measured: 2-3 % SLOWER than halo16 -- the hidden band round trip does not
                           // pay for twice the per-tap address work and 64-byte weight rows; they stay on the halo16 kernel)
#endif
template <typename T, int CIN, int COUT, int HO, int STRIDE, bool RELU, bool RESID, int PCIN>
static int launch_band16(const void* in, const ConvW& w, const void* resid, void* out, int n, hipStream_t s, const void* wgt_p = nullptr,
                         const float* bias = nullptr) {
  constexpr int BM = 256, BN = 128;
  constexpr int LEAD = HO + 1, TRAIL = STRIDE == 1 ? HO + 1 : 0;
  constexpr int NPW = (4 + LEAD + BM + TRAIL + 63) / 64;
  constexpr int LDS = 2 * NPW * 4 * 1024 + HIPAC_B16_NSLOT * 8192;
  auto kern = conv3x3_band16_kernel<T, CIN, COUT, HO, HO, STRIDE, BM, BN, RELU, RESID, PCIN>;
  static bool attr_done[kMaxDevices] = {};
  if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
  const int M = n * HO * HO;
  const int n_mtiles = (M + BM - 1) / BM;
  const int mt8 = (n_mtiles + 7) / 8 * 8;
  const int n_vtiles = mt8 * (COUT / BN);
  dim3 grid(n_vtiles < HIPAC_HALO_GRID ? n_vtiles : HIPAC_HALO_GRID);
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const T*)in, (const T*)w.w, bias ? bias : w.bias, (const T*)resid, (T*)out, M, n, n_mtiles,
                     (const T*)wgt_p);
  return (int)hipGetLastError();
}

template <typename T, int CIN, int COUT, int HI, int WI, int KS, int STRIDE, bool RELU, bool RESID,
          bool OUTF32, bool STEM = false, bool SPLIT = false, bool POOL = false>
static int launch_conv(const void* in, const ConvW& w, const void* resid, void* out, int n, hipStream_t s,
                       const char* zero_page = nullptr) {
  constexpr int PAD = STEM ? 0 : KS / 2;
  constexpr int HO = STEM ? 112 : (HI + 2 * PAD - KS) / STRIDE + 1;
  constexpr int WO = STEM ? 112 : (WI + 2 * PAD - KS) / STRIDE + 1;
  const int M = n * HO * WO;
  if constexpr (STEM || sizeof(T) == 4) {
    // the stem, and every layer of the fp32 parity mode (exact f32 MFMA), run on the v1 kernel
    constexpr int BN = 64;
    dim3 grid((M + 127) / 128, COUT / BN);
    hipLaunchKernelGGL((conv_igemm_kernel<T, CIN, COUT, HI, WI, KS, STRIDE, BN, RELU, RESID, OUTF32, STEM>),
                       grid, dim3(256), 0, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out, M);
  } else if constexpr (HIPAC_USE_C64 && !SPLIT && KS == 3 && STRIDE == 1 && CIN == 64 && COUT == 64 && HI == 56 && !OUTF32) {
    const int n_tiles = n * 49;
    const int n_units = (n_tiles + 1) / 2;
    const int grid = n_units < 512 ? n_units : 512;  // persistent, 2 workgroups per CU
    hipLaunchKernelGGL((conv3x3_c64_kernel<T, RESID, RELU>), dim3(grid), dim3(256), 0, s, (const T*)in, (const T*)w.w,
                       w.bias, (const T*)resid, (T*)out, n_tiles, zero_page);
  } else if constexpr (HIPAC_USE_HALO && HIPAC_HALO_BIG && !SPLIT && KS == 3 && STRIDE == 1 && COUT >= 128) {
    // one 512-register wave per SIMD, each wave a 128 pixel x 128 channel tile
    constexpr int BN = COUT >= 256 ? 256 : 128;
    constexpr int WM = COUT >= 256 ? 2 : 4, WN = 4 / WM;
    constexpr int BM = WM * 128;
    constexpr int A_BYTES = halo_band_pieces(WI, BM) * 1024;
    constexpr int STG = 4 * 32 * (BN / WN * 4 + 16);
    constexpr int NSW = (A_BYTES + (3 * BN * 128 > STG ? 3 * BN * 128 : STG) <= 160 * 1024) ? 3 : 2;
    constexpr int LDS = A_BYTES + (NSW * BN * 128 > STG ? NSW * BN * 128 : STG);
    auto kern = conv3x3_halo_kernel<T, CIN, COUT, HI, WI, BM, BN, NSW, RELU, RESID, OUTF32, false, WM, WN, 1>;
    static bool attr_done[kMaxDevices] = {};
    if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
    const int n_mtiles = (M + BM - 1) / BM;
    const int mt8 = (n_mtiles + 7) / 8 * 8;
    const int n_vtiles = mt8 * (COUT / BN);
    dim3 grid(n_vtiles < 256 ? n_vtiles : 256);  // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out, M, n,
                       n_mtiles, zero_page, (const T*)nullptr);
  } else if constexpr (HIPAC_BAND16_S1 && HIPAC_HALO_MF16 && KS == 3 && STRIDE == 1 && !SPLIT && sizeof(T) == 2 && !OUTF32 && COUT % 128 == 0 && HI == WI) {
    return launch_band16<T, CIN, COUT, HI, 1, RELU, RESID, 0>(in, w, resid, out, n, s);
  } else if constexpr (HIPAC_S2_HALO16 && HIPAC_BAND16 && HIPAC_HALO_MF16 && KS == 3 && STRIDE == 2 && !SPLIT && sizeof(T) == 2 && !OUTF32 && !RESID && CIN <= HIPAC_S2_HALO16_MAXCIN &&
                       COUT % 128 == 0 && HI == WI) {
    // the entry convs of layers 2-4 on halo16's stride-2 form: four parity-plane bands per 64-channel chunk
    constexpr int HO = HI / 2, BM = 256, BN = 128, NSW = 2;
    constexpr int LDS = halo_band_pieces(HO, BM) * 1024 + NSW * BN * 128;
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
    auto kern = conv3x3_halo16_kernel<T, CIN, COUT, HO, HO, BM, BN, NSW, RELU, false, false, 0, false, true>;
    static bool attr_done[kMaxDevices] = {};
    if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
    const int Mo = n * HO * HO;
    const int n_mtiles = (Mo + BM - 1) / BM;
    const int mt8 = (n_mtiles + 7) / 8 * 8;
    const int n_vtiles = mt8 * (COUT / BN);
    dim3 grid(n_vtiles < HIPAC_HALO_GRID ? n_vtiles : HIPAC_HALO_GRID);
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)nullptr, out, Mo, n, n_mtiles, zero_page,
                       (const T*)nullptr);
    return (int)hipGetLastError();
  } else if constexpr (HIPAC_BAND16 && HIPAC_HALO_MF16 && KS == 3 && STRIDE == 2 && !SPLIT && sizeof(T) == 2 && !OUTF32 && !RESID && COUT % 128 == 0 && HI == WI) {
    return launch_band16<T, CIN, COUT, HI / 2, 2, RELU, false, 0>(in, w, nullptr, out, n, s);
  } else if constexpr (HIPAC_USE_HALO && KS == 3 && STRIDE == 1) {
    constexpr int BN = COUT >= 128 ? 128 : 64;
    // 256-pixel tiles (each wave 128 px x 64 ch: 0.75 LDS reads per MFMA, half the weight DMA per
    // FLOP) wherever the band still leaves room for two workgroups per CU; else 128
    constexpr int A256 = halo_band_pieces(WI, 256) * 1024;
    constexpr int BM = (HIPAC_HALO_BM256 && A256 + 2 * BN * 128 <= 80 * 1024) ? 256 : 128;
    constexpr int A_BYTES = halo_band_pieces(WI, BM) * 1024;
    constexpr int STG = 4 * 32 * (BN / 2 * 4 + 16);  // epilogue staging, aliases the ring
    // fp16x3, narrow tiles: double weight steps where two 2-tile ring slots still leave two workgroups per CU (layer1)
    constexpr bool DBLW = SPLIT && HIPAC_SPLIT_DBLW && (A_BYTES + 4 * BN * 128 <= 80 * 1024);
    constexpr int NSW = DBLW ? 2 : ((A_BYTES + 3 * BN * 128 <= 80 * 1024) ? 3 : 2);  // deepest ring that keeps 2 workgroups/CU
    constexpr int RING = NSW * BN * 128 * (DBLW ? 2 : 1);
    constexpr int LDS = A_BYTES + (RING > STG ? RING : STG);
    auto kern = halo_kernel_of<T, CIN, COUT, HI, WI, BM, BN, NSW, RELU, RESID, OUTF32, SPLIT, 0, DBLW, POOL>();
    static bool attr_done[kMaxDevices] = {};  // the attribute is per device; a benign race at worst repeats the call
    if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
    const int n_mtiles = (M + BM - 1) / BM;
    const int mt8 = (n_mtiles + 7) / 8 * 8;
    const int n_vtiles = mt8 * (COUT / BN);
    dim3 grid(n_vtiles < HIPAC_HALO_GRID ? n_vtiles : HIPAC_HALO_GRID);  // persistent; both are multiples of 8
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out, M, n,
                       n_mtiles, zero_page, (const T*)nullptr);
  } else if constexpr (HIPAC_S2_WIDE && KS == 3 && STRIDE == 2 && !SPLIT && COUT % HIPAC_S2_BN == 0 && !RESID && !OUTF32) {
    // plain 3x3 / stride 2 entry convs of layers 3-4 (their projection is folded into the block's second conv): one
    // 8-wave workgroup per CU, every wave a 128 pixel x 64 channel tile -- 0.75 LDS fragment reads per MFMA instead of 1
    // and half the LDS-DMA bytes per FLOP of the 128 x 128 tile
    constexpr int BM = HIPAC_S2_BM, BN = HIPAC_S2_BN, NSTAGE = HIPAC_S2_NSTAGE, WTM = HIPAC_S2_WTM;
    constexpr int THREADS = (BM / WTM) * (BN / 64) * 64;
    constexpr int LDS = NSTAGE * (BM + BN) * 128;
    auto kern = conv_glds_kernel<T, CIN, COUT, HI, WI, KS, STRIDE, BM, BN, NSTAGE, RELU, RESID, OUTF32, false, SPLIT, WTM>;
    static bool attr_done[kMaxDevices] = {};
    if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
    const int n_mtiles = (M + BM - 1) / BM;
    const int mt8 = (n_mtiles + 7) / 8 * 8;
    dim3 grid(mt8 * (COUT / BN));
    hipLaunchKernelGGL(kern, grid, dim3(THREADS), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out,
                       M, n_mtiles, zero_page, (const T*)nullptr, (const float*)nullptr, (void*)nullptr);
  } else {
    using C = TileCfg<COUT>;
    constexpr int BM = C::BM, BN = C::BN, NSTAGE = C::NSTAGE;
    constexpr int THREADS = (BM / 64) * (BN / 64) * 64;
    constexpr int LDS = NSTAGE * (BM + BN) * 128;
    auto kern = conv_glds_kernel<T, CIN, COUT, HI, WI, KS, STRIDE, BM, BN, NSTAGE, RELU, RESID, OUTF32, false, SPLIT>;
    static bool attr_done[kMaxDevices] = {};  // the attribute is per device; a benign race at worst repeats the call
    if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
    const int n_mtiles = (M + BM - 1) / BM;
    const int mt8 = (n_mtiles + 7) / 8 * 8;
    dim3 grid(mt8 * (COUT / BN));
    hipLaunchKernelGGL(kern, grid, dim3(THREADS), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)resid, out,
                       M, n_mtiles, zero_page, (const T*)nullptr, (const float*)nullptr, (void*)nullptr);
  }
  return (int)hipGetLastError();
}

// second conv of a down-sampling block with the 1x1 / stride 2 projection folded in as extra K steps (halo kernel, PCIN):
// tmp = relu(conv1(x)) -> out = relu(conv2(tmp) + proj(x) + b2 + bp)
template <typename T, int CO, int HO, int PCIN>
static int launch_conv_projk(const void* tmp, const ConvW& w2, const ConvW& wp, const float* bias_sum, const void* xblk, void* out,
                             int n, hipStream_t s, const char* zero_page) {
  if constexpr (HIPAC_BAND16_S1 && HIPAC_HALO_MF16 && sizeof(T) == 2)
    return launch_band16<T, CO, CO, HO, 1, true, false, PCIN>(tmp, w2, xblk, out, n, s, wp.w, bias_sum);
  constexpr int BN = 128;
  constexpr int A256 = halo_band_pieces(HO, 256) * 1024;
  constexpr int BM = (HIPAC_HALO_BM256 && A256 + 2 * BN * 128 <= 80 * 1024) ? 256 : 128;
  constexpr int A_BYTES = halo_band_pieces(HO, BM) * 1024;
  constexpr int NSW = (A_BYTES + 3 * BN * 128 <= 80 * 1024) ? 3 : 2;
  constexpr int STG = 4 * 32 * (BN / 2 * 4 + 16);
  constexpr int LDS = A_BYTES + (NSW * BN * 128 > STG ? NSW * BN * 128 : STG);
  auto kern = halo_kernel_of<T, CO, CO, HO, HO, BM, BN, NSW, true, false, false, false, PCIN, false>();
  static bool attr_done[kMaxDevices] = {};
  if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
  const int M = n * HO * HO;
  const int n_mtiles = (M + BM - 1) / BM;
  const int mt8 = (n_mtiles + 7) / 8 * 8;
  const int n_vtiles = mt8 * (CO / BN);
  dim3 grid(n_vtiles < HIPAC_HALO_GRID ? n_vtiles : HIPAC_HALO_GRID);
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const T*)tmp, (const T*)w2.w, bias_sum, (const T*)xblk, out, M, n, n_mtiles,
                     zero_page, (const T*)wp.w);
  return (int)hipGetLastError();
}

// Data gradient of a stride-2 convolution (training) WITHOUT the zero-interleaved gradient map: the fine grid falls into four
// parity classes; input position (2y + PY, 2x + PX) only ever meets the taps kh with kh + PY odd ... i.e. kh = 1 for PY = 0 and
// kh in {0, 2} for PY = 1 (same in x): 1, 2, 2 and 4 taps instead of 9 each -- a quarter of the MFMAs of the 3x3 convolution
// over the zero-interleaved map.  `g` = gradient wrt the conv output on the coarse HC x HC grid [n][HC][HC][CG]; `wc` = the class's
// weights [CX][taps][CG] (pack mode 3 of train_amp.hip); `dx` = [n][2 HC][2 HC][CX], this class's positions written.
template <typename T, int CG, int CX, int HC, int TKH, int TKW, int PY, int PX>
static int launch_dgrad_s2_class(const void* g, const void* wc, const float* zero_bias, void* dx, int n, hipStream_t s,
                                 const char* zero_page) {
  using C = TileCfg<CX>;
  constexpr int BM = C::BM, BN = C::BN, NSTAGE = C::NSTAGE;
  constexpr int THREADS = (BM / 64) * (BN / 64) * 64;
  constexpr int LDS = NSTAGE * (BM + BN) * 128;
  auto kern = conv_glds_kernel<T, CG, CX, HC, HC, 1, 1, BM, BN, NSTAGE, false, false, false, false, false, 64, TKH, TKW, 4 | (PY << 1) | PX>;
  static bool attr_done[kMaxDevices] = {};
  if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
  const int M = n * HC * HC;
  const int n_mtiles = (M + BM - 1) / BM;
  const int mt8 = (n_mtiles + 7) / 8 * 8;
  dim3 grid(mt8 * (CX / BN));
  hipLaunchKernelGGL(kern, grid, dim3(THREADS), LDS, s, (const T*)g, (const T*)wc, zero_bias, (const T*)nullptr, dx, M, n_mtiles,
                     zero_page, (const T*)nullptr, (const float*)nullptr, (void*)nullptr);
  return (int)hipGetLastError();
}
// all four classes of a 3x3 / stride 2 conv (KS3 = true) or the one class of a 1x1 / stride 2 conv (the other positions of dx
// must have been zeroed): wc = the four class blocks back to back (1, 2, 2, 4 taps) or the 1x1 matrix [CX][CG]
template <typename T, int CG, int CX, int HC, bool KS3>
static int launch_dgrad_s2(const void* g, const void* wc, const float* zero_bias, void* dx, int n, hipStream_t s, const char* zero_page) {
  const T* w = (const T*)wc;
  constexpr size_t blk = (size_t)CX * CG;
  if constexpr (!KS3) return launch_dgrad_s2_class<T, CG, CX, HC, 1, 1, 0, 0>(g, w, zero_bias, dx, n, s, zero_page);
  if (int rc = launch_dgrad_s2_class<T, CG, CX, HC, 1, 1, 0, 0>(g, w, zero_bias, dx, n, s, zero_page)) return rc;
  if (int rc = launch_dgrad_s2_class<T, CG, CX, HC, 1, 2, 0, 1>(g, w + blk, zero_bias, dx, n, s, zero_page)) return rc;
  if (int rc = launch_dgrad_s2_class<T, CG, CX, HC, 2, 1, 1, 0>(g, w + 3 * blk, zero_bias, dx, n, s, zero_page)) return rc;
  return launch_dgrad_s2_class<T, CG, CX, HC, 2, 2, 1, 1>(g, w + 5 * blk, zero_bias, dx, n, s, zero_page);
}

// the same four classes on the v1 kernel (the fp32 training step: exact f32 MFMA)
template <typename T, int CG, int CX, int HC, int TKH, int TKW, int PY, int PX>
static int launch_dgrad_s2_class_v1(const void* g, const void* wc, const float* zero_bias, void* dx, int n, hipStream_t s) {
  constexpr int BN = 64;
  const int M = n * HC * HC;
  dim3 grid((M + 127) / 128, CX / BN);
  hipLaunchKernelGGL((conv_igemm_kernel<T, CG, CX, HC, HC, 1, 1, BN, false, false, false, false, TKH, TKW, 4 | (PY << 1) | PX>), grid,
                     dim3(256), 0, s, (const T*)g, (const T*)wc, zero_bias, (const T*)nullptr, dx, M);
  return (int)hipGetLastError();
}
template <typename T, int CG, int CX, int HC, bool KS3>
static int launch_dgrad_s2_v1(const void* g, const void* wc, const float* zero_bias, void* dx, int n, hipStream_t s) {
  const T* w = (const T*)wc;
  constexpr size_t blk = (size_t)CX * CG;
  if constexpr (!KS3) return launch_dgrad_s2_class_v1<T, CG, CX, HC, 1, 1, 0, 0>(g, w, zero_bias, dx, n, s);
  if (int rc = launch_dgrad_s2_class_v1<T, CG, CX, HC, 1, 1, 0, 0>(g, w, zero_bias, dx, n, s)) return rc;
  if (int rc = launch_dgrad_s2_class_v1<T, CG, CX, HC, 1, 2, 0, 1>(g, w + blk, zero_bias, dx, n, s)) return rc;
  if (int rc = launch_dgrad_s2_class_v1<T, CG, CX, HC, 2, 1, 1, 0>(g, w + 3 * blk, zero_bias, dx, n, s)) return rc;
  return launch_dgrad_s2_class_v1<T, CG, CX, HC, 2, 2, 1, 1>(g, w + 5 * blk, zero_bias, dx, n, s);
}

#ifndef HIPAC_USE_S2C64
#define HIPAC_USE_S2C64 1  // layer2's 3x3/2 conv + projection on the register-weight kernel
#endif
#ifndef HIPAC_FUSE_PROJ
#define HIPAC_FUSE_PROJ 1
#endif
#ifndef HIPAC_FUSE_PROJ_MAXCO
#define HIPAC_FUSE_PROJ_MAXCO 256  // layer4 (512): 181 ns fused at 251 VGPRs vs 118 + 36 separate
#endif

// 3x3 / stride 2 conv (+BN+ReLU) of a down-sampling BasicBlock with its 1x1 / stride 2 projection
// shortcut (+BN) riding along (conv_glds_kernel<..., PROJ = true>): x -> (out, out_p)
template <typename T, int CIN, int COUT, int HI, bool SPLIT = false>
static int launch_down(const void* in, const ConvW& w, const ConvW& wp, void* out, void* out_p, int n, hipStream_t s,
                       const char* zero_page) {
  if constexpr (HIPAC_USE_S2C64 && !SPLIT && CIN == 64 && COUT == 128 && HI == 56) {
    const int n_tiles = n * 28;
    const int grid = n_tiles < 512 ? n_tiles : 512;  // persistent, 2 workgroups per CU
    hipLaunchKernelGGL((conv3x3s2_c64_kernel<T>), dim3(grid), dim3(256), 0, s, (const T*)in, (const T*)w.w, w.bias,
                       (const T*)wp.w, wp.bias, (T*)out, (T*)out_p, n_tiles, zero_page);
    return (int)hipGetLastError();
  }
  using C = TileCfg<COUT>;
  constexpr int BM = C::BM, BN = C::BN, NSTAGE = C::NSTAGE;
  constexpr int THREADS = (BM / 64) * (BN / 64) * 64;
  constexpr int LDS = NSTAGE * (BM + BN) * 128;
  constexpr int HO = HI / 2;
  const int M = n * HO * HO;
  auto kern = conv_glds_kernel<T, CIN, COUT, HI, HI, 3, 2, BM, BN, NSTAGE, true, false, false, true, SPLIT>;
  static bool attr_done[kMaxDevices] = {};
  if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
  const int n_mtiles = (M + BM - 1) / BM;
  const int mt8 = (n_mtiles + 7) / 8 * 8;
  dim3 grid(mt8 * (COUT / BN));
  hipLaunchKernelGGL(kern, grid, dim3(THREADS), LDS, s, (const T*)in, (const T*)w.w, w.bias, (const T*)nullptr, out, M,
                     n_mtiles, zero_page, (const T*)wp.w, wp.bias, out_p);
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

// precision fp16q8 (halo16x2.h): the q8 tensor of the pair tensor at workspace offset o lives at Plan::q8 + o / 2
struct Q8Map {
  char* ws;
  size_t q8;
  void* of(const void* pairs) const { return ws + q8 + (size_t)((const char*)pairs - ws) / 2; }
};
#ifndef HIPAC_X3_HALO16
#define HIPAC_X3_HALO16 1  // fp16x3: every 3x3 conv on halo16x2.h's X3 form (16x16x32 MFMA, parity-plane entry convs, folded projection, pooled head);
                           // 0: round 3's SPLIT forms of the 32x32x16 kernels
#endif
template <int CIN, int COUT, int HW, bool RELU, bool RESID, bool Q8OUT, bool POOL, bool OUTF32 = false, bool S2 = false, int PCIN = 0, bool LO16 = true,
          bool X3 = false>  // HW: the OUTPUT map
static int launch_halo16x2(const void* in, const void* in_q, const ConvW& w, const void* resid, void* out, void* out_q, int n, hipStream_t s,
                           const void* resid_q = nullptr, const void* wgt_p = nullptr, const float* bias = nullptr) {
  constexpr int BM = 256, BN = COUT % 128 == 0 ? 128 : 64;
  constexpr int LDS = halo_band_pieces(HW, BM) * 1024 + (BN == 64 ? HIPAC_Q8_NSW64 : 2) * BN * 128;
  auto kern = conv3x3_halo16x2_kernel<CIN, COUT, HW, HW, BN, RELU, RESID, Q8OUT, POOL, OUTF32, S2, PCIN, LO16, X3>;
  static bool attr_done[kMaxDevices] = {};
  if (int rc_attr = ensure_dynamic_lds((const void*)kern, LDS, attr_done)) return rc_attr;
  const int M = n * HW * HW;
  const int n_mtiles = (M + BM - 1) / BM;
  const int mt8 = (n_mtiles + 7) / 8 * 8;
  const int n_vtiles = mt8 * (COUT / BN);
  dim3 grid(n_vtiles < HIPAC_HALO_GRID ? n_vtiles : HIPAC_HALO_GRID);  // persistent; both are multiples of 8
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const _Float16*)in, (const unsigned char*)in_q, (const unsigned char*)w.w, bias ? bias : w.bias,
                     (const _Float16*)resid, out, (unsigned char*)out_q, M, n, n_mtiles, (const unsigned char*)resid_q, (const unsigned char*)wgt_p);
  return (int)hipGetLastError();
}
// the two pair modes' convs through one call: MODE 1 = fp16q8 (byte tensors, flags as given), MODE 2 = fp16x3 on the same kernel (pairs
// only: no q8 output, the lo plane always written)
template <int MODE, int CIN, int COUT, int HW, bool RESID, bool Q8OUT, bool POOL, bool OUTF32 = false, bool S2 = false, int PCIN = 0, bool LO16 = true>
static int launch_pairconv(const void* in, const void* in_q, const ConvW& w, const void* resid, void* out, void* out_q, int n, hipStream_t s,
                           const void* resid_q = nullptr, const void* wgt_p = nullptr, const float* bias = nullptr) {
  if constexpr (MODE == 2)
    return launch_halo16x2<CIN, COUT, HW, true, RESID, false, POOL, OUTF32, S2, PCIN, true, true>(in, nullptr, w, resid, out, nullptr, n, s, nullptr, wgt_p, bias);
  else
    return launch_halo16x2<CIN, COUT, HW, true, RESID, Q8OUT, POOL, OUTF32, S2, PCIN, LO16, false>(in, in_q, w, resid, out, out_q, n, s, resid_q, wgt_p, bias);
}

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
template <typename T, int CI, int CO, int HI, int STRIDE, bool LAST, bool SPLIT = false, int PM = 0>
static int run_stage(const Net& net, int stage, const void* x, void* tmp, void* ds, void* o0, void* o1, int n,
                     hipStream_t s, OpRange& ops, bool fuse_blocks = false, void* pool_part = nullptr, Q8Map qm = Q8Map{nullptr, 0}) {
  constexpr int HO = HI / STRIDE;
  const ConvW(&bw)[2] = net.block[2 * stage];
  const ConvW(&bw1)[2] = net.block[2 * stage + 1];
  const char* z = net.zero_page;
  if constexpr (PM != 0) {
    // the pair modes on halo16x2.h -- PM 1: precision fp16q8 (pair + q8 tensors), PM 2: fp16x3 (pairs only, three f16 products).
    // Stride-2 stages: the entry conv on the parity-plane form, the 1x1 / stride 2 projection shortcut folded into the block's second
    // conv as extra K steps (its op slot is empty).  A block's first conv leaves the lo plane out in fp16q8 (nothing reads it).
    static_assert(SPLIT && sizeof(T) == 2, "the pair layout");
    auto Q = [&](const void* pairs) -> void* { return PM == 1 ? qm.of(pairs) : nullptr; };
    if constexpr (STRIDE == 1) {
      if (ops.take()) HIPAC_TRY((launch_pairconv<PM, CI, CO, HO, false, true, false, false, false, 0, false>(x, Q(x), bw[0], nullptr, tmp, Q(tmp), n, s)));
      if (ops.take()) HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, true, true, false>(tmp, Q(tmp), bw[1], x, o0, Q(o0), n, s)));
    } else {
      if (ops.take()) HIPAC_TRY((launch_pairconv<PM, CI, CO, HO, false, true, false, false, true, 0, false>(x, Q(x), bw[0], nullptr, tmp, Q(tmp), n, s)));
      (void)ops.take();
      if (ops.take())
        HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, false, true, false, false, false, CI>(tmp, Q(tmp), bw[1], x, o0, Q(o0), n, s, Q(x), net.down[stage - 1].w,
                                                                                          net.bias_c2p[stage - 1])));
    }
    if (ops.take()) HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, false, true, false, false, false, 0, false>(o0, Q(o0), bw1[0], nullptr, tmp, Q(tmp), n, s)));
    if (ops.take()) {
      if constexpr (LAST) {
        if (pool_part) HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, true, false, true>(tmp, Q(tmp), bw1[1], o0, pool_part, nullptr, n, s)));
        else HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, true, false, false, true>(tmp, Q(tmp), bw1[1], o0, o1, nullptr, n, s)));
      } else {
        HIPAC_TRY((launch_pairconv<PM, CO, CO, HO, true, true, false>(tmp, Q(tmp), bw1[1], o0, o1, Q(o1), n, s)));  // (its q8 tensor feeds the next stage's entry conv)
      }
    }
    return 0;
  } else {  // (the single-value precisions, and HIPAC_X3_HALO16=0: round 3's SPLIT kernels)
  // second conv of the stage's second block; for the network's last one (LAST) either the fp32 map or, with `pool_part`,
  // the per-image partial sums of the global average pool (halo16.h, POOL)
  auto launch_last = [&](const void* in_, const void* resid_, void* out_) -> int {
    if constexpr (LAST && halo_pool_available<T, SPLIT>()) {
      if (pool_part)
        return launch_conv<T, CO, CO, HO, HO, 3, 1, true, true, true, false, SPLIT, true>(in_, bw1[1], resid_, pool_part, n, s, z);
    }
    return launch_conv<T, CO, CO, HO, HO, 3, 1, true, true, LAST, false, SPLIT>(in_, bw1[1], resid_, out_, n, s, z);
  };
  if constexpr (CI == 64 && CO == 64 && HI == 56 && STRIDE == 1 && sizeof(T) == 2 && !SPLIT) {
    if (fuse_blocks) {
      // layer1: each BasicBlock is one launch (conv1 -> conv2 + shortcut on chip); the conv2 op slots stay empty
      const int per_xcd = 4 * ((n + 7) / 8);                   // strips on the busiest XCD
      const int grid = 8 * (per_xcd < 32 ? per_xcd : 32);      // persistent: one 8-wave workgroup per CU
      auto blk_kern = HIPAC_BLK16 ? block16_c64_kernel<T> : block_c64_kernel<T>;
      if (ops.take()) {
        hipLaunchKernelGGL(blk_kern, dim3(grid), dim3(512), 0, s, (const T*)x, (const T*)bw[0].w,
                           bw[0].bias, (const T*)bw[1].w, bw[1].bias, (T*)o0, n);
        HIPAC_TRY((int)hipGetLastError());
      }
      (void)ops.take();
      if (ops.take()) {
        hipLaunchKernelGGL(blk_kern, dim3(grid), dim3(512), 0, s, (const T*)o0, (const T*)bw1[0].w,
                           bw1[0].bias, (const T*)bw1[1].w, bw1[1].bias, (T*)o1, n);
        HIPAC_TRY((int)hipGetLastError());
      }
      (void)ops.take();
      return 0;
    }
  }
  // block 0
  const void* idt = x;
  if constexpr (STRIDE == 2 && sizeof(T) == 2 && !SPLIT && (CO >= 256 || (HIPAC_BAND16 && HIPAC_HALO_MF16))) {
    if (net.projk && net.bias_c2p[stage - 1]) {
      // layers 3, 4: plain 3x3/2 entry conv; the projection rides in the SECOND conv as extra K steps (its op slot is empty)
      if (ops.take())
        HIPAC_TRY((launch_conv<T, CI, CO, HI, HI, 3, STRIDE, true, false, false, false, false>(x, bw[0], nullptr, tmp, n, s, z)));
      (void)ops.take();
      if (ops.take())
        HIPAC_TRY((launch_conv_projk<T, CO, HO, CI>(tmp, bw[1], net.down[stage - 1], net.bias_c2p[stage - 1], x, o0, n, s, z)));
      if (ops.take()) HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, false, false, false, SPLIT>(o0, bw1[0], nullptr, tmp, n, s, z)));
      if (ops.take()) HIPAC_TRY((launch_last(tmp, o0, o1)));
      return 0;
    }
  }
  if constexpr (STRIDE == 2 && sizeof(T) == 2 && HIPAC_FUSE_PROJ && CO <= HIPAC_FUSE_PROJ_MAXCO) {
    // one launch: conv1 and the projection shortcut (the op slot of the projection stays empty).
    // Not for layer4: its second accumulator set pushes the kernel past 256 registers, i.e. to
    // one workgroup per CU (measured 349 ns/img fused vs 132 + 37 separate).
    if (ops.take()) HIPAC_TRY((launch_down<T, CI, CO, HI, SPLIT>(x, bw[0], net.down[stage - 1], tmp, ds, n, s, z)));
    (void)ops.take();
    idt = ds;
  } else {
    if (ops.take())
      HIPAC_TRY((launch_conv<T, CI, CO, HI, HI, 3, STRIDE, true, false, false, false, SPLIT>(x, bw[0], nullptr, tmp, n, s, z)));
    if constexpr (STRIDE != 1 || CI != CO) {
      if (ops.take())
        HIPAC_TRY((launch_conv<T, CI, CO, HI, HI, 1, STRIDE, false, false, false, false, SPLIT>(x, net.down[stage - 1], nullptr, ds, n, s, z)));
      idt = ds;
    }
  }
  if (ops.take()) HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, true, false, false, SPLIT>(tmp, bw[1], idt, o0, n, s, z)));
  // block 1
  if (ops.take()) HIPAC_TRY((launch_conv<T, CO, CO, HO, HO, 3, 1, true, false, false, false, SPLIT>(o0, bw1[0], nullptr, tmp, n, s, z)));
  if (ops.take()) HIPAC_TRY((launch_last(tmp, o0, o1)));
  return 0;
  }
}

template <typename T, bool SPLIT = false, int PM = 0>  // PM: pair mode (run_stage)
static int run_trunk(const Net& net, const Plan& p, char* ws, const void* xin, int n_early, int img_off, int n_late,
                     hipStream_t s, int first, int last) {
  OpRange ops{first, last, 0};
  const Q8Map qm{ws, p.q8};
  const int ne = n_early, nl = n_late;
  bool fused_done = false;
  if constexpr (SPLIT) {
    if (p.u8_input && p.stem_strip) {
      // fp16x3, uint8 input: the strip kernel with split weights (bytes are exact in fp16); op 1 = nothing
      if (ops.take()) {
        const int n_strips = 2 * ne;
        const int n_pairs = (n_strips + 1) / 2;
        const int sgrid = n_pairs < 256 ? n_pairs : 256;
        hipLaunchKernelGGL((stem_pool_strip2_kernel<T, true, PM == 1>), dim3(sgrid), dim3(512), 0, s, (const unsigned char*)xin,
                           (const T*)net.stem_u8.w, net.stem_u8.bias, (T*)(ws + p.pool), n_strips,
                           ne * kPatch * kPatch * 3, PM == 1 ? (unsigned char*)qm.of(ws + p.pool) : nullptr);
        HIPAC_TRY((int)hipGetLastError());
      }
      (void)ops.take();
    } else {
    // float input: the stem runs on the exact f32 MFMA (fp32 NHWC4 input, fp32 stem map), the pool writes (hi, lo) pairs
    if (ops.take())
      HIPAC_TRY((launch_conv<float, 4, 64, 224, 224, 7, 2, true, false, false, true>(xin, net.stem, nullptr, ws + p.stem, ne, s)));
    if (ops.take()) {
      const long long total = (long long)ne * 56 * 56 * 8;
      hipLaunchKernelGGL((maxpool3x3s2_split_kernel<_Float16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                         (const float*)(ws + p.stem), (_Float16*)(ws + p.pool), ne);
      HIPAC_TRY((int)hipGetLastError());
      if constexpr (PM == 1) HIPAC_TRY(launch_pairs_to_q8(ws + p.pool, qm.of(ws + p.pool), (long long)ne * 56 * 56, 64, s));
    }
    }
    fused_done = true;
  } else if constexpr (sizeof(T) == 2) {
    if (p.fuse_stem) {
    // op 0 = fused stem + max-pool (the 112x112 stem map is never materialised), op 1 = nothing
    if (ops.take()) {
      const int n_tiles = ne * kStemTilesPerImage;
      const int grid = n_tiles < 512 ? n_tiles : 512;  // persistent: 2 workgroups per CU
      if (p.u8_input && p.stem_strip) {
        const int n_strips = 2 * ne;
        const int n_pairs = (n_strips + 1) / 2;
        const int sgrid = n_pairs < 256 ? n_pairs : 256;  // persistent: one 8-wave workgroup (two teams) per CU
        hipLaunchKernelGGL((stem_pool_strip2_kernel<T>), dim3(sgrid), dim3(512), 0, s, (const unsigned char*)xin,
                           (const T*)net.stem_u8.w, net.stem_u8.bias, (T*)(ws + p.pool), n_strips,
                           ne * kPatch * kPatch * 3);
      } else if (p.u8_input)
        hipLaunchKernelGGL((stem_pool_kernel<T, true>), dim3(grid), dim3(256), 0, s, xin, (const T*)net.stem.w,
                           net.stem.bias, (T*)(ws + p.pool), n_tiles, net.lut_t,
                           (long long)ne * kPatch * kPatch * 3);
      else
        hipLaunchKernelGGL((stem_pool_kernel<T, false>), dim3(grid), dim3(256), 0, s, xin, (const T*)net.stem.w,
                           net.stem.bias, (T*)(ws + p.pool), n_tiles, (const unsigned short*)nullptr, 0LL);
      HIPAC_TRY((int)hipGetLastError());
    }
    (void)ops.take();
      fused_done = true;
    }
  }
  if constexpr (!SPLIT)
  if (!fused_done) {
  if (ops.take())
    HIPAC_TRY((launch_conv<T, 4, 64, 224, 224, 7, 2, true, false, false, true>(xin, net.stem, nullptr, ws + p.stem, ne, s)));
  if (ops.take()) {
    const long long total = (long long)ne * 56 * 56 * 8;
    hipLaunchKernelGGL((maxpool3x3s2_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const T*)(ws + p.stem), (T*)(ws + p.pool), ne);
    HIPAC_TRY((int)hipGetLastError());
  }
    }
  // layer2's second block writes straight into this sub-batch's slice of the group buffer
  char* l2out = ws + p.blk[3] + (size_t)img_off * 28 * 28 * 128 * p.esz;
  HIPAC_TRY((run_stage<T, 64, 64, 56, 1, false, SPLIT, PM>(net, 0, ws + p.pool, ws + p.tmp_e, nullptr, ws + p.blk[0], ws + p.blk[1], ne, s, ops,
                                                           p.l1_fused != 0, nullptr, qm)));
  HIPAC_TRY((run_stage<T, 64, 128, 56, 2, false, SPLIT, PM>(net, 1, ws + p.blk[1], ws + p.tmp_e, ws + p.ds_e, ws + p.blk[2], l2out, ne, s, ops, false,
                                                            nullptr, qm)));
  HIPAC_TRY((run_stage<T, 128, 256, 28, 2, false, SPLIT, PM>(net, 2, ws + p.blk[3], ws + p.tmp_l, ws + p.ds_l, ws + p.blk[4], ws + p.blk[5], nl, s, ops,
                                                             false, nullptr, qm)));
  HIPAC_TRY((run_stage<T, 256, 512, 14, 2, true, SPLIT, PM>(net, 3, ws + p.blk[5], ws + p.tmp_l, ws + p.ds_l, ws + p.blk[6], ws + p.blk[7], nl, s, ops,
                                                            false, p.pool_head && (PM != 0 || halo_pool_available<T, SPLIT>()) ? ws + p.part : nullptr, qm)));
  return 0;
}

}  // namespace hipac
