// conv3x3_halo16x2_kernel: every 3x3 convolution (stride 1, and the stride-2 entry convs with the projection shortcut folded into
// the block's second conv) of the two parity modes -- `fp16q8` (round 4; VERDICT r3 item 3 "a parity mode with cheaper cross
// terms"), described first, and `fp16x3` (the X3 form, at the template's comment).  Included by conv_igemm.h after halo16.h, whose
// tile, band, weight ring, lane -> pixel permutation, written-out MFMA stream and direct epilogue it shares -- read that header first.
//
// Arithmetic.  As in fp16x3 every value is the pair hi = rn16(v), lo = rn16(v - hi) of fp16 numbers and a product is
//     x w ~= xhi whi + (xlo whi + xhi wlo)            (the dropped xlo wlo is 2^-22 relative)
// The first term needs the full 11-bit operands: v_mfma_f32_16x16x32_f16.  The bracket is 2^-11 of the result, so its operands
// need ~8 significant bits for the sum to stay exact to ~2^-19: it runs on the block-scaled MX instruction
// v_mfma_scale_f32_16x16x128_f8f6f4 with OCP e4m3 operands (measured, tools/mfma_f8_probe.hip: 2.25x the FLOP/s of the f16
// instruction on random data under the power cap), with CONSTANT scales (tests/tools/prec_mx.py: e4m3's own exponent gives every
// element its dynamic range; logits 5e-5 against the fp32 oracle where fp16x3 has 4e-6 and one fp16 product 1e-3):
//     activations:  hi8 = e4m3(xhi),  lo8 = e4m3(xlo * 2^11)        weights:  whi8 = e4m3(whi * 2^4),  wlo8 = e4m3(wlo * 2^15)
// so both cross products carry the factor 2^15, which the instruction's E8M0 scale operand (2^-15 on the weight side, 2^0 on the
// activation side) removes on the way into the SAME fp32 accumulators the f16 products go to.
//
// Layout.  Activations are the fp16x3 pair tensor [pixel][hi: C | lo: C] (what the residual adds and the other kernels of the
// mode read and write) PLUS a byte tensor q8 [pixel][C / 64][lo8: 64 | hi8: 64]: per 64-channel chunk one 128-byte row -- the
// row format of an fp16 chunk, so the band, its swizzle and the fragment reads are those of halo16.h, and ONE K = 128 MFMA per
// accumulator multiplies a chunk's row by the weight row [whi8: 64 | wlo8: 64]: xlo whi + xhi wlo.  (Whatever the instruction's
// lane -> k map is, the two operands use the same one: tools/mfma_f8_probe.hip.)  Weights: per output channel and tap, per chunk
// [hi16: 64 halves | whi8: 64 | wlo8: 64] = 256 bytes.
//
// K loop.  Per 64-channel chunk c: the hi band (pair tensor, chunk c of the hi plane), 9 taps of 64 f16 MFMAs; then the q8 band of
// chunk c, 9 taps of 32 fp8 MFMAs of twice the cycles: 2 MFMA-time units per term where fp16x3 spends 3.
#pragma once

#ifndef HIPAC_Q8_ABL
#define HIPAC_Q8_ABL 0  // developer builds (wrong results): 1 no epilogue arithmetic / stores, 2 no residual pass, 4 no fp8 steps, 8 no f16 steps, 16 epilogue arithmetic but no pair / byte stores
#endif
#ifndef HIPAC_Q8_DEPHASE
#define HIPAC_Q8_DEPHASE 0  // > 0: the workgroups that take a CU's second slot (dispatch order: XCD-local index 32 .. 63) start N x 8128
                            // cycles late, so that one workgroup's epilogue meets the other's K loop
#endif
#ifndef HIPAC_Q8_PRE_ALL
#define HIPAC_Q8_PRE_ALL 1  // 1: a tile's first NSW weight tiles (every ring slot) are requested BEFORE the previous tile's epilogue stores,
                            // and steps 1 .. NSW - 1 wait for no vector-memory operation: vmcnt retires in issue order, so a weight tile
                            // requested after the stores could only be waited for by draining every one of them (each stays counted until
                            // its data is in L2).  0: NSW - 1 tiles ahead, the first in-loop request at step 0 (halo16.h's schedule).
                            // Measured equal (139.2 k patches/s both): the drain is not what the stores cost
#endif
#ifndef HIPAC_Q8_RESID_AHEAD
#define HIPAC_Q8_RESID_AHEAD 2  // residual fragments in flight in the identity-MFMA pass (6 measured equal: 136.0 vs 136.2 k patches/s)
#endif
#ifndef HIPAC_Q8_AH8
#define HIPAC_Q8_AH8 2  // activation fragments (two ds_read_b128 each) in flight ahead of the fp8 sub-step being issued
#endif
#ifndef HIPAC_Q8_NSW64
#define HIPAC_Q8_NSW64 4  // weight ring slots of the BN = 64 (layer1) form: 2 .. 4
#endif

namespace hipac {

typedef int i32x8 __attribute__((ext_vector_type(8)));

// D += 2^(sa - 127) 2^(sb - 127) A B, A / B e4m3 (cbsz = blgp = 0), K = 128
__device__ __forceinline__ void mfma_f8_scaled(f32x4& c, const i32x8& a, const i32x8& b, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

// four floats -> four OCP e4m3 bytes (round to nearest even, saturating at +-448: the conversion itself turns larger values into NaN)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned cvt4_e4m3(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}
#else
__device__ inline unsigned cvt4_e4m3(float, float, float, float) { return 0; }
#endif
constexpr float kQ8LoScale = (float)(1 << kQ8LoShift);  // activations' lo parts are converted as lo * 2^11 (common.h has the shifts)
constexpr int kQ8ScaleA = 127 - kQ8WloShift, kQ8ScaleB = 127;  // E8M0 scale operands of the MFMA: 2^-15 on the weight side

// pair tensor -> q8 tensor, for the activations the mode's other kernels produce (stem + pool, the stride-2 entry convs)
template <int UNUSED = 0>  // (a template: the header is part of several translation units)
__global__ __launch_bounds__(256) void pairs_to_q8_kernel(const _Float16* __restrict__ in, unsigned char* __restrict__ q, long long n_items,
                                                          int C) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;  // item = 8 channels of one pixel
  if (gid >= n_items) return;
  const int c8 = (int)(gid % (C / 8));
  const long long pix = gid / (C / 8);
  const f16x8 hi = *reinterpret_cast<const f16x8*>(in + pix * 2 * C + c8 * 8);
  const f16x8 lo = *reinterpret_cast<const f16x8*>(in + pix * 2 * C + C + c8 * 8);
  u32x2 h8, l8;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    h8[k] = cvt4_e4m3((float)hi[4 * k], (float)hi[4 * k + 1], (float)hi[4 * k + 2], (float)hi[4 * k + 3]);
    l8[k] = cvt4_e4m3((float)lo[4 * k] * kQ8LoScale, (float)lo[4 * k + 1] * kQ8LoScale, (float)lo[4 * k + 2] * kQ8LoScale,
                      (float)lo[4 * k + 3] * kQ8LoScale);
  }
  unsigned char* row = q + pix * 2 * C + (c8 >> 3) * 128 + (c8 & 7) * 8;
  *reinterpret_cast<u32x2*>(row) = l8;
  *reinterpret_cast<u32x2*>(row + 64) = h8;
}
static int launch_pairs_to_q8(const void* in, void* q, long long n_pix, int C, hipStream_t s) {
  const long long items = n_pix * (C / 8);
  hipLaunchKernelGGL(pairs_to_q8_kernel<0>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (const _Float16*)in, (unsigned char*)q, items, C);
  return (int)hipGetLastError();
}

// Q8OUT: also write the q8 tensor of the output (for a following conv of this kind).  POOL: the network's last conv -- the pooled
// fp32 epilogue of halo16.h (partial sums in `outp`), no pair / q8 output.  OUTF32: the fp32 map float[pixel][COUT] instead of pairs.
// BN: 128 (waves 2 x 2, each 128 px x 64 ch) or 64 (layer1: waves 4 x 1, each 64 px x 64 ch).
// S2: the 3x3 / STRIDE 2 entry conv of a stage (H x W = the OUTPUT map, the input is 2H x 2W): halo16.h's stride-2 form -- per chunk
// and operand kind four parity-plane bands (4 / 2 / 2 / 1 taps), gathered pixel by pixel.
// PCIN > 0: the BasicBlock's 1x1 / stride 2 projection shortcut folded in (halo16.h, PCIN): 2 PCIN / 64 more K steps -- per
// 64-channel chunk of the block input `resid` (pairs [n][2H][2W][PCIN], q8 tensor `resid_q`) its pixel (2y, 2x) gathered at the
// centre tap's slots, first the hi plane x the projection's f16 weights, then the q8 row x its e4m3 weights (`wgt_p`:
// [COUT][PCIN / 64][hi16: 64 | whi8: 64 | wlo8: 64]); `bias` = the conv's + the projection's.
// LO16: write the lo plane of the output pairs.  Only a later residual add reads it (the next conv takes the hi plane and the q8
// tensor), so a block's FIRST conv leaves it out: a third of its output bytes and stores.
// X3 (precision fp16x3 on this kernel): no byte tensors -- the second band of a chunk is the LO plane of the pair tensor and all three
// products run on the f16 MFMA: per chunk the hi band x the tiles [whi | wlo] of each tap (two steps per tap on the same activation
// addresses), then the lo band x whi: 27 steps.  Weight rows [Cout][tap][C / 64][whi: 64 f16 | wlo: 64 f16] (the q8 rows' size).
template <int CIN, int COUT, int H, int W, int BN, bool RELU, bool RESID, bool Q8OUT, bool POOL, bool OUTF32 = false, bool S2 = false, int PCIN = 0,
          bool LO16 = true, bool X3 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo16x2_kernel(const _Float16* __restrict__ in, const unsigned char* __restrict__ in_q,
                                                                  const unsigned char* __restrict__ wgt, const float* __restrict__ bias,
                                                                  const _Float16* __restrict__ resid, void* __restrict__ outp,
                                                                  unsigned char* __restrict__ out_q, int M, int n_img, int n_mtiles,
                                                                  const unsigned char* __restrict__ resid_q = nullptr,
                                                                  const unsigned char* __restrict__ wgt_p = nullptr) {
  using T = _Float16;
  using frag = f16x8;
  constexpr int BM = 256;
  // weight ring depth: BN = 128: two slots, one step (1024 cycles of MFMA per wave) of distance.  BN = 64: four 8 KB slots -- the
  // distance itself measured nothing (a step is 512 cycles; 2 slots: 1 % slower), but 32 KB of ring let BOTH planes of the residual
  // tile arrive in one DMA round trip (hi tile in the band region, lo tile in the ring)
  constexpr int NSW = BN == 64 ? HIPAC_Q8_NSW64 : 2;
  constexpr int CC = CIN / 64;                      // 64-channel chunks
  constexpr int VC = 2 * CC;                        // bands of the K loop: chunk c's hi band (v = 2c), then its q8 band (v = 2c + 1)
  constexpr int KROW = 9 * VC * 128;                // bytes per weight row (output channel)
  constexpr int WN = BN / 64, WM = 4 / WN;
  constexpr int WPX = BM / WM, MT = WPX / 16;       // 128 pixels = 8 sub-tiles per wave (BN = 64: 64 pixels = 4)
  constexpr int WTN = BN / WN, NT = WTN / 16;       // 64 channels = 4 tiles per wave
  constexpr int A_PIECES = halo_band_pieces(W, BM);
  constexpr int A_BYTES = A_PIECES * 1024;
  constexpr int W_BYTES = BN * 128;
  constexpr int WPW = BN / 8 / 4;
  constexpr int NTILES_N = COUT / BN;
  constexpr int PCC = PCIN / 64;                    // chunks of the folded projection (0: none)
  constexpr int SPC = X3 ? 27 : 18;                 // K steps per chunk
  constexpr int SPP = X3 ? 3 : 2;                   // ... per chunk of the projection
  constexpr int NSTEP = CC * SPC + SPP * PCC;
  static_assert(!X3 || (!Q8OUT && LO16), "fp16x3: pairs in, pairs out");
  constexpr int PRE = HIPAC_Q8_PRE_ALL ? NSW : NSW - 1;  // weight tiles requested ahead of a tile's first step
  static_assert(NSTEP > NSW, "K steps");
  constexpr int S_BYTES = NSW * W_BYTES;
  static_assert((BN == 64 || BN == 128) && CIN % 64 == 0 && COUT % BN == 0 && (MT == 8 || MT == 4) && NT == 4, "tile shape");
  static_assert(A_BYTES + S_BYTES <= 80 * 1024, "LDS: two workgroups per CU");
  static_assert(BM + 2 * W + 4 <= A_PIECES * 8, "band slots");
  static_assert(!POOL || (BN == 128 && RELU && !Q8OUT && !OUTF32 && (BM + H * W - 1) / (H * W) + 1 <= kPoolSlots && H * W > 16), "pooled epilogue");
  static_assert(!OUTF32 || !Q8OUT, "the fp32 map has no q8 tensor");
  static_assert(!S2 || (!RESID && !POOL && !OUTF32 && PCIN == 0), "stride-2 form: plain entry conv");
  static_assert(PCIN % 64 == 0 && (PCIN == 0 || !RESID), "folded projection replaces the residual input");
  static_assert(!RESID || (A_BYTES >= BM * 128 && (WN == 1 || S_BYTES >= BM * 128)), "residual tile: one 64-channel chunk per region");

  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  unsigned char* const Abuf = ring;
  unsigned char* const Wbuf = ring + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int n16 = lane & 15, g = lane >> 4;
  const int pn = n16 < 4 ? 2 * n16 : (n16 < 12 ? 2 * n16 - 7 : 2 * n16 - 16);  // perm16(n16)
  using lptr_t = __attribute__((address_space(3))) void*;
  const int prow = lane >> 3, dchunk = lane & 7;
  int sc_a = kQ8ScaleA, sc_b = kQ8ScaleB;  // the MFMA's scale operands live in VGPRs
  asm volatile("" : "+v"(sc_a), "+v"(sc_b));

  // bands (halo16.h: slot q = pixel m0 - W - 3 + q, slots 0, 1 zeros, 16-byte chunk c of slot q at c ^ ((q >> 1) & 7)); the hi band
  // comes from the pair tensor (4 CIN bytes per pixel), the q8 band from the byte tensor (2 CIN bytes per pixel)
  const rsrc_t h_rsrc = make_rsrc(in, (S2 ? 4 : 1) * M * CIN * 4);
  const rsrc_t q_rsrc = make_rsrc(in_q, (S2 ? 4 : 1) * M * CIN * 2);
  const int swz16 = (dchunk ^ ((4 * wave + (prow >> 1)) & 7)) * 16;
  const int h_lane = prow * (CIN * 4) + swz16;
  const int q_lane = prow * (CIN * 2) + swz16;
  // S2: per-lane input pixel (2 r, 2 c) of every band piece this wave issues (slot q = 8 p + prow holds output-grid pixel
  // m0 - W - 3 + q), -1 where the slot is a zero slot or lies outside the batch; made once per tile
  constexpr int NPB = S2 ? (A_PIECES + 3) / 4 : 1;
  int b_pix[NPB];
  auto plane_offsets = [&](int m0_) {
    if constexpr (S2) {
#pragma unroll
      for (int k = 0; k < NPB; ++k) {
        const int q = 8 * (wave + 4 * k) + prow;
        const int u = m0_ - W - 3 + q;
        const int b = u / (H * W), rem = u - b * (H * W), r = rem / W, c = rem - r * W;
        b_pix[k] = (q >= 2 && u >= 0 && u < M) ? (b * (2 * H) + 2 * r) * (2 * W) + 2 * c : -1;
      }
    }
  };
  // band of the plane of tap position k (positions 0-3 plane (1,1), 4-5 (0,1), 6-7 (1,0), 8 (0,0)), band v = 2 chunk + kind
  auto issue_plane_band = [&](int k, int v) {
    if constexpr (S2) {
      const int py = k < 4 ? 1 : (k < 6 ? 0 : (k < 8 ? 1 : 0)), px = k < 6 ? 1 : 0;
      if (!X3 && (v & 1)) {
        const int pofs = (py * 2 * W + px) * (CIN * 2) + (v >> 1) * 128 + swz16;
#pragma unroll
        for (int kk = 0; kk < NPB; ++kk) {
          const int p = wave + 4 * kk;
          if (p < A_PIECES) buffer_load_lds16(q_rsrc, Abuf + p * 1024, b_pix[kk] < 0 ? (int)0x80000000 : b_pix[kk] * (CIN * 2) + pofs, 0);
        }
        asm volatile("" ::: "memory");
        return;
      }
      const int pofs = (py * 2 * W + px) * (CIN * 4) + (v >> 1) * 128 + swz16 + (X3 && (v & 1) ? CIN * 2 : 0);
#pragma unroll
      for (int kk = 0; kk < NPB; ++kk) {
        const int p = wave + 4 * kk;
        if (p < A_PIECES) buffer_load_lds16(h_rsrc, Abuf + p * 1024, b_pix[kk] < 0 ? (int)0x80000000 : b_pix[kk] * (CIN * 4) + pofs, 0);
      }
    }
  };
  auto issue_band_of = [&](int m0_, int v) {
    if constexpr (S2) {
      issue_plane_band(0, v);
      return;
    }
    const int mlast_ = (m0_ + BM <= M ? m0_ + BM : M) - 1;
    const int npieces_ = (mlast_ - m0_ + 1 + 2 * W + 2 + 2 + 7) >> 3;
    if (!X3 && (v & 1)) {
      const int base = (m0_ - W - 3) * (CIN * 2) + (v >> 1) * 128;
      for (int p = wave; p < npieces_; p += 4) {
        int off = q_lane + base + p * (8 * CIN * 2);
        if (p == 0 && prow < 2) off = (int)0x80000000;
        buffer_load_lds16(q_rsrc, Abuf + p * 1024, off, 0);
      }
      asm volatile("" ::: "memory");  // (keeps hipcc from merging the two paths into one with a SELECTED descriptor: halo16.h, issue_w)
      return;
    }
    const int base = (m0_ - W - 3) * (CIN * 4) + (v >> 1) * 128 + (X3 && (v & 1) ? CIN * 2 : 0);  // (X3, odd v: the lo plane)
    for (int p = wave; p < npieces_; p += 4) {
      int off = h_lane + base + p * (8 * CIN * 4);
      if (p == 0 && prow < 2) off = (int)0x80000000;
      buffer_load_lds16(h_rsrc, Abuf + p * 1024, off, 0);
    }
  };

  if constexpr (HIPAC_Q8_DEPHASE > 0) {
    if ((blockIdx.x >> 3) & 32) {
#pragma unroll
      for (int k = 0; k < HIPAC_Q8_DEPHASE; ++k) __builtin_amdgcn_s_sleep(127);
    }
  }
  constexpr int N_EPI_STORES = OUTF32 ? MT * NT : MT * (2 + (LO16 ? 2 : 0) + (Q8OUT ? 2 : 0));  // hi16, lo16 (two 32-channel halves each), hi8, lo8
  static_assert(LO16 || Q8OUT, "an output without lo plane and without q8 tensor is a plain fp16 map");
  static_assert(N_EPI_STORES < 64, "vmcnt range");
  bool prev_full = false;
  for (int vb = blockIdx.x, first_tile = 1;; vb += gridDim.x, first_tile = 0) {
  const int xcd = vb & 7, slot = vb >> 3;
  const int mt_q = (n_mtiles + 7) >> 3;
  const int mt = (slot / NTILES_N) < mt_q ? xcd * mt_q + slot / NTILES_N : n_mtiles;
  const int nt = slot % NTILES_N;
  if (mt >= n_mtiles) break;
  const int m0 = mt * BM, n0 = nt * BN;
  const int mstart = m0 - W - 1;
  // weight DMA: LDS row `row` of the tile takes output channel (row & ~15) | perm16_inv(row & 15)
  int w_off[WPW];
  int wp_off[PCC > 0 ? WPW : 1];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (wave + 4 * i) * 8 + prow;
    const int j = row & 15;
    const int srow = (row & ~15) | ((j & 1) ? (j + 7) >> 1 : (j < 8 ? j >> 1 : (j >> 1) + 8));
    w_off[i] = (n0 + srow) * KROW + (dchunk ^ ((row >> 1) & 7)) * 16;
    if constexpr (PCC > 0) wp_off[i] = (n0 + srow) * (PCC * 256) + (dchunk ^ ((row >> 1) & 7)) * 16;
  }
  const rsrc_t w_rsrc = make_rsrc(wgt, COUT * KROW);
  const rsrc_t wp_rsrc = make_rsrc(PCC > 0 ? wgt_p : wgt, COUT * (PCC > 0 ? PCC * 256 : KROW));
  auto issue_w = [&](int step, int slot_) {  // the 128-byte-per-channel weight tile of K step `step` (decoded below) into ring slot `slot_`
    if (PCC > 0 && step >= CC * SPC) {  // uniform: a projection step, row bytes [pc][256]: (hi, whi) (X3: (hi, wlo)) then the second band's tile
      const int idx = step - CC * SPC;
      const int kofs_p = X3 ? (idx / 3) * 256 + (idx % 3 == 1 ? 128 : 0) : idx * 128;
      static_for<WPW>([&](auto I) {
        constexpr int i = decltype(I)::value;
        buffer_load_lds16(wp_rsrc, Wbuf + slot_ * W_BYTES + (wave + 4 * i) * 1024, wp_off[PCC > 0 ? i : 0], kofs_p);
      });
      asm volatile("" ::: "memory");  // (see issue_band_of)
      return;
    }
    // step = c SPC + j.  q8: j < 9 the hi band's taps (row bytes [0, 128) of the chunk), then the q8 band's ([128, 256)).
    // X3: j < 18 the hi band's (tap, tile) pairs -- tile 0 = whi ([0, 128)), 1 = wlo ([128, 256)) --, then the lo band's taps x whi
    const int c = step / SPC, j = step - c * SPC;
    const int tap = X3 ? (j < 18 ? j >> 1 : j - 18) : (j < 9 ? j : j - 9);
    const int sel = X3 ? (j < 18 ? j & 1 : 0) : (j < 9 ? 0 : 1);
    const int kofs_bytes = (S2 ? b16_tap<2>(tap) : tap) * (VC * 128) + c * 256 + sel * 128;  // (S2: `tap` is the position in plane order)
    static_for<WPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      buffer_load_lds16(w_rsrc, Wbuf + slot_ * W_BYTES + (wave + 4 * i) * 1024, w_off[i], kofs_bytes);
    });
  };

  // ---- consumer side (halo16.h) ----
  const int mw0 = m0 + wm * WPX + pn;
  const int q0 = mw0 - mstart + 2;
  unsigned epk = 0;
  {
    const int rem = mw0 % (H * W);
    int y = rem / W, x = rem - y * W;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      epk |= (unsigned)((x == 0 ? 1 : 0) | (!S2 && x == W - 1 ? 2 : 0) | (y == 0 ? 4 : 0) | (!S2 && y == H - 1 ? 8 : 0)) << (4 * i);
      x += 16 % W, y += 16 / W;
      if (x >= W) x -= W, y += 1;
      if (y >= H) y -= H;
    }
  }
  const int rdw0 = (wn * WTN + pn) * 128 + ((g ^ ((pn >> 1) & 7)) << 4);        // f16 weight fragment, k32 step 0, channel tile 0
  const int rdw8 = (wn * WTN + pn) * 128 + (((2 * g) ^ ((pn >> 1) & 7)) << 4);  // fp8: 16-byte chunks 2 g (and, ^ 16, 2 g + 1)

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)ring;
  // the f16 step: halo16.h's k_step
  auto k_step = [&](const unsigned char* wst, const int (&a_addr)[MT], auto&& mid) {
    constexpr int AH = S2 ? 2 : 3;
    constexpr int NS = 2 * MT;
    constexpr bool W1_FIRST = MT - AH < NT;  // (4 sub-tiles: the second k32 step's weights cannot trail behind the activations)
    frag wf[2][NT];
    frag af[AH + 1];
    const unsigned w0 = lds0 + (unsigned)(wst - ring) + (unsigned)rdw0;
    const unsigned w1 = w0 ^ 64u;
    unsigned aa[NS];
#pragma unroll
    for (int i = 0; i < MT; ++i) aa[i] = lds0 + (unsigned)a_addr[i], aa[MT + i] = aa[i] ^ 64u;
    static_for<NT>([&](auto J) { lds_read16<decltype(J)::value * 2048>(wf[0][decltype(J)::value], w0); });
    if constexpr (W1_FIRST) static_for<NT>([&](auto J) { lds_read16<decltype(J)::value * 2048>(wf[1][decltype(J)::value], w1); });
    static_for<AH>([&](auto S) { lds_read16<0>(af[decltype(S)::value], aa[decltype(S)::value]); });
    __builtin_amdgcn_s_setprio(1);
    static_for<NS>([&](auto S) {
      constexpr int s = decltype(S)::value, kk = s / MT, i = s % MT;
      if constexpr (s + AH < NS) lds_read16<0>(af[(s + AH) % (AH + 1)], aa[s + AH]);
      if constexpr (!W1_FIRST && s >= 1 && s <= NT) lds_read16<(s - 1) * 2048>(wf[1][s - 1], w1);
      constexpr int a_after = (s + AH < NS ? AH : NS - 1 - s);
      constexpr int w_lo = (s - AH > 1 ? s - AH : 1), w_hi = (s < NT ? s : NT);
      constexpr int w_after = !W1_FIRST && w_hi >= w_lo ? w_hi - w_lo + 1 : 0;
      wait_lgkmcnt<a_after + w_after>();
#pragma unroll
      for (int j = 0; j < NT; ++j) Asm16<T>::mfma(acc[i][j], wf[kk][j], af[s % (AH + 1)]);
      if constexpr (s == MT / 2) mid();
    });
    __builtin_amdgcn_s_setprio(0);
  };
  // the fp8 step: one K = 128 MFMA per accumulator; a fragment = 32 bytes = two ds_read_b128 (chunks 2 g, 2 g + 1 of the row) into
  // the two halves of an 8-register operand.  Reads in issue order: W[0..NT) (2 each), A[0..AH), then A[s + AH] in sub-step s;
  // LDS returns in order, so sub-step s waits until only the 2 min(AH, MT - 1 - s) reads issued after A[s] are outstanding
  auto k_step8 = [&](const unsigned char* wst, const int (&a_addr)[MT], auto&& mid) {
    constexpr int AH = S2 ? 1 : HIPAC_Q8_AH8;  // (S2: the per-lane plane offsets need the registers; a sub-step is 128 cycles of MFMA)
    u32x4 wl[NT], wh[NT], al[AH + 1], ah[AH + 1];
    const unsigned w0 = lds0 + (unsigned)(wst - ring) + (unsigned)rdw8;
    const unsigned w1 = w0 ^ 16u;
    unsigned aa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) aa[i] = lds0 + (unsigned)a_addr[i];
    static_for<NT>([&](auto J) {
      lds_read16<decltype(J)::value * 2048>(wl[decltype(J)::value], w0);
      lds_read16<decltype(J)::value * 2048>(wh[decltype(J)::value], w1);
    });
    static_for<AH>([&](auto S) {
      lds_read16<0>(al[decltype(S)::value], aa[decltype(S)::value]);
      lds_read16<0>(ah[decltype(S)::value], aa[decltype(S)::value] ^ 16u);
    });
    __builtin_amdgcn_s_setprio(1);
    static_for<MT>([&](auto S) {
      constexpr int s = decltype(S)::value;
      if constexpr (s + AH < MT) {
        lds_read16<0>(al[(s + AH) % (AH + 1)], aa[s + AH]);
        lds_read16<0>(ah[(s + AH) % (AH + 1)], aa[s + AH] ^ 16u);
      }
      wait_lgkmcnt<2 * (s + AH < MT ? AH : MT - 1 - s)>();
      const i32x8 a = __builtin_bit_cast(i32x8, __builtin_shufflevector(al[s % (AH + 1)], ah[s % (AH + 1)], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const i32x8 w = __builtin_bit_cast(i32x8, __builtin_shufflevector(wl[j], wh[j], 0, 1, 2, 3, 4, 5, 6, 7));
        mfma_f8_scaled(acc[s][j], w, a, sc_a, sc_b);
      }
      if constexpr (s == MT / 2) mid();
    });
    __builtin_amdgcn_s_setprio(0);
  };

  int s = 0;  // K step counter
  if constexpr (S2) {
    if (first_tile) plane_offsets(m0);  // (later tiles: made by the previous tile's epilogue, for its band prefetch)
  }
  if (first_tile) {
    issue_band_of(m0, 0);
#pragma unroll
    for (int ps = 0; ps < PRE; ++ps) issue_w(ps, ps);  // (later tiles: the previous tile's epilogue has requested all of these)
  }
  for (int c = 0; c < CC; ++c) {
    static_for<2>([&](auto KIND) {
      constexpr int kind = decltype(KIND)::value;  // 0: hi band x f16 weights, 1: q8 band x fp8 weights (X3: lo band x f16 weights)
      constexpr int TPT = (X3 && kind == 0) ? 2 : 1;  // weight tiles per tap (X3, hi band: whi then wlo on the same addresses)
      constexpr bool F8 = kind == 1 && !X3;
      if (c > 0 || kind > 0) {
        __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous band
        issue_band_of(m0, 2 * c + kind);
      }
#pragma unroll HIPAC_HALO_TAP_UNROLL
      for (int tap = 0; tap < 9; ++tap) {
        const bool plane_switch = S2 && (tap == 4 || tap == 6 || tap == 8);
        if (plane_switch) {  // the next plane's band (its round trip is exposed: the wait below drains it)
          __builtin_amdgcn_s_barrier();
          issue_plane_band(tap, 2 * c + kind);
        }
        const int tap_w = S2 ? b16_tap<2>(tap) : tap;  // the tap's index in the weights
        const int kh = tap_w / 3, kw = tap_w - kh * 3;
        const int toff = S2 ? (kh == 0 ? -W : 0) + (kw == 0 ? -1 : 0) : (kh - 1) * W + kw - 1;
        const unsigned tapmask = ((kw == 0 ? 1u : 0u) | (!S2 && kw == 2 ? 2u : 0u) | (kh == 0 ? 4u : 0u) | (!S2 && kh == 2 ? 8u : 0u)) * 0x11111111u;
        const int qt = q0 + toff;
        const int x0 = (F8 ? g << 5 : g << 4) ^ (((qt >> 1) & 7) << 4);
        const int a_in = (qt << 7) + x0;
        const int a_zero = ((qt & 1) << 7) + x0;
        unsigned em;
        asm("v_and_b32 %0, %1, %2" : "=v"(em) : "s"(tapmask), "v"(epk));
        int a_addr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a_addr[i] = (em & (0xFu << (4 * i))) ? a_zero : a_in + 2048 * i;
#pragma unroll
        for (int t = 0; t < TPT; ++t, ++s) {
          // W(s) must have landed -- W(s + 1 .. s + NSW - 2) were requested after it and may stay in flight; the band too at its
          // first step and behind a plane switch (it was requested after them all: drain)
          if (s == 0 && prev_full) wait_vmcnt<N_EPI_STORES>();  // the prefetch is older than the previous epilogue's stores
          else if (!(t == 0 && (tap == 0 || plane_switch))) {
            if (HIPAC_Q8_PRE_ALL && s < NSW) {
              // W(s) was requested ahead of the tile and step 0's wait covered it
            } else if (NSW > 2 && s + NSW - 2 < NSTEP) wait_vmcnt<(NSW - 2) * WPW>();
            else wait_vmcnt<0>();
          } else wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
          auto mid = [&] {
            // the slot of step s - 1 was freed by this step's barrier (PRE_ALL: at step 0 there is no such slot, every one was filled ahead)
            if (s + NSW - 1 < NSTEP && (!HIPAC_Q8_PRE_ALL || s > 0)) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
          };
          if constexpr (F8) {
            if constexpr (HIPAC_Q8_ABL & 4) mid();
            else k_step8(wst, a_addr, mid);
          } else {
            if constexpr (HIPAC_Q8_ABL & 8) mid();
            else k_step(wst, a_addr, mid);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    });
  }

  if constexpr (PCC > 0) {
    // ---- the folded projection: per chunk pc the gathered hi rows x f16 weights, then the gathered q8 rows x e4m3 weights; centre
    // tap only, no image-edge cases (pixel (2y, 2x) always exists).  Rows outside the tile's pixels are sent out of range (zeros).
    const rsrc_t ph_rsrc = make_rsrc(resid, 4 * M * PCIN * 4);
    const rsrc_t pq_rsrc = make_rsrc(resid_q, 4 * M * PCIN * 2);
    const int mlast = (m0 + BM <= M ? m0 + BM : M) - 1;
    const int first = W + 3, last = first + (mlast - m0);  // slots that hold the tile's pixels
    for (int pc = 0; pc < PCC; ++pc) {
      static_for<2>([&](auto KIND) {
        constexpr int kind = decltype(KIND)::value;
        constexpr int TPT = (X3 && kind == 0) ? 2 : 1;
        constexpr bool F8 = kind == 1 && !X3;
        __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous band
        for (int p = wave + (first >> 3); p <= (last >> 3); p += 4) {
          const int q = p * 8 + prow;
          const int mm = m0 + q - first;
          const bool ok = q >= first && q <= last;
          const int b = mm / (H * W), rem = mm - b * (H * W), y = rem / W, x = rem - y * W;
          const int pix = (b * (2 * H) + 2 * y) * (2 * W) + 2 * x;
          const int schunk = (dchunk ^ ((q >> 1) & 7)) * 16;
          if constexpr (F8) buffer_load_lds16(pq_rsrc, Abuf + p * 1024, ok ? pix * (PCIN * 2) + pc * 128 + schunk : (int)0x80000000, 0);
          else buffer_load_lds16(ph_rsrc, Abuf + p * 1024, ok ? pix * (PCIN * 4) + pc * 128 + schunk + (kind ? PCIN * 2 : 0) : (int)0x80000000, 0);
        }
        const int a_in = (q0 << 7) + ((F8 ? g << 5 : g << 4) ^ (((q0 >> 1) & 7) << 4));
        int a_addr[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a_addr[i] = a_in + 2048 * i;
#pragma unroll
        for (int t = 0; t < TPT; ++t, ++s) {
          wait_vmcnt<0>();
          __builtin_amdgcn_s_barrier();
          const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
          auto mid = [&] {
            if (s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
          };
          if constexpr (F8) k_step8(wst, a_addr, mid);
          else k_step(wst, a_addr, mid);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      });
    }
  }

  if constexpr (RESID && !(HIPAC_Q8_ABL & 2)) {
    // ---- the residual pair, added on the matrix pipe (halo16.h): D += I x R for the hi plane, then for the lo plane -- both exact
    const rsrc_t r_rsrc = make_rsrc(resid, M * COUT * 4);
    const int r_lane = prow * (COUT * 4) + swz16;
    frag ident[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) ident[o][e] = (g == 2 * o + (n16 >> 3) && e == (n16 & 7)) ? (T)1.0f : (T)0.0f;
    const int r0 = wm * WPX + pn;  // slot of sub-tile 0's pixel
    const unsigned rb0 = lds0 + (unsigned)((wn ? Wbuf : Abuf) - ring) + (unsigned)(r0 * 128 + ((g ^ ((r0 >> 1) & 7)) << 4));
    // (BN = 64 with a 32 KB ring: both planes in ONE round trip, the hi tile into the band region, the lo tile into the ring)
    constexpr bool BOTH = WN == 1 && S_BYTES >= BM * 128;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      if (!BOTH || part == 0) {
        __builtin_amdgcn_s_barrier();  // every wave has finished the last K step (the other plane's reads): band and ring are free
#pragma unroll
        for (int pl = 0; pl < (BOTH ? 2 : 1); ++pl) {
          const int r_base = m0 * (COUT * 4) + (n0 + (BOTH ? pl : part) * COUT) * 2;
#pragma unroll
          for (int cch = 0; cch < WN; ++cch)
#pragma unroll
            for (int k = 0; k < BM / 8 / 4; ++k) {
              const int p = wave + 4 * k;
              buffer_load_lds16(r_rsrc, (cch || pl ? Wbuf : Abuf) + p * 1024, r_lane + r_base + cch * 128 + p * (8 * COUT * 4), 0);
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
      }
      const unsigned rb = BOTH && part ? rb0 + (unsigned)(Wbuf - Abuf) : rb0;
      // fragments (i, kk) in the order (0,0) (0,1) (1,0) ...: RA of them in flight (the K loop's fragment registers are free now)
      constexpr int RA = HIPAC_Q8_RESID_AHEAD;
      static_assert(RA >= 2 && RA <= 8 && RA <= 2 * MT, "residual read-ahead");
      frag rf[RA];
      static_for<RA>([&](auto S) { lds_read16<(decltype(S)::value >> 1) * 2048>(rf[decltype(S)::value], (decltype(S)::value & 1) ? rb ^ 64u : rb); });
      static_for<2 * MT>([&](auto S) {
        constexpr int s2 = decltype(S)::value, i = s2 >> 1, kk = s2 & 1;
        // (the slot of fragment s2 - 1 is free: its two MFMAs were issued in the previous sub-step)
        if constexpr (s2 >= 1 && s2 - 1 + RA < 2 * MT) lds_read16<((s2 - 1 + RA) >> 1) * 2048>(rf[(s2 - 1 + RA) % RA], ((s2 - 1 + RA) & 1) ? rb ^ 64u : rb);
        constexpr int issued = (s2 == 0 ? RA : (s2 - 1 + RA < 2 * MT ? s2 + RA : 2 * MT));  // fragments requested so far
        wait_lgkmcnt<issued - s2 - 1>();
        Asm16<T>::mfma(acc[i][2 * kk], ident[0], rf[s2 % RA]);
        Asm16<T>::mfma(acc[i][2 * kk + 1], ident[1], rf[s2 % RA]);
      });
    }
  }
  // ---- epilogue (direct, halo16.h) ----------------------------------------------------------------
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");  // XDL write -> VALU read of the accumulators
  float4 bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + n0 + wn * WTN + 16 * j + 4 * g);
  const int c_lane = n0 + wn * WTN + 16 * (g & 1) + 8 * (g >> 1);  // + 32 jp: first of the 8 channels of a 16-bit store
  __builtin_amdgcn_s_barrier();  // every wave has left the K loop: band and ring are free
  {
    // the next tile's first band and first weight tile land behind this epilogue
    const int vn = vb + gridDim.x;
    const int mtn = ((vn >> 3) / NTILES_N) < mt_q ? (vn & 7) * mt_q + (vn >> 3) / NTILES_N : n_mtiles;
    if (mtn < n_mtiles) {
      if constexpr (S2) plane_offsets(mtn * BM);
      issue_band_of(mtn * BM, 0);
      const int dn = (((vn >> 3) % NTILES_N) * BN - n0) * KROW;
#pragma unroll
      for (int i = 0; i < WPW; ++i) w_off[i] += dn;
#pragma unroll
      for (int ps = 0; ps < PRE; ++ps) issue_w(ps, ps);
    }
  }
  [[maybe_unused]] f32x4 poolS[NT][2];
  [[maybe_unused]] int pool_slot = 0, pool_bound = 0;
  [[maybe_unused]] auto pool_flush = [&](int slot_) {
    float* dst = reinterpret_cast<float*>(outp) + (((size_t)(mt * WM + wm) * kPoolSlots + slot_) * 2) * COUT + n0 + wn * WTN + 4 * g;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = row16_sum(poolS[j][part][e]);
        if (n16 == 0) *reinterpret_cast<float4*>(dst + part * COUT + 16 * j) = make_float4(t[0], t[1], t[2], t[3]);
      }
  };
  if constexpr (POOL) {
    constexpr int IMG = H * W;
    const int mwave = m0 + wm * WPX;
    pool_slot = mwave / IMG - m0 / IMG;
    pool_bound = (mwave / IMG + 1) * IMG;
#pragma unroll
    for (int j = 0; j < NT; ++j) poolS[j][0] = poolS[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  static_for<(HIPAC_Q8_ABL & 1) ? 0 : MT>([&](auto SUB) {
    constexpr int i = decltype(SUB)::value;
    const int m = mw0 + 16 * i;
    if constexpr (POOL) {
      // exact two-grid sums: halo16.h's pooled epilogue
      f32x4 hi[NT], lo[NT];
      const bool live = m < M;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = live ? fmaxf(v[e], 0.f) : 0.f;
          hi[j][e] = (v[e] + 12288.0f) - 12288.0f;
          lo[j][e] = ((v[e] - hi[j][e]) + 0.0234375f) - 0.0234375f;
        }
      }
      if (m0 + wm * WPX + 16 * i + 15 < pool_bound) {
#pragma unroll
        for (int j = 0; j < NT; ++j) poolS[j][0] += hi[j], poolS[j][1] += lo[j];
      } else {
        const bool in_a = m < pool_bound;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) poolS[j][0][e] += in_a ? hi[j][e] : 0.f, poolS[j][1][e] += in_a ? lo[j][e] : 0.f;
        pool_flush(pool_slot);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) poolS[j][0][e] = in_a ? 0.f : hi[j][e], poolS[j][1][e] = in_a ? 0.f : lo[j][e];
        ++pool_slot, pool_bound += H * W;
      }
      if constexpr (i == MT - 1) {
        if (pool_slot < kPoolSlots) pool_flush(pool_slot);
      }
    } else if constexpr (OUTF32) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (m < M)
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * COUT + n0 + wn * WTN + 16 * j + 4 * g) =
              make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      // lane (n, g): channels 16 j + 4 g .. + 3 of pixel perm16(n).  16-bit planes: v_permlane16_swap on the dwords of tiles
      // (2 jp, 2 jp + 1) -> 16 contiguous bytes per lane (halo16.h).  Byte planes: one dword (4 channels) per tile; the same swap
      // gives 8 contiguous bytes per pair (jp = 0: channels 16 (g & 1) + 8 (g >> 1) .. + 7, jp = 1: 32 more), and v_permlane32_swap
      // of the jp = 0 dwords' upper half with the jp = 1 dwords' lower half joins the two rows g, g ^ 2 that hold neighbouring
      // 8-channel groups: rows 0, 1 end up with channels 16 g .. + 15, rows 2, 3 with channels 32 + 16 (g & 1) .. + 15.
      // Arithmetic per PAIR of values (packed fp32 adds / multiplies, packed conversions): h = rn16(v) as a pair, l = v - h exactly,
      // lo16 = rn16(l); the byte planes come from the unrounded v and l -- e4m3(min(v, 448)), e4m3(clamp(l * 2^11)) -- one rounding
      // each instead of two.
      unsigned PH[NT][2], PL[NT][2], QH[NT], QL[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x2 bb[2] = {f32x2{bv[j].x, bv[j].y}, f32x2{bv[j].z, bv[j].w}};
        int q_h = 0, q_l = 0;
        static_for<2>([&](auto HP) {
          constexpr int hp = decltype(HP)::value;
          const f32x2 u = f32x2{acc[i][j][2 * hp], acc[i][j][2 * hp + 1]} + bb[hp];
          f32x2 v = u;
          if constexpr (RELU) v = f32x2{fmaxf(u[0], 0.f), fmaxf(u[1], 0.f)};
          const f16x2 h2 = __builtin_convertvector(v, f16x2);
          PH[j][hp] = __builtin_bit_cast(unsigned, h2);
          const f32x2 l = v - __builtin_convertvector(h2, f32x2);
          if constexpr (LO16) PL[j][hp] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, f16x2));
          if constexpr (Q8OUT) {
            const f32x2 ls = l * f32x2{kQ8LoScale, kQ8LoScale};
            // (v_med3 on the value BEFORE the ReLU: ReLU and the clamp at 448 in one, and no canonicalising v_max in front of a
            // v_min; the first conversion of a dword takes a dead register as its `old` operand instead of a zero that needs a move)
            const float c0 = __builtin_amdgcn_fmed3f(u[0], RELU ? 0.f : -448.f, 448.f), c1 = __builtin_amdgcn_fmed3f(u[1], RELU ? 0.f : -448.f, 448.f);
            const float d0 = __builtin_amdgcn_fmed3f(ls[0], -448.f, 448.f), d1 = __builtin_amdgcn_fmed3f(ls[1], -448.f, 448.f);
            q_h = __builtin_amdgcn_cvt_pk_fp8_f32(c0, c1, hp == 0 ? __builtin_bit_cast(int, u[0]) : q_h, hp == 1);
            q_l = __builtin_amdgcn_cvt_pk_fp8_f32(d0, d1, hp == 0 ? __builtin_bit_cast(int, ls[0]) : q_l, hp == 1);
          }
        });
        QH[j] = (unsigned)q_h, QL[j] = (unsigned)q_l;
      }
      T* const out_h = reinterpret_cast<T*>(outp) + (size_t)m * (2 * COUT) + c_lane;
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        permlane16_swap(PH[2 * jp][0], PH[2 * jp + 1][0]);
        permlane16_swap(PH[2 * jp][1], PH[2 * jp + 1][1]);
        if constexpr (LO16) {
          permlane16_swap(PL[2 * jp][0], PL[2 * jp + 1][0]);
          permlane16_swap(PL[2 * jp][1], PL[2 * jp + 1][1]);
        }
        if ((HIPAC_Q8_ABL & 16) ? m < 0 : m < M) {
          store16_out<false>(out_h + 32 * jp, u32x4{PH[2 * jp][0], PH[2 * jp][1], PH[2 * jp + 1][0], PH[2 * jp + 1][1]});
          if constexpr (LO16) store16_out<false>(out_h + COUT + 32 * jp, u32x4{PL[2 * jp][0], PL[2 * jp][1], PL[2 * jp + 1][0], PL[2 * jp + 1][1]});
        }
      }
      if constexpr (Q8OUT) {
        permlane16_swap(QH[0], QH[1]), permlane16_swap(QH[2], QH[3]);
        permlane16_swap(QL[0], QL[1]), permlane16_swap(QL[2], QL[3]);
        permlane32_swap(QH[0], QH[2]), permlane32_swap(QH[1], QH[3]);
        permlane32_swap(QL[0], QL[2]), permlane32_swap(QL[1], QL[3]);
        // the wave's 64 channels are one chunk of the q8 tensor: row [lo8: 64 | hi8: 64]
        unsigned char* const qrow = out_q + (size_t)m * (2 * COUT) + ((n0 + wn * WTN) >> 6) * 128 + 32 * (g >> 1) + 16 * (g & 1);
        if ((HIPAC_Q8_ABL & 16) ? m < 0 : m < M) {
          store16_out<false>(qrow, u32x4{QL[0], QL[1], QL[2], QL[3]});
          store16_out<false>(qrow + 64, u32x4{QH[0], QH[1], QH[2], QH[3]});
        }
      }
    }
  });
  prev_full = !POOL && !(HIPAC_Q8_ABL & 1) && (m0 + BM <= M);
  if constexpr (HIPAC_Q8_ABL & 1) {  // keep the accumulators alive
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123.456f) reinterpret_cast<float*>(outp)[0] = t;
  }
  }  // persistent tile loop
}

}  // namespace hipac
