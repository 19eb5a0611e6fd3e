// conv3x3_halo16_kernel: the halo direct convolution of conv_igemm.h (3x3 / stride 1 / pad 1, layers 2-4) on
// v_mfma_f32_16x16x32_{bf16,f16} instead of 32x32x16 (round 4).  Included by conv_igemm.h.
//
// Why a second MFMA shape: under the whole forward this chip sits at its package power limit (sclk ~1.98 GHz of 2.4,
// DESIGN.md section 3), so cycles per FLOP do not decide the rate -- the clock the chip can hold does, and the guide
// (MI355X_MICROARCH.md, "DVFS give-back" item 7) measures 1.12-1.14x the FLOP/s for the 16x16x32 shape over 32x32x16
// on random data with LDS-fed operands at equal cycles.  Same tile, same LDS traffic, same accumulator registers:
//
//   workgroup tile 256 pixels x 128 channels, 4 waves (2 x 2), each wave 128 px x 64 ch
//     = 8 pixel sub-tiles (16 px) x 4 channel sub-tiles (16 ch) = 32 accumulators of 4 registers (128, as before);
//   a K step of 32 channels = 8 activation + 4 weight fragments (ds_read_b128) for 32 MFMAs
//     = 0.75 fragment reads per 32x32x16-equivalent, as before.
//
// Operands (guide section 3, "A/B operand lane maps"): lane l = (n = l & 15, g = l >> 4) holds A[row n][k = 8 g + j] and
// B[k = 8 g + j][col n]; D: lane holds rows 4 g .. 4 g + 3 of column n.  Weights are A (rows = output channels),
// activations B (columns = pixels), so a lane's four accumulator registers are four CONSECUTIVE channels of one pixel.
//
// LDS bank conflicts.  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32 for the
// upper half): with the 16x16x32 operand map a group mixes columns {0-3, 12-15} of k-group g with columns {4-11} of k-group
// g + 1, i.e. two DIFFERENT 16-byte chunks of the 128-byte pixel rows.  In the band image (slot q at q * 128, chunk c at
// c ^ ((q >> 1) & 7)) the 256-byte bank row holds one even and one odd slot, so the group is conflict-free for any pair of
// chunks iff its two column sets fall on slots of different parity.  Hence the lane -> pixel map is not the identity:
//     perm16(n) = 2 n (n < 4), 2 n - 7 (4 <= n < 12), 2 n - 16 (n >= 12)
// columns {0-3, 12-15} take the even pixels of a 16-pixel sub-tile, columns {4-11} the odd ones; eight same-parity slots
// with one chunk cover the eight chunk positions of their half of the bank row ((q >> 1) & 7 takes every value once).  The
// weight ring has the same row format, so LDS row j of a 16-row group holds output channel perm16_inv(j) -- applied
// to the SOURCE row of the weight DMA (computed once per tile), which keeps D's rows in natural channel order.
//
// Everything else (contiguous band per 64-channel chunk by LDS-DMA, zero slots by parity for image-edge taps, 2-slot weight
// ring one tap ahead through a buffer descriptor, persistent workgroups, per-wave staged epilogue, folded projection
// PCIN) is the 32x32 kernel's design; see the comment above conv3x3_halo_kernel.
#pragma once

namespace hipac {

template <typename T> struct Elem16;
template <> struct Elem16<__bf16> {
  static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem16<_Float16> {
  static __device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// The K step's instruction stream is written out (HIPAC_H16_ASM): hipcc orders every ds_read_b128 right in front of an
// `s_waitcnt lgkmcnt(0)` (it does not count LDS reads in flight in this loop), which with 16-cycle MFMAs in groups of four
// leaves ~64 cycles of cover for an LDS round trip, and it renames accumulators between MFMAs (extra live ranges, s_nop
// padding).  Here: reads HIPAC_H16_AHEAD sub-tiles ahead, counted waits, accumulators updated in place.  asm volatile
// statements keep their order among themselves; nothing but these statements touches the fragments.
template <typename T> struct Asm16;
template <> struct Asm16<__bf16> {
  static __device__ __forceinline__ void mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
};
template <> struct Asm16<_Float16> {
  static __device__ __forceinline__ void mfma(f32x4& c, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  }
};
template <int OFF, typename F>
__device__ __forceinline__ void lds_read16(F& f, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
#if defined(__HIP_DEVICE_COMPILE__)
// rows = 16 lanes: a.row1 <-> b.row0, a.row3 <-> b.row2 (tools/permswap16_probe.hip)
__device__ __forceinline__ void permlane16_swap(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
#else
__device__ inline void permlane16_swap(unsigned&, unsigned&) {}
#endif
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
  static_assert(N >= 0 && N < 16, "lgkmcnt range");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
}

// tap order inside a half-chunk.  Stride 1: row-major.  Stride 2: by plane -- (1,1): taps (0,0) (0,2) (2,0) (2,2); (0,1): (1,0)
// (1,2); (1,0): (0,1) (2,1); (0,0): (1,1) -- so that a plane's taps are consecutive and the band changes at positions 0, 4, 6, 8.
// (Arithmetic on a packed constant, not a table: a table indexed at run time becomes a scalar LOAD, and scalar loads share the
// lgkmcnt counter the written-out MFMA stream counts its LDS reads with.)
template <int STRIDE> __host__ __device__ constexpr int b16_tap(int k) { return STRIDE == 1 ? k : (int)((0x471538620ull >> (4 * k)) & 15); }
template <int STRIDE> __host__ __device__ constexpr int b16_kh(int k) { return b16_tap<STRIDE>(k) / 3; }
template <int STRIDE> __host__ __device__ constexpr int b16_kw(int k) { return b16_tap<STRIDE>(k) % 3; }
__host__ __device__ constexpr int perm16(int n) { return n < 4 ? 2 * n : (n < 12 ? 2 * n - 7 : 2 * n - 16); }
__host__ __device__ constexpr int perm16_inv(int j) { return (j & 1) ? (j + 7) / 2 : (j < 8 ? j / 2 : j / 2 + 8); }

#ifndef HIPAC_H16_ASM
#define HIPAC_H16_ASM 1   // 1: the K step as written-out asm statements (see Asm16); 0: builtins, hipcc's schedule
#endif
#ifndef HIPAC_H16_DIRECT
#define HIPAC_H16_DIRECT 1  // 1: epilogue straight from the accumulators (v_permlane16_swap pairs 16-lane rows into 16-byte
                            // items), the next tile's band AND first weight tiles prefetched behind it; 0: staged through LDS
#endif
#ifndef HIPAC_H16_RESID_MFMA
#define HIPAC_H16_RESID_MFMA 1  // 1: the residual is added by the matrix pipe (identity "weights" over the residual tile brought
                                // into LDS by DMA, after the last tap); 0: loaded into registers and added in the epilogue
#endif
#ifndef HIPAC_H16_EPI_PRIO
#define HIPAC_H16_EPI_PRIO 0  // s_setprio level of the epilogue's store loop (the K loop runs at 1)
#endif
#ifndef HIPAC_H16_STAGE16
#define HIPAC_H16_STAGE16 0  // 1: 16-bit outputs leave through a 4 KB LDS transpose per wave as whole cache lines (see STAGE16; measured equal)
#endif
#ifndef HIPAC_H16_XCD_CHUNKS
#define HIPAC_H16_XCD_CHUNKS 1  // 1: M-tiles dealt to the XCDs in contiguous runs (neighbouring tiles share W + 1 band pixels: L2 hits;
                                // entry convs -2..-3 %, the rest +-0), 0: round-robin
#endif
#ifndef HIPAC_H16_ABL
#define HIPAC_H16_ABL 0  // developer builds (wrong results): 1 no per-step barrier, 2 no weight DMA in the K loop, 4 no fragment waits, 8 no wait for the weight DMA, 16 no image-edge selects, 32 no stores, 64 (stride-2 form) no band reload at the plane switches, 128 stores wrapped into a 16 K-pixel window (the instructions without the HBM write traffic)
#endif
#ifndef HIPAC_H16_SB
#define HIPAC_H16_SB 1
#endif
#ifndef HIPAC_H16_AHEAD
#define HIPAC_H16_AHEAD (HIPAC_H16_ASM ? 3 : 2)  // activation fragments in flight ahead of the sub-tile whose MFMAs are being issued
#endif

// POOL (the network's last conv: OUTF32, RELU): instead of storing the fp32 map [n][H*W][COUT] (0.1 MB per patch, read back by
// the head kernel), the epilogue reduces the tile's pixels per IMAGE -- the global average pool's sum -- and writes partial
// sums float[n_mtiles][WM][kPoolSlots][2][COUT] (`outp`; [..][0] = grid-2^-10 parts, [..][1] = remainders, see the epilogue):
// slot r of M-tile mt = image (mt * BM) / (H*W) + r.  Per lane the eight
// pixels are summed in order into the accumulator set of the sub-tile's first image (A) or of the next one (B; a 16-pixel
// sub-tile meets at most two images), a set is reduced over the 16 lanes of a row (four DPP rotations) and written when the
// walk leaves its image: no atomics, one fixed order, so the features are reproducible bit for bit.  hipac_capi.hip's
// head_pool_kernel adds the <= 2 x WM partial sums of an image in a fixed order, divides by H*W and applies the fc.
constexpr int kPoolSlots = 7;  // images a 256-pixel tile of 49-pixel maps can meet (1 + 5 x 49 + 10)

#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row, in every lane (lane n adds its partners in the order n+8, n+4, n+2, n+1 mod 16)
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  return dpp_add<0x121>(v);  // row_ror:1
}
#else
__device__ inline float row16_sum(float v) { return v; }
#endif

template <typename T, int CIN, int COUT, int H, int W, int BM, int BN, int NSW, bool RELU, bool RESID, bool OUTF32, int PCIN = 0,
          bool POOL = false, bool S2 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo16_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                                const float* __restrict__ bias, const T* __restrict__ resid,
                                                                void* __restrict__ outp, int M, int n_img, int n_mtiles,
                                                                const char* __restrict__ zero_page,
                                                                const T* __restrict__ wgt_p = nullptr) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int CC = CIN / 64;                      // 64-channel chunks of the K loop
  constexpr int KTOT = 9 * CIN;
  constexpr int WM = 2, WN = 2;                     // 4 waves: pixel parts x channel parts
  constexpr int WPX = BM / WM;                      // pixels per wave
  constexpr int MT = WPX / 16;                      // 16-pixel sub-tiles per wave (8)
  constexpr int G32 = WPX / 32;                     // 32-pixel epilogue groups per wave
  constexpr int WTN = BN / WN, NT = WTN / 16;       // channels per wave, 16-wide tiles per wave (4)
  constexpr int A_PIECES = halo_band_pieces(W, BM);
  constexpr int A_BYTES = A_PIECES * 1024;
  constexpr int W_BYTES = BN * 128;
  constexpr int WPW = BN / 8 / 4;                   // W pieces per wave per tap
  constexpr int NTILES_N = COUT / BN;
  constexpr int PCC = PCIN / 64;                    // projection K steps (0: no folded projection)
  constexpr int NSTEP = 9 * CC + PCC;
  constexpr bool RESID_MFMA = RESID && HIPAC_H16_RESID_MFMA && HIPAC_H16_ASM && HIPAC_H16_DIRECT;  // see "the residual, added on the matrix pipe"
  constexpr bool EPI_RESID = RESID && !RESID_MFMA;  // the epilogue loads and adds the residual itself
  static_assert(sizeof(T) == 2, "16-bit operands");
  // S2: a 3x3 / STRIDE 2 conv (H x W = the OUTPUT map, the input is 2H x 2W): the nine taps fall on the four parity planes of the
  // input (band16.h has the geometry), each a stride-1 problem on the output grid, so a K step is unchanged and only the
  // band differs -- per 64-channel chunk FOUR bands (planes (1,1), (0,1), (1,0), (0,0) with 4 / 2 / 2 / 1 taps), gathered
  // pixel by pixel (full 128-byte rows), one after the other into the single band buffer.  Each band's DMA round trip is
  // exposed, which is why layers 3-4 use band16.h's double-buffered half-chunk bands instead; for CIN = 64 (layer2's entry)
  // a half-chunk band is a HALF cache line per pixel and the DMA engine, not the exposure, bounds that kernel.
  static_assert(!S2 || (PCIN == 0 && !RESID && !POOL && !OUTF32 && HIPAC_H16_DIRECT && HIPAC_H16_ASM), "stride-2 form: plain entry conv");
  static_assert(!POOL || (OUTF32 && RELU && HIPAC_H16_DIRECT && (BM + H * W - 1) / (H * W) + 1 <= kPoolSlots && H * W > 16),
                "pooled epilogue: the fp32 form of the direct epilogue, maps of more than 16 pixels");
  static_assert(PCIN % 64 == 0 && (PCIN == 0 || !RESID), "folded projection replaces the residual input");
  static_assert((BM == 128 || BM == 256) && WTN % 16 == 0 && COUT % BN == 0 && CIN % 64 == 0 && MT <= 8, "tile shape");
  static_assert((BN / 8) % 4 == 0, "W piece split");
  static_assert(NSW == 2 || NSW == 3, "weight ring depth");
  constexpr int SROWW = WTN * 4 + 16;               // staging row: WTN fp32 + pad
  constexpr int STG_BYTES = 4 * 32 * SROWW;         // 4 waves x [32 px][WTN fp32 + pad] epilogue staging
  constexpr int S_BYTES = NSW * W_BYTES > STG_BYTES ? NSW * W_BYTES : STG_BYTES;  // ring, aliased by the staging
  static_assert(A_BYTES + S_BYTES <= 80 * 1024, "LDS: two workgroups per CU");
  // the last sub-tile's taps may read past the tile's own band (tail tiles are not clamped): still inside the band region
  static_assert(BM + 2 * W + 4 <= A_PIECES * 8, "band slots");

  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  unsigned char* const Abuf = ring;
  unsigned char* const Wbuf = ring + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int n16 = lane & 15, g = lane >> 4;
  const int pn = n16 < 4 ? 2 * n16 : (n16 < 12 ? 2 * n16 - 7 : 2 * n16 - 16);  // perm16(n16)

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const char* in_b = reinterpret_cast<const char*>(in);
  const char* w_b = reinterpret_cast<const char*>(wgt);
  const int prow = lane >> 3, dchunk = lane & 7;

  // band of the tile starting at pixel m0_: the contiguous pixel range [m0_ - W - 1, mlast_ + W + 1], slot q = pixel
  // m0_ - W - 3 + q (slots 0, 1 = zeros).  Through a buffer descriptor over the activation map: pixels before the first image
  // (negative offsets wrap to huge unsigned ones) and after the last one read as zeros by the range check, slots 0 and 1 are
  // sent out of range explicitly, and a lane's offset inside a piece does not depend on the piece -- a wave's pieces
  // p = wave, wave + 4, ... all have p's parity, so the swizzle term (q >> 1) & 7 = (4 p + (prow >> 1)) & 7 is a per-lane
  // constant: one add per piece.  (Slots past the band's end receive whatever pixels follow; no stored result reads them.)
  const rsrc_t a_rsrc = make_rsrc(in_b, S2 ? n_img * (4 * H * W * CIN * 2) : M * CIN * 2);
  const int a_lane = (prow * CIN + (dchunk ^ ((4 * wave + (prow >> 1)) & 7)) * 8) * 2;
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
  // band of the plane of tap position k (band16.h's order: positions 0-3 plane (1,1), 4-5 (0,1), 6-7 (1,0), 8 (0,0)), chunk cc
  auto issue_plane_band = [&](int k, int cc) {
    if constexpr (S2) {
      const int py = k < 4 ? 1 : (k < 6 ? 0 : (k < 8 ? 1 : 0)), px = k < 6 ? 1 : 0;
      const int pofs = (py * 2 * W + px) * CIN * 2 + cc * 128 + (dchunk ^ ((4 * wave + (prow >> 1)) & 7)) * 16;
#pragma unroll
      for (int kk = 0; kk < NPB; ++kk) {
        const int p = wave + 4 * kk;
        if (p < A_PIECES) buffer_load_lds16(a_rsrc, Abuf + p * 1024, b_pix[kk] < 0 ? (int)0x80000000 : b_pix[kk] * (CIN * 2) + pofs, 0);
      }
    }
  };
  auto issue_band_of = [&](int m0_, int cc) {
    const int mlast_ = (m0_ + BM <= M ? m0_ + BM : M) - 1;
    const int npieces_ = (mlast_ - m0_ + 1 + 2 * W + 2 + 2 + 7) >> 3;
    const int base = ((m0_ - W - 3) * CIN + cc * 64) * 2;  // byte offset of slot 0's pixel (may be negative)
    for (int p = wave; p < npieces_; p += 4) {
      int off = a_lane + base + p * (8 * CIN * 2);
      if (p == 0 && prow < 2) off = (int)0x80000000;
      buffer_load_lds16(a_rsrc, Abuf + p * 1024, off, 0);
    }
  };

  // DIRECT: the next tile's band and first weights are requested BEFORE the epilogue's stores, and vector-memory operations
  // retire in order: at the next tile's first step it is enough to wait until only those stores are outstanding -- if every
  // one of them was issued, i.e. the tile was full (a store whose lanes are all past M may be branched around)
  constexpr int N_EPI_STORES = MT * (OUTF32 ? NT : NT / 2);
  bool prev_full = false;
  for (int vb = blockIdx.x, first_tile = 1;; vb += gridDim.x, first_tile = 0) {
  const int xcd = vb & 7, slot = vb >> 3;
#if HIPAC_H16_XCD_CHUNKS
  // every XCD takes a contiguous run of M-tiles: neighbouring tiles share W + 1 band pixels, which then hit that XCD's L2
  const int mt_q = (n_mtiles + 7) >> 3;
  const int mt = (slot / NTILES_N) < mt_q ? xcd * mt_q + slot / NTILES_N : n_mtiles;
#else
  const int mt = (slot / NTILES_N) * 8 + xcd;
#endif
  const int nt = slot % NTILES_N;
  if (mt >= n_mtiles) break;  // mt grows with vb on a fixed XCD: nothing valid follows
  const int m0 = mt * BM, n0 = nt * BN;
  const int mlast = (m0 + BM <= M ? m0 + BM : M) - 1;
  const int mstart = m0 - W - 1;
  constexpr bool DIRECT = HIPAC_H16_DIRECT != 0;
  HALO_STAMP(t_start);
  if (!DIRECT && !first_tile) __builtin_amdgcn_s_barrier();  // the previous tile's staging reads are done: ring is free
  // weight DMA: LDS row `row` of the tile takes output channel (row & ~15) | perm16_inv(row & 15)
  int w_off[WPW];
  int wp_off[PCC > 0 ? WPW : 1];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int row = (wave + 4 * i) * 8 + prow;
    const int j = row & 15;
    const int srow = (row & ~15) | ((j & 1) ? (j + 7) >> 1 : (j < 8 ? j >> 1 : (j >> 1) + 8));
    w_off[i] = ((n0 + srow) * KTOT + (dchunk ^ ((row >> 1) & 7)) * 8) * 2;
    if constexpr (PCC > 0) wp_off[i] = ((n0 + srow) * PCIN + (dchunk ^ ((row >> 1) & 7)) * 8) * 2;
  }
  const rsrc_t w_rsrc = make_rsrc(w_b, COUT * KTOT * 2);
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
    const int kofs_bytes = ((S2 ? b16_tap<2>(tap) : tap) * CIN + cc * 64) * 2;  // (S2: `tap` is the position in plane order)
    static_for<WPW>([&](auto I) {
      constexpr int i = decltype(I)::value;
      buffer_load_lds16(w_rsrc, Wbuf + slot_ * W_BYTES + (wave + 4 * i) * 1024, w_off[i], kofs_bytes);
    });
  };
  // folded projection: the block input's pixel (2y, 2x) of every output pixel of the tile, 64 channels of chunk pc, at the
  // slot the centre tap reads for that output pixel
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

  // ---- consumer side ----
  // slot of this lane's pixel of sub-tile 0 for the centre tap; sub-tile i is 16 slots on.  Lanes past the end of a tail
  // tile are NOT clamped: they read slots the band DMA did not fill (inside the band region; MFMA columns are independent
  // and those columns are never stored).
  const int mw0 = m0 + wm * WPX + pn;
  const int q0 = mw0 - mstart + 2;
  // image-edge flags of the lane's pixel in each sub-tile, 4 bits each: bit 0: x == 0, 1: x == W-1, 2: y == 0, 3: y == H-1
  unsigned epk = 0;
  {
    // (x, y) of sub-tile 0's pixel by division, the others by stepping 16 pixels on: 16 = (16 / W) rows + (16 % W) columns
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
  HALO_STAMP(t_setup);
  const int ck0 = g << 4;                                                    // k32 step 0: chunk g (step 1: chunk 4 + g)
  const int rdw0 = (wn * WTN + pn) * 128 + ((g ^ ((pn >> 1) & 7)) << 4);     // weight fragment, k32 step 0, channel tile 0

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // One K step = 64 channels of one tap = two k32 sub-steps of MT x NT MFMAs.  a_addr[i] = byte address of the lane's
  // fragment of sub-tile i for k32 sub-step 0 (sub-step 1: ^ 64).  `mid` runs once, inside the MFMA stream.
#if HIPAC_H16_ASM
  const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)ring;  // LDS byte address of the band region
  auto k_step = [&](const unsigned char* wst, const int (&a_addr)[MT], auto&& mid) {
    constexpr int AH = HIPAC_H16_AHEAD;
    constexpr int NS = 2 * MT;  // (k32 sub-step, pixel sub-tile) pairs in issue order
    static_assert(AH >= 1 && AH <= 3 && MT >= 2 + NT, "read-ahead schedule");
    frag wf[2][NT];
    frag af[AH + 1];
    const unsigned w0 = lds0 + (unsigned)(wst - ring) + (unsigned)rdw0;
    const unsigned w1 = w0 ^ 64u;
    unsigned aa[NS];
#pragma unroll
    for (int i = 0; i < MT; ++i) aa[i] = lds0 + (unsigned)a_addr[i], aa[MT + i] = aa[i] ^ 64u;
    // reads in issue order: W0[0..NT), A[0..AH), then per sub-step s: A[s + AH], and W1[s - 1] for s = 1..NT
    static_for<NT>([&](auto J) { lds_read16<decltype(J)::value * 2048>(wf[0][decltype(J)::value], w0); });
    static_for<AH>([&](auto S) { lds_read16<0>(af[decltype(S)::value], aa[decltype(S)::value]); });
    __builtin_amdgcn_s_setprio(1);
    static_for<NS>([&](auto S) {
      constexpr int s = decltype(S)::value, kk = s / MT, i = s % MT;
      if constexpr (s + AH < NS) lds_read16<0>(af[(s + AH) % (AH + 1)], aa[s + AH]);
      if constexpr (s >= 1 && s <= NT) lds_read16<(s - 1) * 2048>(wf[1][s - 1], w1);
      // LDS returns in order: wait until only the reads issued AFTER A[s] are outstanding.  Those are A[s+1 .. min(s+AH,
      // NS-1)] and the W1 reads of sub-steps t in [max(1, s - AH) .. min(s, NT)] (A[s] was issued first thing in sub-step
      // s - AH, or in the prologue for s < AH).  W0 is older than A[0]; every W1 fragment (issued by sub-step NT) is older than
      // A[MT] (issued in sub-step MT - AH), the first fragment whose MFMAs need W1.
      constexpr int a_after = (s + AH < NS ? AH : NS - 1 - s);
      constexpr int w_lo = (s - AH > 1 ? s - AH : 1), w_hi = (s < NT ? s : NT);
      constexpr int w_after = w_hi >= w_lo ? w_hi - w_lo + 1 : 0;
      static_assert(MT - AH >= NT, "W1 fragments are older than the first activation fragment that needs them");
      if constexpr (!(HIPAC_H16_ABL & 4)) wait_lgkmcnt<a_after + w_after>();
#pragma unroll
      for (int j = 0; j < NT; ++j) Asm16<T>::mfma(acc[i][j], wf[kk][j], af[s % (AH + 1)]);
      if constexpr (s == MT / 2) mid();
    });
    __builtin_amdgcn_s_setprio(0);
  };
#else
  auto k_step = [&](const unsigned char* wst, const int (&a_addr)[MT], auto&& mid) {
    frag wf[2][NT];
    frag af[HIPAC_H16_AHEAD + 1];
    constexpr int NS = 2 * MT;  // (k32 sub-step, pixel sub-tile) pairs in issue order
    auto a_of = [&](int s) -> frag {
      const int kk = s / MT, i = s % MT;
      return *reinterpret_cast<const frag*>(Abuf + (kk ? a_addr[i] ^ 64 : a_addr[i]));
    };
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[0][j] = *reinterpret_cast<const frag*>(wst + j * 2048 + rdw0);
#pragma unroll
    for (int s = 0; s < HIPAC_H16_AHEAD; ++s) af[s] = a_of(s);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int kk = s / MT, i = s % MT;
      if (s + HIPAC_H16_AHEAD < NS) af[(s + HIPAC_H16_AHEAD) % (HIPAC_H16_AHEAD + 1)] = a_of(s + HIPAC_H16_AHEAD);
      if (kk == 0 && i >= 2 && i < 2 + NT)  // the second sub-step's weight fragments, one per sub-tile
        wf[1][i - 2] = *reinterpret_cast<const frag*>(wst + (i - 2) * 2048 + (rdw0 ^ 64));
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = Elem16<T>::mfma(wf[kk][j], af[s % (HIPAC_H16_AHEAD + 1)], acc[i][j]);
      if (s == MT / 2) mid();
#if HIPAC_H16_SB
      __builtin_amdgcn_sched_barrier(0);  // keep the reads HIPAC_H16_AHEAD sub-tiles ahead of their MFMAs, as written
#endif
    }
    __builtin_amdgcn_s_setprio(0);
  };
#endif

  // Epilogue geometry (per WAVE, no workgroup barriers): 32-pixel groups through the wave's private fp32 staging
  // [32 px][WTN] and out as 16-byte items (8 channels): item = lane + 64k -> pixel item / CPW, channel group lane % CPW.
  constexpr int CPW = WTN / 8;                      // 8-channel items per pixel (wave's channel half)
  constexpr int IPT = 32 * CPW / 64;                // items per lane and group
  static_assert(64 % CPW == 0 && (32 * CPW) % 64 == 0, "epilogue items");
  const int e_c0 = n0 + wn * WTN + (lane % CPW) * 8;  // first of this lane's 8 output channels
  const int e_px = lane / CPW;                        // pixel of item k: e_px + k * (64 / CPW)
  frag rv[2][RESID ? IPT : 1];
  auto load_resid = [&](auto SUB) {
    constexpr int i = decltype(SUB)::value;
    if constexpr (RESID) {
#pragma unroll
      for (int k = 0; k < IPT; ++k) {
        int m = m0 + wm * WPX + i * 32 + e_px + k * (64 / CPW);
        m = m < M ? m : M - 1;  // unconditional load from a valid row (tail rows are never stored)
        rv[i & 1][k] = *reinterpret_cast<const frag*>(resid + (size_t)m * COUT + e_c0);
      }
    }
  };

#ifdef HIPAC_HALO_STAMPS
  unsigned long long t_first = 0;
#endif
  int s = 0;  // K step counter
  if constexpr (S2) plane_offsets(m0);
  if (first_tile) {
    if constexpr (S2) issue_plane_band(0, 0);
    else issue_band_of(m0, 0);
  }
  if (!DIRECT || first_tile) {  // (DIRECT: the previous tile's epilogue has requested them)
#pragma unroll
    for (int pstep = 0; pstep < NSW - 1; ++pstep)
      if (pstep < NSTEP) issue_w(pstep, pstep);
  }
  for (int cc = 0; cc < CC; ++cc) {
    if (cc > 0) {
      __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous chunk's band
      if constexpr (S2) issue_plane_band(0, cc);
      else issue_band_of(m0, cc);
    }
#pragma unroll HIPAC_HALO_TAP_UNROLL
    for (int tap = 0; tap < 9; ++tap, ++s) {
      if (S2 && !(HIPAC_H16_ABL & 64) && (tap == 4 || tap == 6 || tap == 8)) {  // the next plane's band (its round trip is exposed: the wait below drains it)
        __builtin_amdgcn_s_barrier();
        issue_plane_band(tap, cc);
      }
      // W(s) must have landed; the band too at tap 0 (it was issued AFTER W(s+1..), so drain everything)
      if (NSW == 3 && tap != 0 && s + 1 < NSTEP) wait_vmcnt<WPW>();
      else if (DIRECT && s == 0 && prev_full) wait_vmcnt<N_EPI_STORES>();  // the prefetch is older than the epilogue's stores
      else if (!(HIPAC_H16_ABL & 8) || tap == 0) wait_vmcnt<0>();  // (ablation 8: weight tiles are not waited for)
      if (!(HIPAC_H16_ABL & 1) || tap == 0) __builtin_amdgcn_s_barrier();
#ifdef HIPAC_HALO_STAMPS
      if (s == 0) {
        t_first = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
#endif
      const int tap_w = S2 ? b16_tap<2>(tap) : tap;  // the tap's index in the weights
      const int kh = tap_w / 3, kw = tap_w - kh * 3;
      const int toff = S2 ? (kh == 0 ? -W : 0) + (kw == 0 ? -1 : 0) : (kh - 1) * W + kw - 1;
      const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
      const unsigned tapmask = ((kw == 0 ? 1u : 0u) | (!S2 && kw == 2 ? 2u : 0u) | (kh == 0 ? 4u : 0u) | (!S2 && kh == 2 ? 8u : 0u)) * 0x11111111u;
      // out-of-image taps read a zero pixel: slot 0 or 1 by the parity of the slot the lane would have read, at the chunk
      // position its swizzle selects -- the same 16-byte bank group as the in-image address.  (qt + 16 i) has the parity and
      // the swizzle of qt: one swizzle term per tap.
      const int qt = q0 + toff;
      const int x0 = ck0 ^ (((qt >> 1) & 7) << 4);
      const int a_in = (qt << 7) + x0;
      const int a_zero = ((qt & 1) << 7) + x0;
      // (through asm: hipcc would otherwise re-associate (epk & tapmask) & nibble_i into eight pre-masked copies of epk that
      // live in registers across the whole tile)
      unsigned em;
      asm("v_and_b32 %0, %1, %2" : "=v"(em) : "s"(tapmask), "v"(epk));
      int a_addr[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a_addr[i] = (HIPAC_H16_ABL & 16) ? a_in + 2048 * i : ((em & (0xFu << (4 * i))) ? a_zero : a_in + 2048 * i);
      k_step(wst, a_addr, [&] {
        // the next step's weight DMA is issued from inside the MFMA stream (its slot was freed by this step's barrier)
        if (!(HIPAC_H16_ABL & 2) && s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if constexpr (PCC > 0) {
    // ---- the folded projection: PCC more K steps, centre tap only (no image-edge cases: pixel (2y, 2x) always exists)
    for (int pc = 0; pc < PCC; ++pc, ++s) {
      __builtin_amdgcn_s_barrier();  // every wave has finished reading the previous band
      issue_gather(pc);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (s + NSW - 1 < NSTEP) issue_w(s + NSW - 1, (s + NSW - 1) % NSW);
      const unsigned char* wst = Wbuf + (s % NSW) * W_BYTES;
      const int a_in = (q0 << 7) + (ck0 ^ (((q0 >> 1) & 7) << 4));
      int a_addr[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a_addr[i] = a_in + 2048 * i;
      k_step(wst, a_addr, [] {});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  if constexpr (RESID_MFMA) {
    // ---- the residual, added on the matrix pipe.  A direct epilogue would have to LOAD it in the accumulator layout: 16 bytes
    // per lane with neighbouring lanes on different pixels -- four times the address-coalescer cycles of a contiguous read,
    // in bursts of 16 loads per wave that queue in front of the other workgroup's weight DMA (measured: 6-8 k cycles to issue
    // them, 7-10 k more in the store loop; the residual convs' epilogues were 2x the others').  Instead the tile's residual,
    // [256 px][128 ch], comes in by LDS-DMA in the band's own row format (full 128-byte lines per pixel, no registers): the
    // 64 channels of the wn = 0 waves into the band region, those of the wn = 1 waves into the weight ring (both are free
    // now), and every accumulator takes ONE more MFMA: D += I x R with I the 16 x 32 identity fragment that picks the tile's
    // 16 channels out of the 32 of a k32 step.  1.0 x r is exact and lands in the fp32 accumulator: the same single rounding
    // as the epilogue's `+ (float)r`.  32 MFMAs + 16 fragment reads per wave, one DMA round trip.
    __builtin_amdgcn_s_barrier();  // every wave has finished the last K step: band and ring are free
    const rsrc_t r_rsrc = make_rsrc(reinterpret_cast<const char*>(resid), M * COUT * 2);
    {
      // piece p of chunk c = slots 8p .. 8p+7 (slot = pixel - m0); a wave's pieces share p's parity: per-lane swizzle term
      const int r_lane = (prow * COUT + (dchunk ^ ((4 * wave + (prow >> 1)) & 7)) * 8) * 2;
      const int r_base = (m0 * COUT + n0) * 2;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < BM / 8 / 4; ++k) {
          const int p = wave + 4 * k;
          buffer_load_lds16(r_rsrc, (c ? Wbuf : Abuf) + p * 1024, r_lane + r_base + c * 128 + p * (8 * COUT * 2), 0);
        }
    }
    static_assert(A_BYTES >= BM * 128 && S_BYTES >= BM * 128 && BN == 128, "residual tile: one 64-channel chunk per region");
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // identity fragments: lane (n, g) holds k = 8 g .. 8 g + 7 of row n; row n of an EVEN 16-channel tile is k = n of its
    // k32 step (channels 32 kk .. + 31), of an ODD tile k = 16 + n
    frag ident[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int e = 0; e < 8; ++e) ident[o][e] = (g == 2 * o + (n16 >> 3) && e == (n16 & 7)) ? (T)1.0f : (T)0.0f;
    {
      const int r0 = wm * WPX + pn;  // slot of sub-tile 0's pixel
      const unsigned rb = lds0 + (unsigned)((wn ? Wbuf : Abuf) - ring) + (unsigned)(r0 * 128 + ((g ^ ((r0 >> 1) & 7)) << 4));
      frag rf[4];
      // reads: (i, kk) in the order (0,0) (0,1) (1,0) ...; two sub-tiles in flight
      static_for<2>([&](auto S) { lds_read16<(decltype(S)::value >> 1) * 2048>(rf[decltype(S)::value], (decltype(S)::value & 1) ? rb ^ 64u : rb); });
      static_for<2 * MT>([&](auto S) {
        constexpr int s2 = decltype(S)::value, i = s2 >> 1, kk = s2 & 1;
        if constexpr (s2 + 2 < 2 * MT) lds_read16<((s2 + 2) >> 1) * 2048>(rf[(s2 + 2) & 3], ((s2 + 2) & 1) ? rb ^ 64u : rb);
        wait_lgkmcnt<(s2 + 2 < 2 * MT) ? 2 : (2 * MT - 1 - s2)>();
        Asm16<T>::mfma(acc[i][2 * kk], ident[0], rf[s2 & 3]);
        Asm16<T>::mfma(acc[i][2 * kk + 1], ident[1], rf[s2 & 3]);
      });
    }
  }
  // ---- epilogue -------------------------------------------------------------------------------
  HALO_STAMP(t_loop);
#if HIPAC_H16_ASM
  // the accumulators were last written by MFMAs hipcc does not know about: the wait states it would have put in front of
  // their first reader (XDL write -> VALU / LDS read) are spelled out, once per tile
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#endif
  if constexpr (DIRECT) {
    // ---- direct epilogue: no LDS.  Lane (n, g) holds channels 16 j + 4 g .. + 3 of pixel perm16(n) of every sub-tile: two
    // packed dwords per 16-wide tile j.  v_permlane16_swap on the dwords of tiles (2 jp, 2 jp + 1) leaves every lane with
    // 16 contiguous bytes -- row g of the wave stores channels 32 jp + 16 (g & 1) + 8 (g >> 1) .. + 7 -- so one store
    // instruction writes 64 contiguous bytes per pixel (the 32x32 form of this idea gave 32-byte runs and lost to the
    // staged form).  The residual arrives in the stored layout and is un-paired by the same swap (an involution).
    float4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + n0 + wn * WTN + 16 * j + 4 * g);
    const int c_lane = n0 + wn * WTN + 16 * (g & 1) + 8 * (g >> 1);  // + 32 jp: first of the 8 channels this lane stores
    // STAGE16: the 16-byte items (one pixel each, neighbouring lanes on DIFFERENT pixels: every store instruction touches 16
    // half lines and costs the address coalescer 4x the cycles of a contiguous one -- removing the stores altogether made the
    // layer2 / layer3 convs 16-22 % faster, most of it the OTHER workgroup's weight DMA queueing behind these bursts) go through
    // LDS once more, 32 pixels at a time, and leave as whole 128-byte lines
    constexpr bool STAGE16 = HIPAC_H16_STAGE16 && !OUTF32 && NSW == 2 && WTN == 64;
    unsigned char* const Sw16 = Wbuf + W_BYTES + wave * 4096;
    // ALL of the tile's residual loads are requested up front (the K loop's fragment registers are free now): vector-memory
    // operations retire in order, so a load requested later would queue behind the next tile's band and weight DMA below
    u32x4 rq[MT][EPI_RESID && !OUTF32 ? NT / 2 : 1];
    u32x2 rq32[MT][EPI_RESID && OUTF32 ? NT : 1];
    auto load_resid_d = [&](auto SUB) {
      constexpr int i = decltype(SUB)::value;
      if constexpr (EPI_RESID) {
        int m = mw0 + 16 * i;
        m = m < M ? m : M - 1;  // unconditional load from a valid row (tail rows are never stored)
        if constexpr (OUTF32) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
            rq32[i][j] = *reinterpret_cast<const u32x2*>(resid + (size_t)m * COUT + n0 + wn * WTN + 16 * j + 4 * g);
        } else {
#pragma unroll
          for (int jp = 0; jp < NT / 2; ++jp)
            rq[i][jp] = *reinterpret_cast<const u32x4*>(resid + (size_t)m * COUT + c_lane + 32 * jp);
        }
      }
    };
    static_for<MT>([&](auto SUB) { load_resid_d(SUB); });
    __builtin_amdgcn_s_barrier();  // every wave has left the K loop: band and ring are free
    HALO_STAMP(t_bar);
    {
      // the next tile's first band chunk and first weight tile(s) land behind this epilogue
      const int vn = vb + gridDim.x;
#if HIPAC_H16_XCD_CHUNKS
      const int mtn = ((vn >> 3) / NTILES_N) < ((n_mtiles + 7) >> 3) ? (vn & 7) * ((n_mtiles + 7) >> 3) + (vn >> 3) / NTILES_N : n_mtiles;
#else
      const int mtn = ((vn >> 3) / NTILES_N) * 8 + (vn & 7);
#endif
      if (mtn < n_mtiles) {
        if constexpr (S2) {
          plane_offsets(mtn * BM);
          issue_plane_band(0, 0);
        } else {
          issue_band_of(mtn * BM, 0);
        }
        const int dn = (((vn >> 3) % NTILES_N) * BN - n0) * KTOT * 2;  // the next tile's weight rows relative to this one's
#pragma unroll
        for (int i = 0; i < WPW; ++i) w_off[i] += dn;
#pragma unroll
        for (int pstep = 0; pstep < NSW - 1; ++pstep)
          if (pstep < 9 * CC) issue_w(pstep, pstep);
      }
    }
    HALO_STAMP(t_pref);
#if HIPAC_H16_EPI_PRIO
    __builtin_amdgcn_s_setprio(HIPAC_H16_EPI_PRIO);
#endif
    // pooled epilogue state: S[..][0] = sums of the values rounded to the grid 2^-10, S[..][1] = sums of the remainders rounded
    // to the grid 2^-29, for the image the walk over the wave's pixels is in
    [[maybe_unused]] f32x4 poolS[NT][2];
    [[maybe_unused]] int pool_slot = 0, pool_bound = 0;  // that image's slot; first pixel of the image after it
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
    static_for<MT>([&](auto SUB) {
      constexpr int i = decltype(SUB)::value;
      const int m = mw0 + 16 * i;
      if constexpr (POOL) {
        // An EXACT sum, so that an image's features do not depend on where in the batch (which lanes, which tile) it sits:
        // v = hi + lo + r with hi on the grid 2^-10 (v + C - C rounds to the grid of C's ulp), lo = (v - hi) on the grid
        // 2^-29, |r| <= 2^-30.  Sums of such terms are exact in fp32 while sum(hi) < 2^14 (|lo| <= 2^-11: 49 of them stay
        // below 2^-5 = 2^24 grid steps), whatever the order -- per-lane sums, the DPP tree, head_pool_kernel.
        f32x4 hi[NT], lo[NT];
        const bool live = m < M;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
          if constexpr (EPI_RESID) {
            const typename E::vec4 rr = __builtin_bit_cast(typename E::vec4, rq32[i][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = live ? fmaxf(v[e], 0.f) : 0.f;
            hi[j][e] = (v[e] + 12288.0f) - 12288.0f;                       // C = 1.5 x 2^13: ulp 2^-10
            lo[j][e] = ((v[e] - hi[j][e]) + 0.0234375f) - 0.0234375f;     // C = 1.5 x 2^-6: ulp 2^-29
          }
        }
        if (m0 + wm * WPX + 16 * i + 15 < pool_bound) {  // (uniform) the whole sub-tile lies in the current image
#pragma unroll
          for (int j = 0; j < NT; ++j) poolS[j][0] += hi[j], poolS[j][1] += lo[j];
        } else {  // an image ends inside this sub-tile (or right in front of it): finish it, start the next one
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
          // the image the walk ends in (slot <= kPoolSlots - 1); if the wave's last pixel closed an image, an all-zero set
          // for a slot nobody reads -- unless it would lie outside the buffer
          if (pool_slot < kPoolSlots) pool_flush(pool_slot);
        }
      } else if constexpr (OUTF32) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
          if constexpr (EPI_RESID) {
            const typename E::vec4 rr = __builtin_bit_cast(typename E::vec4, rq32[i][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
          }
          if constexpr (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (m < M)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(outp) + (size_t)m * COUT + n0 + wn * WTN + 16 * j + 4 * g) =
                make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
          unsigned R[4] = {0u, 0u, 0u, 0u};
          if constexpr (EPI_RESID) {
            R[0] = rq[i][jp][0], R[1] = rq[i][jp][1], R[2] = rq[i][jp][2], R[3] = rq[i][jp][3];
            permlane16_swap(R[0], R[2]);  // -> (R[0], R[1]) = this lane's 4 channels of tile 2 jp, (R[2], R[3]) = of tile 2 jp + 1
            permlane16_swap(R[1], R[3]);
          }
          unsigned P[2][2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jp + jj;
            float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
            if constexpr (EPI_RESID) {
              const typename E::vec4 rr = __builtin_bit_cast(typename E::vec4, u32x2{R[2 * jj], R[2 * jj + 1]});
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
            }
            if constexpr (RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            P[jj][0] = PackPair<T>::pack_rn(v[0], v[1]);
            P[jj][1] = PackPair<T>::pack_rn(v[2], v[3]);
          }
          permlane16_swap(P[0][0], P[1][0]);
          permlane16_swap(P[0][1], P[1][1]);
          if constexpr (STAGE16) {
            // through the wave's 4 KB of ring slot 1 (slot 0 is receiving the next tile's first weight tile): [32 px][128 B],
            // 16-byte position c ^ (px & 7) -- the eight lanes a ds_write_b128 serves together are eight pixels of different
            // px & 7 (perm16), i.e. eight different positions = all 32 banks
            const int spx = (i & 1) * 16 + pn;
            const int sch = 4 * jp + 2 * (g & 1) + (g >> 1);  // the 16-byte chunk of the wave's 128 bytes this lane holds
            *reinterpret_cast<u32x4*>(Sw16 + spx * 128 + ((sch ^ (spx & 7)) << 4)) = u32x4{P[0][0], P[0][1], P[1][0], P[1][1]};
          } else if ((HIPAC_H16_ABL & 32) ? m < 0 : m < M)  // (ablation 32: no stores)
            store16_out<HIPAC_NT_STORES && (H * W >= 784)>(reinterpret_cast<T*>(outp) + (size_t)((HIPAC_H16_ABL & 128) ? (m & 0x3fff) : m) * COUT + c_lane + 32 * jp,
                                                           u32x4{P[0][0], P[0][1], P[1][0], P[1][1]});  // (ablation 128: every store into one L2-resident window)
        }
        if constexpr (STAGE16 && (i & 1)) {
          // the 32-pixel group is staged: out as whole 128-byte lines -- 8 lanes per pixel, 16 bytes each
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int spx = (lane >> 3) + 8 * k, sch = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(Sw16 + spx * 128 + ((sch ^ (spx & 7)) << 4));
            const int ms = m0 + wm * WPX + (i >> 1) * 32 + spx;
            if ((HIPAC_H16_ABL & 32) ? ms < 0 : ms < M)
              *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(outp) + (size_t)ms * COUT + n0 + wn * WTN + sch * 8) = v;
          }
        }
      }
    });
#ifdef HIPAC_HALO_STAMPS
    HALO_STAMP(t_end);
    if (tid == 0) {
      atomicAdd(&g_halo_stamps[0], t_first - t_start);
      atomicAdd(&g_halo_stamps[1], t_loop - t_first);
      atomicAdd(&g_halo_stamps[2], t_end - t_loop);
      atomicAdd(&g_halo_stamps[3], 1ull);
      atomicAdd(&g_halo_stamps[4], t_bar - t_loop);
      atomicAdd(&g_halo_stamps[5], t_pref - t_bar);
      atomicAdd(&g_halo_stamps[6], t_end - t_pref);
      atomicAdd(&g_halo_stamps[7], t_setup - t_start);
    }
#endif
#if HIPAC_H16_EPI_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    prev_full = !POOL && (m0 + BM <= M);  // (POOL: the number of stores depends on the images the tile meets)
    continue;  // next tile
  }
  // the residual of the first 32-pixel group is requested here, not from inside the last K step as in the 32x32 kernel: its
  // 16 registers do not fit beside the K loop's, and the latency has the barrier, the band prefetch and the staging round
  // trip of group 0 to hide behind
  load_resid(std::integral_constant<int, 0>{});
  const float4 b_lo = *reinterpret_cast<const float4*>(bias + e_c0);
  const float4 b_hi = *reinterpret_cast<const float4*>(bias + e_c0 + 4);
  __builtin_amdgcn_s_barrier();  // every wave has left the K loop: band and ring are free
  HALO_STAMP(t_bar);
  {
    // prefetch the next tile's first band chunk; it lands behind this epilogue
    const int vn = vb + gridDim.x;
#if HIPAC_H16_XCD_CHUNKS
    const int mtn = ((vn >> 3) / NTILES_N) < ((n_mtiles + 7) >> 3) ? (vn & 7) * ((n_mtiles + 7) >> 3) + (vn >> 3) / NTILES_N : n_mtiles;
#else
    const int mtn = ((vn >> 3) / NTILES_N) * 8 + (vn & 7);
#endif
    if (mtn < n_mtiles) issue_band_of(mtn * BM, 0);
  }
  HALO_STAMP(t_pref);
  unsigned char* const Sl = Wbuf + wave * (32 * SROWW);  // this wave's private staging
  static_for<G32>([&](auto SUB) {
    constexpr int i = decltype(SUB)::value;
    if constexpr (i + 1 < G32) load_resid(std::integral_constant<int, i + 1>{});
    // accumulators -> fp32 rows: sub-tiles 2i (staging pixels 0-15) and 2i+1 (16-31); a lane holds channels 4g .. 4g+3 of
    // every 16-wide tile of pixel pn (LDS operations of one wave complete in order: no barrier)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        *reinterpret_cast<f32x4*>(Sl + (hh * 16 + pn) * SROWW + (j * 16 + 4 * g) * 4) = acc[2 * i + hh][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      const int px = e_px + k * (64 / CPW);
      const int m = m0 + wm * WPX + i * 32 + px;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(Sl + px * SROWW + (lane % CPW) * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(Sl + px * SROWW + (lane % CPW) * 32 + 16);
      if (m < M) {
        float v[8] = {lo[0] + b_lo.x, lo[1] + b_lo.y, lo[2] + b_lo.z, lo[3] + b_lo.w,
                      hi[0] + b_hi.x, hi[1] + b_hi.y, hi[2] + b_hi.z, hi[3] + b_hi.w};
        const size_t o = (size_t)m * COUT + e_c0;
        if constexpr (RESID) {
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
        } else {
          frag ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (T)v[e];
          *reinterpret_cast<frag*>(reinterpret_cast<T*>(outp) + o) = ov;
        }
      }
    }
    // the next group overwrites the staging rows: this wave's reads above must have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
#ifdef HIPAC_HALO_STAMPS
  HALO_STAMP(t_end);
  if (tid == 0) {
    atomicAdd(&g_halo_stamps[0], t_first - t_start);  // prologue: band + first weight tile in flight
    atomicAdd(&g_halo_stamps[1], t_loop - t_first);   // K loop
    atomicAdd(&g_halo_stamps[2], t_end - t_loop);     // epilogue (all of it)
    atomicAdd(&g_halo_stamps[3], 1ull);
    atomicAdd(&g_halo_stamps[4], t_bar - t_loop);     // ... of which: waiting for the other waves to leave the K loop
    atomicAdd(&g_halo_stamps[5], t_pref - t_bar);     // ... issuing the next tile's band
    atomicAdd(&g_halo_stamps[6], t_end - t_pref);     // ... the staged groups
  }
#endif
  }  // persistent tile loop
}

}  // namespace hipac
