// layer1 BasicBlock (56x56, 64 -> 64 -> 64 channels) in ONE kernel (gfx950), included by conv_igemm.h:
//     out = relu(conv2(relu(conv1(x) + b1)) + b2 + x)
// The two 3x3 convolutions of a layer1 block are HBM-bound when launched separately (each streams a
// 200 MB activation in and out per 512 patches, conv2 a second one for the shortcut: 4.5 - 5 TB/s measured);
// here the intermediate map never leaves the CU, so a block costs one read of x and one write of out.
//
//   * ONE 8-wave workgroup per CU walks DOWN a strip of 14 output columns (4 strips per image): x rows
//     arrive by LDS-DMA into a ring of rows (18 columns = 14 + a halo of 2 on each side), team A (waves 0-3)
//     turns them into rows of the intermediate map I = relu(conv1(x) + b1) (16 columns = 14 + 1 + 1), rounded to
//     T exactly as the unfused path stores it, in a second ring; team B (waves 4-7) follows ten rows behind
//     and turns I rows into output rows.  Nothing is computed twice in y; in x conv1 computes 16 columns for
//     14 and conv2 uses 14 of its 16 lanes: 1.14x the MFMAs of the unfused pair.
//   * every lane keeps ITS weight fragments of ITS convolution in registers (36 fragments = 144 VGPRs: wave =
//     (team, channel half, sub-tile half)), so LDS serves only activation fragments, one ds_read_b128 per MFMA;
//     the two waves of a SIMD are one A wave and one B wave.
//   * a step = 8 rows: A computes I rows 8m .. 8m+7 of its strip (4 sub-tiles of 2 rows x 16 columns, two per
//     wave) while B computes output rows 8m-10 .. 8m-3 (at m = 1 the three sub-tiles that exist, at the step
//     after the strip's last one the five that remain -- A is then already in the next strip).  ONE workgroup
//     barrier per step.
//   * rings: row L = y + 1 of a strip (L = 0 and 57 are the zero rows above and below the image) lives at ring
//     row L mod 30, and L = 30, 31 ALSO at rows 30, 31, so the 4 rows a sub-tile's window reads (start row even)
//     are always contiguous: window base + immediate offsets, no wrap inside the loop.  Every strip starts at
//     ring row 0 again; with 30 rows neither A's writes nor the DMA of the rows two steps ahead (or of the next
//     strip's first rows) ever meet a row that B or A still reads (DESIGN.md has the table).
//   * bank conflicts: 16-byte chunk c of the pixel in column j sits at chunk c ^ ((j >> 1) & 7); a ds_read_b128
//     lane group is 16 consecutive columns of one row = 16 distinct 16-byte bank groups.  The DMA applies the
//     swizzle on the source side (LDS-DMA writes lane-linear).  x pixels outside the image are zero-filled by
//     the buffer range check; I columns outside the image are written as zeros.
//   * shortcut: the x pixels a B wave adds are fetched by LDS-DMA (L2 hits: the rows went through the x ring one
//     step earlier) into per-wave staging at the start of the step.
//   * output: the accumulator layout gives a lane 4 consecutive channels per register group; one
//     v_permlane32_swap per packed dword pairs the lane halves so that every lane stores 16 contiguous bytes.
// The MFMA order (tap-major, k16 within a tap), the bias / shortcut association and every rounding are those of
// conv3x3_c64_kernel, so the fused block is bit-identical to the unfused pair (tests/test_gpu_resnet.py).
#pragma once

namespace hipac {

#ifndef HIPAC_BLK_PF
#define HIPAC_BLK_PF 2  // LDS fragment reads run this many k16 steps ahead of their MFMAs
#endif

#ifndef HIPAC_BLK_ABL
#define HIPAC_BLK_ABL 0  // developer builds (wrong results): 1 no B epilogue, 2 no A epilogue, 4 no x DMA, 8 no shortcut, 16 no conv MFMAs, 32 no stores
#endif
#ifndef HIPAC_BLK_PRIO
#define HIPAC_BLK_PRIO 1  // s_setprio around the MFMA loops
#endif

// lb[i][kw]: LDS byte ADDRESS (not an offset into a buffer: the buffer base is folded in, so that the only VALU
// per fragment read is the xor that selects the k16 chunk) of sub-tile i's window, tap column kw, this lane
template <typename T, int NSUB, int PITCH>
__device__ __forceinline__ void c64_strip_mfma(const typename Elem<T>::frag (&wreg)[9][4], const int (&lb)[NSUB][3],
                                               const float* __restrict__ bl, f32x16 (&acc)[NSUB]) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int PF = HIPAC_BLK_PF;
  frag ring[PF + 1][NSUB];
  // the bias is the initial accumulator (register group q = channels 8q + 4h .. +3 of the wave's 32): LDS -> acc
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 8 * q);
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = b[e];
  }
  auto rd_step = [&](auto S) {
    constexpr int st = decltype(S)::value;
    constexpr int kh = c64_step_kh(st), kw = c64_step_kw(st), kk = c64_step_kk(st);
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
      ring[st % (PF + 1)][i] = *(const __attribute__((address_space(3))) frag*)(size_t)(unsigned)((lb[i][kw] ^ (kk << 5)) + kh * PITCH);
  };
#if HIPAC_BLK_PRIO
  __builtin_amdgcn_s_setprio(1);
#endif
#if HIPAC_BLK_ABL & 16
  return;
#endif
  static_for<PF>([&](auto S) { rd_step(S); });
  static_for<36>([&](auto S) {
    constexpr int st = decltype(S)::value;
    if constexpr (st + PF < 36) rd_step(std::integral_constant<int, st + PF>{});
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
      acc[i] = E::mfma(wreg[3 * c64_step_kh(st) + c64_step_kw(st)][c64_step_kk(st)], ring[st % (PF + 1)][i], acc[i]);
    __builtin_amdgcn_sched_barrier(0);  // pin the read-ahead
  });
#if HIPAC_BLK_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

// Fragment-sharing form of the loop above (-DHIPAC_BLK_SHARE=1 -DHIPAC_C64_ORDER=1; OFF by default: measured slower, see the
// end of this comment): LDS serves 40 fragment reads per 72 MFMAs instead of 72.
//   * rows: the window rows of a wave's sub-tiles overlap -- sub-tile i, tap row kh reads row pair R = y0 + 2i + kh -- so
//     the loop walks the DISTINCT row pairs rho = 0 .. 2 NSUB and feeds every (sub-tile, kh) that reads that pair
//     (rho = 2: sub-tile 0's kh = 2 and sub-tile 1's kh = 0);
//   * columns: a 16-lane DPP row is 16 consecutive window columns of one (row, k-half), so of the three tap columns
//     only two are read -- A = columns lx, B = columns lx + 2 -- and kw = 1 (columns lx + 1) is A shifted down one lane
//     (row_shl:1; lane 15 has no source and keeps its old value) overwritten by B shifted up one lane (row_shr:1; lane 0
//     has no source and keeps A's column 1): 8 v_mov_dpp instead of a 1 KB LDS read.
// Every accumulator still sees its taps in the order kh = 0, 1, 2; within a tap row the order is (k16 step, kw) instead
// of (kw, k16 step) -- conv3x3_c64_kernel uses the same order (HIPAC_C64_ORDER), so fused == unfused stays bit for bit.
// Measured on one box (tools/l1bench.py, ns per patch and block): 514 as above (72 reads), 527 with rows + columns shared (40
// reads, but three dependent MFMAs in a row on one accumulator where the plain loop alternates two), 540 with the columns
// shared only (48 reads, accumulators alternating): the v_mov_dpp feeding an MFMA operand cost more than the LDS reads they
// replace -- layer1 is not bound by its fragment reads.
#ifndef HIPAC_BLK_SHARE
#define HIPAC_BLK_SHARE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
template <typename F>
__device__ __forceinline__ F frag_mid_column(const F& a, const F& b) {
  u32x4 A = __builtin_bit_cast(u32x4, a), B = __builtin_bit_cast(u32x4, b), S;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = __builtin_amdgcn_update_dpp((int)A[e], (int)A[e], 0x101 /* row_shl:1 */, 0xf, 0xf, false);
    S[e] = (unsigned)__builtin_amdgcn_update_dpp(t, (int)B[e], 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  }
  return __builtin_bit_cast(F, S);
}
#else
template <typename F>
__device__ inline F frag_mid_column(const F& a, const F&) { return a; }
#endif

template <typename T, int NSUB, int PITCH>
__device__ __forceinline__ void c64_strip_mfma_share(const typename Elem<T>::frag (&wreg)[9][4], const int (&lb)[NSUB][3],
                                                     const float* __restrict__ bl, f32x16 (&acc)[NSUB]) {
  using E = Elem<T>;
  using frag = typename E::frag;
  constexpr int PF = HIPAC_BLK_PF;
  constexpr int NRHO = 2 * NSUB + 1, NST = 4 * NRHO;  // distinct row pairs; steps = (row pair, k16 step)
  frag ra[PF + 1], rb[PF + 1];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 8 * q);
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][4 * q + e] = b[e];
  }
  auto rd_step = [&](auto S) {
    constexpr int st = decltype(S)::value;
    constexpr int rho = st / 4, kk = st % 4;
    // row pair rho through the window of the sub-tile that owns it (each window is contiguous in the ring on its own)
    constexpr int i = rho <= 2 ? 0 : (rho - 1) / 2, row = rho - 2 * i;
    ra[st % (PF + 1)] = *(const __attribute__((address_space(3))) frag*)(size_t)(unsigned)((lb[i][0] ^ (kk << 5)) + row * PITCH);
    rb[st % (PF + 1)] = *(const __attribute__((address_space(3))) frag*)(size_t)(unsigned)((lb[i][2] ^ (kk << 5)) + row * PITCH);
  };
#if HIPAC_BLK_PRIO
  __builtin_amdgcn_s_setprio(1);
#endif
#if HIPAC_BLK_ABL & 16
  return;
#endif
  static_for<PF>([&](auto S) { rd_step(S); });
  static_for<NST>([&](auto S) {
    constexpr int st = decltype(S)::value;
    constexpr int rho = st / 4, kk = st % 4;
    if constexpr (st + PF < NST) rd_step(std::integral_constant<int, st + PF>{});
    const frag fa = ra[st % (PF + 1)], fb = rb[st % (PF + 1)];
    const frag fm = frag_mid_column(fa, fb);
    static_for<NSUB>([&](auto I) {
      constexpr int i = decltype(I)::value, kh = rho - 2 * i;
      if constexpr (kh >= 0 && kh < 3) {
        acc[i] = E::mfma(wreg[3 * kh + 0][kk], fa, acc[i]);
        acc[i] = E::mfma(wreg[3 * kh + 1][kk], fm, acc[i]);
        acc[i] = E::mfma(wreg[3 * kh + 2][kk], fb, acc[i]);
      }
    });
    __builtin_amdgcn_sched_barrier(0);  // pin the read-ahead
  });
#if HIPAC_BLK_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

typedef __attribute__((ext_vector_type(2))) short s16x2;
// ReLU on a dword of two T (bf16 | fp16): the sign bit decides, so it is a packed signed-integer max with 0
// (== rounding the fp32 ReLU: rounding keeps the sign, -0 becomes +0)
__device__ __forceinline__ unsigned relu_pk(unsigned v) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0}));
}

#ifdef HIPAC_HALO_STAMPS
static __device__ unsigned long long g_blk_stamps[4];  // barrier cycles (A, B), wave-steps (A, B)
#endif
#if HIPAC_BLK_SHARE
static_assert(HIPAC_C64_ORDER == 1, "the fragment-sharing loop accumulates in (kh, k16 step, kw) order");
#define C64_STRIP_MFMA c64_strip_mfma_share
#else
#define C64_STRIP_MFMA c64_strip_mfma
#endif
constexpr int kBlkRing = 30;              // ring rows (rows 30, 31 of each buffer repeat L = 30, 31)
constexpr int kBlkXPitch = 18 * 128;      // x ring: 18 columns x 64 channels
constexpr int kBlkIPitch = 16 * 128;      // I ring: 16 columns
constexpr int kBlkXBytes = 32 * kBlkXPitch + 512;   // + 4 slots: the last 8-slot DMA piece of a 2-row group overshoots
constexpr int kBlkIBytes = 32 * kBlkIPitch + 256;   // + 2 slots: idle lanes (columns 14, 15) read 2 pixels past their row
constexpr int kBlkRsBytes = 2 * 6144 + 2 * 4096;    // shortcut staging: 3 sub-tiles for the mh = 0 B waves, 2 for mh = 1

template <typename T>
__global__ __launch_bounds__(512, 2) void block_c64_kernel(const T* __restrict__ in, const T* __restrict__ w1,
                                                           const float* __restrict__ b1, const T* __restrict__ w2,
                                                           const float* __restrict__ b2, T* __restrict__ out,
                                                           int n_img) {
  using E = Elem<T>;
  using frag = typename E::frag;
  using vec4 = typename E::vec4;
  constexpr int H = 56, W = 56, C = 64, RING = kBlkRing, XP = kBlkXPitch, IP = kBlkIPitch;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kBlkXBytes + kBlkIBytes + kBlkRsBytes + 512];
  unsigned char* const Xr = smem;
  unsigned char* const Ir = smem + kBlkXBytes;
  float* const Bl = reinterpret_cast<float*>(smem + kBlkXBytes + kBlkIBytes + kBlkRsBytes);  // b1[64], b2[64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3, nh = tw & 1, mh = tw >> 1;
  const int r = lane & 31, h = lane >> 5, ly = r >> 4, lx = r & 15;
  unsigned char* const Rw = smem + kBlkXBytes + kBlkIBytes + (mh == 0 ? nh * 6144 : 2 * 6144 + nh * 4096);

  if (tid < 64) Bl[tid] = b1[tid];
  else if (tid < 128) Bl[tid] = b2[tid - 64];

  // strips of this workgroup: image b lives on XCD b % 8 (workgroup ids congruent mod 8 share an XCD and its L2),
  // the 4 strips of an image go to 4 consecutive slots of that XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int n_x = n_img > xcd ? (n_img - xcd + 7) >> 3 : 0;
  const int nst = 4 * n_x;
  const int ns = nst > slot ? (nst - slot + nslots - 1) / nslots : 0;
  if (ns == 0) return;  // the whole workgroup

  // weights of this wave's convolution, channels nh*32 + r, all 9 taps x 64 input channels, in registers
  frag wreg[9][4];
  {
    const char* wb = reinterpret_cast<const char*>(team ? w2 : w1) + (size_t)(nh * 32 + r) * (9 * C * 2) + 16 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wreg[tap][kk] = *reinterpret_cast<const frag*>(wb + tap * 128 + kk * 32);
  }

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const rsrc_t x_rsrc = make_rsrc(in, n_img * (H * W * C * 2));

  // ---- x ring DMA (team A): one piece = 8 ring slots (1 KB, lane-linear), piece p of a row group covers its slots
  // 8p .. 8p+7 (slot = row * 18 + column).  Everything about a lane's 16 bytes that does not depend on the strip
  // is computed once: byte offset relative to the group's first pixel, with 4 flags in the low nibble
  // (column j < 2 / j >= 16 / group row 0 / group row >= 1) that the strip and the group turn into "outside the image".
  constexpr int NKP = 5;  // pieces per wave and 18-piece group: p = tw + 4 kp (waves 0, 1: 5, waves 2, 3: 4)
  int relo[NKP];
#pragma unroll
  for (int kp = 0; kp < NKP; ++kp) {
    const int q = 8 * (tw + 4 * kp) + (lane >> 3);
    const int rr = (q * 3641) >> 16;  // q / 18 for q < 160
    const int j = q - 18 * rr;
    const int c = (lane & 7) ^ ((j >> 1) & 7);
    relo[kp] = (((rr * W + j) << 7) + (c << 4)) | (j < 2 ? 1 : 0) | (j >= 16 ? 2 : 0) | (rr == 0 ? 4 : 0) | (rr >= 1 ? 8 : 0);
  }
  // rows L0 .. (np pieces) of image `img`, strip column x0, into ring rows rb ..
  auto x_rows = [&](int img, int x0, int L0, int rb, int np, int rowmask) {
    const int sb = ((img * H + L0 - 1) * W + x0 - 2) << 7;
    const int smask = (x0 == 0 ? 1 : 0) | (x0 == 42 ? 2 : 0) | rowmask;
    static_for<NKP>([&](auto KP) {
      constexpr int kp = decltype(KP)::value;
      const int p = tw + 4 * kp;
      if (p < np && !(HIPAC_BLK_ABL & 4)) {
        asm volatile("" : "+v"(relo[kp]));  // keep offset and flags in ONE register (no hoisted, split copies)
        const int off = (relo[kp] & smask) ? (int)0x80000000 : (relo[kp] & ~15) + sb;  // out of range: zeros
        buffer_load_lds16(x_rsrc, Xr + rb * XP + p * 1024, off, 0);
      }
    });
  };
  // group g = rows L = 8g .. 8g+7 (g = 7: L = 56, 57 only); ring rows 0, 8, 16, 24, 2, 10, 18, 26
  auto x_group = [&](int img, int x0, int g) {
    const int rb = 8 * g >= RING ? 8 * g - RING : 8 * g;
    x_rows(img, x0, 8 * g, rb, g == 7 ? 5 : 18, g == 0 ? 4 : (g == 7 ? 8 : 0));
    if (g == 3) x_rows(img, x0, RING, 0, 5, 0);  // L = 30, 31 landed in rows 30, 31 (window extension): home rows 0, 1 too
  };
  auto strip_img = [&](int k) { return (((slot + k * nslots) >> 2) << 3) + xcd; };
  auto strip_x0 = [&](int k) { return ((slot + k * nslots) & 3) * 14; };

  // per-lane address parts
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  int la[3];  // activation fragment of tap column kw: (row ly, column lx + kw) of the window, swizzled chunk
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int j = lx + kw;
    la[kw] = ((ly * (team ? 16 : 18) + j) << 7) + (((((j >> 1) & 7)) ^ h) << 4) + (team ? kBlkXBytes : 0) + lds_base;
  }
  const int iw_lane = (lx << 7) + (h << 3);          // team A: this lane's 8 bytes inside its I pixel ...
  const int iw_pos = (nh << 2) ^ ((lx >> 1) & 7);    // ... chunk position of channel group q: iw_pos ^ q
  const float* const bl = Bl + team * 64 + nh * 32 + 4 * h;  // this lane's bias values: bl[8q + e]

#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(wreg[tap][kk]));  // weight loads retire before the loop
  if (team == 0) {
    x_group(strip_img(0), strip_x0(0), 0);
    x_group(strip_img(0), strip_x0(0), 1);
    wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int n_steps = 7 * ns;
#ifdef HIPAC_HALO_STAMPS
  unsigned long long z_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, z_bar = 0;
#endif
  // team B: the epilogue of a step's LAST MFMA group is deferred to the top of the next step, so that after a
  // barrier one wave of every SIMD (A) starts in its MFMA loop and the other (B) in its vector work
  f32x16 acc[2];
  int pend_n = 0, pend_y = 0;  // B: deferred sub-tiles (0, 1 or 2) and their first row
  // shortcut pixels of a sub-tile by LDS-DMA (L2 hits: the rows went through the x ring a step earlier) into the wave's
  // staging slot [channel group q = 0..3][pixel r][16 B]: piece kq holds q = 2kq (lanes 0-31) and 2kq+1 (lanes 32-63), each
  // lane fetches the 16 bytes whose halves the two lanes (r, 0), (r, 1) add in the epilogue (8-byte reads, lane-linear)
  int rs_off[2];  // this lane's source offset in piece kq, relative to the sub-tile's (row y0, column x0)
#pragma unroll
  for (int kq = 0; kq < 2; ++kq) rs_off[kq] = ((ly * W + (lx < 14 ? lx : 13)) * C + nh * 32 + 8 * (2 * kq + h)) * 2;
  auto fetch_resid = [&](int pix0, int y0, int sl) {
    const int sb = (pix0 + y0 * W) << 7;
    if (HIPAC_BLK_ABL & 8) return;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) buffer_load_lds16(x_rsrc, Rw + sl * 2048 + kq * 1024, rs_off[kq], sb);
  };
  const int out_lane = ((ly * W + lx) * C + nh * 32 + 8 * h) * 2;  // byte offset of this lane's 16 bytes (+ 32 for qp = 1)
  // conv2 epilogue: + shortcut (fp32), round to T, ReLU, pair the lane halves, 16-byte stores
  auto epilogue_b = [&](const f32x16& a, int pix0, int y0, int sl) {
    unsigned P[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vec4 rv = {};
      if (!(HIPAC_BLK_ABL & 8)) rv = *reinterpret_cast<const vec4*>(Rw + sl * 2048 + q * 512 + r * 16 + h * 8);
      P[q][0] = relu_pk(PackPair<T>::pack_rn(a[4 * q + 0] + (float)rv[0], a[4 * q + 1] + (float)rv[1]));
      P[q][1] = relu_pk(PackPair<T>::pack_rn(a[4 * q + 2] + (float)rv[2], a[4 * q + 3] + (float)rv[3]));
    }
    char* const dst = reinterpret_cast<char*>(out) + ((size_t)(unsigned)(pix0 + y0 * W) << 7) + out_lane;
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      // lane halves: (q = 2qp, q = 2qp+1) x (h = 0, 1) -> h = 0 keeps channels 16qp .. +7, h = 1 channels 16qp+8 .. +15
      permlane32_swap(P[2 * qp][0], P[2 * qp + 1][0]);
      permlane32_swap(P[2 * qp][1], P[2 * qp + 1][1]);
      const u32x4 o = {P[2 * qp][0], P[2 * qp][1], P[2 * qp + 1][0], P[2 * qp + 1][1]};
#if HIPAC_BLK_ABL & 32
      asm volatile("" ::"v"(o), "v"(dst));
#else
      if (lx < 14) *reinterpret_cast<u32x4*>(dst + 32 * qp) = o;
#endif
    }
  };
  int pend_pix = 0, pend_sl = 0;
  auto flush_b = [&]() {  // the deferred epilogue(s); their shortcut pixels were requested a step ago
    if (pend_n) {
      wait_vmcnt<0>();
#if HIPAC_BLK_ABL & 1
      asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#else
      epilogue_b(acc[0], pend_pix, pend_y, pend_sl);
      if (pend_n == 2) epilogue_b(acc[1], pend_pix, pend_y + 2, pend_sl + 1);
#endif
      pend_n = 0;
    }
  };

  // one loop per team (same number of barriers in both): what a team keeps across steps is live in its loop only
  auto step_barrier = [&]() {
    HALO_STAMP(s_t0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // step boundary: I rows written <-> read, x rows landed <-> read, for both teams
#ifdef HIPAC_HALO_STAMPS
    HALO_STAMP(s_t1);
    z_bar += s_t1 - s_t0;
#endif
  };
  if (team == 0) {
    for (int G = 0; G <= n_steps; ++G) {
      if (G < n_steps) {
        // ------------------------------ team A: conv1, I rows 8m .. 8m+7 ------------------------------
        const int k = G / 7, m = G - 7 * k;
        const int img = strip_img(k), x0 = strip_x0(k);
        HALO_STAMP(a_t0);
        if (m <= 4) x_group(img, x0, m + 2);
        else if (m == 5) {
          x_group(img, x0, 7);
          if (k + 1 < ns) x_group(strip_img(k + 1), strip_x0(k + 1), 0);
        } else if (k + 1 < ns) x_group(strip_img(k + 1), strip_x0(k + 1), 1);
        if (m == 0 || m == 6)  // the zero row above (L = 0) / below (L = 57) the image
          *reinterpret_cast<u32x2*>(Ir + (m == 0 ? 0 : 27) * IP + tw * 512 + lane * 8) = u32x2{0u, 0u};
        const int y0a = 8 * m + 4 * mh;
        int lb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int y0 = y0a + 2 * i;
          const int wbase = (y0 >= RING ? y0 - RING : y0) * XP;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) lb[i][kw] = la[kw] + wbase;
        }
        HALO_STAMP(a_t1);
        C64_STRIP_MFMA<T, 2, XP>(wreg, lb, bl, acc);
        HALO_STAMP(a_t2);
#if HIPAC_BLK_ABL & 2
        asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#else
        // epilogue: ReLU, round to T, into the I ring (columns outside the image as zeros)
        const bool col_ok = !((x0 == 0 && lx == 0) || (x0 == 42 && lx == 15));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int Lr = y0a + 2 * i + 1 + ly;
          const bool twice = Lr == RING || Lr == RING + 1;
          unsigned char* const dst = Ir + (Lr >= RING ? Lr - RING : Lr) * IP + iw_lane;
          unsigned char* const dst2 = Ir + Lr * IP + iw_lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u32x2 pk;
            pk[0] = relu_pk(PackPair<T>::pack_rn(acc[i][4 * q + 0], acc[i][4 * q + 1]));
            pk[1] = relu_pk(PackPair<T>::pack_rn(acc[i][4 * q + 2], acc[i][4 * q + 3]));
            if (!col_ok) pk = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(dst + ((iw_pos ^ q) << 4)) = pk;
            if (twice) *reinterpret_cast<u32x2*>(dst2 + ((iw_pos ^ q) << 4)) = pk;
          }
        }
#endif
#ifdef HIPAC_HALO_STAMPS
        HALO_STAMP(a_t3);
        wait_vmcnt<0>();
        HALO_STAMP(a_t4);
        z_sum[0] += a_t1 - a_t0, z_sum[1] += a_t2 - a_t1, z_sum[2] += a_t3 - a_t2, z_sum[3] += a_t4 - a_t3;
#endif
      }
      wait_vmcnt<0>();  // the rows requested at the top of this step have landed
      step_barrier();
    }
  } else {
    for (int G = 0; G <= n_steps; ++G) {
      // ------------------------------ team B: conv2 + shortcut, ten rows behind ------------------------------
      HALO_STAMP(b_t0);
      flush_b();
      HALO_STAMP(b_t1);
      if (G >= 1) {
        const int k = (G - 1) / 7, mp = G - 7 * k;  // 1 .. 7
        // sub-tiles (2 output rows each) of this wave: a pair and / or a single one
        int yp = 8 * mp - 10 + 4 * mh, ys = -1;
        bool has_pair = true;
        if (mp == 1) {
          if (mh == 0) yp = 0;
          else has_pair = false, ys = 4;
        } else if (mp == 7) {
          if (mh == 0) yp = 46, ys = 50;
          else yp = 52;
        }
        const int pix0 = strip_img(k) * (H * W) + strip_x0(k);  // pixel index of (row 0, column x0)
        if (has_pair) {
          fetch_resid(pix0, yp, 0);
          fetch_resid(pix0, yp + 2, 1);
        }
        if (ys >= 0) fetch_resid(pix0, ys, has_pair ? 2 : 0);
        pend_pix = pix0;
        if (has_pair) {
          int lb[2][3];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int y0 = yp + 2 * i;
            const int wbase = (y0 >= RING ? y0 - RING : y0) * IP;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) lb[i][kw] = la[kw] + wbase;
          }
          HALO_STAMP(b_t2);
          C64_STRIP_MFMA<T, 2, IP>(wreg, lb, bl, acc);
#ifdef HIPAC_HALO_STAMPS
          HALO_STAMP(b_t3);
          z_sum[5] += b_t3 - b_t2, z_sum[7] += b_t2 - b_t1;
#endif
          pend_n = 2, pend_y = yp, pend_sl = 0;
        }
        if (ys >= 0) {
          flush_b();  // a pair of this very step (only the last step of a strip, mh = 0): its epilogue is not deferred
          int lb[1][3];
          const int wbase = (ys >= RING ? ys - RING : ys) * IP;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) lb[0][kw] = la[kw] + wbase;
          f32x16(&acc1)[1] = reinterpret_cast<f32x16(&)[1]>(acc[0]);
          C64_STRIP_MFMA<T, 1, IP>(wreg, lb, bl, acc1);
          pend_n = 1, pend_y = ys, pend_sl = has_pair ? 2 : 0;
        }
      }
#ifdef HIPAC_HALO_STAMPS
      z_sum[4] += b_t1 - b_t0;
#endif
      step_barrier();
    }
    flush_b();
  }
#ifdef HIPAC_HALO_STAMPS
  // developer build: A waves report (DMA issue, MFMA loop, epilogue, DMA wait), B waves (deferred epilogue incl. the
  // wait for its shortcut pixels, MFMA pair loop, -, step setup)
  if (lane == 0) {
    if (team == 0) {
      for (int i = 0; i < 4; ++i) atomicAdd(&g_halo_stamps[i], z_sum[i]);
    } else {
      for (int i = 4; i < 8; ++i) atomicAdd(&g_halo_stamps[i], z_sum[i]);
    }
    atomicAdd(&g_blk_stamps[team], z_bar);
    atomicAdd(&g_blk_stamps[2 + team], (unsigned long long)(n_steps + 1));
  }
#endif
}

}  // namespace hipac
