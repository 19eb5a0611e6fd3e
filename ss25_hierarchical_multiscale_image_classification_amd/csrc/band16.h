// conv3x3_band16_kernel: 3x3 convolutions of layers 2-4, stride 1 AND stride 2, as shifted reads of pixel BANDS held in
// LDS, on v_mfma_f32_16x16x32 (round 4).  Included by conv_igemm.h after halo16.h, whose lane maps, written-out MFMA stream,
// direct epilogue and residual-by-MFMA step it shares; what is new is the band itself:
//
//   * HALF-CHUNK bands.  halo16 holds the band of one 64-channel chunk (128-byte rows, 35-40 KB) in ONE buffer: the next
//     chunk's band can only be requested once the last tap has been read, so its DMA round trip is exposed once per nine
//     steps (stamped: 8 % of a layer4 tile).  Here a band holds 32 channels (64-byte rows, 17-20 KB) and there are TWO
//     buffers: the K loop walks (half-chunk, tap) and the band of the next segment lands while the current one is read.
//     A K step is therefore a HALF-step: 32 channels of one tap = 8 activation + 4 weight fragments, 32 MFMAs per wave,
//     its weights one 8 KB ring slot ([128 rows][64 B]); the ring has four slots, filled three half-steps ahead.
//   * PLANES.  A band is a run of consecutive pixels of a PLANE.  Stride 1: the plane is the input map and the nine taps
//     are shifts (kh - 1) W + (kw - 1).  Stride 2: output (oy, ox) reads input (2 oy + kh - 1, 2 ox + kw - 1), i.e. pixel
//     (oy + dy, ox + dx) of the parity plane (py, px) = (kh != 1, kw != 1) with dy = -(kh == 0), dx = -(kw == 0): four
//     planes with 4 / 2 / 2 / 1 taps, each a stride-1 problem on the OUTPUT grid.  The planes are never materialised: the
//     band DMA gathers pixel (2 r + py, 2 c + px) per slot (per-lane source offsets computed once per tile; every slot is a
//     full 64-byte run).  The folded 1x1 / stride 2 projection shortcut (PCIN) is plane (0, 0) of the block input with one
//     tap.  Bank conflicts: slot q at q * 64, 16-byte position c ^ ((q >> 2) & 3); the 16x16x32 lane groups mix two k-groups
//     of opposite pixel parity (perm16), whose slots differ in q & 1, and the four same-parity slots that share q & 3 differ
//     in (q >> 2) & 3: conflict-free for any chunk pair, as in halo16.
//
// Geometry per workgroup: 256 output pixels x 128 channels, 4 waves of 128 x 64 (as halo16).
#pragma once

namespace hipac {

#ifndef HIPAC_B16_ABL
#define HIPAC_B16_ABL 0  // developer builds (wrong results): 1 no band DMA inside the K loop, 2 no weight DMA inside the K loop, 4 no waits for either
#endif
#ifndef HIPAC_B16_NSLOT
#define HIPAC_B16_NSLOT 4  // weight ring: 8 KB slots
#endif

// (b16_tap / b16_kh / b16_kw: the tap order inside a half-chunk, defined in halo16.h, which shares it)
template <int STRIDE> __host__ __device__ constexpr bool b16_seg_start(int k) { return STRIDE == 1 ? k == 0 : (k == 0 || k == 4 || k == 6 || k == 8); }
template <int STRIDE> __host__ __device__ constexpr int b16_next_start(int k) {  // position of the next segment's first tap (9: next half-chunk)
  if (STRIDE == 1) return 9;
  return k < 4 ? 4 : (k < 6 ? 6 : (k < 8 ? 8 : 9));
}

template <typename T, int CIN, int COUT, int HO, int WO, int STRIDE, int BM, int BN, bool RELU, bool RESID, int PCIN = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_band16_kernel(const T* __restrict__ in, const T* __restrict__ wgt,
                                                                const float* __restrict__ bias, const T* __restrict__ resid,
                                                                T* __restrict__ outp, int M, int n_img, int n_mtiles,
                                                                const T* __restrict__ wgt_p = nullptr) {
  using E = Elem<T>;
  using frag = typename E::frag;
  static_assert(sizeof(T) == 2 && (STRIDE == 1 || STRIDE == 2) && BM == 256 && BN == 128, "tile shape");
  static_assert(CIN % 32 == 0 && PCIN % 32 == 0 && COUT % BN == 0 && (PCIN == 0 || (!RESID && STRIDE == 1)), "channels");
  constexpr int HI = HO * STRIDE, WI = WO * STRIDE;  // input map
  constexpr int NHC = CIN / 32, NPJ = PCIN / 32;     // half-chunks of the conv / of the folded projection
  constexpr int NH = 9 * NHC + NPJ;                  // half-steps per tile
  constexpr int KTOT = 9 * CIN;
  constexpr int WM = 2, WN = 2, WPX = BM / WM, MT = WPX / 16, WTN = BN / WN, NT = WTN / 16;
  constexpr int NTILES_N = COUT / BN;
  constexpr int NSLOT = HIPAC_B16_NSLOT;
  static_assert(NSLOT == 4, "the vmcnt constants below are written for a lead of three half-steps");
  // band: slots 0..3 = zeros, then the plane pixels [m0 - LEAD, m0 + BM + TRAIL)
  constexpr int LEAD = WO + 1, TRAIL = STRIDE == 1 ? WO + 1 : 0;
  constexpr int NSLOTS_B = 4 + LEAD + BM + TRAIL;
  constexpr int NPW = (NSLOTS_B + 63) / 64;          // 1 KB pieces (16 slots) per wave and band: every wave issues this many
  constexpr int BAND_BYTES = NPW * 4 * 1024;
  constexpr int RING_OFF = 2 * BAND_BYTES, RING_BYTES = NSLOT * 8192;
  static_assert(RING_OFF + RING_BYTES <= 80 * 1024, "LDS: two workgroups per CU");
  static_assert(RING_OFF >= BM * 128 && RING_BYTES >= BM * 128, "the residual tile (two 64-channel chunks) fits the two regions");
  constexpr bool GATHER = STRIDE == 2 || PCIN > 0;

  extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];
  using lptr_t = __attribute__((address_space(3))) void*;
  const unsigned lds0 = (unsigned)(unsigned long long)(lptr_t)ring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int n16 = lane & 15, g = lane >> 4;
  const int pn = n16 < 4 ? 2 * n16 : (n16 < 12 ? 2 * n16 - 7 : 2 * n16 - 16);  // perm16(n16)

  const rsrc_t a_rsrc = make_rsrc(in, n_img * (HI * WI * CIN * 2));
  const rsrc_t p_rsrc = make_rsrc(PCIN > 0 ? resid : in, n_img * (PCIN > 0 ? 4 * HO * WO * PCIN * 2 : HI * WI * CIN * 2));
  const rsrc_t w_rsrc = make_rsrc(wgt, COUT * KTOT * 2);
  const rsrc_t wp_rsrc = make_rsrc(PCIN > 0 ? wgt_p : wgt, COUT * (PCIN > 0 ? PCIN : KTOT) * 2);

  // ---- band DMA: piece = 16 slots x 64 B; lane -> slot 16 p + (lane >> 2), 16-byte position lane & 3, which holds source
  // chunk (lane & 3) ^ ((slot >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3) for every piece
  const int b_sl = lane >> 2;                                  // slot inside the piece
  const int b_ch = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;      // byte offset of the lane's 16 bytes inside the 64-byte run
  const int q0 = 4 + LEAD + wm * WPX + pn;                     // slot of this lane's pixel of sub-tile 0 (tap offset 0)
  const int rdw = (wn * WTN + pn) * 64 + ((g ^ ((pn >> 2) & 3)) << 4);  // weight fragment of channel tile 0 inside a ring slot
  const int w_rr = lane >> 2;                                  // row inside a 16-row weight piece
  const int w_ch = ((lane & 3) ^ ((w_rr >> 2) & 3)) * 16;
  const int w_srow = (w_rr & 1) ? (w_rr + 7) >> 1 : (w_rr < 8 ? w_rr >> 1 : (w_rr >> 1) + 8);  // perm16_inv(w_rr)

  frag ident[2];  // the residual step's identity fragments (halo16.h)
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int e = 0; e < 8; ++e) ident[o][e] = (RESID && g == 2 * o + (n16 >> 3) && e == (n16 & 7)) ? (T)1.0f : (T)0.0f;

  constexpr int N_EPI_STORES = MT * (NT / 2);
  bool prev_full = false;
  for (int vb = blockIdx.x, first_tile = 1;; vb += gridDim.x, first_tile = 0) {
    const int xcd = vb & 7, slotv = vb >> 3;
    const int mt = (slotv / NTILES_N) * 8 + xcd;
    const int nt = slotv % NTILES_N;
    if (mt >= n_mtiles) break;
    const int m0 = mt * BM, n0 = nt * BN;
    HALO_STAMP(t_start);
#ifdef HIPAC_HALO_STAMPS
    unsigned long long t_first = 0;
#endif

    // per-lane source offsets of the band pieces this wave issues (bytes, relative to the plane's first pixel and to
    // channel 0): piece p = wave + 4 k covers slots 16 p ..; slot q holds plane pixel u = m0 - LEAD + q - 4
    int b_off[NPW];
    auto band_offsets = [&](int m0_) {
#pragma unroll
      for (int k = 0; k < NPW; ++k) {
        const int q = 16 * (wave + 4 * k) + b_sl;
        const int u = m0_ - LEAD + q - 4;
        const bool ok = q >= 4 && u >= 0 && u < M;
        if constexpr (GATHER) {
          const int b = u / (HO * WO), rem = u - b * (HO * WO), r = rem / WO, c = rem - r * WO;
          // stride 2: input pixel (2 r, 2 c) (+ the plane's (py, px) as a scalar offset); a stride-1 conv gathers only for its
          // folded projection, whose plane lives in the block input [n][2 HO][2 WO][PCIN]
          const int pix = (b * (2 * HO) + 2 * r) * (2 * WO) + 2 * c;
          b_off[k] = ok ? pix : -1;  // (pixel index; multiplied by the row pitch where it is used)
        } else {
          b_off[k] = ok ? u : -1;
        }
      }
    };
    band_offsets(m0);
    // band of plane (py, px), channels [ch0, ch0 + 32) of `rs` (pitch CP elements per pixel) into buffer `buf`
    auto issue_band = [&](const rsrc_t rs, int CP, int plane_pix, int ch0, int buf, bool gather) {
#pragma unroll
      for (int k = 0; k < NPW; ++k) {
        int off;
        if (gather || !GATHER) off = b_off[k] < 0 ? (int)0x80000000 : ((b_off[k] + plane_pix) * CP + ch0) * 2 + b_ch;
        else off = 0;
        buffer_load_lds16(rs, ring + buf * BAND_BYTES + (wave + 4 * k) * 1024, off, 0);
      }
    };
    // a stride-1 conv with a folded projection needs BOTH kinds of offsets: contiguous for its own plane, gathered for the
    // projection's.  Its own are cheap to make on the fly:
    auto issue_band_s1 = [&](int m0_, int ch0, int buf) {
#pragma unroll
      for (int k = 0; k < NPW; ++k) {
        const int q = 16 * (wave + 4 * k) + b_sl;
        const int u = m0_ - LEAD + q - 4;
        const int off = (q >= 4 && u >= 0 && u < M) ? (u * CIN + ch0) * 2 + b_ch : (int)0x80000000;
        buffer_load_lds16(a_rsrc, ring + buf * BAND_BYTES + (wave + 4 * k) * 1024, off, 0);
      }
    };
    // the band of segment (half-chunk hc, first tap position k) -- or of projection half-chunk hc - NHC -- into buffer `buf`
    auto issue_segment = [&](int m0_, int hc, int k, int buf) {
      if (PCIN > 0 && hc >= NHC) {
        issue_band(p_rsrc, PCIN, 0, (hc - NHC) * 32, buf, true);
        asm volatile("" ::: "memory");
        return;
      }
      if constexpr (STRIDE == 1) {
        if constexpr (PCIN > 0) issue_band_s1(m0_, hc * 32, buf);
        else issue_band(a_rsrc, CIN, 0, hc * 32, buf, false);
      } else {
        const int py = k < 4 ? 1 : (k < 6 ? 0 : (k < 8 ? 1 : 0)), px = k < 4 ? 1 : (k < 6 ? 1 : 0);
        issue_band(a_rsrc, CIN, py * WI + px, hc * 32, buf, true);
      }
    };
    // weights of half-step (half-chunk hc, position k) into ring slot `sl`: 8 pieces of 16 rows, two per wave
    auto issue_w = [&](int hc, int k, int sl, int n0_) {
      if (PCIN > 0 && hc >= NHC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int p = wave + 4 * i;
          buffer_load_lds16(wp_rsrc, ring + RING_OFF + sl * 8192 + p * 1024, ((n0_ + 16 * p + w_srow) * PCIN + (hc - NHC) * 32) * 2 + w_ch, 0);
        }
        asm volatile("" ::: "memory");
        return;
      }
      const int tap = b16_tap<STRIDE>(k);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = wave + 4 * i;
        buffer_load_lds16(w_rsrc, ring + RING_OFF + sl * 8192 + p * 1024, ((n0_ + 16 * p + w_srow) * KTOT + tap * CIN + hc * 32) * 2 + w_ch, 0);
      }
    };
    auto tap_of_flat = [&](int h, int& hc, int& k) {  // flat half-step index -> (half-chunk, position); projection: (NHC + pj, 0)
      if (h < 9 * NHC) hc = h / 9, k = h - 9 * hc;
      else hc = NHC + (h - 9 * NHC), k = 0;
    };

    // image-edge flags per sub-tile, 4 bits each (halo16.h): bit 0 x == 0, 1 x == WO-1, 2 y == 0, 3 y == HO-1 (stride 2: bits 0, 2)
    unsigned epk = 0;
    {
      const int mw0 = m0 + wm * WPX + pn;
      const int rem = mw0 % (HO * WO);
      int y = rem / WO, x = rem - y * WO;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        epk |= (unsigned)((x == 0 ? 1 : 0) | (STRIDE == 1 && x == WO - 1 ? 2 : 0) | (y == 0 ? 4 : 0) | (STRIDE == 1 && y == HO - 1 ? 8 : 0)) << (4 * i);
        x += 16 % WO, y += 16 / WO;
        if (x >= WO) x -= WO, y += 1;
        if (y >= HO) y -= HO;
      }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (first_tile) {  // (later tiles: requested behind the previous tile's epilogue)
      issue_segment(m0, 0, 0, 0);
#pragma unroll
      for (int h = 0; h < NSLOT - 1; ++h) {
        int hc, k;
        tap_of_flat(h, hc, k);
        issue_w(hc, k, h, n0);
      }
    }

    int cur = 1;      // band buffer of the current segment (toggled at every segment start: the first one uses 0)
    int hflat = 0;    // flat half-step index
    // one half-step.  K = position inside the half-chunk (compile time), PROJ = a projection half-step.
    auto half_step = [&](auto KK, auto PROJ, int hc) {
      constexpr int k = decltype(KK)::value;
      constexpr bool proj = decltype(PROJ)::value;
      constexpr bool seg_start = proj || b16_seg_start<STRIDE>(k);
      // ---- wait: the weights of this half-step (requested three half-steps ago) and, at a segment start, its band
      // (requested at the previous segment's start).  Vector-memory operations retire in order: allow only what was
      // requested AFTER the youngest thing needed now -- the weights of the next two half-steps (2 pieces each), plus the
      // band requested in one of the last two half-steps where that band is not itself needed now.
      if (hflat == 0) {
        if (!first_tile && prev_full) wait_vmcnt<N_EPI_STORES>();  // the prefetch is older than the epilogue's stores
        else wait_vmcnt<0>();
      } else if (HIPAC_B16_ABL & 4) {  // (ablation: requests are not waited for)
      } else if (hflat >= NH - 2) {
        wait_vmcnt<0>();  // (the ring runs dry at the end of the tile: fewer requests are in flight than the constants assume)
      } else if constexpr (proj) {
        wait_vmcnt<2>();  // band requested one half-step ago (previous segment = 1 half-step): only W(h+2)... conservatively 2
      } else if constexpr (STRIDE == 1) {
        if constexpr (k == 1 || k == 2) wait_vmcnt<4 + NPW>();  // the next band was requested 1-2 half-steps ago
        else wait_vmcnt<4>();
      } else {
        // bands are requested at positions 0, 4, 6, 8 (for the segments starting at 4, 6, 8, next 0)
        if constexpr (k == 0) wait_vmcnt<2>();                 // band requested at 8, one half-step ago, and needed now
        else if constexpr (k == 1) wait_vmcnt<4 + 2 * NPW>();  // requests at 8 and 0
        else if constexpr (k == 2 || k == 5 || k == 7) wait_vmcnt<4 + NPW>();
        else wait_vmcnt<4>();                                  // 3; 4, 6, 8: their band is at least two half-steps old
      }
      __builtin_amdgcn_s_barrier();
#ifdef HIPAC_HALO_STAMPS
      if (hflat == 0) {
        t_first = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
      }
#endif
      if constexpr (seg_start) {
        cur ^= 1;
        // the band of the NEXT segment into the other buffer (every wave has left the segment that used it)
        int hc2 = hc, k2 = proj ? 9 : b16_next_start<STRIDE>(k);
        if (k2 == 9) hc2 = hc + 1, k2 = 0;
        if (HIPAC_B16_ABL & 1) {
        } else if (hc2 < NHC + NPJ) issue_segment(m0, hc2, k2, cur ^ 1);
        else {  // nothing follows in this tile: keep the request count uniform with out-of-range pieces (zero fill)
#pragma unroll
          for (int kk = 0; kk < NPW; ++kk) buffer_load_lds16(a_rsrc, ring + (cur ^ 1) * BAND_BYTES + (wave + 4 * kk) * 1024, (int)0x80000000, 0);
        }
      }
      // ---- addresses
      constexpr int kh = proj ? 1 : b16_kh<STRIDE>(k), kw = proj ? 1 : b16_kw<STRIDE>(k);
      constexpr int toff = STRIDE == 1 ? (kh - 1) * WO + (kw - 1) : (kh == 0 ? -WO : 0) + (kw == 0 ? -1 : 0);
      constexpr unsigned tapmask = proj ? 0u : ((kw == 0 ? 1u : 0u) | (STRIDE == 1 && kw == 2 ? 2u : 0u) | (kh == 0 ? 4u : 0u) |
                                                (STRIDE == 1 && kh == 2 ? 8u : 0u)) * 0x11111111u;
      const int qt = q0 + toff;
      const unsigned sw = (unsigned)((g ^ (qt >> 2)) & 3) << 4;
      const unsigned bufb = lds0 + (unsigned)(cur * BAND_BYTES);
      const unsigned a_in = bufb + ((unsigned)qt << 6) + sw;
      const unsigned a_zero = bufb + ((unsigned)(qt & 3) << 6) + sw;
      unsigned aa[MT];
      if constexpr (tapmask != 0) {
        unsigned em;
        asm("v_and_b32 %0, %1, %2" : "=v"(em) : "s"(tapmask), "v"(epk));
#pragma unroll
        for (int i = 0; i < MT; ++i) aa[i] = (em & (0xFu << (4 * i))) ? a_zero : a_in + 1024 * i;
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i) aa[i] = a_in + 1024 * i;
      }
      const unsigned wb = lds0 + (unsigned)(RING_OFF + (hflat % NSLOT) * 8192 + rdw);
      // ---- the stream: W[0..NT), A[0..3), then per sub-tile s: A[s + 3]; counted waits; accumulators in place
      frag wf[NT], af[4];
      static_for<NT>([&](auto J) { lds_read16<decltype(J)::value * 1024>(wf[decltype(J)::value], wb); });
      static_for<3>([&](auto S) { lds_read16<0>(af[decltype(S)::value], aa[decltype(S)::value]); });
      __builtin_amdgcn_s_setprio(1);
      static_for<MT>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr (s + 3 < MT) lds_read16<0>(af[(s + 3) & 3], aa[s + 3]);
        wait_lgkmcnt<(s + 3 < MT) ? 3 : (MT - 1 - s)>();
#pragma unroll
        for (int j = 0; j < NT; ++j) Asm16<T>::mfma(acc[s][j], wf[j], af[s & 3]);
        if constexpr (s == MT / 2) {
          // the weights three half-steps on, into the slot the previous half-step used (freed by this half-step's barrier)
          int hc3, k3;
          tap_of_flat(hflat + NSLOT - 1, hc3, k3);
          if (!(HIPAC_B16_ABL & 2) && hflat + NSLOT - 1 < NH) issue_w(hc3, k3, (hflat + NSLOT - 1) % NSLOT, n0);
        }
      });
      __builtin_amdgcn_s_setprio(0);
      ++hflat;
    };

    for (int hc = 0; hc < NHC; ++hc)
      static_for<9>([&](auto KK) { half_step(KK, std::false_type{}, hc); });
    if constexpr (NPJ > 0) {
      for (int pj = 0; pj < NPJ; ++pj) half_step(std::integral_constant<int, 0>{}, std::true_type{}, NHC + pj);
    }

    HALO_STAMP(t_loop);
    // ---- the residual, added on the matrix pipe (halo16.h): [256 px][128 ch] by LDS-DMA in 128-byte rows, channels of the
    // wn = 0 waves into the band buffers' region, those of the wn = 1 waves into the ring
    if constexpr (RESID) {
      __builtin_amdgcn_s_barrier();
      const rsrc_t r_rsrc = make_rsrc(resid, M * COUT * 2);
      {
        const int prow = lane >> 3, dchunk = lane & 7;
        const int r_lane = (prow * COUT + (dchunk ^ ((4 * wave + (prow >> 1)) & 7)) * 8) * 2;
        const int r_base = (m0 * COUT + n0) * 2;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int kk = 0; kk < BM / 8 / 4; ++kk) {
            const int p = wave + 4 * kk;
            buffer_load_lds16(r_rsrc, ring + (c ? RING_OFF : 0) + p * 1024, r_lane + r_base + c * 128 + p * (8 * COUT * 2), 0);
          }
      }
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const int r0 = wm * WPX + pn;
      const unsigned rb = lds0 + (unsigned)(wn ? RING_OFF : 0) + (unsigned)(r0 * 128 + ((g ^ ((r0 >> 1) & 7)) << 4));
      frag rf[4];
      static_for<2>([&](auto S) { lds_read16<(decltype(S)::value >> 1) * 2048>(rf[decltype(S)::value], (decltype(S)::value & 1) ? rb ^ 64u : rb); });
      static_for<2 * MT>([&](auto S) {
        constexpr int s2 = decltype(S)::value, i = s2 >> 1, kk = s2 & 1;
        if constexpr (s2 + 2 < 2 * MT) lds_read16<((s2 + 2) >> 1) * 2048>(rf[(s2 + 2) & 3], ((s2 + 2) & 1) ? rb ^ 64u : rb);
        wait_lgkmcnt<(s2 + 2 < 2 * MT) ? 2 : (2 * MT - 1 - s2)>();
        Asm16<T>::mfma(acc[i][2 * kk], ident[0], rf[s2 & 3]);
        Asm16<T>::mfma(acc[i][2 * kk + 1], ident[1], rf[s2 & 3]);
      });
    }
    // the accumulators were last written by MFMAs hipcc does not know about (halo16.h)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");

    // ---- direct epilogue (halo16.h): bias, ReLU, round, v_permlane16_swap pairs 16-lane rows into 16-byte items
    float4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + n0 + wn * WTN + 16 * j + 4 * g);
    const int c_lane = n0 + wn * WTN + 16 * (g & 1) + 8 * (g >> 1);
    HALO_STAMP(t_res);
    __builtin_amdgcn_s_barrier();  // every wave has left the K loop (and the residual tile): buffers and ring are free
    HALO_STAMP(t_bar);
    {
      const int vn = vb + gridDim.x;
      const int mtn = ((vn >> 3) / NTILES_N) * 8 + (vn & 7);
      if (mtn < n_mtiles) {
        const int m0n = mtn * BM, n0n = ((vn >> 3) % NTILES_N) * BN;
        band_offsets(m0n);
        issue_segment(m0n, 0, 0, 0);
#pragma unroll
        for (int h = 0; h < NSLOT - 1; ++h) {
          int hc, k;
          tap_of_flat(h, hc, k);
          issue_w(hc, k, h, n0n);
        }
      }
    }
    HALO_STAMP(t_pref);
    const int mw0 = m0 + wm * WPX + pn;
    static_for<MT>([&](auto SUB) {
      constexpr int i = decltype(SUB)::value;
      const int m = mw0 + 16 * i;
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        unsigned P[2][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          float v[4] = {acc[i][j][0] + bv[j].x, acc[i][j][1] + bv[j].y, acc[i][j][2] + bv[j].z, acc[i][j][3] + bv[j].w};
          if constexpr (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          P[jj][0] = PackPair<T>::pack_rn(v[0], v[1]);
          P[jj][1] = PackPair<T>::pack_rn(v[2], v[3]);
        }
        permlane16_swap(P[0][0], P[1][0]);
        permlane16_swap(P[0][1], P[1][1]);
        if (m < M) *reinterpret_cast<u32x4*>(outp + (size_t)m * COUT + c_lane + 32 * jp) = u32x4{P[0][0], P[0][1], P[1][0], P[1][1]};
      }
    });
#ifdef HIPAC_HALO_STAMPS
    HALO_STAMP(t_end);
    if (tid == 0) {
      atomicAdd(&g_halo_stamps[0], t_first - t_start);  // tile setup + wait for the prefetched band / weights / the other waves
      atomicAdd(&g_halo_stamps[1], t_loop - t_first);   // K loop
      atomicAdd(&g_halo_stamps[2], t_end - t_loop);     // residual step + epilogue
      atomicAdd(&g_halo_stamps[3], 1ull);
      atomicAdd(&g_halo_stamps[4], t_bar - t_loop);     // residual step + barrier
      atomicAdd(&g_halo_stamps[5], t_pref - t_bar);     // next tile's offsets, band and weight requests
      atomicAdd(&g_halo_stamps[6], t_end - t_pref);     // stores
    }
#endif
    prev_full = (m0 + BM <= M);
  }
}

}  // namespace hipac
