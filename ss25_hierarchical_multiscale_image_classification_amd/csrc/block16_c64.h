// block16_c64_kernel: the fused layer1 BasicBlock of block_c64.h (x ring -> conv1 -> I ring -> conv2 + shortcut, two teams of
// four waves walking down a strip of 14 columns; read that header first) on v_mfma_f32_16x16x32 (round 4).  Included by
// conv_igemm.h after halo16.h (Asm16, lds_read16, wait_lgkmcnt, permlane16_swap, perm16).
//
// What changes with the MFMA shape:
//   * a sub-tile is ONE row of 16 columns (was two rows): a wave's step = 4 rows x 16 columns x 32 channels = 4 x 2
//     accumulators of 4 registers (32, as before).  Lane (n, g): column perm16(n) of the row, k-group g; D gives it channels
//     16 t + 4 g .. + 3 of that pixel.
//   * weights in registers as before (144 VGPRs): fragment [tap][k32 step][16-channel tile] = row 16 t + n, channels
//     32 kk + 8 g .. + 7.
//   * an activation fragment (row, tap column, k32 step) is read once and feeds BOTH channel tiles: 72 reads for 144 MFMAs
//     (0.5 per 32x32x16-equivalent, was 1) -- with independent accumulators, unlike round 3's fragment-sharing experiment.
//     Bank conflicts: the rings keep their row format (column j at j * 128, chunk c at c ^ ((j >> 1) & 7)); the 16x16x32 lane
//     groups mix two k-groups, which perm16 puts on columns of opposite parity (halo16.h).
//   * the MFMA stream is written out (asm statements): reads one group (4 rows) ahead, counted lgkmcnt, in-place accumulators.
//   * conv2's output leaves as 16-byte items through v_permlane16_swap (the two channel tiles of a lane pair up): 64 contiguous
//     bytes per pixel and store; the shortcut pixels arrive by LDS-DMA in [pixel][64 B] rows (16-byte parts swizzled by
//     (pixel >> 2) & 3 on the source side: conflict-free 8-byte reads in the accumulator layout).
// The accumulation order differs from conv3x3_c64_kernel's (32 channels per MFMA instead of 16): fused == unfused holds to
// rounding now, not bit for bit (tests/test_gpu_resnet.py).
#pragma once

namespace hipac {

#ifndef HIPAC_BLK16_AHEAD
#define HIPAC_BLK16_AHEAD 1
#endif

// one step of a wave: rows 0 .. 2 NP - 1 (NP pairs of rows; pair p's 4-row window starts at lb[p][.]), all 9 taps x 64 channels.
// lb[p][kw]: LDS byte address of (window row 0, column perm16(n) + kw, chunk g swizzled) of pair p.
template <typename T, int NP, int PITCH>
__device__ __forceinline__ void c64_strip16_mfma(const typename Elem<T>::frag (&wreg)[9][2][2], const int (&lb)[NP][3],
                                                 const float* __restrict__ bl, f32x4 (&acc)[2 * NP][2]) {
  using frag = typename Elem<T>::frag;
  constexpr int NR = 2 * NP;  // rows
  // the bias is the initial accumulator: channels 16 t + 4 g .. + 3 (bl already points at this lane's 4 g)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 16 * t);
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i][t] = b;
  }
  // the bias values came through LDS reads hipcc waits for itself; from here on the LDS reads are counted by hand
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int AH = HIPAC_BLK16_AHEAD;  // groups of NR fragment reads in flight ahead of the group whose MFMAs are being issued
  frag fr[AH + 1][NR];
  // group s = (tap, k32 step): row i of pair p reads (lb[p][kw] ^ (kk << 6)) + ((i & 1) + kh) * PITCH
  auto rd = [&](auto S, auto I) {
    constexpr int s = decltype(S)::value, i = decltype(I)::value;
    constexpr int tap = s >> 1, kk = s & 1, kh = tap / 3, kw = tap % 3;
    lds_read16<((i & 1) + kh) * PITCH>(fr[s % (AH + 1)][i], (unsigned)(lb[i >> 1][kw] ^ (kk << 6)));
  };
  __builtin_amdgcn_s_setprio(1);
  static_for<AH>([&](auto S) { static_for<NR>([&](auto I) { rd(S, I); }); });
  static_for<18>([&](auto S) {
    constexpr int s = decltype(S)::value, tap = s >> 1, kk = s & 1;
    static_for<NR>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if constexpr (s + AH < 18) rd(std::integral_constant<int, s + AH>{}, I);
      // outstanding behind fragment (s, i): the rest of group s, the groups s+1 .. s+AH-1 that exist, and rows 0 .. i of group s+AH
      constexpr int full = (18 - 1 - s < AH - 1 ? 18 - 1 - s : AH - 1);  // whole groups behind it
      wait_lgkmcnt<(NR - 1 - i) + full * NR + (s + AH < 18 ? i + 1 : 0)>();
      Asm16<T>::mfma(acc[i][0], wreg[tap][kk][0], fr[s % (AH + 1)][i]);
      Asm16<T>::mfma(acc[i][1], wreg[tap][kk][1], fr[s % (AH + 1)][i]);
    });
  });
  __builtin_amdgcn_s_setprio(0);
  // the accumulators were last written by MFMAs hipcc does not know about: its wait states in front of their first reader
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
}

template <typename T>
__global__ __launch_bounds__(512, 2) void block16_c64_kernel(const T* __restrict__ in, const T* __restrict__ w1,
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
  const int n16 = lane & 15, g = lane >> 4;
  const int lx = n16 < 4 ? 2 * n16 : (n16 < 12 ? 2 * n16 - 7 : 2 * n16 - 16);  // perm16(n16): this lane's column of a row
  unsigned char* const Rw = smem + kBlkXBytes + kBlkIBytes + (mh == 0 ? nh * 6144 : 2 * 6144 + nh * 4096);

  if (tid < 64) Bl[tid] = b1[tid];
  else if (tid < 128) Bl[tid] = b2[tid - 64];

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int n_x = n_img > xcd ? (n_img - xcd + 7) >> 3 : 0;
  const int nst = 4 * n_x;
  const int ns = nst > slot ? (nst - slot + nslots - 1) / nslots : 0;
  if (ns == 0) return;  // the whole workgroup

  // weights of this wave's convolution in registers: [tap][k32 step][16-channel tile]: row nh*32 + 16 t + n, channels 32 kk + 8 g ..
  frag wreg[9][2][2];
  {
    const char* wb = reinterpret_cast<const char*>(team ? w2 : w1) + (size_t)(nh * 32 + n16) * (9 * C * 2) + 16 * g;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t) wreg[tap][kk][t] = *reinterpret_cast<const frag*>(wb + t * 16 * (9 * C * 2) + tap * 128 + kk * 64);
  }

  const rsrc_t x_rsrc = make_rsrc(in, n_img * (H * W * C * 2));

  // ---- x ring DMA (team A): exactly block_c64.h's (lane-linear 1 KB pieces; nothing depends on the MFMA shape)
  constexpr int NKP = 5;
  int relo[NKP];
#pragma unroll
  for (int kp = 0; kp < NKP; ++kp) {
    const int q = 8 * (tw + 4 * kp) + (lane >> 3);
    const int rr = (q * 3641) >> 16;  // q / 18 for q < 160
    const int j = q - 18 * rr;
    const int c = (lane & 7) ^ ((j >> 1) & 7);
    relo[kp] = (((rr * W + j) << 7) + (c << 4)) | (j < 2 ? 1 : 0) | (j >= 16 ? 2 : 0) | (rr == 0 ? 4 : 0) | (rr >= 1 ? 8 : 0);
  }
  auto x_rows = [&](int img, int x0, int L0, int rb, int np, int rowmask) {
    const int sb = ((img * H + L0 - 1) * W + x0 - 2) << 7;
    const int smask = (x0 == 0 ? 1 : 0) | (x0 == 42 ? 2 : 0) | rowmask;
    static_for<NKP>([&](auto KP) {
      constexpr int kp = decltype(KP)::value;
      const int p = tw + 4 * kp;
      if (p < np) {
        asm volatile("" : "+v"(relo[kp]));  // keep offset and flags in ONE register (no hoisted, split copies)
        const int off = (relo[kp] & smask) ? (int)0x80000000 : (relo[kp] & ~15) + sb;  // out of range: zeros
        buffer_load_lds16(x_rsrc, Xr + rb * XP + p * 1024, off, 0);
      }
    });
  };
  auto x_group = [&](int img, int x0, int gq) {
    const int rb = 8 * gq >= RING ? 8 * gq - RING : 8 * gq;
    x_rows(img, x0, 8 * gq, rb, gq == 7 ? 5 : 18, gq == 0 ? 4 : (gq == 7 ? 8 : 0));
    if (gq == 3) x_rows(img, x0, RING, 0, 5, 0);
  };
  auto strip_img = [&](int k) { return (((slot + k * nslots) >> 2) << 3) + xcd; };
  auto strip_x0 = [&](int k) { return ((slot + k * nslots) & 3) * 14; };

  // per-lane address parts
  const int lds_base = (int)(unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  int la[3];  // activation fragment of tap column kw: (window row 0, column lx + kw), chunk g swizzled (k32 step 1: ^ 64)
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int j = lx + kw;
    la[kw] = (j << 7) + (((((j >> 1) & 7)) ^ g) << 4) + (team ? kBlkXBytes : 0) + lds_base;
  }
  // team A: this lane's 8 bytes (channels 16 t + 4 g .. + 3 of its 32) inside its I pixel: chunk 4 nh + 2 t + (g >> 1), half g & 1
  const int iw_lane = (lx << 7) + ((g & 1) << 3);
  const int iw_pos = ((nh << 2) + (g >> 1)) ^ ((lx >> 1) & 7);  // tile t: ^ (2 t)  (bit 1 of the chunk index)
  const float* const bl = Bl + team * 64 + nh * 32 + 4 * g;     // this lane's bias values: bl[16 t + e]

#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(wreg[tap][kk][t]));  // weight loads retire before the loop
  if (team == 0) {
    x_group(strip_img(0), strip_x0(0), 0);
    x_group(strip_img(0), strip_x0(0), 1);
    wait_vmcnt<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int n_steps = 7 * ns;
  f32x4 acc[4][2];             // [row][channel tile]
  int pend_n = 0, pend_y = 0;  // B: deferred row pairs (0, 1 or 2) and their first row
  // shortcut pixels of a row pair by LDS-DMA into the wave's staging slot [row 0 | row 1][16 px][64 B]: one piece per row,
  // lane -> pixel lane >> 2, part lane & 3, which holds source part (lane & 3) ^ ((pixel >> 2) & 3)
  const int rs_px = lane >> 2;
  const int rs_off = (((rs_px < 14 ? rs_px : 13)) * C + nh * 32 + 8 * ((lane & 3) ^ ((rs_px >> 2) & 3))) * 2;
  auto fetch_resid = [&](int pix0, int y0, int sl) {
    const int sb = (pix0 + y0 * W) << 7;
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) buffer_load_lds16(x_rsrc, Rw + sl * 2048 + rho * 1024, rs_off + rho * (W * C * 2), sb);
  };
  // this lane's 8 shortcut bytes of (row rho, tile t): part 2 t + (g >> 1) at its swizzled position, half g & 1
  const int rs_rd = lx * 64 + 8 * (g & 1);
  const int rs_sw = (lx >> 2) & 3;
  const int out_lane = (lx * C + nh * 32 + 16 * (g & 1) + 8 * (g >> 1)) * 2;  // byte offset of this lane's 16 output bytes in its row
  // conv2 epilogue of a row pair (rows y0, y0 + 1 = acc rows a0, a0 + 1): + shortcut (fp32), round to T, ReLU, pair the two
  // channel tiles of neighbouring 16-lane rows, 16-byte stores
  auto epilogue_b = [&](auto A0, int pix0, int y0, int sl) {
    constexpr int a0 = decltype(A0)::value;
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      unsigned P[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const vec4 rv = *reinterpret_cast<const vec4*>(Rw + sl * 2048 + rho * 1024 + rs_rd + (((2 * t + (g >> 1)) ^ rs_sw) << 4));
        const f32x4& a = acc[a0 + rho][t];
        P[t][0] = relu_pk(PackPair<T>::pack_rn(a[0] + (float)rv[0], a[1] + (float)rv[1]));
        P[t][1] = relu_pk(PackPair<T>::pack_rn(a[2] + (float)rv[2], a[3] + (float)rv[3]));
      }
      permlane16_swap(P[0][0], P[1][0]);
      permlane16_swap(P[0][1], P[1][1]);
      char* const dst = reinterpret_cast<char*>(out) + ((size_t)(unsigned)(pix0 + (y0 + rho) * W) << 7) + out_lane;
      if (lx < 14) store16_out<HIPAC_NT_STORES != 0>(dst, u32x4{P[0][0], P[0][1], P[1][0], P[1][1]});
    }
  };
  int pend_pix = 0, pend_sl = 0;
  auto flush_b = [&]() {  // the deferred epilogue(s); their shortcut pixels were requested a step ago
    if (pend_n) {
      wait_vmcnt<0>();
      epilogue_b(std::integral_constant<int, 0>{}, pend_pix, pend_y, pend_sl);
      if (pend_n == 2) epilogue_b(std::integral_constant<int, 2>{}, pend_pix, pend_y + 2, pend_sl + 1);
      pend_n = 0;
    }
  };
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // step boundary: I rows written <-> read, x rows landed <-> read, for both teams
  };
  if (team == 0) {
    for (int G = 0; G <= n_steps; ++G) {
      if (G < n_steps) {
        // ------------------------------ team A: conv1, I rows 8m .. 8m+7 (this wave: 4 of them) ------------------------------
        const int k = G / 7, m = G - 7 * k;
        const int img = strip_img(k), x0 = strip_x0(k);
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
        for (int p = 0; p < 2; ++p) {
          const int y0 = y0a + 2 * p;
          const int wbase = (y0 >= RING ? y0 - RING : y0) * XP;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) lb[p][kw] = la[kw] + wbase;
        }
        c64_strip16_mfma<T, 2, XP>(wreg, lb, bl, acc);
        // epilogue: ReLU, round to T, into the I ring (columns outside the image as zeros)
        const bool col_ok = !((x0 == 0 && lx == 0) || (x0 == 42 && lx == 15));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int Lr = y0a + i + 1;  // (uniform: a sub-tile is one row)
          const bool twice = Lr == RING || Lr == RING + 1;
          unsigned char* const dst = Ir + (Lr >= RING ? Lr - RING : Lr) * IP + iw_lane;
          unsigned char* const dst2 = Ir + Lr * IP + iw_lane;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            u32x2 pk;
            pk[0] = relu_pk(PackPair<T>::pack_rn(acc[i][t][0], acc[i][t][1]));
            pk[1] = relu_pk(PackPair<T>::pack_rn(acc[i][t][2], acc[i][t][3]));
            if (!col_ok) pk = u32x2{0u, 0u};
            *reinterpret_cast<u32x2*>(dst + ((iw_pos ^ (2 * t)) << 4)) = pk;
            if (twice) *reinterpret_cast<u32x2*>(dst2 + ((iw_pos ^ (2 * t)) << 4)) = pk;
          }
        }
      }
      wait_vmcnt<0>();  // the rows requested at the top of this step have landed
      step_barrier();
    }
  } else {
    for (int G = 0; G <= n_steps; ++G) {
      // ------------------------------ team B: conv2 + shortcut, ten rows behind ------------------------------
      flush_b();
      if (G >= 1) {
        const int k = (G - 1) / 7, mp = G - 7 * k;  // 1 .. 7
        // row pairs of this wave: a double pair (4 rows) and / or a single one (2 rows)
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
          for (int p = 0; p < 2; ++p) {
            const int y0 = yp + 2 * p;
            const int wbase = (y0 >= RING ? y0 - RING : y0) * IP;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) lb[p][kw] = la[kw] + wbase;
          }
          c64_strip16_mfma<T, 2, IP>(wreg, lb, bl, acc);
          pend_n = 2, pend_y = yp, pend_sl = 0;
        }
        if (ys >= 0) {
          flush_b();  // a pair of this very step (only the last step of a strip, mh = 0): its epilogue is not deferred
          int lb[1][3];
          const int wbase = (ys >= RING ? ys - RING : ys) * IP;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) lb[0][kw] = la[kw] + wbase;
          f32x4(&acc1)[2][2] = reinterpret_cast<f32x4(&)[2][2]>(acc[0]);
          c64_strip16_mfma<T, 1, IP>(wreg, lb, bl, acc1);
          pend_n = 1, pend_y = ys, pend_sl = has_pair ? 2 : 0;
        }
      }
      step_barrier();
    }
    flush_b();
  }
}

}  // namespace hipac
