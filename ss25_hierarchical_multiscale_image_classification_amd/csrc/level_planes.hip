// Level resampler ("planes"): the whole-level form of the tile preprocess for windows on
// the reference's 224-pixel lattice (src/main.py:682-683: x, y in range(0, ..., 224)).
//
// With window size P = 224*s (s = 2, 4, 8) and origins on multiples of 224, column j of a
// window at x = 224*i resamples source columns that depend only on the GLOBAL output column
// g = (224/s)*i + j -- except j = 0 and j = 223, whose Pillow kernels are clamped to the
// window and therefore depend on i.  The same holds for rows.  So instead of resampling each
// window on its own (each level-0 pixel up to 64 times at stride 224) the level is resampled
// ONCE into an image D of (GY + 2*NY) rows x (G + 2*NX) columns:
//     columns [0, G)            interior kernel at global column g
//     columns [G, G+NX)         left-edge kernel (j = 0) of window column i
//     columns [G+NX, G+2NX)     right-edge kernel (j = 223) of window column i
// and likewise for rows; a window is then a gather from D.  Arithmetic is Pillow's two-pass
// 8bpc resampler exactly (horizontal pass, uint8, vertical pass, uint8; 22-bit weights): the
// interior weights are (2t+1) * 2^22 / (2 s^2) exactly, so the interior accumulation is the
// small-integer dot product (s^2 + sum w_t p_t) >> log2(2 s^2), done with v_dot4_u32_u8.
// The 224x224 "cell" sums needed for the whiteness test (a window = s x s cells) fall out of
// the same pass over the source.  Pixels are kept as RGBX dwords in the H and D images.
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "common.h"

namespace hipac {

constexpr int kLat = 224;  // window lattice and output size

struct PlaneGeom {
  int s, ng;          // scale, 224/s
  int NX, NY;         // window columns / rows on the lattice (origins < W, < H)
  int G, GY;          // dense output columns / rows
  int HW;             // H/D image row length in pixels = G + 2*NX
  int HROWS;          // source rows covered by the horizontal pass = 224*(NY-1) + P
  int DROWS;          // D image rows = GY + 2*NY
  int NGRP;           // 8-pixel source groups per row = (224*(NX-1) + P) / 8
  int NCX, NCY;       // cell grid
};

static PlaneGeom make_geom(int W, int H, int P) {
  PlaneGeom g;
  g.s = P / kLat;
  g.ng = kLat / g.s;
  g.NX = (W + kLat - 1) / kLat;
  g.NY = (H + kLat - 1) / kLat;
  g.G = g.ng * (g.NX - 1) + kLat;
  g.GY = g.ng * (g.NY - 1) + kLat;
  g.HW = g.G + 2 * g.NX;
  g.HROWS = kLat * (g.NY - 1) + P;
  g.DROWS = g.GY + 2 * g.NY;
  g.NGRP = (kLat * (g.NX - 1) + P) / 8;
  g.NCX = g.NX + g.s - 1;
  g.NCY = g.NY + g.s - 1;
  return g;
}

// Pillow's clip8 of a 22-bit fixed-point accumulator.  The empty asm keeps hipcc (ROCm 7.2)
// from fusing two neighbouring clamps into gfx950's v_ashr_pk_u8_i32: that instruction leaves
// non-zero bits above bit 15 of its result, which the surrounding `| (b << 16)` then picks up
// (observed: a wrong blue channel; reproduced in isolation).
__device__ __forceinline__ unsigned clip8p(int acc) {
  int v = acc >> 22;
  v = v < 0 ? 0 : v;
  v = v > 255 ? 255 : v;
  asm volatile("" : "+v"(v));
  return (unsigned)v;
}

// weight dword for the interior kernel: bytes of cover dword `dw` (cover starts CB0 bytes
// before the group's first byte), channel ch, output k of the group
template <int S>
__host__ __device__ constexpr unsigned interior_wdword(int k, int ch, int dw) {
  constexpr int CB0 = S == 8 ? 12 : (S == 4 ? 8 : 4);
  unsigned w = 0;
  for (int b = 0; b < 4; ++b) {
    const int byte = dw * 4 + b - CB0;            // relative to the group's first byte (pixel 8h)
    const int px = byte >= 0 ? byte / 3 : -((-byte + 2) / 3);  // floor(byte / 3)
    const int c = byte - px * 3;
    const int t = px - (S * k - S / 2);           // tap index of output k (taps start at 8h + S*k - S/2)
    if (c == ch && t >= 0 && t < 2 * S) {
      const int wt = t < S ? 2 * t + 1 : 4 * S - 1 - 2 * t;
      w |= (unsigned)wt << (8 * b);
    }
  }
  return w;
}

// byte `bo` of the cover (compile-time offset)
template <int BO, int NDW>
__device__ __forceinline__ int cover_byte(const unsigned (&d)[NDW]) {
  static_assert(BO >= 0 && BO < 4 * NDW, "inside the cover");
  return (int)((d[BO / 4] >> (8 * (BO % 4))) & 255u);
}

// Pillow's clamped kernel for window column 0 / 223: 3S/2 taps starting P0 pixels after the
// group's first pixel, general 22-bit weights kt[0..3S/2)
template <int S, int P0, int NDW>
__device__ __forceinline__ unsigned edge_pixel(const unsigned (&d)[NDW], const int* __restrict__ kt) {
  constexpr int CB0 = S == 8 ? 12 : (S == 4 ? 8 : 4);
  constexpr int ECNT = 3 * S / 2;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
  static_for<ECNT>([&](auto TT) {
    constexpr int t = decltype(TT)::value;
    constexpr int bo = CB0 + 3 * (P0 + t);
    const int kv = kt[t];
    // 24-bit multiplies (v_mad_i32_i24, full rate; a 32-bit v_mul_lo is quarter rate): byte < 2^8, kv < 2^23
    a0 += __mul24(cover_byte<bo>(d), kv);
    a1 += __mul24(cover_byte<bo + 1>(d), kv);
    a2 += __mul24(cover_byte<bo + 2>(d), kv);
  });
  return clip8p(a0) | (clip8p(a1) << 8) | (clip8p(a2) << 16) | 0xff000000u;
}

// ---------------------------------------------------------------------------------------
// K1: horizontal pass.  thread = one group of 8 source pixels (24 bytes) x a strip of 16 rows.
// ---------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void hpass_kernel(const uint8_t* __restrict__ level, int W, int H,
                                                    long long pitch, PlaneGeom gm,
                                                    const int* __restrict__ bounds, const int* __restrict__ kk,
                                                    int ksize, unsigned* __restrict__ himg,
                                                    unsigned* __restrict__ cells) {
  constexpr int CB0 = S == 8 ? 12 : (S == 4 ? 8 : 4);   // cover start, bytes before the group
  constexpr int NDW = S == 8 ? 12 : (S == 4 ? 10 : 8);  // cover length in dwords
  constexpr int NOUT = 8 / S;                           // dense outputs per group
  constexpr int LOG2 = S == 8 ? 7 : (S == 4 ? 5 : 3);   // log2(2 S^2)
  constexpr int RS = 16;                                // rows per strip (divides 224)
  constexpr int GPC = kLat / 8;                         // groups per cell = 28
  const int h = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = h < gm.NGRP;
  const int r0 = blockIdx.y * RS;
  const long long cb = 24LL * h - CB0;
  const long long wbytes = 3LL * W;
  // window-edge duties of this group
  const int hm = h % GPC;
  const int win_i = h / GPC;
  const bool is_left = live && hm == 0 && win_i < gm.NX;
  // right edge of window i ends at pixel 224 i + P: group index 28 i + P/8 - 1
  const int hr = h - (S * kLat / 8 - 1);
  const bool is_right = live && hr >= 0 && hr % GPC == 0 && hr / GPC < gm.NX;
  const int right_i = hr / GPC;
  unsigned gsum = 0;
  for (int rr = 0; rr < RS; ++rr) {
    const int y = r0 + rr;
    if (y >= gm.HROWS) break;
    unsigned d[NDW];
    const bool row_ok = y < H;
    const uint8_t* rowp = level + (long long)y * pitch;
    // fast path (almost every thread): the whole cover lies inside the row's real pixels ->
    // 16-byte (and one 8-byte) loads, no masking.  The base is only 4-byte aligned, which is
    // all global_load_dwordx4 needs.
    if (live && row_ok && cb >= 0 && cb + 4 * NDW <= wbytes) {
      const uint8_t* q = rowp + cb;
#pragma unroll
      for (int k = 0; k + 4 <= NDW; k += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(q + 4 * k);
        d[k] = v[0];
        d[k + 1] = v[1];
        d[k + 2] = v[2];
        d[k + 3] = v[3];
      }
      if constexpr (NDW % 4 == 2) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(q + 4 * (NDW - 2));
        d[NDW - 2] = v[0];
        d[NDW - 1] = v[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDW; ++k) {
        const long long o = cb + 4 * k;
        unsigned v = 0xffffffffu;
        if (live && row_ok && o >= 0 && o + 4 <= pitch) {
          v = *reinterpret_cast<const unsigned*>(rowp + o);
          const long long nv = wbytes - o;  // valid bytes in this dword
          if (nv < 4) v |= nv <= 0 ? 0xffffffffu : (0xffffffffu << (8 * (int)nv));
        }
        d[k] = v;
      }
    }
    if (live) {
      // whiteness: bytes [24h, 24h+24) = cover dwords CB0/4 .. CB0/4+5
#pragma unroll
      for (int k = 0; k < 6; ++k) gsum = __builtin_amdgcn_sad_u8(d[CB0 / 4 + k], 0u, gsum);
      // dense interior outputs (weight dwords are compile-time constants; zero ones vanish)
      unsigned* hrow = himg + (long long)y * gm.HW;
      static_for<NOUT>([&](auto K) {
        constexpr int k = decltype(K)::value;
        unsigned px = 0xff000000u;
        static_for<3>([&](auto CH) {
          constexpr int ch = decltype(CH)::value;
          unsigned x = 0;
          static_for<NDW>([&](auto DW) {
            constexpr int dw = decltype(DW)::value;
            constexpr unsigned wt = interior_wdword<S>(k, ch, dw);
            if constexpr (wt != 0) x = __builtin_amdgcn_udot4(d[dw], wt, x, false);
          });
          px |= ((S * S + x) >> LOG2) << (8 * ch);
        });
        hrow[h * NOUT + k] = px;
      });
      // sparse edge outputs: Pillow's clamped kernels for window columns 0 and 223 (general
      // 22-bit weights from the table; tap positions are fixed by S: 3S/2 taps at the window's
      // first / last pixels), bytes extracted from the cover at compile-time offsets
      if (is_left) hrow[gm.G + win_i] = edge_pixel<S, 0>(d, kk);
      if (is_right) hrow[gm.G + gm.NX + right_i] = edge_pixel<S, 8 - 3 * S / 2>(d, kk + 223 * ksize);
    }
  }
  // cell sums: segmented reduction over the lanes of a wave that share a cell column
  const int cx = live ? h / GPC : -1 - lane;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(gsum, off, 64);
    const int cup = __shfl_up(cx, off, 64);
    if (lane >= off && cup == cx) gsum += up;
  }
  const int cnext = __shfl_down(cx, 1, 64);
  if (live && (lane == 63 || cnext != cx)) atomicAdd(&cells[(r0 / kLat) * gm.NCX + cx], gsum);
}

// ---------------------------------------------------------------------------------------
// K2: vertical pass over the H image.  thread = one D pixel.
// ---------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void vpass_kernel(const unsigned* __restrict__ himg, PlaneGeom gm,
                                                    const int* __restrict__ bounds, const int* __restrict__ kk,
                                                    int ksize, unsigned* __restrict__ dimg) {
  constexpr int LOG2 = S == 8 ? 7 : (S == 4 ? 5 : 3);
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int rr = blockIdx.y;
  if (col >= gm.HW) return;
  unsigned out;
  if (rr < gm.GY) {
    const int row0 = S * rr - S / 2;
    if (row0 < 0 || row0 + 2 * S > gm.HROWS) {
      out = 0xff000000u;  // first / last dense row: never used (rows 0 and 223 of a window are edge rows)
    } else {
      unsigned rb = 0, g = 0;  // R and B accumulate in 16-bit lanes of one dword (max 255 * 2 S^2 < 65536)
#pragma unroll
      for (int t = 0; t < 2 * S; ++t) {
        const unsigned wt = t < S ? 2 * t + 1 : 4 * S - 1 - 2 * t;
        const unsigned px = himg[(long long)(row0 + t) * gm.HW + col];
        rb += (px & 0x00ff00ffu) * wt;
        g += ((px >> 8) & 0xffu) * wt;
      }
      const unsigned r = ((rb & 0xffffu) + S * S) >> LOG2;
      const unsigned b = ((rb >> 16) + S * S) >> LOG2;
      const unsigned gg = (g + S * S) >> LOG2;
      out = r | (gg << 8) | (b << 16) | 0xff000000u;
    }
  } else {
    const bool top = rr < gm.GY + gm.NY;
    const int iy = top ? rr - gm.GY : rr - gm.GY - gm.NY;
    const int j = top ? 0 : 223;
    const int row0 = kLat * iy + bounds[2 * j], cnt = bounds[2 * j + 1];
    const int* k = kk + j * ksize;
    int a0 = 1 << 21, a1 = a0, a2 = a0;
    for (int t = 0; t < cnt; ++t) {
      const unsigned px = himg[(long long)(row0 + t) * gm.HW + col];
      const int kv = k[t];
      a0 += (int)(px & 255u) * kv;
      a1 += (int)((px >> 8) & 255u) * kv;
      a2 += (int)((px >> 16) & 255u) * kv;
    }
    out = clip8p(a0) | (clip8p(a1) << 8) | (clip8p(a2) << 16) | 0xff000000u;
  }
  dimg[(long long)rr * gm.HW + col] = out;
}

// ---------------------------------------------------------------------------------------
// K1+K2 fused ("planes_kernel", the one that runs): horizontal AND vertical pass in one sweep over the source.
// thread = one group of 8 source pixels x a strip of 112 OWNED source rows (+ S/2 halo rows on either side:
// 120 of 112 rows read, 1.07x): the horizontal result of every row (NOUT dense pixels, plus the clamped
// left / right edge pixel for the groups that carry one) never leaves the registers -- each row adds into
// two running vertical sums per output column (the rising half of output row rr and the falling half of
// rr - 1: Pillow's triangle weights 2t + 1 and 2S - 1 - 2t), and a D pixel is emitted every S rows.  The
// clamped top / bottom window rows (general 22-bit weights) accumulate the same way over the 3S/2 rows they
// cover; strips start at multiples of 112 rows, so those ranges (rows 0..3S/2-1 and 224-3S/2..223 modulo
// 224) never straddle a strip.  The H image (1.6 GB written and read back per 50 000^2 level) no longer
// exists: source bytes once (+7 %), D image written once.
// ---------------------------------------------------------------------------------------
#ifndef HIPAC_PL_ABL
#define HIPAC_PL_ABL 0  // developer builds (wrong results): 1 = no edge columns, 2 = loads only, 4 = no loads
#endif
constexpr int kEdgeRows = 8;    // rows per edge batch
constexpr int kEdgeSlots = 8;   // edge groups of a wave: left edges in slots 0-3, right edges in 4-7 (at most 3 + 3 occur)

// Vertical state of ONE D column: running sums of the current block of S rows (R = sum (2t+1) p, P = sum p),
// the previous block's R, and the accumulators of the clamped top / bottom window-row kernel.
struct VCol {
  unsigned a_rb, a_g, p_rb, p_g, b_rb, b_g;
  int e0, e1, e2;
  __device__ __forceinline__ void reset() {
    a_rb = a_g = p_rb = p_g = b_rb = b_g = 0u;
    e0 = e1 = e2 = 1 << 21;
  }
};

// Feed the horizontally resampled pixel `px` (RGBX) of strip row rr (source row y) into column state `v` and write the
// D pixels that complete: the dense row every S rows, the clamped window rows 0 / 223 after their 3S/2 rows.
// Inside a block of S rows (t = rr mod S) a row feeds the rising half of output rr with weight 2t + 1 and the falling
// half of output rr - 1 with 2S - 1 - 2t = 2S - (2t + 1): only R and P are kept, the falling part is 2S P - R.
// R and B share a dword as 16-bit lanes (<= 255 * 2 S^2 < 65536; 2S P >= R in every lane: no borrow).
template <int S>
__device__ __forceinline__ void vcol_step(VCol& v, unsigned px, int rr, int y, int y_own, bool owned, int dcol,
                                          const PlaneGeom& gm, const int* __restrict__ kk, int ksize, int top0, int bot0,
                                          unsigned* __restrict__ dimg) {
  constexpr int LOG2 = S == 8 ? 7 : (S == 4 ? 5 : 3);
  constexpr int ECNT = 3 * S / 2;
  const unsigned t = (unsigned)(rr % S), wa = 2 * t + 1;
  const unsigned rb = px & 0x00ff00ffu, g = (px >> 8) & 0xffu;
  v.a_rb += __umul24(rb, wa), v.a_g += __umul24(g, wa);
  v.p_rb += rb, v.p_g += g;
  if (t == S - 1) {  // output row rr_out = (y_own + rr + 1) / S - 2 completes (uniform)
    const int rr_out = (y_own + rr + 1 - 2 * S) / S;
    const bool emit = rr >= S && rr_out < gm.GY;  // the first block of a strip only starts the next row's rising half
    const int row0 = S * rr_out - S / 2;
    const bool unused = row0 < 0 || row0 + 2 * S > gm.HROWS;  // first / last dense row: never gathered
    const unsigned tot_rb = v.b_rb + (2 * S * v.p_rb - v.a_rb), tot_g = v.b_g + (2 * S * v.p_g - v.a_g);
    const unsigned r = ((tot_rb & 0xffffu) + S * S) >> LOG2, b = ((tot_rb >> 16) + S * S) >> LOG2;
    const unsigned gg = (tot_g + S * S) >> LOG2;
    if (emit && dcol >= 0)
      dimg[(long long)rr_out * gm.HW + dcol] = unused ? 0xff000000u : (r | (gg << 8) | (b << 16) | 0xff000000u);
    v.b_rb = v.a_rb, v.b_g = v.a_g, v.a_rb = v.a_g = v.p_rb = v.p_g = 0u;
  }
  // clamped window rows (top: window row 0, bottom: window row 223), general 22-bit weights
  if (owned) {
    const int ym = y % kLat;
    int te = -1, iy = 0, jrow = 0;
    if (ym >= top0 && ym < top0 + ECNT) te = ym - top0, iy = y / kLat, jrow = 0;
    else {
      const int yb = y - bot0;  // = 224 iy + t
      if (yb >= 0 && yb % kLat < ECNT) te = yb % kLat, iy = yb / kLat, jrow = 223;
    }
    if (te >= 0 && iy < gm.NY) {
      const int kv = kk[jrow * ksize + te];
      v.e0 += __mul24((int)(px & 255u), kv), v.e1 += __mul24((int)((px >> 8) & 255u), kv);
      v.e2 += __mul24((int)((px >> 16) & 255u), kv);
      if (te == ECNT - 1) {
        if (dcol >= 0)
          dimg[(long long)(gm.GY + (jrow ? gm.NY : 0) + iy) * gm.HW + dcol] =
              clip8p(v.e0) | (clip8p(v.e1) << 8) | (clip8p(v.e2) << 16) | 0xff000000u;
        v.e0 = v.e1 = v.e2 = 1 << 21;
      }
    }
  }
}

// EDGE: the wave also produces the clamped left / right window columns (j = 0, 223) of the groups that carry one
// (group 0 / group 27 of every 28).  Only 4-6 of a wave's 64 lanes do, so the general-weight kernel (36 multiply-adds
// per pixel) is NOT run per row under a mostly empty exec mask: the edge lanes park their raw cover in LDS, every
// kEdgeRows rows the whole wave computes the kEdgeRows x kEdgeSlots edge pixels at once (lane = (row, slot)), and the
// owning lanes then run the vertical step of their one edge column over those rows.  (Ablation on a 50 000^2 level: the
// per-row form cost 1.5 ms of VALU time next to 0.75 ms for everything else and 1.6 ms of streaming.)
template <int S, bool EDGE>
__device__ __forceinline__ void planes_strip(const uint8_t* __restrict__ level, int W, int H, long long pitch,
                                             const PlaneGeom& gm, const int* __restrict__ bounds,
                                             const int* __restrict__ kk, int ksize, unsigned* __restrict__ dimg,
                                             unsigned* __restrict__ cells, int h, bool live, unsigned* __restrict__ lds_wave) {
  constexpr int CB0 = S == 8 ? 12 : (S == 4 ? 8 : 4);
  constexpr int NDW = S == 8 ? 12 : (S == 4 ? 10 : 8);
  constexpr int NOUT = 8 / S;
  constexpr int LOG2 = S == 8 ? 7 : (S == 4 ? 5 : 3);
  constexpr int OWN = 112;                 // owned rows per strip
  constexpr int GPC = kLat / 8;
  const int lane = threadIdx.x & 63;
  const int y_own = blockIdx.y * OWN;
  const long long cb = 24LL * h - CB0;
  const long long wbytes = 3LL * W;
  const int hm = h % GPC, win_i = h / GPC;
  const bool is_left = EDGE && live && hm == 0 && win_i < gm.NX;
  const int hr = h - (S * kLat / 8 - 1);
  const bool is_right = EDGE && live && hr >= 0 && hr % GPC == 0 && hr / GPC < gm.NX;
  const int right_i = hr / GPC;
  const bool is_edge = is_left || is_right;
  // D columns of this thread's outputs (-1: none)
  int dcol[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; ++k) dcol[k] = live ? h * NOUT + k : -1;
  const int ecol = is_left ? gm.G + win_i : (is_right ? gm.G + gm.NX + right_i : -1);
  // LDS slot of an edge lane: its rank among the wave's left (0..) / right (4..) edge lanes
  int eslot = 0;
  if constexpr (EDGE) {
    const unsigned long long ml = __ballot(is_left), mr = __ballot(is_right);
    const unsigned long long below = (1ull << lane) - 1ull;
    eslot = is_left ? __popcll(ml & below) : 4 + __popcll(mr & below);
    eslot &= kEdgeSlots - 1;
  }
  unsigned* const raw = lds_wave;                                      // [kEdgeRows][kEdgeSlots][NDW]
  unsigned* const epx = lds_wave + kEdgeRows * kEdgeSlots * NDW;       // [kEdgeRows][kEdgeSlots]

  VCol vc[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) vc[c].reset();
  // the vertical state of the edge column is touched once per batch: it lives in LDS between batches (9 VGPRs = one
  // wave per SIMD of occupancy at S = 8)
#ifndef HIPAC_PL_VE_LDS
#define HIPAC_PL_VE_LDS 0  // 1: the edge column's vertical state lives in LDS between batches (measured: no register gain)
#endif
  VCol* const ve_home = reinterpret_cast<VCol*>(epx + kEdgeRows * kEdgeSlots) + eslot;
#if HIPAC_PL_VE_LDS
  if constexpr (EDGE) {
    if (is_edge) {
      VCol z;
      z.reset();
      *ve_home = z;
    }
  }
#else
  VCol ve;
  ve.reset();
#endif
  unsigned gsum = 0;
  const int top0 = bounds[0], bot0 = bounds[2 * 223];  // first source row (relative to the window) of the edge kernels
  auto load_row = [&](int rr, unsigned (&d)[NDW]) {
    const int y = y_own - S / 2 + rr;
    const bool row_ok = y >= 0 && y < H && rr < OWN + S;
    const uint8_t* rowp = level + (long long)(row_ok ? y : 0) * pitch;
    if (live && row_ok && cb >= 0 && cb + 4 * NDW <= wbytes) {
      const uint8_t* q = rowp + cb;
#pragma unroll
      for (int k = 0; k + 4 <= NDW; k += 4) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(q + 4 * k);
        d[k] = v[0], d[k + 1] = v[1], d[k + 2] = v[2], d[k + 3] = v[3];
      }
      if constexpr (NDW % 4 == 2) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(q + 4 * (NDW - 2));
        d[NDW - 2] = v[0], d[NDW - 1] = v[1];
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDW; ++k) {
        const long long o = cb + 4 * k;
        unsigned v = 0xffffffffu;
        if (live && row_ok && o >= 0 && o + 4 <= pitch) {
          v = *reinterpret_cast<const unsigned*>(rowp + o);
          const long long nv = wbytes - o;
          if (nv < 4) v |= nv <= 0 ? 0xffffffffu : (0xffffffffu << (8 * (int)nv));
        }
        d[k] = v;
      }
    }
  };
  auto is_owned = [&](int rr) { return rr >= S / 2 && rr < S / 2 + OWN && y_own - S / 2 + rr < gm.HROWS; };
  auto process_row = [&](int rr, const unsigned (&d)[NDW]) {
    const int y = y_own - S / 2 + rr;
    const bool owned = is_owned(rr);
    // whiteness sums over the OWNED rows only (halo rows belong to the neighbouring strips)
    if (live && owned) {
#pragma unroll
      for (int k = 0; k < 6; ++k) gsum = __builtin_amdgcn_sad_u8(d[CB0 / 4 + k], 0u, gsum);
    }
    // horizontal pass of this row: the thread's dense pixels, RGBX, each straight into its column's vertical step
    static_for<NOUT>([&](auto K) {
      constexpr int k = decltype(K)::value;
      unsigned p = 0xff000000u;
      static_for<3>([&](auto CH) {
        constexpr int ch = decltype(CH)::value;
        unsigned x = 0;
        static_for<NDW>([&](auto DW) {
          constexpr int dw = decltype(DW)::value;
          constexpr unsigned wt = interior_wdword<S>(k, ch, dw);
          if constexpr (wt != 0) x = __builtin_amdgcn_udot4(d[dw], wt, x, false);
        });
        p |= ((S * S + x) >> LOG2) << (8 * ch);
      });
      vcol_step<S>(vc[k], p, rr, y, y_own, owned, dcol[k], gm, kk, ksize, top0, bot0, dimg);
    });
  };
  // Prefetch (S = 2 only): row rr + 1 is requested before row rr is processed -- two register sets, the inner loop
  // unrolled by 2 so that they alternate.  Measured on a 50 000^2 level: 448-pixel windows 3.29 -> 2.99 ms, but 1792
  // 2.27 -> 2.48 (the registers cost occupancy) and 896 +-0; a 3-deep ring was slower everywhere.
#ifndef HIPAC_PL_PREFETCH
#define HIPAC_PL_PREFETCH -1  // -1: by window size, 0 / 1: force
#endif
  constexpr bool PF = HIPAC_PL_PREFETCH < 0 ? S == 2 : HIPAC_PL_PREFETCH != 0;
  unsigned dbuf[PF ? 2 : 1][NDW];
  // one row: BUF = register set holding it (compile-time, so the two sets never need dynamic indexing)
  auto row_body = [&](int rb, int i, auto BUF) {
    constexpr int buf = decltype(BUF)::value;
    const int rr = rb + i;
    unsigned(&d)[NDW] = dbuf[buf];
#if HIPAC_PL_ABL & 4
#pragma unroll
    for (int k = 0; k < NDW; ++k) d[k] = 0x01020304u * (unsigned)(rr + k + lane);
#else
    if constexpr (PF) {
      // no request is in flight across the edge batch (its registers would add to the batch's peak): a block's first
      // row is requested at the top of the block
      if (rr + 1 < OWN + S && i + 1 < kEdgeRows) load_row(rr + 1, dbuf[PF ? 1 - buf : 0]);
    } else {
      load_row(rr, d);
    }
#endif
#if HIPAC_PL_ABL & 2
#pragma unroll
    for (int k = 0; k < NDW; ++k) gsum += d[k];
#else
    process_row(rr, d);
    if constexpr (EDGE && !(HIPAC_PL_ABL & 1)) {
      if (is_edge) {  // park the raw cover of this row
        unsigned* dst = raw + (i * kEdgeSlots + eslot) * NDW;
#pragma unroll
        for (int k = 0; k < NDW; ++k) dst[k] = d[k];
      }
    }
#endif
  };
  static_assert((OWN + S) % 2 == 0 && kEdgeRows % 2 == 0, "rows are walked in pairs");
  for (int rb = 0; rb < OWN + S; rb += kEdgeRows) {
    if constexpr (PF) {
      if constexpr (!(HIPAC_PL_ABL & 4)) load_row(rb, dbuf[0]);
#pragma unroll 1
      for (int i = 0; i < kEdgeRows; i += 2) {
        if (rb + i >= OWN + S) break;
        row_body(rb, i, std::integral_constant<int, 0>{});
        row_body(rb, i + 1, std::integral_constant<int, 1>{});
      }
    } else {
#pragma unroll 1
      for (int i = 0; i < kEdgeRows; ++i) {
        if (rb + i >= OWN + S) break;
        row_body(rb, i, std::integral_constant<int, 0>{});
      }
    }
#if !(HIPAC_PL_ABL & 3)
    if constexpr (EDGE) {
      __builtin_amdgcn_wave_barrier();
      {  // the wave's kEdgeRows x kEdgeSlots edge pixels, one per lane (unused slots compute on stale bytes; never read)
        const int e = lane & (kEdgeSlots - 1);
        unsigned dd[NDW];
        const unsigned* src = raw + lane * NDW;  // = (row lane / kEdgeSlots, slot e)
#pragma unroll
        for (int k = 0; k < NDW; ++k) dd[k] = src[k];
        const unsigned px = e < 4 ? edge_pixel<S, 0>(dd, kk) : edge_pixel<S, 8 - 3 * S / 2>(dd, kk + 223 * ksize);
        epx[lane] = px;
      }
      __builtin_amdgcn_wave_barrier();
      if (is_edge) {  // vertical step of this lane's edge column over the batch's rows
#if HIPAC_PL_VE_LDS
        VCol ve = *ve_home;
#endif
        for (int i = 0; i < kEdgeRows; ++i) {
          const int rr = rb + i;
          if (rr >= OWN + S) break;
          vcol_step<S>(ve, epx[i * kEdgeSlots + eslot], rr, y_own - S / 2 + rr, y_own, is_owned(rr), ecol, gm, kk, ksize, top0,
                       bot0, dimg);
        }
#if HIPAC_PL_VE_LDS
        *ve_home = ve;
#endif
      }
      __builtin_amdgcn_wave_barrier();
    }
#endif
  }
  // cell sums: segmented reduction over the lanes of a wave that share a cell column
  const int cx = live ? h / GPC : -1 - lane;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned up = __shfl_up(gsum, off, 64);
    const int cup = __shfl_up(cx, off, 64);
    if (lane >= off && cup == cx) gsum += up;
  }
  const int cnext = __shfl_down(cx, 1, 64);
  if (live && (lane == 63 || cnext != cx)) atomicAdd(&cells[(y_own / kLat) * gm.NCX + cx], gsum);
}

#ifndef HIPAC_PL_WAVES
#define HIPAC_PL_WAVES 0
#endif
template <int S>
__global__ __launch_bounds__(256)
#if HIPAC_PL_WAVES
__attribute__((amdgpu_waves_per_eu(HIPAC_PL_WAVES, HIPAC_PL_WAVES)))
#endif
void planes_kernel(const uint8_t* __restrict__ level, int W, int H, long long pitch,
                                                     PlaneGeom gm, const int* __restrict__ bounds,
                                                     const int* __restrict__ kk, int ksize,
                                                     unsigned* __restrict__ dimg, unsigned* __restrict__ cells,
                                                     int n_edge_waves) {
  constexpr int GPC = kLat / 8;
  constexpr int NDW = S == 8 ? 12 : (S == 4 ? 10 : 8);
  constexpr int LDS_WAVE = kEdgeRows * kEdgeSlots * (NDW + 1) + kEdgeSlots * 12;  // dwords: parked covers, edge pixels, edge column states
  __shared__ unsigned lds_edge[4 * LDS_WAVE];
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave-uniform by construction
  unsigned* const lds_wave = lds_edge + (threadIdx.x >> 6) * LDS_WAVE;
  if (n_edge_waves < 0) {  // natural order: every wave carries its 4-6 edge groups (measured faster, see the launch)
    const int h = wave * 64 + lane;
#if HIPAC_PL_ABL & 1
    planes_strip<S, false>(level, W, H, pitch, gm, bounds, kk, ksize, dimg, cells, h, h < gm.NGRP, lds_wave);  // no edge columns
    return;
#endif
    planes_strip<S, true>(level, W, H, pitch, gm, bounds, kk, ksize, dimg, cells, h, h < gm.NGRP, lds_wave);
  } else if (__builtin_amdgcn_readfirstlane(wave) < n_edge_waves) {
    const int e = wave * 64 + lane;                       // edge group index: (window column, left | right)
    const int h = (e >> 1) * GPC + ((e & 1) ? GPC - 1 : 0);
    planes_strip<S, true>(level, W, H, pitch, gm, bounds, kk, ksize, dimg, cells, h, h < gm.NGRP, lds_wave);
  } else {
    const int j = (wave - n_edge_waves) * 64 + lane;      // interior group index: 26 of every 28
    const int h = (j / (GPC - 2)) * GPC + 1 + j % (GPC - 2);
    planes_strip<S, false>(level, W, H, pitch, gm, bounds, kk, ksize, dimg, cells, h, h < gm.NGRP, lds_wave);
  }
}

// K3: window sums from cell sums; keep = sum <= 240*3*P*P
__global__ void window_stats_kernel(const unsigned* __restrict__ cells, PlaneGeom gm, const int* __restrict__ xy,
                                    int n, unsigned* __restrict__ sums, unsigned char* __restrict__ keep,
                                    unsigned threshold) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int i = xy[2 * w] / kLat, iy = xy[2 * w + 1] / kLat;
  unsigned s = 0;
  for (int a = 0; a < gm.s; ++a)
    for (int b = 0; b < gm.s; ++b) s += cells[(iy + a) * gm.NCX + i + b];
  if (sums) sums[w] = s;
  if (keep) keep[w] = s <= threshold ? 1 : 0;
}

// K4: gather windows from D.  block = (row strip of 8, window); thread item = 4 output pixels.
__global__ __launch_bounds__(256) void gather_kernel(const unsigned* __restrict__ dimg, PlaneGeom gm,
                                                     const int* __restrict__ xy, unsigned char* __restrict__ out) {
  const int w = blockIdx.y;
  const int i = xy[2 * w] / kLat, iy = xy[2 * w + 1] / kLat;
  const int gx0 = gm.ng * i, gy0 = gm.ng * iy;
  for (int item = threadIdx.x; item < 8 * 56; item += 256) {
    const int jy = blockIdx.x * 8 + item / 56, q = item % 56;
    const int rr = jy == 0 ? gm.GY + iy : (jy == 223 ? gm.GY + gm.NY + iy : gy0 + jy);
    const unsigned* drow = dimg + (long long)rr * gm.HW;
    unsigned p[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int jx = 4 * q + e;
      const int cc = jx == 0 ? gm.G + i : (jx == 223 ? gm.G + gm.NX + i : gx0 + jx);
      p[e] = drow[cc];
    }
    // 4 RGBX pixels -> 12 bytes RGBRGBRGBRGB
    const unsigned o0 = (p[0] & 0x00ffffffu) | (p[1] << 24);
    const unsigned o1 = ((p[1] >> 8) & 0x0000ffffu) | (p[2] << 16);
    const unsigned o2 = ((p[2] >> 16) & 0x000000ffu) | (p[3] << 8);
    unsigned* dst = reinterpret_cast<unsigned*>(out + ((size_t)w * kLat + jy) * (kLat * 3) + q * 12);
    dst[0] = o0;
    dst[1] = o1;
    dst[2] = o2;
  }
}

// mask -> per-cell "any pixel > 0" flags (cells of 224 x 224, zero outside the mask).  One workgroup per
// cell: thread -> (16-byte piece of the cell's 14, row group of 18); rows are OR-ed as 16-byte vectors
// (`vec` = pointer and pitch allow it), the piece cut by the level's right edge byte by byte.
__global__ __launch_bounds__(256) void mask_cells_kernel(const uint8_t* __restrict__ mask, int W, int H,
                                                         long long pitch, int ncx, int vec,
                                                         unsigned char* __restrict__ cellany) {
  __shared__ int any_s;
  const int cx = blockIdx.x, cy = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) any_s = 0;
  __syncthreads();
  const int x0 = cx * kLat, y0 = cy * kLat;
  const int cols = min(kLat, W - x0), rows = min(kLat, H - y0);
  unsigned acc = 0;
  constexpr int PIECES = kLat / 16;  // 14
  const int pc = tid % PIECES, rg = tid / PIECES;  // 18 row groups (252 threads)
  if (rg < 18 && cols > 0 && rows > 0) {
    const int c0 = pc * 16;
    const uint8_t* base = mask + (long long)y0 * pitch + x0 + c0;
    if (vec && c0 + 16 <= cols) {
      u32x4 v = {0u, 0u, 0u, 0u};
      for (int ry = rg; ry < rows; ry += 18) v |= *reinterpret_cast<const u32x4*>(base + (long long)ry * pitch);
      acc = v[0] | v[1] | v[2] | v[3];
    } else if (c0 < cols) {
      const int nb = min(16, cols - c0);
      for (int ry = rg; ry < rows; ry += 18)
        for (int k = 0; k < nb; ++k) acc |= base[(long long)ry * pitch + k];
    }
  }
  if (acc) atomicOr(&any_s, 1);
  __syncthreads();
  if (tid == 0) cellany[cy * ncx + cx] = (unsigned char)any_s;
}

__global__ void window_labels_cells_kernel(const unsigned char* __restrict__ cellany, int ncx, int ncy, int s,
                                           const int* __restrict__ xy, int n, unsigned char* __restrict__ labels) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n) return;
  const int i = xy[2 * w] / kLat, iy = xy[2 * w + 1] / kLat;
  unsigned char any = 0;
  for (int a = 0; a < s; ++a)
    for (int b = 0; b < s; ++b)
      if (iy + a < ncy && i + b < ncx) any |= cellany[(iy + a) * ncx + i + b];
  labels[w] = any ? 1 : 0;
}

}  // namespace hipac

using namespace hipac;

extern "C" {

static int check_planes_args(int W, int H, int P) {
  HIPAC_REQUIRE(W > 0 && H > 0, HIPAC_EINVAL, "level planes: bad geometry");
  HIPAC_REQUIRE(P == 448 || P == 896 || P == 1792, HIPAC_EINVAL, "level planes: P %d (448, 896 or 1792)", P);
  return 0;
}

int hipac_level_planes_sizes(int W, int H, int P, size_t* himg_bytes, size_t* dimg_bytes, size_t* cell_bytes) {
  int rc = check_planes_args(W, H, P);
  if (rc) return rc;
  const PlaneGeom g = make_geom(W, H, P);
  if (himg_bytes) *himg_bytes = (size_t)g.HROWS * g.HW * 4;
  if (dimg_bytes) *dimg_bytes = (size_t)g.DROWS * g.HW * 4;
  if (cell_bytes) *cell_bytes = (size_t)g.NCX * g.NCY * 4;
  return 0;
}

int hipac_level_build_planes(const uint8_t* level, int W, int H, int64_t pitch, int P, const int32_t* coeff_bounds,
                             const int32_t* coeff_kk, int ksize, void* himg, void* dimg, uint32_t* cells,
                             void* stream) {
  int rc = check_planes_args(W, H, P);
  if (rc) return rc;
  HIPAC_REQUIRE(level && coeff_bounds && coeff_kk && himg && dimg && cells, HIPAC_EINVAL,
                "level_build_planes: null argument");
  HIPAC_REQUIRE(pitch >= (int64_t)W * 3 && pitch % 4 == 0 && ((uintptr_t)level & 3) == 0, HIPAC_EINVAL,
                "level_build_planes: level base and pitch must be 4-byte aligned (RGB, 3 bytes per pixel)");
  const PlaneGeom g = make_geom(W, H, P);
  HIPAC_REQUIRE(ksize == 2 * g.s + 1, HIPAC_EINVAL, "level_build_planes: ksize %d != %d", ksize, 2 * g.s + 1);
  {
    // the kernels hard-wire Pillow's tap geometry for integer scales; re-derive it on the host and refuse
    // to run if it ever differs (bounds of columns 0, 1 and 223)
    std::vector<int32_t> b(2 * 224), k((size_t)224 * ksize);
    HIPAC_REQUIRE(hipac_resample_coeffs(P, 224, b.data(), k.data(), ksize) == ksize, HIPAC_EINVAL,
                  "level_build_planes: coefficient table");
    const int s_ = g.s;
    bool ok = b[0] == 0 && b[1] == 3 * s_ / 2 && b[2 * 223] == P - 3 * s_ / 2 && b[2 * 223 + 1] == 3 * s_ / 2;
    for (int j = 1; j < 223 && ok; ++j) {
      ok = b[2 * j] == s_ * j - s_ / 2 && b[2 * j + 1] == 2 * s_;
      for (int t = 0; t < 2 * s_ && ok; ++t) {
        const int wt = t < s_ ? 2 * t + 1 : 4 * s_ - 1 - 2 * t;
        ok = k[(size_t)j * ksize + t] == wt * ((1 << 22) / (2 * s_ * s_));
      }
    }
    HIPAC_REQUIRE(ok, HIPAC_EUNSUPPORTED, "level_build_planes: resampling table does not have the expected structure");
  }
  hipStream_t s = (hipStream_t)stream;
  HIPAC_CHECK_HIP(hipMemsetAsync(cells, 0, (size_t)g.NCX * g.NCY * 4, s));
  unsigned* hi = (unsigned*)himg;
  unsigned* di = (unsigned*)dimg;
  const char* env = getenv("HIPAC_PLANES_FUSED");
  if (!env || atoi(env) != 0) {
    // fused horizontal + vertical pass (the H image is not touched)
    // HIPAC_PLANES_SPLIT=1 deals the edge groups (h mod 28 in {0, 27}) to waves of their own; measured slower
    // (50 000^2: 2.44 vs 2.46 ms at P = 1792, 2.55 vs 2.18 at 896, 3.22 vs 2.68 at 448: the edge waves' loads
    // do not coalesce and they finish last), so the natural order is the default
    const char* env_split = getenv("HIPAC_PLANES_SPLIT");
    const bool split = env_split && atoi(env_split) != 0;
    const int n_cols28 = (g.NGRP + 27) / 28;                       // blocks of 28 groups
    const int n_edge_waves = split ? (2 * n_cols28 + 63) / 64 : -1;
    const int n_int_waves = (26 * n_cols28 + 63) / 64;
    dim3 gf(split ? (n_edge_waves + n_int_waves + 3) / 4 : (g.NGRP + 255) / 256, (g.HROWS + 111) / 112);
    switch (g.s) {
      case 8: hipLaunchKernelGGL((planes_kernel<8>), gf, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds, coeff_kk, ksize, di, cells, n_edge_waves); break;
      case 4: hipLaunchKernelGGL((planes_kernel<4>), gf, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds, coeff_kk, ksize, di, cells, n_edge_waves); break;
      default: hipLaunchKernelGGL((planes_kernel<2>), gf, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds, coeff_kk, ksize, di, cells, n_edge_waves);
    }
    HIPAC_CHECK_HIP(hipGetLastError());
    return 0;
  }
  dim3 g1((g.NGRP + 255) / 256, (g.HROWS + 15) / 16);
  dim3 g2((g.HW + 255) / 256, g.DROWS);
  switch (g.s) {
    case 8:
      hipLaunchKernelGGL((hpass_kernel<8>), g1, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds,
                         coeff_kk, ksize, hi, cells);
      hipLaunchKernelGGL((vpass_kernel<8>), g2, dim3(256), 0, s, hi, g, coeff_bounds, coeff_kk, ksize, di);
      break;
    case 4:
      hipLaunchKernelGGL((hpass_kernel<4>), g1, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds,
                         coeff_kk, ksize, hi, cells);
      hipLaunchKernelGGL((vpass_kernel<4>), g2, dim3(256), 0, s, hi, g, coeff_bounds, coeff_kk, ksize, di);
      break;
    default:
      hipLaunchKernelGGL((hpass_kernel<2>), g1, dim3(256), 0, s, level, W, H, (long long)pitch, g, coeff_bounds,
                         coeff_kk, ksize, hi, cells);
      hipLaunchKernelGGL((vpass_kernel<2>), g2, dim3(256), 0, s, hi, g, coeff_bounds, coeff_kk, ksize, di);
  }
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipac_level_window_stats(const uint32_t* cells, int W, int H, int P, const int32_t* xy, int n, uint32_t* sums,
                             uint8_t* keep, void* stream) {
  int rc = check_planes_args(W, H, P);
  if (rc) return rc;
  HIPAC_REQUIRE(cells && xy && n >= 0, HIPAC_EINVAL, "level_window_stats: bad argument");
  if (n == 0) return 0;
  const PlaneGeom g = make_geom(W, H, P);
  const unsigned thr = 240u * 3u * (unsigned)P * (unsigned)P;
  hipLaunchKernelGGL(window_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cells, g, xy, n,
                     sums, keep, thr);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipac_level_gather(const void* dimg, int W, int H, int P, const int32_t* xy, int n, uint8_t* out,
                       void* stream) {
  int rc = check_planes_args(W, H, P);
  if (rc) return rc;
  HIPAC_REQUIRE(dimg && xy && out && n >= 0, HIPAC_EINVAL, "level_gather: bad argument");
  if (n == 0) return 0;
  const PlaneGeom g = make_geom(W, H, P);
  for (int w0 = 0; w0 < n; w0 += 65535) {  // gridDim.y limit
    const int nw = n - w0 < 65535 ? n - w0 : 65535;
    hipLaunchKernelGGL(gather_kernel, dim3(kLat / 8, nw), dim3(256), 0, (hipStream_t)stream, (const unsigned*)dimg, g,
                       xy + 2 * (size_t)w0, out + (size_t)w0 * kLat * kLat * 3);
  }
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

extern "C" {

/* Cell form of hipac_window_labels for lattice windows: one pass over the mask, then s x s
 * cell flags per window.  cellany: uint8[ceil(H/224)][ceil(W/224)] (caller-owned). */
int hipac_mask_cells(const uint8_t* mask, int W, int H, int64_t pitch, uint8_t* cellany, void* stream) {
  HIPAC_REQUIRE(mask && cellany && W > 0 && H > 0 && pitch >= W, HIPAC_EINVAL, "mask_cells: bad argument");
  const int ncx = (W + kLat - 1) / kLat, ncy = (H + kLat - 1) / kLat;
  const int vec = (((uintptr_t)mask | (uintptr_t)pitch) & 15) == 0;
  hipLaunchKernelGGL(mask_cells_kernel, dim3(ncx, ncy), dim3(256), 0, (hipStream_t)stream, mask, W, H,
                     (long long)pitch, ncx, vec, cellany);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

int hipac_window_labels_cells(const uint8_t* cellany, int W, int H, int P, const int32_t* xy, int n,
                              uint8_t* labels, void* stream) {
  HIPAC_REQUIRE(cellany && xy && labels && n >= 0 && W > 0 && H > 0, HIPAC_EINVAL, "window_labels_cells: bad argument");
  HIPAC_REQUIRE(P >= kLat && P % kLat == 0, HIPAC_EINVAL, "window_labels_cells: P %d", P);
  if (n == 0) return 0;
  const int ncx = (W + kLat - 1) / kLat, ncy = (H + kLat - 1) / kLat;
  hipLaunchKernelGGL(window_labels_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cellany,
                     ncx, ncy, P / kLat, xy, n, labels);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
