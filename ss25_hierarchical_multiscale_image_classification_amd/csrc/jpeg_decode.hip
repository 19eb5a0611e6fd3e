// Baseline-JPEG tile decoder on the device: how a tiled pyramidal TIFF (the CAMELYON16 container; the reference opens it with
// openslide.OpenSlide, src/main.py:650, and reads windows with read_region, :693) gets into HBM without the host decoding a
// pixel.  Decoding the tiles on host threads (Pillow / libjpeg-turbo, 16 threads) delivers 0.18 Gpx/s -- a 100 000^2 slide
// takes 74 s to load and 0.53 s to scan; the tiles are independent JPEG streams, tens of thousands per level, so the
// sequential part of JPEG (the Huffman bit stream) parallelises over TILES:
//   * host (C++): parse each tile's headers (SOI, DQT, DHT, SOF0, DRI, APP14, SOS) on top of the directory's JPEGTables,
//     derive the Huffman lookup tables exactly as libjpeg's jdhuff.c does, deduplicate the (tables, frame) configurations
//     (one per level in practice) and hand the kernels tile descriptors; the compressed bytes are read on the device;
//   * jpeg_huffman_kernel: one LANE per tile walks its entropy-coded segment (byte un-stuffing, 8-bit look-ahead table,
//     slow path by maxcode, DC prediction, restart markers) and writes de-zigzagged int16 coefficients;
//   * jpeg_idct_kernel: one lane per 8x8 block: dequantise + libjpeg's jidctint.c (the "ISLOW" integer IDCT, libjpeg's and
//     Pillow's default), range limit -> uint8 sample planes;
//   * jpeg_store_kernel: one lane per pixel: h2v2 / h2v1 "fancy" (triangle) chroma upsampling of jdsample.c with its alternating
//     rounding and edge replication, YCbCr -> RGB with jdcolor.c's fixed-point tables, straight into the level image in HBM.
// Everything is integer arithmetic restated from libjpeg and checked against Pillow's decoder bit for bit
// (tests/test_gpu_jpeg.py; the formulas were first pinned by a pure-Python restatement on the CPU).  Supported: baseline
// sequential, 8 bit, three components, 4:2:0, 4:2:2 or 4:4:4, Huffman table ids 0 / 1, tile sides that are multiples of 16;
// anything else is reported per tile (status 1) and the caller decodes that tile on the host as before.  Bound: the Huffman
// kernel's serial bit walk (~0.6 us per symbol at one stream per wave: a chain of ~150 dependent instructions with an LDS
// look-up in it; 80 % of a load, its coefficient stores are 5 % of that): latency, not HBM.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace hipac {

struct DevHuff {
  unsigned short look[256];  // codes of <= 8 bits: (length << 8) | symbol, 0 = longer
  int maxcode[18];           // largest code of length k (-1 if none); [17] = sentinel
  int valoff[17];            // huffval index of the first code of length k, minus that code
  unsigned char vals[256];
};
struct JpegCfg {
  DevHuff dc[2], ac[2];
  unsigned short q[3][64];   // natural order, per component
  int hs[3], vs[3];          // sampling factors
  int td[3], ta[3];          // table ids per component
  int bw[3], bh[3];          // blocks per component plane (width, height)
  int blk_off[3];            // first block of the component in a tile's coefficient array
  int plane_off[3];          // byte offset of the component's plane in a tile's sample area
  int mcus_x, mcus_y, dri, convert;  // convert: 1 = YCbCr -> RGB
  int n_blocks, plane_bytes;
  int mcu_blocks;            // blocks per MCU (6 for 4:2:0, 3 for 4:4:4)
  unsigned char mcu_ci[8], mcu_by[8], mcu_bx[8];  // component and position inside the MCU of its k-th block
};
struct TileDesc {
  long long off;  // first byte of the entropy-coded segment in the file
  long long end;  // one past the last byte of the tile's stream
  int cfg, x, y, lvl;
};
struct LevelDesc {
  unsigned char* pixels;
  long long pitch;
  int W, H, tile_w, tile_h;
};

static const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                          41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                          30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
__constant__ unsigned char d_zigzag[64];

// ---- Huffman decoding, one LANE per tile ------------------------------------------------------------------------------------
// The 64 lanes of a wave walk 64 different bit streams in lock-step, so the loop is FLAT: one iteration = one Huffman symbol
// of every lane, whatever block / component / MCU that lane is in (its position is lane state, not loop structure) -- a
// loop nest over MCUs and blocks would make every lane wait, block by block, for the lane with the most coefficients.
// Bits: a 64-bit buffer refilled 32 bits at a time when no 0xFF is among the next four bytes (the usual case), byte by byte
// otherwise (0xFF00 -> 0xFF; a marker feeds zeros and is not stepped over: libjpeg's "insufficient data" rule).
// The Huffman tables of the workgroup's first tile sit in LDS (every tile of a level shares them in practice; a lane whose
// tile uses another configuration reads its own from global memory), and so does the zigzag order.
__device__ __forceinline__ int huff_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// `lanes` tiles per wave: with few tiles (a slide has thousands, the chip holds 16 384 waves at 16 per SIMD) a wave walks FEWER
// streams -- its lanes then rarely sit on different sides of a branch, and the SIMDs that would idle hold the other waves,
// whose latencies overlap.  64 only when there are more tiles than the chip has wave slots.
__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const unsigned char* __restrict__ file, const TileDesc* __restrict__ tiles,
                                                         const JpegCfg* __restrict__ cfgs, int n_tiles, short* __restrict__ coef,
                                                         long long coef_stride, int lanes) {
  __shared__ DevHuff sh[4];  // dc[0], dc[1], ac[0], ac[1] of the first tile's configuration
  __shared__ unsigned char zz[64];
  const int t0 = blockIdx.x * lanes;
  const int cfg0 = tiles[t0].cfg;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&cfgs[cfg0].dc[0]);  // dc[2] and ac[2] are contiguous
    unsigned* dst = reinterpret_cast<unsigned*>(&sh[0]);
    for (int i = threadIdx.x; i < (int)(4 * sizeof(DevHuff) / 4); i += 64) dst[i] = src[i];
    zz[threadIdx.x] = d_zigzag[threadIdx.x];
  }
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if ((int)threadIdx.x >= lanes || t >= n_tiles) return;
  const TileDesc td = tiles[t];
  const JpegCfg& c = cfgs[td.cfg];
  const bool shared_tables = td.cfg == cfg0;
  short* const out = coef + (size_t)t * coef_stride;
  const unsigned char* p = file + td.off;
  const unsigned char* const end = file + td.end;
  unsigned long long buf = 0;
  int n = 0;
  const unsigned char* pw = p;  // where the prefetched dword `wq` was read: the stream is fetched one refill ahead
  unsigned wq;
  __builtin_memcpy(&wq, p, 4);
  int pred0 = 0, pred1 = 0, pred2 = 0;
  const int dri = c.dri, mcus_x = c.mcus_x, n_mcus = c.mcus_x * c.mcus_y, mcu_blocks = c.mcu_blocks;
  // everything the walk needs from the configuration lives in registers: a load from it would sit in the dependent chain
  // of every symbol (one wave per SIMD at best: nothing hides a latency here)
  unsigned long long mcu_pack = 0;  // per block of the MCU: component (2 bits), row (1), column (1)
  for (int b = 0; b < mcu_blocks; ++b)
    mcu_pack |= (unsigned long long)(c.mcu_ci[b] | (c.mcu_by[b] << 2) | (c.mcu_bx[b] << 3)) << (4 * b);
  unsigned tab_pack = 0;  // per component: DC table id (bit 0), AC table id (bit 1)
  for (int q = 0; q < 3; ++q) tab_pack |= (unsigned)(c.td[q] | (c.ta[q] << 1)) << (2 * q);
  const int hs0 = c.hs[0], vs0 = c.vs[0], bw0 = c.bw[0], bw1 = c.bw[1], bo1 = c.blk_off[1], bo2 = c.blk_off[2];
  auto block_ptr = [&](int bi_, int mx_, int my_, int& ci_) -> short* {
    const int e = (int)(mcu_pack >> (4 * bi_)) & 15;
    ci_ = e & 3;
    const int by = (e >> 2) & 1, bx = (e >> 3) & 1;
    // component 0 has (hs0, vs0) blocks per MCU, the chroma components one
    const size_t b = ci_ == 0 ? (size_t)(my_ * vs0 + by) * bw0 + mx_ * hs0 + bx : (size_t)(ci_ == 1 ? bo1 : bo2) + (size_t)my_ * bw1 + mx_;
    return out + b * 64;
  };
  int m = 0, mx = 0, my = 0, bi = 0, left = dri, k = 0;
  int ci;
  short* blk = block_ptr(0, 0, 0, ci);
  while (m < n_mcus) {
    // ---- at least 32 bits (a code is <= 16 bits, its extra bits <= 15)
    if (n < 32) {
      bool fast = false;
      if (p + 4 <= end) {
        if (pw != p) {  // (after the byte-wise path) the dword in hand is not the one at p
          __builtin_memcpy(&wq, p, 4);
          pw = p;
        }
        const unsigned w = wq;
        const unsigned x = ~w;  // a byte of w is 0xFF  <=>  that byte of x is 0
        if (!((x - 0x01010101u) & ~x & 0x80808080u)) {
          buf = (buf << 32) | (unsigned long long)__builtin_bswap32(w);
          n += 32;
          p += 4;
          pw = p;
          __builtin_memcpy(&wq, p, 4);  // requested now, needed ~5 symbols later (the file image has slack behind its end)
          fast = true;
        }
      }
      if (!fast) {
        while (n <= 48) {
          unsigned b = p < end ? *p : 0u;
          if (b == 0xFFu) {
            const unsigned nb = p + 1 < end ? p[1] : 0xD9u;
            if (nb == 0) p += 2;
            else b = 0;  // a marker: stay in front of it
          } else if (p < end) {
            ++p;
          }
          buf = (buf << 8) | b;
          n += 8;
        }
      }
    }
    // ---- one symbol
    const bool is_dc = k == 0;
    const int tid_dc = (int)(tab_pack >> (2 * ci)) & 1, tid_ac = (int)(tab_pack >> (2 * ci + 1)) & 1;
    const DevHuff& tb = shared_tables ? sh[is_dc ? tid_dc : 2 + tid_ac] : (is_dc ? c.dc[tid_dc] : c.ac[tid_ac]);
    int sym;
    {
      const unsigned lk = tb.look[(unsigned)(buf >> (n - 8)) & 255u];
      if (lk) {
        n -= (int)(lk >> 8);
        sym = (int)(lk & 255u);
      } else {
        int l = 9;
        int code = (int)((buf >> (n - 9)) & 511u);
        while (l <= 16 && code > tb.maxcode[l]) {
          ++l;
          code = (int)((buf >> (n - l)) & ((1u << l) - 1u));
        }
        if (l > 16) {
          n -= 16;
          sym = 0;  // corrupt data: libjpeg warns and takes 0
        } else {
          n -= l;
          sym = tb.vals[(code + tb.valoff[l]) & 255];
        }
      }
    }
    const int sz = sym & 15;
    int v = 0;
    if (sz) {
      v = huff_extend((int)((buf >> (n - sz)) & ((1u << sz) - 1u)), sz);
      n -= sz;
    }
    if (is_dc) {
      int pr = ci == 0 ? pred0 : (ci == 1 ? pred1 : pred2);
      pr += v;
      if (ci == 0) pred0 = pr;
      else if (ci == 1) pred1 = pr;
      else pred2 = pr;
      blk[0] = (short)pr;
      k = 1;
    } else {
      const int r = sym >> 4;
      if (sz == 0) {
        k = r == 15 ? k + 16 : 64;
      } else {
        k += r;
        if (k < 64) blk[zz[k]] = (short)v;
        ++k;
      }
    }
    if (k >= 64) {  // next block of this lane's tile
      k = 0;
      if (++bi == mcu_blocks) {
        bi = 0;
        ++m;
        if (++mx == mcus_x) mx = 0, ++my;
        if (dri && --left == 0 && m < n_mcus) {  // restart interval: byte-align, step over RSTn, reset the predictions
          n = 0;
          buf = 0;
          if (p + 1 < end && p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7) p += 2;
          pred0 = pred1 = pred2 = 0;
          left = dri;
        }
      }
      blk = block_ptr(bi, mx, my, ci);
    }
  }
}

// ---- jidctint.c (ISLOW): CONST_BITS 13, PASS1_BITS 2 -----------------------------------------------------------------
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ void idct_1d(const int (&v)[8], int shift, int (&o)[8]) {
  int z2 = v[2], z3 = v[6];
  int z1 = (z2 + z3) * 4433;
  int tmp2 = z1 + z3 * (-15137);
  int tmp3 = z1 + z2 * 6270;
  z2 = v[0], z3 = v[4];
  int tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = v[7], tmp1 = v[5], tmp2 = v[3], tmp3 = v[1];
  z1 = tmp0 + tmp3, z2 = tmp1 + tmp2, z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * 9633;
  tmp0 *= 2446, tmp1 *= 16819, tmp2 *= 25172, tmp3 *= 12299;
  z1 *= -7373, z2 *= -20995, z3 *= -16069, z4 *= -3196;
  z3 += z5, z4 += z5;
  tmp0 += z1 + z3, tmp1 += z2 + z4, tmp2 += z2 + z3, tmp3 += z1 + z4;
  o[0] = descale(tmp10 + tmp3, shift), o[7] = descale(tmp10 - tmp3, shift);
  o[1] = descale(tmp11 + tmp2, shift), o[6] = descale(tmp11 - tmp2, shift);
  o[2] = descale(tmp12 + tmp1, shift), o[5] = descale(tmp12 - tmp1, shift);
  o[3] = descale(tmp13 + tmp0, shift), o[4] = descale(tmp13 - tmp0, shift);
}
__device__ __forceinline__ unsigned char range_limit_idct(int x) {  // libjpeg's range_limit table behind RANGE_MASK (1023), + CENTERJSAMPLE
  const int i = x & 1023;
  return (unsigned char)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const short* __restrict__ coef, long long coef_stride, const TileDesc* __restrict__ tiles,
                                                        const JpegCfg* __restrict__ cfgs, int n_tiles, int max_blocks,
                                                        unsigned char* __restrict__ planes, long long plane_stride) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const int t = (int)(g / max_blocks), b = (int)(g - (long long)t * max_blocks);
  if (t >= n_tiles) return;
  const JpegCfg& c = cfgs[tiles[t].cfg];
  if (b >= c.n_blocks) return;
  const int ci = b >= c.blk_off[2] ? 2 : (b >= c.blk_off[1] ? 1 : 0);
  const int bi = b - c.blk_off[ci], by = bi / c.bw[ci], bx = bi - by * c.bw[ci];
  const short* in = coef + (size_t)t * coef_stride + (size_t)b * 64;
  const unsigned short* q = c.q[ci];
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int v[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (int)in[r * 8 + col] * (int)q[r * 8 + col];
    idct_1d(v, 11, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
  }
  unsigned char* dst = planes + (size_t)t * plane_stride + c.plane_off[ci] + ((size_t)by * 8) * (c.bw[ci] * 8) + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int v[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ws[r * 8 + k];
    idct_1d(v, 18, o);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) lo |= (unsigned)range_limit_idct(o[k]) << (8 * k), hi |= (unsigned)range_limit_idct(o[4 + k]) << (8 * k);
    *reinterpret_cast<u32x2*>(dst + (size_t)r * (c.bw[ci] * 8)) = u32x2{lo, hi};
  }
}

// ---- jdsample.c h2v2_fancy_upsample + jdcolor.c ycc_rgb_convert ------------------------------------------------------
__device__ __forceinline__ int fancy_h2v2(const unsigned char* __restrict__ pl, int cw, int ch, int x, int y) {
  const int r = y >> 1, cx = x >> 1;
  const int rf = (y & 1) ? (r + 1 < ch ? r + 1 : ch - 1) : (r > 0 ? r - 1 : 0);  // the next-nearest row, replicated at the edges
  const unsigned char* near = pl + (size_t)r * cw;
  const unsigned char* far = pl + (size_t)rf * cw;
  const int cs = near[cx] * 3 + far[cx];
  if (x & 1) {
    const int xn = cx + 1 < cw ? cx + 1 : cx;
    return (cs * 3 + (near[xn] * 3 + far[xn]) + 7) >> 4;
  }
  const int xl = cx > 0 ? cx - 1 : cx;
  return (cs * 3 + (near[xl] * 3 + far[xl]) + 8) >> 4;
}
// jdsample.c h2v1_fancy_upsample (4:2:2): 3/4 nearer + 1/4 further column, rounding 1 / 2 alternating, the outermost columns copied
__device__ __forceinline__ int fancy_h2v1(const unsigned char* __restrict__ pl, int cw, int x, int y) {
  const unsigned char* row = pl + (size_t)y * cw;
  const int cx = x >> 1, v = row[cx];
  if (x & 1) return cx + 1 < cw ? (v * 3 + row[cx + 1] + 2) >> 2 : v;
  return cx > 0 ? (v * 3 + row[cx - 1] + 1) >> 2 : v;
}
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ __launch_bounds__(256) void jpeg_store_kernel(const unsigned char* __restrict__ planes, long long plane_stride,
                                                         const TileDesc* __restrict__ tiles, const JpegCfg* __restrict__ cfgs,
                                                         const LevelDesc* __restrict__ levels, int n_tiles) {
  const int t = blockIdx.y;
  const TileDesc td = tiles[t];
  const LevelDesc lv = levels[td.lvl];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lv.tile_w * lv.tile_h) return;
  const int y = p / lv.tile_w, x = p - y * lv.tile_w;
  const int gx = td.x + x, gy = td.y + y;
  if (gx >= lv.W || gy >= lv.H) return;
  const JpegCfg& c = cfgs[td.cfg];
  const unsigned char* base = planes + (size_t)t * plane_stride;
  const int yw = c.bw[0] * 8;
  const int Y = base[c.plane_off[0] + (size_t)y * yw + x];
  int cb, cr;
  if (c.hs[0] == 2 && c.vs[0] == 2) {
    const int cw = c.bw[1] * 8, ch = c.bh[1] * 8;
    cb = fancy_h2v2(base + c.plane_off[1], cw, ch, x, y);
    cr = fancy_h2v2(base + c.plane_off[2], cw, ch, x, y);
  } else if (c.hs[0] == 2) {
    const int cw = c.bw[1] * 8;
    cb = fancy_h2v1(base + c.plane_off[1], cw, x, y);
    cr = fancy_h2v1(base + c.plane_off[2], cw, x, y);
  } else {
    cb = base[c.plane_off[1] + (size_t)y * yw + x];
    cr = base[c.plane_off[2] + (size_t)y * yw + x];
  }
  int R = Y, G = cb, B = cr;
  if (c.convert) {
    const int xb = cb - 128, xr = cr - 128;
    R = clamp255(Y + ((91881 * xr + 32768) >> 16));
    G = clamp255(Y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
    B = clamp255(Y + ((116130 * xb + 32768) >> 16));
  }
  unsigned char* o = lv.pixels + (size_t)gy * lv.pitch + (size_t)gx * 3;
  o[0] = (unsigned char)R, o[1] = (unsigned char)G, o[2] = (unsigned char)B;
}

// ---- host: headers ------------------------------------------------------------------------------------------------------
struct HostTables {
  unsigned short q[4][64];
  bool q_ok[4];
  unsigned char bits[2][4][17];
  unsigned char vals[2][4][256];
  bool h_ok[2][4];
};
struct HostFrame {
  int w, h, nc, dri, adobe;
  int id[3], hs[3], vs[3], tq[3], td[3], ta[3];
  long long scan;  // offset of the entropy-coded data, -1: none seen
};

static int be16(const unsigned char* p) { return (p[0] << 8) | p[1]; }

// walks the marker segments of [p, p + n); tables accumulate in T.  Returns 0, or -1 malformed, -2 unsupported.
static int parse_stream(const unsigned char* p, long long n, HostTables& T, HostFrame& F) {
  long long i = 0;
  if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return -1;
  i = 2;
  while (i + 4 <= n) {
    if (p[i] != 0xFF) return -1;
    while (i < n && p[i] == 0xFF) ++i;  // fill bytes
    if (i >= n) return -1;
    const int m = p[i++];
    if (m == 0xD9) return 0;  // EOI (a tables-only stream)
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (i + 2 > n) return -1;
    const int L = be16(p + i);
    if (L < 2 || i + L > n) return -1;
    const unsigned char* s = p + i + 2;
    const int sl = L - 2;
    if (m == 0xDB) {
      int j = 0;
      while (j < sl) {
        const int pq = s[j] >> 4, tq = s[j] & 15;
        ++j;
        if (tq > 3 || pq > 1 || j + (pq ? 128 : 64) > sl) return -1;
        for (int k = 0; k < 64; ++k) {
          const int v = pq ? be16(s + j + 2 * k) : s[j + k];
          T.q[tq][kZigzag[k]] = (unsigned short)v;
        }
        T.q_ok[tq] = true;
        j += pq ? 128 : 64;
      }
    } else if (m == 0xC4) {
      int j = 0;
      while (j < sl) {
        const int tc = s[j] >> 4, th = s[j] & 15;
        ++j;
        if (tc > 1 || th > 3 || j + 16 > sl) return -1;
        int cnt = 0;
        T.bits[tc][th][0] = 0;
        for (int k = 1; k <= 16; ++k) cnt += (T.bits[tc][th][k] = s[j + k - 1]);
        j += 16;
        if (cnt > 256 || j + cnt > sl) return -1;
        std::memset(T.vals[tc][th], 0, 256);
        std::memcpy(T.vals[tc][th], s + j, (size_t)cnt);
        T.h_ok[tc][th] = true;
        j += cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (sl < 6) return -1;
      if (s[0] != 8) return -2;
      F.h = be16(s + 1), F.w = be16(s + 3), F.nc = s[5];
      if (F.nc != 3) return -2;
      if (sl < 6 + 3 * F.nc) return -1;
      for (int c = 0; c < 3; ++c) F.id[c] = s[6 + 3 * c], F.hs[c] = s[7 + 3 * c] >> 4, F.vs[c] = s[7 + 3 * c] & 15, F.tq[c] = s[8 + 3 * c];
    } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
      return -2;  // progressive, lossless, arithmetic ...
    } else if (m == 0xDD) {
      if (sl < 2) return -1;
      F.dri = be16(s);
    } else if (m == 0xEE) {
      if (sl >= 12 && std::memcmp(s, "Adobe", 5) == 0) F.adobe = s[11];
    } else if (m == 0xDA) {
      if (sl < 1 || s[0] != 3 || sl < 1 + 2 * 3 + 3) return -2;  // one interleaved scan of all three components
      for (int k = 0; k < 3; ++k) {
        const int cid = s[1 + 2 * k];
        int c = -1;
        for (int q = 0; q < 3; ++q)
          if (F.id[q] == cid) c = q;
        if (c != k) return -2;
        F.td[c] = s[2 + 2 * k] >> 4, F.ta[c] = s[2 + 2 * k] & 15;
      }
      if (s[7] != 0 || s[8] != 63 || s[9] != 0) return -2;  // Ss, Se, Ah/Al of a sequential scan
      F.scan = i + L;
      return 0;
    }
    i += L;
  }
  return 0;
}

static bool derive(const unsigned char* bits, const unsigned char* vals, DevHuff& d) {  // jdhuff.c jpeg_make_d_derived_tbl
  char huffsize[257];
  unsigned huffcode[257];
  int p = 0;
  for (int l = 1; l <= 16; ++l) {
    const int n = bits[l];
    if (p + n > 256) return false;
    for (int k = 0; k < n; ++k) huffsize[p++] = (char)l;
  }
  huffsize[p] = 0;
  const int numsymbols = p;
  unsigned code = 0;
  int si = huffsize[0];
  p = 0;
  while (huffsize[p]) {
    while ((int)huffsize[p] == si) huffcode[p++] = code++;
    if ((int)code >= (1 << si)) return false;  // jdhuff.c: no code may be all ones
    code <<= 1;
    ++si;
  }
  p = 0;
  for (int l = 1; l <= 16; ++l) {
    if (bits[l]) {
      d.valoff[l] = p - (int)huffcode[p];
      p += bits[l];
      d.maxcode[l] = (int)huffcode[p - 1];
    } else {
      d.maxcode[l] = -1;
      d.valoff[l] = 0;
    }
  }
  d.valoff[0] = 0;
  d.maxcode[0] = -1;
  d.maxcode[17] = 0xFFFFF;
  std::memset(d.look, 0, sizeof(d.look));
  p = 0;
  for (int l = 1; l <= 8; ++l)
    for (int k = 1; k <= bits[l]; ++k, ++p) {
      const int look = (int)huffcode[p] << (8 - l);
      for (int ctr = 0; ctr < (1 << (8 - l)); ++ctr) d.look[look + ctr] = (unsigned short)((l << 8) | vals[p]);
    }
  std::memcpy(d.vals, vals, 256);
  (void)numsymbols;
  return true;
}

}  // namespace hipac

using namespace hipac;

extern "C" {

size_t hipac_jpeg_workspace_bytes(int tile_w, int tile_h, int n_tiles) {
  if (tile_w <= 0 || tile_h <= 0 || n_tiles <= 0) return 0;
  const size_t px = (size_t)tile_w * tile_h;
  // worst case 4:4:4: three full planes; coefficients 2 bytes per sample; descriptors and configurations on top
  return (size_t)n_tiles * (px * 3 * 2 + px * 3 + 16 + sizeof(TileDesc)) + 64 * sizeof(JpegCfg) + 64 * sizeof(LevelDesc) + 4096;
}

int hipac_jpeg_decode_tiles(const uint8_t* file_host, const uint8_t* file_dev, int64_t file_bytes, const hipac_jpeg_level* levels,
                            int n_levels, const int64_t* tile_off, const int64_t* tile_len, const int32_t* tile_xyl, int n_tiles,
                            void* workspace, size_t workspace_bytes, uint8_t* status_host, void* stream) {
  HIPAC_REQUIRE(file_host && file_dev && levels && tile_off && tile_len && tile_xyl && workspace && status_host, HIPAC_EINVAL,
                "jpeg_decode_tiles: null argument");
  HIPAC_REQUIRE(n_tiles > 0 && n_tiles <= 65535 && n_levels > 0 && n_levels <= 64, HIPAC_EINVAL, "jpeg_decode_tiles: bad counts");
  int max_tw = 0, max_th = 0;
  for (int l = 0; l < n_levels; ++l) {
    const hipac_jpeg_level& L = levels[l];
    HIPAC_REQUIRE(L.pixels && L.W > 0 && L.H > 0 && L.tile_w > 0 && L.tile_h > 0 && L.pitch_bytes >= (int64_t)L.W * 3, HIPAC_EINVAL,
                  "jpeg_decode_tiles: bad geometry of level %d", l);
    max_tw = L.tile_w > max_tw ? L.tile_w : max_tw, max_th = L.tile_h > max_th ? L.tile_h : max_th;
  }
  HIPAC_REQUIRE(workspace_bytes >= hipac_jpeg_workspace_bytes(max_tw, max_th, n_tiles), HIPAC_EINVAL,
                "jpeg_decode_tiles: workspace too small (hipac_jpeg_workspace_bytes of the largest tile)");
  hipStream_t s = (hipStream_t)stream;
  std::vector<HostTables> base((size_t)n_levels);
  for (int l = 0; l < n_levels; ++l) {
    std::memset(&base[l], 0, sizeof(HostTables));
    if (levels[l].jpeg_tables && levels[l].jpeg_tables_len > 0) {
      HostFrame dummy;
      std::memset(&dummy, 0, sizeof(dummy));
      dummy.scan = -1, dummy.adobe = -1;
      HIPAC_REQUIRE(parse_stream(levels[l].jpeg_tables, levels[l].jpeg_tables_len, base[l], dummy) == 0, HIPAC_EINVAL,
                    "jpeg_decode_tiles: malformed JPEGTables of level %d", l);
    }
  }
  std::vector<JpegCfg> cfgs;
  std::vector<TileDesc> descs;
  std::vector<int> which;  // tile index of every descriptor
  // consecutive tiles of a level nearly always carry the same tables and frame: remember the last ones and their configuration
  HostTables last_T;
  HostFrame last_F;
  int last_lvl = -1, last_ci = -1;
  std::memset(&last_T, 0, sizeof(last_T));
  std::memset(&last_F, 0, sizeof(last_F));
  for (int t = 0; t < n_tiles; ++t) {
    status_host[t] = 1;
    const long long off = tile_off[t], len = tile_len[t];
    const int lvl = tile_xyl[3 * t + 2];
    HIPAC_REQUIRE(lvl >= 0 && lvl < n_levels, HIPAC_EINVAL, "jpeg_decode_tiles: tile %d: level %d", t, lvl);
    if (len <= 0) {
      status_host[t] = 2;  // a missing tile: nothing to decode, the level keeps its zeros
      continue;
    }
    HIPAC_REQUIRE(off >= 0 && off + len <= file_bytes, HIPAC_EINVAL, "jpeg_decode_tiles: tile %d lies outside the file", t);
    HIPAC_REQUIRE(tile_xyl[3 * t] >= 0 && tile_xyl[3 * t + 1] >= 0, HIPAC_EINVAL, "jpeg_decode_tiles: tile %d: negative origin", t);
    const hipac_jpeg_level& L = levels[lvl];
    const int tile_w = L.tile_w, tile_h = L.tile_h;
    if (tile_w % 16 || tile_h % 16) continue;
    HostTables T = base[lvl];
    HostFrame F;
    std::memset(&F, 0, sizeof(F));
    F.scan = -1, F.adobe = -1;
    if (parse_stream(file_host + off, len, T, F) != 0 || F.scan < 0) continue;
    if (F.w != tile_w || F.h != tile_h) continue;
    {
      HostFrame Fs = F;
      Fs.scan = 0;  // the scan offset differs per tile, the rest of the frame decides the configuration
      if (last_ci >= 0 && lvl == last_lvl && std::memcmp(&Fs, &last_F, sizeof(Fs)) == 0 && std::memcmp(&T, &last_T, sizeof(T)) == 0) {
        descs.push_back(TileDesc{off + F.scan, off + len, last_ci, tile_xyl[3 * t], tile_xyl[3 * t + 1], lvl});
        which.push_back(t);
        continue;
      }
    }
    const bool s420 = F.hs[0] == 2 && F.vs[0] == 2 && F.hs[1] == 1 && F.vs[1] == 1 && F.hs[2] == 1 && F.vs[2] == 1;
    const bool s444 = F.hs[0] == 1 && F.vs[0] == 1 && F.hs[1] == 1 && F.vs[1] == 1 && F.hs[2] == 1 && F.vs[2] == 1;
    const bool s422 = F.hs[0] == 2 && F.vs[0] == 1 && F.hs[1] == 1 && F.vs[1] == 1 && F.hs[2] == 1 && F.vs[2] == 1;
    if (!s420 && !s444 && !s422) continue;
    bool ok = true;
    JpegCfg c;
    std::memset(&c, 0, sizeof(c));
    for (int k = 0; k < 3 && ok; ++k) {
      ok = F.tq[k] <= 3 && T.q_ok[F.tq[k]] && F.td[k] <= 1 && F.ta[k] <= 1 && T.h_ok[0][F.td[k]] && T.h_ok[1][F.ta[k]];
      if (!ok) break;
      std::memcpy(c.q[k], T.q[F.tq[k]], sizeof(c.q[k]));
      c.hs[k] = F.hs[k], c.vs[k] = F.vs[k], c.td[k] = F.td[k], c.ta[k] = F.ta[k];
    }
    if (!ok) continue;
    for (int id = 0; id < 2 && ok; ++id) {
      if (T.h_ok[0][id]) ok = ok && derive(T.bits[0][id], T.vals[0][id], c.dc[id]);
      if (T.h_ok[1][id]) ok = ok && derive(T.bits[1][id], T.vals[1][id], c.ac[id]);
    }
    if (!ok) continue;
    const int hmax = F.hs[0], vmax = F.vs[0];
    c.mcus_x = tile_w / (8 * hmax), c.mcus_y = tile_h / (8 * vmax), c.dri = F.dri;
    // TIFF photometric 6 = YCbCr samples; 2 = RGB samples in the JPEG stream (an Adobe marker with transform 0 says the same)
    c.convert = (L.photometric == 6 && F.adobe != 0) ? 1 : 0;
    int blk = 0, pl = 0, mb = 0;
    for (int k = 0; k < 3; ++k) {
      c.bw[k] = c.mcus_x * F.hs[k], c.bh[k] = c.mcus_y * F.vs[k];
      c.blk_off[k] = blk, c.plane_off[k] = pl;
      blk += c.bw[k] * c.bh[k], pl += c.bw[k] * c.bh[k] * 64;
      for (int by = 0; by < F.vs[k]; ++by)
        for (int bx = 0; bx < F.hs[k]; ++bx, ++mb) c.mcu_ci[mb] = (unsigned char)k, c.mcu_by[mb] = (unsigned char)by, c.mcu_bx[mb] = (unsigned char)bx;
    }
    c.n_blocks = blk, c.plane_bytes = pl, c.mcu_blocks = mb;
    int ci = -1;
    for (size_t k = 0; k < cfgs.size(); ++k)
      if (std::memcmp(&cfgs[k], &c, sizeof(c)) == 0) ci = (int)k;
    if (ci < 0) {
      if (cfgs.size() >= 64) continue;  // more distinct table sets than planned for: the host decodes this tile
      cfgs.push_back(c);
      ci = (int)cfgs.size() - 1;
    }
    descs.push_back(TileDesc{off + F.scan, off + len, ci, tile_xyl[3 * t], tile_xyl[3 * t + 1], lvl});
    which.push_back(t);
    last_T = T, last_F = F, last_F.scan = 0, last_lvl = lvl, last_ci = ci;
  }
  const int nd = (int)descs.size();
  if (nd == 0) return 0;
  std::vector<LevelDesc> ldesc((size_t)n_levels);
  for (int l = 0; l < n_levels; ++l)
    ldesc[l] = LevelDesc{levels[l].pixels, (long long)levels[l].pitch_bytes, levels[l].W, levels[l].H, levels[l].tile_w, levels[l].tile_h};
  // workspace: [cfgs 64][levels 64][descs nd][coef nd * max_blocks * 64 int16][planes nd * max_plane]
  int max_blocks = 0, max_plane = 0;
  for (const JpegCfg& c : cfgs) {
    max_blocks = c.n_blocks > max_blocks ? c.n_blocks : max_blocks;
    max_plane = c.plane_bytes > max_plane ? c.plane_bytes : max_plane;
  }
  char* ws = (char*)workspace;
  JpegCfg* d_cfg = (JpegCfg*)ws;
  size_t o = (64 * sizeof(JpegCfg) + 255) / 256 * 256;
  LevelDesc* d_lvl = (LevelDesc*)(ws + o);
  o += (64 * sizeof(LevelDesc) + 255) / 256 * 256;
  TileDesc* d_desc = (TileDesc*)(ws + o);
  o += ((size_t)nd * sizeof(TileDesc) + 255) / 256 * 256;
  short* d_coef = (short*)(ws + o);
  const long long coef_stride = (long long)max_blocks * 64;
  o += (size_t)nd * coef_stride * 2;
  unsigned char* d_planes = (unsigned char*)(ws + o);
  const long long plane_stride = (max_plane + 15) / 16 * 16;
  o += (size_t)nd * plane_stride;
  HIPAC_REQUIRE(o <= workspace_bytes, HIPAC_EINVAL, "jpeg_decode_tiles: workspace too small");
  static bool zz_done[64] = {};
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev < 0 || !zz_done[dev]) {
      HIPAC_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(d_zigzag), kZigzag, 64));
      if (dev >= 0) zz_done[dev] = true;
    }
  }
  HIPAC_CHECK_HIP(hipMemcpyAsync(d_cfg, cfgs.data(), cfgs.size() * sizeof(JpegCfg), hipMemcpyHostToDevice, s));
  HIPAC_CHECK_HIP(hipMemcpyAsync(d_lvl, ldesc.data(), ldesc.size() * sizeof(LevelDesc), hipMemcpyHostToDevice, s));
  HIPAC_CHECK_HIP(hipMemcpyAsync(d_desc, descs.data(), (size_t)nd * sizeof(TileDesc), hipMemcpyHostToDevice, s));
  HIPAC_CHECK_HIP(hipMemsetAsync(d_coef, 0, (size_t)nd * coef_stride * 2, s));
  int lanes = (nd + 16383) / 16384;  // tiles per wave: 16 waves on every SIMD before a wave takes a second stream
  lanes = lanes < 1 ? 1 : (lanes > 64 ? 64 : lanes);
  if (const char* e = getenv("HIPAC_JPEG_LANES")) lanes = atoi(e) >= 1 && atoi(e) <= 64 ? atoi(e) : lanes;  // developer knob
  hipLaunchKernelGGL(jpeg_huffman_kernel, dim3((unsigned)((nd + lanes - 1) / lanes)), dim3(64), 0, s, (const unsigned char*)file_dev,
                     (const TileDesc*)d_desc, (const JpegCfg*)d_cfg, nd, d_coef, coef_stride, lanes);
  const long long nblk = (long long)nd * max_blocks;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s, (const short*)d_coef, coef_stride,
                     (const TileDesc*)d_desc, (const JpegCfg*)d_cfg, nd, max_blocks, d_planes, plane_stride);
  hipLaunchKernelGGL(jpeg_store_kernel, dim3((unsigned)((max_tw * max_th + 255) / 256), (unsigned)nd), dim3(256), 0, s,
                     (const unsigned char*)d_planes, plane_stride, (const TileDesc*)d_desc, (const JpegCfg*)d_cfg,
                     (const LevelDesc*)d_lvl, nd);
  HIPAC_CHECK_HIP(hipGetLastError());
  // the descriptors were copied from this function's own (pageable) vectors: they must outlive the copies, and a slide load
  // is a bulk operation, so the call simply waits for its work
  HIPAC_CHECK_HIP(hipStreamSynchronize(s));
  for (int k = 0; k < nd; ++k) status_host[which[k]] = 0;
  return 0;
}

}  // extern "C"
