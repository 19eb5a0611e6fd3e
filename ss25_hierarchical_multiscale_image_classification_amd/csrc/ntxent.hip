// NT-Xent loss, forward + gradient (SURVEY a-12; reference: src/models/simclr.py:31-54).
//   z = cat(z_i, z_j) [2N, D];  zn = z / max(|z|, 1e-12)  (F.normalize);  S = zn zn^T / t, diag = -inf
//   loss = mean_i( logsumexp_j S_ij - S_{i, pair(i)} ),  pair(i) = (i + N) mod 2N
//   dL/dzn_i = 1/(2N t) * [ sum_{j != i} (e^{S_ij - lse_i} + e^{S_ij - lse_j}) zn_j - 2 zn_pair(i) ]
//   dL/dz_i  = (dzn_i - zn_i (zn_i . dzn_i)) / max(|z_i|, 1e-12)
// fp32 throughout.  2N = 2048, D = 128 is 1 GFLOP: a rounding error next to the encoder's 22 TFLOP
// per step, so the kernels are plain LDS-tiled FMA code (rows x column tiles), not MFMA.
#include "common.h"

namespace hipac {

constexpr int kNtR = 8;    // rows per workgroup (2N = 2048 rows -> 256 workgroups: one per CU)
constexpr int kNtC = 64;   // columns per tile
constexpr int kTPR = 256 / kNtR;   // threads per row in the row-wise phases (32: a half wave)
constexpr int kDPT = 256 / kTPR;   // dims per thread in the backward accumulation (D <= 256)

__global__ __launch_bounds__(256) void ntx_normalize_kernel(const float* __restrict__ z, int rows, int D,
                                                            float* __restrict__ zn, float* __restrict__ inv_norm) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = z[(size_t)row * D + k];
    s = fmaf(v, v, s);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int k = lane; k < D; k += 64) zn[(size_t)row * D + k] = z[(size_t)row * D + k] * inv;
  if (lane == 0) inv_norm[row] = inv;
}

// S tile [kNtR][kNtC] of rows r0.. x columns c0.. into LDS `st` (pitch kNtC + 1); rows / columns beyond
// `rows` and the diagonal become -inf.  zi: [kNtR][D] rows (LDS), zj: [kNtC][D + 1] columns (LDS).
__device__ __forceinline__ void ntx_tile_scores(const float* zi, const float* zj, int D, int r0, int c0, int rows,
                                                float inv_t, float* st) {
  for (int p = threadIdx.x; p < kNtR * kNtC; p += 256) {
    const int a = p / kNtC, b = p - a * kNtC;
    const float* x = zi + a * D;
    const float* y = zj + b * (D + 1);
    float acc = 0.f;
    for (int k = 0; k < D; ++k) acc = fmaf(x[k], y[k], acc);
    const int i = r0 + a, j = c0 + b;
    st[a * (kNtC + 1) + b] = (i < rows && j < rows && i != j) ? acc * inv_t : -INFINITY;
  }
}

__device__ __forceinline__ void ntx_load_tiles(const float* zn, int D, int rows, int r0, int c0, bool load_rows,
                                               float* zi, float* zj) {
  if (load_rows)
    for (int p = threadIdx.x; p < kNtR * D; p += 256) {
      const int a = p / D, k = p - a * D;
      zi[p] = r0 + a < rows ? zn[(size_t)(r0 + a) * D + k] : 0.f;
    }
  for (int p = threadIdx.x; p < kNtC * D; p += 256) {
    const int b = p / D, k = p - b * D;
    zj[b * (D + 1) + k] = c0 + b < rows ? zn[(size_t)(c0 + b) * D + k] : 0.f;
  }
}

// forward: lse[i], and loss += sum_i (lse_i - S_{i,pair(i)}) / rows
__global__ __launch_bounds__(256) void ntx_forward_kernel(const float* __restrict__ zn, int rows, int N, int D,
                                                          float inv_t, float* __restrict__ lse,
                                                          float* __restrict__ terms) {
  extern __shared__ float sm[];
  float* zi = sm;                          // [kNtR][D]
  float* zj = zi + kNtR * D;               // [kNtC][D + 1]
  float* st = zj + kNtC * (D + 1);         // [kNtR][kNtC + 1]
  float* red = st + kNtR * (kNtC + 1);     // [kNtR][3]: running max, running sum, positive
  const int r0 = blockIdx.x * kNtR;
  if (threadIdx.x < kNtR) {
    red[threadIdx.x * 3 + 0] = -INFINITY;
    red[threadIdx.x * 3 + 1] = 0.f;
    red[threadIdx.x * 3 + 2] = 0.f;
  }
  for (int c0 = 0; c0 < rows; c0 += kNtC) {
    __syncthreads();
    ntx_load_tiles(zn, D, rows, r0, c0, c0 == 0, zi, zj);
    __syncthreads();
    ntx_tile_scores(zi, zj, D, r0, c0, rows, inv_t, st);
    __syncthreads();
    // kTPR lanes per row: online logsumexp over the tile's 64 columns
    const int a = threadIdx.x / kTPR, l = threadIdx.x % kTPR;
    float m = -INFINITY;
    for (int b = l; b < kNtC; b += kTPR) m = fmaxf(m, st[a * (kNtC + 1) + b]);
    for (int o = kTPR / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kTPR));
    const float m_old = red[a * 3 + 0];
    const float m_new = fmaxf(m_old, m);
    float s = 0.f;
    if (m_new > -INFINITY)
      for (int b = l; b < kNtC; b += kTPR) s += expf(st[a * (kNtC + 1) + b] - m_new);
    for (int o = kTPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kTPR);
    const int i = r0 + a, pj = i < N ? i + N : i - N;  // pair(i)
    __syncthreads();
    if (l == 0) {
      red[a * 3 + 1] = (m_old > -INFINITY ? red[a * 3 + 1] * expf(m_old - m_new) : 0.f) + s;
      red[a * 3 + 0] = m_new;
      if (i < rows && pj >= c0 && pj < c0 + kNtC) red[a * 3 + 2] = st[a * (kNtC + 1) + (pj - c0)];
    }
  }
  __syncthreads();
  if (threadIdx.x < kNtR && r0 + (int)threadIdx.x < rows) {
    const int a = threadIdx.x;
    const float l = red[a * 3 + 0] + logf(red[a * 3 + 1]);
    lse[r0 + a] = l;
    terms[r0 + a] = l - red[a * 3 + 2];  // summed in row order by ntx_loss_kernel (no atomics: the loss value is reproducible)
  }
}

__global__ __launch_bounds__(256) void ntx_loss_kernel(const float* __restrict__ terms, int rows, float* __restrict__ loss) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) s += terms[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)rows;
}

// backward: dz rows r0 .. r0 + kNtR - 1
__global__ __launch_bounds__(256) void ntx_backward_kernel(const float* __restrict__ zn,
                                                           const float* __restrict__ inv_norm,
                                                           const float* __restrict__ lse, int rows, int N, int D,
                                                           float inv_t, float gscale, float* __restrict__ dz) {
  extern __shared__ float sm[];
  float* zi = sm;
  float* zj = zi + kNtR * D;
  float* st = zj + kNtC * (D + 1);
  float* ls = st + kNtR * (kNtC + 1);      // lse of the tile's columns [kNtC]
  float* lr = ls + kNtC;                   // lse of the workgroup's rows [kNtR]
  const int r0 = blockIdx.x * kNtR;
  // thread -> (row a, dims l + kTPR * t): kTPR threads per row, D / kTPR dims each (D <= 256)
  const int a = threadIdx.x / kTPR, l = threadIdx.x % kTPR;
  float g[kDPT];
#pragma unroll
  for (int t = 0; t < kDPT; ++t) g[t] = 0.f;
  const int i = r0 + a;
  if (threadIdx.x < kNtR) lr[threadIdx.x] = r0 + (int)threadIdx.x < rows ? lse[r0 + threadIdx.x] : 0.f;
  for (int c0 = 0; c0 < rows; c0 += kNtC) {
    __syncthreads();
    ntx_load_tiles(zn, D, rows, r0, c0, c0 == 0, zi, zj);
    if (threadIdx.x < kNtC) ls[threadIdx.x] = c0 + (int)threadIdx.x < rows ? lse[c0 + threadIdx.x] : 0.f;
    __syncthreads();
    ntx_tile_scores(zi, zj, D, r0, c0, rows, inv_t, st);
    __syncthreads();
    // coefficient of zn_j in dzn_i, in place: e^{S - lse_i} + e^{S - lse_j} - 2 [j == pair(i)]
    for (int p = threadIdx.x; p < kNtR * kNtC; p += 256) {
      const int aa = p / kNtC, b = p - aa * kNtC;
      const int ii = r0 + aa, jj = c0 + b;
      const float sv = st[aa * (kNtC + 1) + b];
      float c = 0.f;
      if (sv > -INFINITY) {
        c = expf(sv - lr[aa]) + expf(sv - ls[b]);
        if (jj == (ii < N ? ii + N : ii - N)) c -= 2.f;
      }
      st[aa * (kNtC + 1) + b] = c;
    }
    __syncthreads();
    if (i < rows) {
      for (int b = 0; b < kNtC; ++b) {
        const float c = st[a * (kNtC + 1) + b];
#pragma unroll
        for (int t = 0; t < kDPT; ++t) {
          const int d = l + kTPR * t;
          if (d < D) g[t] = fmaf(c, zj[b * (D + 1) + d], g[t]);
        }
      }
    }
  }
  if (i >= rows) return;  // whole kTPR-lane groups leave together (a = row)
  // dzn = g * gscale;  dz = (dzn - zn (zn . dzn)) * inv_norm
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < kDPT; ++t) {
    const int d = l + kTPR * t;
    if (d < D) dot = fmaf(zi[a * D + d], g[t] * gscale, dot);
  }
  for (int o = kTPR / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, kTPR);
  const float inv = inv_norm[i];
#pragma unroll
  for (int t = 0; t < kDPT; ++t) {
    const int d = l + kTPR * t;
    if (d < D) dz[(size_t)i * D + d] = (g[t] * gscale - zi[a * D + d] * dot) * inv;
  }
}

}  // namespace hipac

extern "C" size_t hipac_ntxent_scratch_bytes(int n, int d) {
  if (n <= 0 || d <= 0) return 0;
  return ((size_t)2 * n * d + (size_t)6 * n) * sizeof(float);  // zn [2N][D], inv_norm [2N], lse [2N], per-row loss terms [2N]
}

extern "C" int hipac_ntxent_fwd_bwd(const float* z, int n, int d, float temperature, float* loss, float* dz,
                                    void* scratch, size_t scratch_bytes, void* stream) {
  using namespace hipac;
  HIPAC_REQUIRE(z && loss && scratch, HIPAC_EINVAL, "ntxent: null argument");
  HIPAC_REQUIRE(n > 0 && d > 0 && d <= 256, HIPAC_EINVAL, "ntxent: n %d, d %d (d <= 256)", n, d);
  HIPAC_REQUIRE(temperature > 0.f, HIPAC_EINVAL, "ntxent: temperature %g", (double)temperature);
  HIPAC_REQUIRE(scratch_bytes >= hipac_ntxent_scratch_bytes(n, d), HIPAC_EWORKSPACE, "ntxent: scratch %zu < %zu",
                scratch_bytes, hipac_ntxent_scratch_bytes(n, d));
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * n;
  float* zn = (float*)scratch;
  float* inv_norm = zn + (size_t)rows * d;
  float* lse = inv_norm + rows;
  float* terms = lse + rows;
  hipLaunchKernelGGL(ntx_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, z, rows, d, zn, inv_norm);
  const size_t lds =
      ((size_t)kNtR * d + (size_t)kNtC * (d + 1) + (size_t)kNtR * (kNtC + 1) + kNtC + kNtR) * sizeof(float);
  const int blocks = (rows + kNtR - 1) / kNtR;
  hipLaunchKernelGGL(ntx_forward_kernel, dim3(blocks), dim3(256), lds, s, zn, rows, n, d, 1.f / temperature, lse, terms);
  hipLaunchKernelGGL(ntx_loss_kernel, dim3(1), dim3(256), 0, s, (const float*)terms, rows, loss);
  if (dz)
    hipLaunchKernelGGL(ntx_backward_kernel, dim3(blocks), dim3(256), lds, s, zn, inv_norm, lse, rows, n, d,
                       1.f / temperature, 1.f / ((float)rows * temperature), dz);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}
