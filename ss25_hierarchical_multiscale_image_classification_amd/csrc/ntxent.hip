// NT-Xent loss, forward + gradient (SURVEY a-12; reference: src/models/simclr.py:31-54).
//   z = cat(z_i, z_j) [2N, D];  zn = z / max(|z|, 1e-12)  (F.normalize);  S = zn zn^T / t, diag = -inf
//   loss = mean_i( logsumexp_j S_ij - S_{i, pair(i)} ),  pair(i) = (i + N) mod 2N
//   dL/dzn_i = 1/(2N t) * [ sum_{j != i} (e^{S_ij - lse_i} + e^{S_ij - lse_j}) zn_j - 2 zn_pair(i) ]
//   dL/dz_i  = (dzn_i - zn_i (zn_i . dzn_i)) / max(|z_i|, 1e-12)
// fp32 throughout.  The two matrix products -- zn zn^T [2N x 2N] and (coefficients) zn [2N x D] -- run on the exact f32 MFMA
// (launch_gemm_f32, the strided GEMM of the linear layers); the similarity matrix lives in the scratch (16 MB at 2N = 2048:
// L2 / MALL resident) and the row-wise steps stream it: log-sum-exp per row, the coefficient matrix in place, the projection
// back through F.normalize.  Round 2's LDS-tiled FMA kernels recomputed every tile's scores with two LDS reads per FMA on half
// the CUs: 1.6 ms per step at 2N = 2048 against ~0.15 ms now.  No atomics: the loss is summed in row order.
#include "common.h"

namespace hipac {

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

// one workgroup per row i of the raw products P = zn zn^T: lse_i over j != i of P_ij / t, and the row's loss term
__global__ __launch_bounds__(256) void ntx_lse_kernel(const float* __restrict__ P, int rows, int N, float inv_t,
                                                      float* __restrict__ lse, float* __restrict__ terms) {
  __shared__ float red[8];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float* row = P + (size_t)i * rows;
  float m = -INFINITY;
  for (int j = tid; j < rows; j += 256)
    if (j != i) m = fmaxf(m, row[j] * inv_t);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int j = tid; j < rows; j += 256)
    if (j != i) s += expf(row[j] * inv_t - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
  __syncthreads();
  if (tid == 0) {
    const float l = m + logf(((red[4] + red[5]) + red[6]) + red[7]);
    lse[i] = l;
    terms[i] = l - row[i < N ? i + N : i - N] * inv_t;
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

// in place: P_ij -> coefficient of zn_j in dzn_i = e^{S_ij - lse_i} + e^{S_ij - lse_j} - 2 [j == pair(i)], 0 on the diagonal
__global__ __launch_bounds__(256) void ntx_coeff_kernel(float* __restrict__ P, const float* __restrict__ lse, int rows, int N,
                                                        float inv_t) {
  const long long total = (long long)rows * rows;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const int i = (int)(g / rows), j = (int)(g - (long long)i * rows);
    float c = 0.f;
    if (i != j) {
      const float sv = P[g] * inv_t;
      c = expf(sv - lse[i]) + expf(sv - lse[j]);
      if (j == (i < N ? i + N : i - N)) c -= 2.f;
    }
    P[g] = c;
  }
}

// dz_i = (g_i gscale - zn_i (zn_i . g_i gscale)) * inv_norm_i; one wave per row (g may alias dz)
__global__ __launch_bounds__(256) void ntx_project_kernel(const float* g, const float* __restrict__ zn,
                                                          const float* __restrict__ inv_norm, int rows, int D, float gscale,
                                                          float* dz) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float dot = 0.f;
  for (int k = lane; k < D; k += 64) dot = fmaf(zn[(size_t)row * D + k], g[(size_t)row * D + k] * gscale, dot);
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  const float inv = inv_norm[row];
  for (int k = lane; k < D; k += 64)
    dz[(size_t)row * D + k] = (g[(size_t)row * D + k] * gscale - zn[(size_t)row * D + k] * dot) * inv;
}

}  // namespace hipac

extern "C" size_t hipac_ntxent_scratch_bytes(int n, int d) {
  if (n <= 0 || d <= 0) return 0;
  // zn [2N][D], inv_norm [2N], lse [2N], per-row loss terms [2N], the similarity / coefficient matrix [2N][2N]
  return ((size_t)2 * n * d + (size_t)6 * n + (size_t)4 * n * n) * sizeof(float);
}

extern "C" int hipac_ntxent_fwd_bwd(const float* z, int n, int d, float temperature, float* loss, float* dz,
                                    void* scratch, size_t scratch_bytes, void* stream) {
  using namespace hipac;
  HIPAC_REQUIRE(z && loss && scratch, HIPAC_EINVAL, "ntxent: null argument");
  HIPAC_REQUIRE(n > 0 && n <= 16384 && d > 0 && d <= 4096, HIPAC_EINVAL, "ntxent: n %d, d %d", n, d);
  HIPAC_REQUIRE(temperature > 0.f, HIPAC_EINVAL, "ntxent: temperature %g", (double)temperature);
  HIPAC_REQUIRE(scratch_bytes >= hipac_ntxent_scratch_bytes(n, d), HIPAC_EWORKSPACE, "ntxent: scratch %zu < %zu",
                scratch_bytes, hipac_ntxent_scratch_bytes(n, d));
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * n;
  float* zn = (float*)scratch;
  float* inv_norm = zn + (size_t)rows * d;
  float* lse = inv_norm + rows;
  float* terms = lse + rows;
  float* P = terms + rows;
  const float inv_t = 1.f / temperature;
  hipLaunchKernelGGL(ntx_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, z, rows, d, zn, inv_norm);
  // P[i][j] = zn_i . zn_j
  if (int rc = launch_gemm_f32(zn, d, 1, zn, d, 1, P, rows, rows, rows, d, s)) return rc;
  hipLaunchKernelGGL(ntx_lse_kernel, dim3(rows), dim3(256), 0, s, (const float*)P, rows, n, inv_t, lse, terms);
  hipLaunchKernelGGL(ntx_loss_kernel, dim3(1), dim3(256), 0, s, (const float*)terms, rows, loss);
  if (dz) {
    const long long total = (long long)rows * rows;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(ntx_coeff_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, s, P, (const float*)lse, rows, n,
                       inv_t);
    // g[i][k] = sum_j C_ij zn[j][k]:  A(i, j) = C, B(k, j) = zn[j][k]
    if (int rc = launch_gemm_f32(P, rows, 1, zn, 1, d, dz, d, rows, d, rows, s)) return rc;
    hipLaunchKernelGGL(ntx_project_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)dz, (const float*)zn,
                       (const float*)inv_norm, rows, d, 1.f / ((float)rows * temperature), dz);
  }
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}
