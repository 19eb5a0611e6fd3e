// MIL head: attention scores per patch, then per bag: softmax over the bag,
// weighted (or mean / max) pooling of the 512-d features and the two-layer classifier.
// Reference: src/models/mil_classifier.py:12-18 (MILAttentionPooling.forward), :38-45
// (MILClassifier.forward).  Everything is fp32; bound: HBM (each feature row is read twice:
// once for its score, once for the pooling -- 4 KB per patch), the arithmetic is ~0.13 MFLOP
// per patch against 3.6 GFLOP for the ResNet that produced the row.
#include "common.h"

namespace hipac {

// scores[i] = U . tanh(V x_i + bV) + bU.  Workgroup = A threads (one hidden unit each) x G patches:
// the G feature rows sit in LDS (read as broadcasts), thread j streams row j of V in 16-byte pieces.
constexpr int kMilG = 16;

__global__ __launch_bounds__(256) void mil_scores_kernel(const float* __restrict__ feats, int n, int F, int A,
                                                         const float* __restrict__ Vw, const float* __restrict__ Vb,
                                                         const float* __restrict__ Uw, const float* __restrict__ Ub,
                                                         float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [kMilG][F] then [kMilG][blockDim.x / 64]
  const int g0 = blockIdx.x * kMilG;
  const int tid = threadIdx.x;
  for (int i = tid; i < kMilG * (F / 4); i += blockDim.x) {
    const int g = i / (F / 4), c = i - g * (F / 4);
    const int row = g0 + g < n ? g0 + g : n - 1;
    reinterpret_cast<f32x4*>(xs)[i] = reinterpret_cast<const f32x4*>(feats + (size_t)row * F)[c];
  }
  __syncthreads();
  float acc[kMilG];
#pragma unroll
  for (int g = 0; g < kMilG; ++g) acc[g] = 0.f;
  if (tid < A) {
    const f32x4* vr = reinterpret_cast<const f32x4*>(Vw + (size_t)tid * F);
    for (int c = 0; c < F / 4; ++c) {
      const f32x4 v = vr[c];
#pragma unroll
      for (int g = 0; g < kMilG; ++g) {
        const f32x4 x = reinterpret_cast<const f32x4*>(xs + g * F)[c];
        acc[g] = fmaf(v[0], x[0], acc[g]);
        acc[g] = fmaf(v[1], x[1], acc[g]);
        acc[g] = fmaf(v[2], x[2], acc[g]);
        acc[g] = fmaf(v[3], x[3], acc[g]);
      }
    }
    const float vb = Vb[tid], u = Uw[tid];
#pragma unroll
    for (int g = 0; g < kMilG; ++g) acc[g] = u * tanhf(acc[g] + vb);
  }
  // sum over hidden units: wave shuffle, then across waves through LDS
  float* part = xs + kMilG * F;
  const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int g = 0; g < kMilG; ++g) {
    float v = acc[g];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) part[g * nw + wave] = v;
  }
  __syncthreads();
  if (tid < kMilG && g0 + tid < n) {
    float v = Ub[0];
    for (int w = 0; w < nw; ++w) v += part[tid * nw + w];
    scores[g0 + tid] = v;
  }
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    const float t = __shfl_down(v, o, 64);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  __syncthreads();  // red may still be read from a previous call
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

// one workgroup (256 threads) per bag
__global__ __launch_bounds__(256) void mil_bag_kernel(const float* __restrict__ feats, const int32_t* __restrict__ offs,
                                                      int F, int Hd, int Cn, int pooling,
                                                      float* scores_w, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, float* __restrict__ logits,
                                                      float* __restrict__ attn, float* __restrict__ pooled_out) {
  __shared__ float red[4];
  __shared__ float pooled[2048];
  __shared__ float hid[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int o0 = offs[b], o1 = offs[b + 1];
  float m = 0.f, inv = 0.f;
  (void)m;
  if (pooling == HIPAC_MIL_ATTENTION) {
    float mx = -INFINITY;
    for (int i = o0 + tid; i < o1; i += 256) mx = fmaxf(mx, scores_w[i]);
    m = block_reduce(mx, true, red);
    float z = 0.f;
    for (int i = o0 + tid; i < o1; i += 256) z += expf(scores_w[i] - m);
    inv = 1.f / block_reduce(z, false, red);
    // scores[] becomes the softmax weight in place (scratch), so the pooling below reads it back
    for (int i = o0 + tid; i < o1; i += 256) {
      const float w = expf(scores_w[i] - m) * inv;
      scores_w[i] = w;
      if (attn) attn[i] = w;
    }
    __threadfence_block();
    __syncthreads();
  } else if (pooling == HIPAC_MIL_MEAN) {
    inv = 1.f / (float)(o1 - o0);
  }
  // pooling: thread -> features tid, tid+256, ...; rows streamed, coalesced across threads
  for (int f = tid; f < F; f += 256) {
    float a = pooling == HIPAC_MIL_MAX ? -INFINITY : 0.f;
    for (int i = o0; i < o1; ++i) {
      const float x = feats[(size_t)i * F + f];
      if (pooling == HIPAC_MIL_ATTENTION) a = fmaf(scores_w[i], x, a);
      else if (pooling == HIPAC_MIL_MEAN) a += x;
      else a = fmaxf(a, x);
    }
    if (pooling == HIPAC_MIL_MEAN) a *= inv;
    pooled[f] = a;
    if (pooled_out) pooled_out[(size_t)b * F + f] = a;
  }
  __syncthreads();
  // classifier: Linear(F, Hd) + ReLU + Linear(Hd, Cn)
  for (int j = tid; j < Hd; j += 256) {
    const f32x4* wr = reinterpret_cast<const f32x4*>(W1 + (size_t)j * F);
    float a = 0.f;
    for (int c = 0; c < F / 4; ++c) {
      const f32x4 w = wr[c];
      a = fmaf(w[0], pooled[4 * c + 0], a);
      a = fmaf(w[1], pooled[4 * c + 1], a);
      a = fmaf(w[2], pooled[4 * c + 2], a);
      a = fmaf(w[3], pooled[4 * c + 3], a);
    }
    hid[j] = fmaxf(a + b1[j], 0.f);
  }
  __syncthreads();
  if (tid < Cn) {
    float a = b2[tid];
    for (int j = 0; j < Hd; ++j) a = fmaf(W2[tid * Hd + j], hid[j], a);
    logits[b * Cn + tid] = a;
  }
}

}  // namespace hipac

extern "C" int hipac_mil_forward(const hipac_mil_params_t* p, int pooling, const float* feats,
                                 const int32_t* bag_offsets, int n, int n_bags, float* logits, float* attn,
                                 float* pooled, float* scores, void* stream) {
  using namespace hipac;
  HIPAC_REQUIRE(p && feats && bag_offsets && logits, HIPAC_EINVAL, "mil_forward: null argument");
  HIPAC_REQUIRE(pooling >= HIPAC_MIL_ATTENTION && pooling <= HIPAC_MIL_MAX, HIPAC_EINVAL, "mil_forward: pooling %d",
                pooling);
  HIPAC_REQUIRE(n > 0 && n_bags > 0, HIPAC_EINVAL, "mil_forward: n %d, n_bags %d", n, n_bags);
  HIPAC_REQUIRE(p->feature_dim > 0 && p->feature_dim % 4 == 0 && p->feature_dim <= 2048, HIPAC_EINVAL,
                "mil_forward: feature_dim %d", p->feature_dim);
  HIPAC_REQUIRE(p->hidden_dim > 0 && p->hidden_dim <= 256 && p->num_classes > 0 && p->num_classes <= 16, HIPAC_EINVAL,
                "mil_forward: hidden_dim %d, num_classes %d", p->hidden_dim, p->num_classes);
  HIPAC_REQUIRE(p->fc1_w && p->fc1_b && p->fc2_w && p->fc2_b, HIPAC_EINVAL, "mil_forward: classifier weights missing");
  HIPAC_REQUIRE(((uintptr_t)feats & 15) == 0 && ((uintptr_t)p->fc1_w & 15) == 0, HIPAC_EINVAL,
                "mil_forward: feats / fc1_w must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (pooling == HIPAC_MIL_ATTENTION) {
    HIPAC_REQUIRE(p->attn_V_w && p->attn_V_b && p->attn_U_w && p->attn_U_b && scores, HIPAC_EINVAL,
                  "mil_forward: attention weights / scores scratch missing");
    HIPAC_REQUIRE(p->attn_dim > 0 && p->attn_dim <= 256 && ((uintptr_t)p->attn_V_w & 15) == 0, HIPAC_EINVAL,
                  "mil_forward: attn_dim %d", p->attn_dim);
    const int threads = (p->attn_dim + 63) / 64 * 64;
    const size_t lds = (size_t)kMilG * p->feature_dim * 4 + (size_t)kMilG * (threads / 64) * 4;
    HIPAC_REQUIRE(lds <= 64 * 1024, HIPAC_EINVAL, "mil_forward: feature_dim %d too large for the score kernel",
                  p->feature_dim);
    hipLaunchKernelGGL(mil_scores_kernel, dim3((n + kMilG - 1) / kMilG), dim3(threads), lds, s, feats, n, p->feature_dim,
                       p->attn_dim, p->attn_V_w, p->attn_V_b, p->attn_U_w, p->attn_U_b, scores);
    HIPAC_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(mil_bag_kernel, dim3(n_bags), dim3(256), 0, s, feats, bag_offsets, p->feature_dim, p->hidden_dim,
                     p->num_classes, pooling, scores, p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, logits, attn, pooled);
  HIPAC_CHECK_HIP(hipGetLastError());
  return 0;
}
