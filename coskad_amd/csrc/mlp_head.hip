// The tail of the `mlp` projector (reference models/common/components.py:209-226: [Linear -> BatchNorm1d -> ReLU] per
// hidden size + a final Linear; selected by 5 of the reference's 7 yamls, models/sts/ae.py:147-164).  The first, wide
// Linear (13 056 -> h) runs on the bottleneck kernels (bottleneck.hip, PReLU of the encoder fused into the load); this
// file is the block behind it on the [B, h] activations:
//     y2 = gamma * (y1 - mean) * invstd + beta,   a = relu(y2),   z = a . W2^T + b2
// forward (train: batch statistics over B + running-statistics update, eval: running statistics) and backward
// (dy1, dgamma, dbeta, dW2, db2).  h, latent <= 64.  Reductions over the batch run one block per feature with fp64
// accumulation in a fixed order (deterministic, no atomics).
#include "common.h"

namespace coskad {
namespace mlp {

constexpr int HMAX = 64;

__device__ __forceinline__ double block_sum(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if (t < w) sh[t] += sh[t + w];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// grid = H blocks: block k owns feature k.  stat[k] = mean, stat[H + k] = invstd (what normalised this batch).
__global__ __launch_bounds__(256) void k_stats(const float* __restrict__ y1, float* __restrict__ stat,
                                               float* __restrict__ rmean, float* __restrict__ rvar,
                                               long long* __restrict__ nbt, float momentum, float eps, int training,
                                               int B, int H) {
  __shared__ double sh[256];
  const int k = blockIdx.x;
  if (!training) {
    if (threadIdx.x == 0) {
      stat[k] = rmean[k];
      stat[H + k] = 1.f / sqrtf(rvar[k] + eps);
    }
    return;
  }
  double s = 0.0;
  for (int n = threadIdx.x; n < B; n += 256) s += (double)y1[(size_t)n * H + k];
  const double mean = block_sum(s, sh) / B;
  double q = 0.0;
  for (int n = threadIdx.x; n < B; n += 256) {
    const double d = (double)y1[(size_t)n * H + k] - mean;
    q += d * d;
  }
  const double ss = block_sum(q, sh);
  if (threadIdx.x == 0) {
    const double var = ss / B;                               // biased: what normalises (nn.BatchNorm1d)
    stat[k] = (float)mean;
    stat[H + k] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      const double unb = B > 1 ? ss / (B - 1) : var;         // unbiased: what the running estimate tracks
      rmean[k] = (float)((1.0 - momentum) * rmean[k] + momentum * mean);
      rvar[k] = (float)((1.0 - momentum) * rvar[k] + momentum * unb);
    }
    if (k == 0 && nbt) *nbt += 1;
  }
}

// one thread per row: z[n] = W2 . relu(bn(y1[n])) + b2
__global__ __launch_bounds__(256) void k_apply(const float* __restrict__ y1, const float* __restrict__ stat,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                               float* __restrict__ z, int B, int H, int L) {
  __shared__ float sc[HMAX], sf[HMAX], w2[HMAX * HMAX], bb[HMAX];
  for (int e = threadIdx.x; e < H; e += 256) {
    sc[e] = gamma[e] * stat[H + e];
    sf[e] = beta[e] - gamma[e] * stat[H + e] * stat[e];
  }
  for (int e = threadIdx.x; e < L * H; e += 256) w2[e] = W2[e];
  for (int e = threadIdx.x; e < L; e += 256) bb[e] = b2 ? b2[e] : 0.f;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  float a[HMAX];                          // statically indexed (registers): loops over HMAX with a guard
#pragma unroll
  for (int k = 0; k < HMAX; ++k) {
    const float y2 = k < H ? fmaf(y1[(size_t)n * H + k], sc[k], sf[k]) : 0.f;
    a[k] = y2 > 0.f ? y2 : 0.f;
  }
  for (int l = 0; l < L; ++l) {
    float s = bb[l];
#pragma unroll
    for (int k = 0; k < HMAX; ++k)
      if (k < H) s = fmaf(a[k], w2[l * H + k], s);
    z[(size_t)n * L + l] = s;
  }
}

// grid = H + 1 blocks.  block k < H: dgamma[k], dbeta[k], dW2[:, k];  block H: db2.
__global__ __launch_bounds__(256) void k_bwd_reduce(const float* __restrict__ y1, const float* __restrict__ stat,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ W2, const float* __restrict__ dz,
                                                    float* __restrict__ red /* [2H]: sum dy2, sum dy2*xhat */,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    float* __restrict__ dW2, float* __restrict__ db2, int accumulate,
                                                    int B, int H, int L) {
  __shared__ double sh[256];
  const int k = blockIdx.x;
  if (k == H) {
    for (int l = 0; l < L; ++l) {
      double s = 0.0;
      for (int n = threadIdx.x; n < B; n += 256) s += (double)dz[(size_t)n * L + l];
      const double t = block_sum(s, sh);
      if (threadIdx.x == 0 && db2) db2[l] = accumulate ? db2[l] + (float)t : (float)t;
    }
    return;
  }
  const float mean = stat[k], inv = stat[H + k], g = gamma[k], bt = beta[k];
  double sb = 0.0, sg = 0.0;
  for (int n = threadIdx.x; n < B; n += 256) {
    const float xh = (y1[(size_t)n * H + k] - mean) * inv;
    const float y2 = fmaf(g, xh, bt);
    float da = 0.f;
    for (int l = 0; l < L; ++l) da = fmaf(dz[(size_t)n * L + l], W2[l * H + k], da);
    const float dy2 = y2 > 0.f ? da : 0.f;
    sb += (double)dy2;
    sg += (double)dy2 * (double)xh;
  }
  const double tb = block_sum(sb, sh), tg = block_sum(sg, sh);
  if (threadIdx.x == 0) {
    red[k] = (float)tb;
    red[H + k] = (float)tg;
    dbeta[k] = accumulate ? dbeta[k] + (float)tb : (float)tb;
    dgamma[k] = accumulate ? dgamma[k] + (float)tg : (float)tg;
  }
  for (int l = 0; l < L; ++l) {                      // dW2[l][k] = sum_n dz[n][l] * a[n][k]
    double s = 0.0;
    for (int n = threadIdx.x; n < B; n += 256) {
      const float y2 = fmaf(g, (y1[(size_t)n * H + k] - mean) * inv, bt);
      s += (double)dz[(size_t)n * L + l] * (double)(y2 > 0.f ? y2 : 0.f);
    }
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) dW2[l * H + k] = accumulate ? dW2[l * H + k] + (float)t : (float)t;
  }
}

// one thread per row: dy1 (train: through the batch statistics; eval: plain affine)
__global__ __launch_bounds__(256) void k_bwd_apply(const float* __restrict__ y1, const float* __restrict__ stat,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ W2, const float* __restrict__ dz,
                                                   const float* __restrict__ red, float* __restrict__ dy1, int training,
                                                   int B, int H, int L) {
  __shared__ float w2[HMAX * HMAX];
  for (int e = threadIdx.x; e < L * H; e += 256) w2[e] = W2[e];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  float d[HMAX];
#pragma unroll
  for (int l = 0; l < HMAX; ++l) d[l] = l < L ? dz[(size_t)n * L + l] : 0.f;
  const float invB = 1.f / (float)B;
  for (int k = 0; k < H; ++k) {
    const float inv = stat[H + k], g = gamma[k];
    const float xh = (y1[(size_t)n * H + k] - stat[k]) * inv;
    const float y2 = fmaf(g, xh, beta[k]);
    float da = 0.f;
#pragma unroll
    for (int l = 0; l < HMAX; ++l)
      if (l < L) da = fmaf(d[l], w2[l * H + k], da);
    const float dy2 = y2 > 0.f ? da : 0.f;
    float r = dy2;
    if (training) r = dy2 - red[k] * invB - xh * red[H + k] * invB;
    dy1[(size_t)n * H + k] = inv * g * r;
  }
}

}  // namespace mlp
}  // namespace coskad

using namespace coskad;

extern "C" {

/* stat: [2H] floats (mean, invstd of this call); red (backward scratch): [2H] floats */
int coskad_mlp_head_fwd_f32(const float* y1, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float momentum, float eps, int training,
                            const float* W2, const float* b2, float* z, float* stat, int B, int H, int L,
                            hipStream_t stream) {
  if (!y1 || !gamma || !beta || !W2 || !z || !stat) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: null pointer");
  if (!training && (!running_mean || !running_var)) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: eval mode needs the running statistics");
  if (B <= 0 || H <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: B=%d H=%d L=%d", B, H, L);
  if (H > mlp::HMAX || L > mlp::HMAX) return fail(COSKAD_ERR_SHAPE, "mlp_head_fwd: hidden=%d / out=%d > %d not supported", H, L, mlp::HMAX);
  hipLaunchKernelGGL(mlp::k_stats, dim3(H), dim3(256), 0, stream, y1, stat, running_mean, running_var, num_batches_tracked,
                     momentum, eps, training, B, H);
  int rc;
  if ((rc = check_launch("mlp_stats"))) return rc;
  hipLaunchKernelGGL(mlp::k_apply, dim3(ceil_div(B, 256)), dim3(256), 0, stream, y1, stat, gamma, beta, W2, b2, z, B, H, L);
  return check_launch("mlp_apply");
}

int coskad_mlp_head_bwd_f32(const float* y1, const float* stat, const float* gamma, const float* beta, const float* W2,
                            const float* dz, float* dy1, float* dgamma, float* dbeta, float* dW2, float* db2, float* red,
                            int training, int accumulate, int B, int H, int L, hipStream_t stream) {
  if (!y1 || !stat || !gamma || !beta || !W2 || !dz || !dy1 || !dgamma || !dbeta || !dW2 || !red)
    return fail(COSKAD_ERR_ARG, "mlp_head_bwd: null pointer");
  if (B <= 0 || H <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "mlp_head_bwd: B=%d H=%d L=%d", B, H, L);
  if (H > mlp::HMAX || L > mlp::HMAX) return fail(COSKAD_ERR_SHAPE, "mlp_head_bwd: hidden=%d / out=%d > %d not supported", H, L, mlp::HMAX);
  hipLaunchKernelGGL(mlp::k_bwd_reduce, dim3(H + 1), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, red, dgamma, dbeta,
                     dW2, db2, accumulate, B, H, L);
  int rc;
  if ((rc = check_launch("mlp_bwd_reduce"))) return rc;
  hipLaunchKernelGGL(mlp::k_bwd_apply, dim3(ceil_div(B, 256)), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, red, dy1,
                     training, B, H, L);
  return check_launch("mlp_bwd_apply");
}

}  // extern "C"
