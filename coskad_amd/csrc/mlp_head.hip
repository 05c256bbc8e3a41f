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

// ---- hidden, out <= 16 (what the reference's yamls select: `mlp` with one hidden layer of the latent size) ------------------------
// The kernels above give a feature ONE block and walk the batch with it: 17 blocks, 18 block-wide fp64 tree sums in a row --
// 104 us for the backward reduction of 256 KB at B = 4096, 166 us for the four launches.  Here a block owns RB = 64 rows: tiles
// of y1 / dz in LDS, every sum over the rows is one thread's loop in a fixed order, blocks write partial rows and the second
// kernel of each direction adds them (fp64, fixed order) -- deterministic as before, 64 blocks instead of 17, no tree sums.
constexpr int HP = 16, RB = 64;
// sum over the blocks' partial rows, eight loads in flight, combined in a fixed order (a plain loop is a chain of dependent
// L2 round trips: 25 us for 64 rows)
template <typename TP>
__device__ __forceinline__ double sum_parts(const TP* __restrict__ part, int nblk, int stride, int e) {
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int p = 0;
  for (; p + 8 <= nblk; p += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += (double)part[(size_t)(p + u) * stride + e];
  }
  for (; p < nblk; ++p) a[0] += (double)part[(size_t)p * stride + e];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
constexpr int FWD_PART = 2 * HP;                     // per block: sum y [16], sum y^2 [16] (doubles)
constexpr int BWD_PART = 2 * HP + HP + HP * HP;      // per block: sum dy2 [16], sum dy2 xhat [16], db2 [16], dW2 [16][16] (floats)

// partial[blk][32] (fp64): sums of y and y^2 over the block's rows
__global__ __launch_bounds__(256) void k_stats16(const float* __restrict__ y1, double* __restrict__ part, int B, int H) {
  __shared__ double sh[2][16][HP + 1];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int r0 = blockIdx.x * RB;
  double s = 0.0, q = 0.0;
  if (col < H)
    for (int r = r0 + rl; r < min(B, r0 + RB); r += 16) {
      const double v = (double)y1[(size_t)r * H + col];
      s += v;
      q += v * v;
    }
  sh[0][rl][col] = s;
  sh[1][rl][col] = q;
  __syncthreads();
  if (threadIdx.x < 2 * HP) {
    const int w = threadIdx.x >> 4, c = threadIdx.x & 15;
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[w][k][c];
    part[(size_t)blockIdx.x * FWD_PART + threadIdx.x] = t;
  }
}

// every block adds the partials (fp64, fixed order) -> mean / invstd; block 0 also stores them and updates the running statistics;
// then z[n] = W2 . relu(bn(y1[n])) + b2 for the block's rows, one thread per (row, quarter of the outputs)
__global__ __launch_bounds__(256) void k_apply16(const float* __restrict__ y1, const double* __restrict__ part, int nblk,
                                                 float* __restrict__ stat, float* __restrict__ rmean, float* __restrict__ rvar,
                                                 long long* __restrict__ nbt, float momentum, float eps, int training,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ z,
                                                 int B, int H, int L) {
  __shared__ float sc[HP], sf[HP], w2[HP][HP + 1], bb[HP], ya[RB][HP + 1];
  __shared__ double tot[FWD_PART];
  if (training) {
    if (threadIdx.x < FWD_PART) tot[threadIdx.x] = sum_parts(part, nblk, FWD_PART, threadIdx.x);
    __syncthreads();
  }
  if (threadIdx.x < HP) {
    const int k = threadIdx.x;
    float mean = 0.f, inv = 0.f;
    if (k < H) {
      if (training) {
        const double m = tot[k] / B;
        double ss = tot[HP + k] - m * tot[k];               // sum (y - mean)^2
        if (ss < 0.0) ss = 0.0;
        const double var = ss / B;
        mean = (float)m;
        inv = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.x == 0) {
          if (rmean) {
            const double unb = B > 1 ? ss / (B - 1) : var;
            rmean[k] = (float)((1.0 - momentum) * rmean[k] + momentum * m);
            rvar[k] = (float)((1.0 - momentum) * rvar[k] + momentum * unb);
          }
          if (k == 0 && nbt) *nbt += 1;
        }
      } else {
        mean = rmean[k];
        inv = 1.f / sqrtf(rvar[k] + eps);
      }
      if (blockIdx.x == 0) { stat[k] = mean; stat[H + k] = inv; }
    }
    sc[k] = k < H ? gamma[k] * inv : 0.f;
    sf[k] = k < H ? beta[k] - gamma[k] * inv * mean : 0.f;
    bb[k] = (k < L && b2) ? b2[k] : 0.f;
  }
  {
    const int l = threadIdx.x >> 4, k = threadIdx.x & 15;
    w2[l][k] = (l < L && k < H) ? W2[l * H + k] : 0.f;
  }
  __syncthreads();
  const int r0 = blockIdx.x * RB;
  {
    const int col = threadIdx.x & 15;
    for (int rr = threadIdx.x >> 4; rr < RB; rr += 16) {
      const int r = r0 + rr;
      const float y2 = (r < B && col < H) ? fmaf(y1[(size_t)r * H + col], sc[col], sf[col]) : 0.f;
      ya[rr][col] = y2 > 0.f ? y2 : 0.f;
    }
  }
  __syncthreads();
  {
    const int l = threadIdx.x & 15;
    for (int rr = threadIdx.x >> 4; rr < RB; rr += 16) {
      const int r = r0 + rr;
      float acc = bb[l];
#pragma unroll
      for (int k = 0; k < HP; ++k) acc = fmaf(ya[rr][k], w2[l][k], acc);
      if (r < B && l < L) z[(size_t)r * L + l] = acc;
    }
  }
}

// per block of RB rows: partial [sum dy2][sum dy2 xhat][db2][dW2] (floats, each a fixed-order sum over the block's rows)
__global__ __launch_bounds__(256) void k_bwd_reduce16(const float* __restrict__ y1, const float* __restrict__ stat,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ W2, const float* __restrict__ dz,
                                                      float* __restrict__ part, int B, int H, int L) {
  __shared__ float w2[HP][HP + 1], dzl[RB][HP + 1], al[RB][HP + 1], d2[RB][HP + 1], dx[RB][HP + 1];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int r0 = blockIdx.x * RB;
  w2[rl][col] = (rl < L && col < H) ? W2[rl * H + col] : 0.f;
  for (int rr = rl; rr < RB; rr += 16) {
    const int r = r0 + rr;
    dzl[rr][col] = (r < B && col < L) ? dz[(size_t)r * L + col] : 0.f;
  }
  __syncthreads();
  const float mean = col < H ? stat[col] : 0.f, inv = col < H ? stat[H + col] : 0.f;
  const float g = col < H ? gamma[col] : 0.f, bt = col < H ? beta[col] : 0.f;
  for (int rr = rl; rr < RB; rr += 16) {
    const int r = r0 + rr;
    const bool ok = r < B && col < H;
    const float xh = ok ? (y1[(size_t)r * H + col] - mean) * inv : 0.f;
    const float y2 = fmaf(g, xh, bt);
    float da = 0.f;
#pragma unroll
    for (int l = 0; l < HP; ++l) da = fmaf(dzl[rr][l], w2[l][col], da);
    const float dy2 = (ok && y2 > 0.f) ? da : 0.f;
    al[rr][col] = (ok && y2 > 0.f) ? y2 : 0.f;
    d2[rr][col] = dy2;
    dx[rr][col] = dy2 * xh;
  }
  __syncthreads();
  float* dst = part + (size_t)blockIdx.x * BWD_PART;
  {                                                       // dW2[l][k] = sum_n dz[n][l] a[n][k]: thread (l, k) walks the rows
    const int l = rl, k = col;
    float t = 0.f;
#pragma unroll 8
    for (int n = 0; n < RB; ++n) t = fmaf(dzl[n][l], al[n][k], t);
    dst[3 * HP + l * HP + k] = t;
  }
  if (threadIdx.x < 3 * HP) {
    const int w = threadIdx.x >> 4, c = threadIdx.x & 15;
    float t = 0.f;
    for (int n = 0; n < RB; ++n) t += w == 0 ? d2[n][c] : (w == 1 ? dx[n][c] : dzl[n][c]);
    dst[w * HP + c] = t;
  }
}

// every block adds the partials of sum dy2 / sum dy2 xhat (fp64, fixed order); block 0 adds all of them and writes the parameter
// gradients; then dy1 for the block's rows
__global__ __launch_bounds__(256) void k_bwd_apply16(const float* __restrict__ y1, const float* __restrict__ stat,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ W2, const float* __restrict__ dz,
                                                     const float* __restrict__ part, int nblk, float* __restrict__ dy1,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dW2, float* __restrict__ db2, int training, int accumulate,
                                                     int B, int H, int L) {
  __shared__ float w2[HP][HP + 1], dzl[RB][HP + 1], red[2 * HP];
  const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int r0 = blockIdx.x * RB;
  const int nsum = blockIdx.x == 0 ? BWD_PART : 2 * HP;
  for (int e = threadIdx.x; e < nsum; e += 256) {
    const double t = sum_parts(part, nblk, BWD_PART, e);
    if (e < 2 * HP) red[e] = (float)t;
    if (blockIdx.x == 0) {
      float* out = nullptr;
      if (e < HP) { if (e < H) out = dbeta + e; }
      else if (e < 2 * HP) { if (e - HP < H) out = dgamma + (e - HP); }
      else if (e < 3 * HP) { if (e - 2 * HP < L && db2) out = db2 + (e - 2 * HP); }
      else { const int l = (e - 3 * HP) >> 4, k = (e - 3 * HP) & 15; if (l < L && k < H) out = dW2 + l * H + k; }
      if (out) *out = accumulate ? *out + (float)t : (float)t;
    }
  }
  w2[rl][col] = (rl < L && col < H) ? W2[rl * H + col] : 0.f;
  for (int rr = rl; rr < RB; rr += 16) {
    const int r = r0 + rr;
    dzl[rr][col] = (r < B && col < L) ? dz[(size_t)r * L + col] : 0.f;
  }
  __syncthreads();
  if (col >= H) return;
  const float mean = stat[col], inv = stat[H + col], g = gamma[col], bt = beta[col];
  const float invB = 1.f / (float)B;
  for (int rr = rl; rr < RB; rr += 16) {
    const int r = r0 + rr;
    if (r >= B) break;
    const float xh = (y1[(size_t)r * H + col] - mean) * inv;
    const float y2 = fmaf(g, xh, bt);
    float da = 0.f;
#pragma unroll
    for (int l = 0; l < HP; ++l) da = fmaf(dzl[rr][l], w2[l][col], da);
    const float dy2 = y2 > 0.f ? da : 0.f;
    float rres = dy2;
    if (training) rres = dy2 - red[col] * invB - xh * red[HP + col] * invB;
    dy1[(size_t)r * H + col] = inv * g * rres;
  }
}

}  // namespace mlp
}  // namespace coskad

using namespace coskad;

extern "C" {

/* floats of the `stat` buffer of coskad_mlp_head_fwd_f32 and of the `red` scratch of coskad_mlp_head_bwd_f32: [2H] values
 * (stat: mean, invstd of this call -- what the backward reads) + the per-block partial sums of the fast path (hidden, out <= 16) */
size_t coskad_mlp_head_ws_floats(int B, int H, int L) {
  size_t n = 2 * (size_t)H + 2;
  if (H <= mlp::HP && L <= mlp::HP && B > 0) {
    const size_t nblk = (size_t)ceil_div(B, mlp::RB);
    n += nblk * (2 * mlp::FWD_PART > mlp::BWD_PART ? 2 * mlp::FWD_PART : mlp::BWD_PART);
  }
  return n;
}

/* stat, red: coskad_mlp_head_ws_floats(B, H, L) floats, 8-byte aligned */
int coskad_mlp_head_fwd_f32(const float* y1, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float momentum, float eps, int training,
                            const float* W2, const float* b2, float* z, float* stat, int B, int H, int L,
                            hipStream_t stream) {
  if (!y1 || !gamma || !beta || !W2 || !z || !stat) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: null pointer");
  if (!training && (!running_mean || !running_var)) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: eval mode needs the running statistics");
  if (B <= 0 || H <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: B=%d H=%d L=%d", B, H, L);
  if (H > mlp::HMAX || L > mlp::HMAX) return fail(COSKAD_ERR_SHAPE, "mlp_head_fwd: hidden=%d / out=%d > %d not supported", H, L, mlp::HMAX);
  int rc;
  if (H <= mlp::HP && L <= mlp::HP) {
    if ((size_t)stat & 7) return fail(COSKAD_ERR_ARG, "mlp_head_fwd: stat must be 8-byte aligned");
    const int nblk = ceil_div(B, mlp::RB);
    double* part = reinterpret_cast<double*>(stat + (2 * H + 2) / 2 * 2);
    if (training) {
      hipLaunchKernelGGL(mlp::k_stats16, dim3(nblk), dim3(256), 0, stream, y1, part, B, H);
      if ((rc = check_launch("mlp_stats16"))) return rc;
    }
    hipLaunchKernelGGL(mlp::k_apply16, dim3(nblk), dim3(256), 0, stream, y1, part, nblk, stat, running_mean, running_var,
                       num_batches_tracked, momentum, eps, training, gamma, beta, W2, b2, z, B, H, L);
    return check_launch("mlp_apply16");
  }
  hipLaunchKernelGGL(mlp::k_stats, dim3(H), dim3(256), 0, stream, y1, stat, running_mean, running_var, num_batches_tracked,
                     momentum, eps, training, B, H);
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
  int rc;
  if (H <= mlp::HP && L <= mlp::HP) {
    const int nblk = ceil_div(B, mlp::RB);
    float* part = red + (2 * H + 2) / 2 * 2;
    hipLaunchKernelGGL(mlp::k_bwd_reduce16, dim3(nblk), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, part, B, H, L);
    if ((rc = check_launch("mlp_bwd_reduce16"))) return rc;
    hipLaunchKernelGGL(mlp::k_bwd_apply16, dim3(nblk), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, part, nblk, dy1, dgamma,
                       dbeta, dW2, db2, training, accumulate, B, H, L);
    return check_launch("mlp_bwd_apply16");
  }
  hipLaunchKernelGGL(mlp::k_bwd_reduce, dim3(H + 1), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, red, dgamma, dbeta,
                     dW2, db2, accumulate, B, H, L);
  if ((rc = check_launch("mlp_bwd_reduce"))) return rc;
  hipLaunchKernelGGL(mlp::k_bwd_apply, dim3(ceil_div(B, 256)), dim3(256), 0, stream, y1, stat, gamma, beta, W2, dz, red, dy1,
                     training, B, H, L);
  return check_launch("mlp_bwd_apply");
}

}  // extern "C"
