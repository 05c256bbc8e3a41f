// BatchNorm2d + residual add + PReLU of an ST_GCNN layer (reference models/graph_layers/stsgcn.py:56-80,106-110) as
// stand-alone kernels on [N, C, P] tensors, for layers whose channel counts are beyond the LDS-resident tile kernels
// (C > 64: the `C = 2 -> 256` stack of BASELINE.json's north_star; 64 input channels on the 25-joint layout).  The 1x1
// convolutions of those layers are GEMMs on csrc/gemm.hip, the space-time mixing is coskad_gcn_f32; together they replace
// the convolution / BatchNorm library calls of the composed path, all in the native NCHW layout.
//
//   out = PReLU( st * Ct + ht  +  sr * Cr + hr )        st = gamma_t * invstd_t, ht = beta_t - st * mean_t (likewise r;
//                                                        identity residual: sr = 1, hr = 0, no statistics)
// Reductions over (N, P) run as (channel, slice) blocks with fp64 partials summed in a fixed order (deterministic).
#include "common.h"

namespace coskad {
namespace wide {

// Train-mode nn.Dropout(p) of the tcn branch (stsgcn.py:66): a counter-based mask -- element i of the [N, C, P] tensor is kept
// iff a 24-bit hash of (seed, i) is >= p, kept values are scaled by 1 / (1 - p).  Nothing is stored: forward and backward
// recompute it from the seed.  p == 0: no dropout.
__host__ __device__ __forceinline__ float drop_mask(unsigned long long seed, unsigned long long i, float p, float scale) {
  unsigned long long x = i * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
  return u >= p ? scale : 0.f;
}

__global__ void k_drop_mask(float* __restrict__ out, size_t n, unsigned long long seed, float p, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = p > 0.f ? drop_mask(seed, i, p, scale) : 1.f;
}

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

// part[(sl * C + c) * 2 + {0,1}] = sum, sum of squares of x[:, c, :] over the clips n = sl, sl + S, ...
__global__ __launch_bounds__(256) void k_stats_part(const float* __restrict__ x, double* __restrict__ part, int Nb, int C, int P,
                                                    int S) {
  __shared__ double sh[256];
  const int c = blockIdx.x, sl = blockIdx.y;
  double s = 0.0, q = 0.0;
  for (int n = sl; n < Nb; n += S) {
    const float* row = x + ((size_t)n * C + c) * P;
    for (int p = threadIdx.x; p < P; p += 256) {
      const double v = row[p];
      s += v;
      q += v * v;
    }
  }
  const double ts = block_sum(s, sh), tq = block_sum(q, sh);
  if (threadIdx.x == 0) {
    part[((size_t)sl * C + c) * 2] = ts;
    part[((size_t)sl * C + c) * 2 + 1] = tq;
  }
}

// stat[c] = mean, stat[C + c] = invstd; running statistics updated (unbiased variance), like nn.BatchNorm2d in train mode
__global__ void k_stats_final(const double* __restrict__ part, int S, float* __restrict__ stat, float* __restrict__ rmean,
                              float* __restrict__ rvar, long long* __restrict__ nbt, float momentum, float eps, int training,
                              double count, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (!training) {
    stat[c] = rmean[c];
    stat[C + c] = 1.f / sqrtf(rvar[c] + eps);
    return;
  }
  double s = 0.0, q = 0.0;
  for (int k = 0; k < S; ++k) {
    s += part[((size_t)k * C + c) * 2];
    q += part[((size_t)k * C + c) * 2 + 1];
  }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stat[c] = (float)mean;
  stat[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
  if (c == 0 && nbt) *nbt += 1;
}

// out = PReLU(st*Ct + ht + sr*Cr + hr); one block row per (clip, channel) row of P floats
__global__ __launch_bounds__(256) void k_apply(const float* __restrict__ Ct, const float* __restrict__ Cr,
                                               const float* __restrict__ stat_t, const float* __restrict__ gt,
                                               const float* __restrict__ bt, const float* __restrict__ stat_r,
                                               const float* __restrict__ gr, const float* __restrict__ br,
                                               const float* __restrict__ slope, float* __restrict__ out, size_t rows, int C,
                                               int P, float drop_p, unsigned long long seed) {
  const float a = slope[0];
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int c = (int)(row % C);
    const float st = gt[c] * stat_t[C + c], ht = bt[c] - st * stat_t[c];
    float sr = 1.f, hr = 0.f;
    if (stat_r) { sr = gr[c] * stat_r[C + c]; hr = br[c] - sr * stat_r[c]; }
    const float* ct = Ct + row * P;
    const float* cr = Cr + row * P;
    float* o = out + row * P;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float m = drop_p > 0.f ? drop_mask(seed, row * P + p, drop_p, dscale) : 1.f;
      const float u = m * fmaf(st, ct[p], ht) + fmaf(sr, cr[p], hr);
      o[p] = u > 0.f ? u : a * u;
    }
  }
}

// backward reductions per channel: part[(sl*C + c)*5 + {0..4}] = sum dU, sum m*dU*Ct, sum dU*Cr, sum dOut*U[U<0], sum m*dU
// (m: the dropout mask of the tcn branch, 1 without dropout)
__global__ __launch_bounds__(256) void k_bwd_part(const float* __restrict__ Ct, const float* __restrict__ Cr,
                                                  const float* __restrict__ dOut, const float* __restrict__ stat_t,
                                                  const float* __restrict__ gt, const float* __restrict__ bt,
                                                  const float* __restrict__ stat_r, const float* __restrict__ gr,
                                                  const float* __restrict__ br, const float* __restrict__ slope,
                                                  double* __restrict__ part, int Nb, int C, int P, int S, float drop_p,
                                                  unsigned long long seed) {
  __shared__ double sh[256];
  const int c = blockIdx.x, sl = blockIdx.y;
  const float a = slope[0];
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const float st = gt[c] * stat_t[C + c], ht = bt[c] - st * stat_t[c];
  float sr = 1.f, hr = 0.f;
  if (stat_r) { sr = gr[c] * stat_r[C + c]; hr = br[c] - sr * stat_r[c]; }
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
  for (int n = sl; n < Nb; n += S) {
    const size_t base = ((size_t)n * C + c) * P;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float ct = Ct[base + p], cr = Cr[base + p], g = dOut[base + p];
      const float m = drop_p > 0.f ? drop_mask(seed, base + p, drop_p, dscale) : 1.f;
      const float u = m * fmaf(st, ct, ht) + fmaf(sr, cr, hr);
      const float du = u > 0.f ? g : a * g;
      s0 += (double)du;
      s1 += (double)(m * du) * (double)ct;
      s2 += (double)du * (double)cr;
      if (u < 0.f) s3 += (double)g * (double)u;
      s4 += (double)(m * du);
    }
  }
  const double t0 = block_sum(s0, sh), t1 = block_sum(s1, sh), t2 = block_sum(s2, sh), t3 = block_sum(s3, sh), t4 = block_sum(s4, sh);
  if (threadIdx.x == 0) {
    double* o = part + ((size_t)sl * C + c) * 5;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
  }
}

// coef[c*6 + ..] = {kt_du, kt_c, kt_0, kr_du, kr_c, kr_0}: dCt = kt_du*dU + kt_c*Ct + kt_0 (likewise r); parameter gradients
__global__ void k_bwd_final(const double* __restrict__ part, int S, const float* __restrict__ stat_t,
                            const float* __restrict__ gt, const float* __restrict__ stat_r, const float* __restrict__ gr,
                            float* __restrict__ coef, float* __restrict__ dgt, float* __restrict__ dbt,
                            float* __restrict__ dgr, float* __restrict__ dbr, double* __restrict__ dslope_part, int training,
                            double count, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s0r = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s0t = 0.0;
  for (int k = 0; k < S; ++k) {
    const double* o = part + ((size_t)k * C + c) * 5;
    s0r += o[0]; s1 += o[1]; s2 += o[2]; s3 += o[3]; s0t += o[4];
  }
  dslope_part[c] = s3;
  auto branch = [&](const float* stat, const float* g, double s0, double sduc, float* dg, float* db, float* k) {
    const double mean = stat[c], inv = stat[C + c], gam = g[c];
    const double sxh = (sduc - mean * s0) * inv;           // sum dU * xhat
    dg[c] = (float)sxh;
    db[c] = (float)s0;
    if (training) {                                        // dC = gam*inv * (dU - s0/n - xhat * sxh/n),  xhat = (C - mean)*inv
      const double f = gam * inv;
      k[0] = (float)f;
      k[1] = (float)(-f * inv * sxh / count);
      k[2] = (float)(-f * s0 / count + f * inv * mean * sxh / count);
    } else {
      k[0] = (float)(gam * inv); k[1] = 0.f; k[2] = 0.f;
    }
  };
  branch(stat_t, gt, s0t, s1, dgt, dbt, coef + c * 6);
  if (stat_r) branch(stat_r, gr, s0r, s2, dgr, dbr, coef + c * 6 + 3);
  else { coef[c * 6 + 3] = 1.f; coef[c * 6 + 4] = 0.f; coef[c * 6 + 5] = 0.f; }
}

__global__ void k_sum_d(const double* __restrict__ v, int n, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    out[0] = (float)s;
  }
}

// dCt, dCr from dOut (recomputes U for the PReLU derivative)
__global__ __launch_bounds__(256) void k_bwd_apply(const float* __restrict__ Ct, const float* __restrict__ Cr,
                                                   const float* __restrict__ dOut, const float* __restrict__ stat_t,
                                                   const float* __restrict__ gt, const float* __restrict__ bt,
                                                   const float* __restrict__ stat_r, const float* __restrict__ gr,
                                                   const float* __restrict__ br, const float* __restrict__ slope,
                                                   const float* __restrict__ coef, float* __restrict__ dCt,
                                                   float* __restrict__ dCr, size_t rows, int C, int P, float drop_p,
                                                   unsigned long long seed) {
  const float a = slope[0];
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int c = (int)(row % C);
    const float st = gt[c] * stat_t[C + c], ht = bt[c] - st * stat_t[c];
    float sr = 1.f, hr = 0.f;
    if (stat_r) { sr = gr[c] * stat_r[C + c]; hr = br[c] - sr * stat_r[c]; }
    const float* k = coef + c * 6;
    const size_t base = row * P;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float ct = Ct[base + p], cr = Cr[base + p], g = dOut[base + p];
      const float m = drop_p > 0.f ? drop_mask(seed, base + p, drop_p, dscale) : 1.f;
      const float u = m * fmaf(st, ct, ht) + fmaf(sr, cr, hr);
      const float du = u > 0.f ? g : a * g;
      dCt[base + p] = fmaf(k[0], m * du, fmaf(k[1], ct, k[2]));
      dCr[base + p] = fmaf(k[3], du, fmaf(k[4], cr, k[5]));
    }
  }
}

}  // namespace wide
}  // namespace coskad

using namespace coskad;

extern "C" {

static int wide_slices(int Nb) { return Nb < 64 ? Nb : 64; }

size_t coskad_bn2_ws_bytes(int Nb, int C) { return (size_t)wide_slices(Nb > 0 ? Nb : 1) * C * 5 * sizeof(double) + (size_t)C * sizeof(double); }

/* per-channel statistics of x [Nb, C, P]: stat [2C] = (mean, 1/sqrt(var + eps)); training: batch statistics + running update */
int coskad_bn2_stats_f32(const float* x, float* stat, float* running_mean, float* running_var, long long* num_batches_tracked,
                         float momentum, float eps, int training, void* ws, size_t ws_bytes, int Nb, int C, int P,
                         hipStream_t stream) {
  if (!x || !stat || !ws) return fail(COSKAD_ERR_ARG, "bn2_stats: null pointer");
  if (!training && (!running_mean || !running_var)) return fail(COSKAD_ERR_ARG, "bn2_stats: eval mode needs running statistics");
  if (Nb <= 0 || C <= 0 || P <= 0 || C > 65535) return fail(COSKAD_ERR_ARG, "bn2_stats: Nb=%d C=%d P=%d", Nb, C, P);
  if (ws_bytes < coskad_bn2_ws_bytes(Nb, C)) return fail(COSKAD_ERR_WORKSPACE, "bn2_stats: workspace too small");
  const int S = wide_slices(Nb);
  double* part = reinterpret_cast<double*>(ws);
  int rc;
  if (training) {
    hipLaunchKernelGGL(wide::k_stats_part, dim3(C, S), dim3(256), 0, stream, x, part, Nb, C, P, S);
    if ((rc = check_launch("bn2_stats_part"))) return rc;
  }
  hipLaunchKernelGGL(wide::k_stats_final, dim3(ceil_div(C, 256)), dim3(256), 0, stream, part, S, stat, running_mean, running_var,
                     num_batches_tracked, momentum, eps, training, (double)Nb * P, C);
  return check_launch("bn2_stats_final");
}

/* out = PReLU(BN_t(Ct) + BN_r(Cr)); stat_r == NULL: identity residual (out = PReLU(BN_t(Ct) + Cr)) */
int coskad_bn2_apply_prelu_f32(const float* Ct, const float* Cr, const float* stat_t, const float* gamma_t, const float* beta_t,
                               const float* stat_r, const float* gamma_r, const float* beta_r, const float* slope, float* out,
                               int Nb, int C, int P, hipStream_t stream, float drop_p, unsigned long long drop_seed) {
  if (!Ct || !Cr || !stat_t || !gamma_t || !beta_t || !slope || !out) return fail(COSKAD_ERR_ARG, "bn2_apply: null pointer");
  if (stat_r && (!gamma_r || !beta_r)) return fail(COSKAD_ERR_ARG, "bn2_apply: residual affine missing");
  if (Nb <= 0 || C <= 0 || P <= 0) return fail(COSKAD_ERR_ARG, "bn2_apply: bad sizes");
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(COSKAD_ERR_ARG, "bn2_apply: dropout probability %g outside [0, 1)", (double)drop_p);
  const size_t rows = (size_t)Nb * C;
  hipLaunchKernelGGL(wide::k_apply, dim3((unsigned)(rows < 65536 ? rows : 65536)), dim3(256), 0, stream, Ct, Cr, stat_t, gamma_t,
                     beta_t, stat_r, gamma_r, beta_r, slope, out, rows, C, P, drop_p, drop_seed);
  return check_launch("bn2_apply");
}

/* backward of coskad_bn2_apply_prelu_f32 (+ both BatchNorms): dCt, dCr [Nb,C,P]; dgamma / dbeta of both branches [C]
 * (residual ones may be NULL with stat_r == NULL); dslope [1] */
int coskad_bn2_bwd_f32(const float* Ct, const float* Cr, const float* dOut, const float* stat_t, const float* gamma_t,
                       const float* beta_t, const float* stat_r, const float* gamma_r, const float* beta_r, const float* slope,
                       float* dCt, float* dCr, float* dgamma_t, float* dbeta_t, float* dgamma_r, float* dbeta_r, float* dslope,
                       int training, void* ws, size_t ws_bytes, int Nb, int C, int P, hipStream_t stream, float drop_p,
                       unsigned long long drop_seed) {
  if (!Ct || !Cr || !dOut || !stat_t || !gamma_t || !beta_t || !slope || !dCt || !dCr || !dgamma_t || !dbeta_t || !dslope || !ws)
    return fail(COSKAD_ERR_ARG, "bn2_bwd: null pointer");
  if (stat_r && (!gamma_r || !beta_r || !dgamma_r || !dbeta_r)) return fail(COSKAD_ERR_ARG, "bn2_bwd: residual tensors missing");
  if (Nb <= 0 || C <= 0 || P <= 0 || C > 65535) return fail(COSKAD_ERR_ARG, "bn2_bwd: bad sizes");
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(COSKAD_ERR_ARG, "bn2_bwd: dropout probability %g outside [0, 1)", (double)drop_p);
  const size_t need = coskad_bn2_ws_bytes(Nb, C) + (size_t)C * 6 * sizeof(float);
  if (ws_bytes < need) return fail(COSKAD_ERR_WORKSPACE, "bn2_bwd: workspace %zu < %zu", ws_bytes, need);
  const int S = wide_slices(Nb);
  double* part = reinterpret_cast<double*>(ws);
  double* dsl = part + (size_t)S * C * 5;
  float* coef = reinterpret_cast<float*>(dsl + C);
  int rc;
  hipLaunchKernelGGL(wide::k_bwd_part, dim3(C, S), dim3(256), 0, stream, Ct, Cr, dOut, stat_t, gamma_t, beta_t, stat_r, gamma_r,
                     beta_r, slope, part, Nb, C, P, S, drop_p, drop_seed);
  if ((rc = check_launch("bn2_bwd_part"))) return rc;
  hipLaunchKernelGGL(wide::k_bwd_final, dim3(ceil_div(C, 256)), dim3(256), 0, stream, part, S, stat_t, gamma_t, stat_r, gamma_r,
                     coef, dgamma_t, dbeta_t, dgamma_r, dbeta_r, dsl, training, (double)Nb * P, C);
  if ((rc = check_launch("bn2_bwd_final"))) return rc;
  hipLaunchKernelGGL(wide::k_sum_d, dim3(1), dim3(64), 0, stream, dsl, C, dslope);
  if ((rc = check_launch("bn2_bwd_slope"))) return rc;
  const size_t rows = (size_t)Nb * C;
  hipLaunchKernelGGL(wide::k_bwd_apply, dim3((unsigned)(rows < 65536 ? rows : 65536)), dim3(256), 0, stream, Ct, Cr, dOut, stat_t,
                     gamma_t, beta_t, stat_r, gamma_r, beta_r, slope, coef, dCt, dCr, rows, C, P, drop_p, drop_seed);
  return check_launch("bn2_bwd_apply");
}

/* the mask the two calls above apply to the tcn branch for (drop_p, drop_seed): out[i] in {0, 1 / (1 - p)} (tests, oracles) */
int coskad_dropout_mask_f32(float* out, size_t n, float drop_p, unsigned long long drop_seed, hipStream_t stream) {
  if (!out || n == 0) return fail(COSKAD_ERR_ARG, "dropout_mask: null pointer / empty");
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(COSKAD_ERR_ARG, "dropout_mask: probability %g outside [0, 1)", (double)drop_p);
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(wide::k_drop_mask, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, out, n, drop_seed,
                     drop_p, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f);
  return check_launch("dropout_mask");
}

size_t coskad_bn2_bwd_ws_bytes(int Nb, int C) { return coskad_bn2_ws_bytes(Nb, C) + (size_t)C * 6 * sizeof(float); }

}  // extern "C"
