// The statistics algebra of coskad_amd/lowrank.py (rev_btlnk + the decoder's first layer folded into K = latent + 1 images) as two
// small kernels instead of ~35 element-wise torch launches each way.  One workgroup per output channel c; X [2, K, Co, TV] holds the
// rank-K tensors of the two BatchNorm branches (P_l = Wt gcn(Hb_l), Q_l = Wr Hb_l; reference models/graph_layers/stsgcn.py:106-110
// behind models/sts/ae.py:223-227), G [K, K] = sum_n zt zt^T (fp64) the latents' Gram matrix:
//
//   forward   mean_rc = 1/N sum_l G[l, K-1] sum_p X_rl[c, p];   E2_rc = 1/N sum_lm G[l, m] sum_p X_rl[c, p] X_rm[c, p]   (fp64)
//             a_rc = gamma_rc / sqrt(E2 - mean^2 + eps);  running statistics (momentum, unbiased variance, + the conv bias on the mean)
//             M_l[c, p] = sum_r a_rc X_rl[c, p]  (+ sum_r (beta_rc - a_rc mean_rc) on the constant image l = K - 1)
//             -> written in the rev_btlnk kernels' weight layout: Mw [Co TV, K - 1], Mb [Co TV]
//   backward  from dMw, dMb: d gamma, d beta, dX_rl = a_r dM_l + dmean_r G[l, K-1] / N + 2 dvar_r / N sum_m G[l, m] X_rm,
//             and this channel's share of dG
// K = 9 (latent 8: what the reference's autoencoder / VAE configs use); other latent sizes keep the torch algebra.
#include "common.h"

namespace coskad {
namespace lrf {

constexpr int K = 9, NP = K * (K + 1) / 2, kThreads = 256, MAXPOS = 2;   // positions per thread: T V <= 512

__device__ __forceinline__ double wave_total_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block sum of NV doubles per thread -> out[NV] in LDS (valid for every thread after the call)
template <int NV>
__device__ __forceinline__ void block_sum(const double (&v)[NV], double (*red)[NV], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double t = wave_total_d(v[i]);
    if (lane == 0) red[wave][i] = t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += kThreads) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  __syncthreads();
}

struct BnPtrs {
  const float* gamma[2];
  const float* beta[2];
  const float* cbias[2];      // conv biases in front of the BatchNorms (NULL: none): they only shift the running means
  float* rmean[2];
  float* rvar[2];
  long long* nbt[2];
  float momentum[2];
  float eps[2];
};

__global__ __launch_bounds__(kThreads) void k_fold_fwd(const float* __restrict__ X, const double* __restrict__ G, BnPtrs bn, double n_pos,
                                                        float* __restrict__ Mw, float* __restrict__ Mb, double* __restrict__ saved,
                                                        double* __restrict__ xbar_out, double* __restrict__ xx_out, int Co, int TV) {
  // saved [3][2][Co] = mean, istd, a;  xbar_out [2][K][Co];  xx_out [2][Co][K][K]
  constexpr int NV = 2 * (K + NP);
  __shared__ double red[kThreads / 64][NV];
  __shared__ double tot[NV];
  __shared__ double gl[K * K];
  __shared__ float coef[3];          // a_0, a_1, shift sum
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < K * K; e += kThreads) gl[e] = G[e];
  float x[MAXPOS][2][K];
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = 0.0;
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int p = tid + kThreads * j;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int l = 0; l < K; ++l) x[j][r][l] = p < TV ? X[(((size_t)r * K + l) * Co + c) * TV + p] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int q = 0;
#pragma unroll
      for (int l = 0; l < K; ++l) {
        v[r * (K + NP) + l] += (double)x[j][r][l];
#pragma unroll
        for (int m = 0; m <= l; ++m) v[r * (K + NP) + K + q++] += (double)x[j][r][l] * (double)x[j][r][m];
      }
    }
  }
  block_sum<NV>(v, red, tot);
  if (tid < 2) {
    const int r = tid;
    const double* sx = tot + r * (K + NP);
    const double* sxx = sx + K;
    double mean = 0.0, e2 = 0.0;
    int q = 0;
    for (int l = 0; l < K; ++l) {
      mean += gl[l * K + K - 1] * sx[l];
      xbar_out[((size_t)r * K + l) * Co + c] = sx[l];
      for (int m = 0; m <= l; ++m) {
        const double t = sxx[q++];
        e2 += (l == m ? gl[l * K + m] : gl[l * K + m] + gl[m * K + l]) * t;
        xx_out[(((size_t)r * Co + c) * K + l) * K + m] = t;
        xx_out[(((size_t)r * Co + c) * K + m) * K + l] = t;
      }
    }
    mean /= n_pos;
    e2 /= n_pos;
    double var = e2 - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double istd = 1.0 / sqrt(var + (double)bn.eps[r]);
    const double a = (double)bn.gamma[r][c] * istd;
    saved[(0 * 2 + r) * Co + c] = mean;
    saved[(1 * 2 + r) * Co + c] = istd;
    saved[(2 * 2 + r) * Co + c] = a;
    if (bn.rmean[r]) {                 // nn.BatchNorm2d: momentum average, unbiased variance; the conv bias shifts the mean only
      const double unb = n_pos > 1.0 ? n_pos / (n_pos - 1.0) : 1.0;
      const float mom = bn.momentum[r];
      const float mb = (float)mean + (bn.cbias[r] ? bn.cbias[r][c] : 0.f);
      bn.rmean[r][c] = (1.f - mom) * bn.rmean[r][c] + mom * mb;
      bn.rvar[r][c] = (1.f - mom) * bn.rvar[r][c] + mom * (float)(var * unb);
      if (c == 0 && bn.nbt[r]) bn.nbt[r][0] += 1;
    }
    red[0][r] = (double)bn.beta[r][c] - a * mean;      // (red is free: block_sum's readers passed its last barrier)
    coef[r] = (float)a;
  }
  __syncthreads();
  if (tid == 0) coef[2] = (float)(red[0][0] + red[0][1]);
  __syncthreads();
  const float a0 = coef[0], a1 = coef[1], sh = coef[2];
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int p = tid + kThreads * j;
    if (p < TV) {
      float* dst = Mw + ((size_t)c * TV + p) * (K - 1);
#pragma unroll
      for (int l = 0; l < K - 1; ++l) dst[l] = a0 * x[j][0][l] + a1 * x[j][1][l];
      Mb[(size_t)c * TV + p] = a0 * x[j][0][K - 1] + a1 * x[j][1][K - 1] + sh;
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_fold_bwd(const float* __restrict__ X, const double* __restrict__ G, const float* __restrict__ dMw,
                                                        const float* __restrict__ dMb, const double* __restrict__ saved,
                                                        const double* __restrict__ xbar, const double* __restrict__ xx,
                                                        const float* __restrict__ gamma0, const float* __restrict__ gamma1, double n_pos,
                                                        float* __restrict__ dX, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        double* __restrict__ dGc, int Co, int TV) {
  // dgamma [2][Co], dbeta [Co] (both BatchNorms' d beta), dGc [Co][K][K]: this channel's share of dG
  __shared__ double red[kThreads / 64][3];
  __shared__ double tot[3];
  __shared__ float gl[K * K];
  __shared__ float cf[2][3];         // per branch: a, dmean / N, 2 dvar / N
  __shared__ double dsh[2][2];       // per branch: dmean, dvar (fp64, for dG)
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int e = tid; e < K * K; e += kThreads) gl[e] = (float)G[e];
  float x[MAXPOS][2][K], dm[MAXPOS][K];
  double v[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int p = tid + kThreads * j;
    const bool ok = p < TV;
#pragma unroll
    for (int l = 0; l < K - 1; ++l) dm[j][l] = ok ? dMw[((size_t)c * TV + p) * (K - 1) + l] : 0.f;
    dm[j][K - 1] = ok ? dMb[(size_t)c * TV + p] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int l = 0; l < K; ++l) x[j][r][l] = ok ? X[(((size_t)r * K + l) * Co + c) * TV + p] : 0.f;
    v[0] += (double)dm[j][K - 1];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float s = 0.f;
#pragma unroll
      for (int l = 0; l < K; ++l) s = fmaf(dm[j][l], x[j][r][l], s);
      v[1 + r] += (double)s;
    }
  }
  block_sum<3>(v, red, tot);
  if (tid < 2) {
    const int r = tid;
    const double mean = saved[(0 * 2 + r) * Co + c], istd = saved[(1 * 2 + r) * Co + c], a = saved[(2 * 2 + r) * Co + c];
    const double S1 = tot[0], da = tot[1 + r] - mean * S1;       // M = a X (+ shift = beta - a mean on the constant image)
    const double g = (double)(r ? gamma1 : gamma0)[c];
    const double dvar = -0.5 * da * g * istd * istd * istd;      // a = gamma (var + eps)^(-1/2)
    const double dmean = -a * S1 - 2.0 * mean * dvar;            // var = E[x^2] - mean^2
    dgamma[r * Co + c] = (float)(da * istd);
    if (r == 0) dbeta[c] = (float)S1;
    cf[r][0] = (float)a;
    cf[r][1] = (float)(dmean / n_pos);
    cf[r][2] = (float)(2.0 * dvar / n_pos);
    dsh[r][0] = dmean;
    dsh[r][1] = dvar;
  }
  __syncthreads();
  // this channel's share of dG: dG[l][m] += dvar_r XX_r[l][m] / N;  dG[l][K-1] += dmean_r xbar_r[l] / N
  for (int e = tid; e < K * K; e += kThreads) {
    const int l = e / K, m = e - l * K;
    double t = 0.0;
    for (int r = 0; r < 2; ++r) {
      t += dsh[r][1] * xx[(((size_t)r * Co + c) * K + l) * K + m];
      if (m == K - 1) t += dsh[r][0] * xbar[((size_t)r * K + l) * Co + c];
    }
    dGc[((size_t)c * K + l) * K + m] = t / n_pos;
  }
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int p = tid + kThreads * j;
    if (p < TV) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int l = 0; l < K; ++l) {
          float gx = 0.f;
#pragma unroll
          for (int m = 0; m < K; ++m) gx = fmaf(gl[l * K + m], x[j][r][m], gx);
          dX[(((size_t)r * K + l) * Co + c) * TV + p] = cf[r][0] * dm[j][l] + cf[r][1] * gl[l * K + K - 1] + cf[r][2] * gx;
        }
    }
  }
}

}  // namespace lrf
}  // namespace coskad

using namespace coskad;

extern "C" {

/* 1 when the fold kernels take the shape: latent 8 (K = 9 images), T V <= 512 */
int coskad_lowrank_fold_ok(int latent, int TV) { return latent == lrf::K - 1 && TV > 0 && TV <= lrf::kThreads * lrf::MAXPOS; }

/* X [2, 9, Co, TV] (branch 0: the tcn BatchNorm's rank-9 tensor, 1: the residual one's), G [9, 9] fp64; per branch r: gamma / beta
 * [Co], conv bias [Co] or NULL, running mean / var [Co] and num_batches_tracked or NULL, momentum, eps.  Out: Mw [Co TV, 8], Mb [Co TV];
 * saved [3][2][Co], xbar [2][9][Co], xx [2][Co][9][9] (fp64: what coskad_lowrank_fold_bwd_f32 reads back). */
int coskad_lowrank_fold_fwd_f32(const float* X, const double* G, const float* gamma0, const float* beta0, const float* cbias0,
                                float* rmean0, float* rvar0, long long* nbt0, float momentum0, float eps0, const float* gamma1,
                                const float* beta1, const float* cbias1, float* rmean1, float* rvar1, long long* nbt1, float momentum1,
                                float eps1, double n_pos, float* Mw, float* Mb, double* saved, double* xbar, double* xx, int Co, int TV,
                                hipStream_t stream) {
  if (!X || !G || !gamma0 || !beta0 || !gamma1 || !beta1 || !Mw || !Mb || !saved || !xbar || !xx)
    return fail(COSKAD_ERR_ARG, "lowrank_fold_fwd: null pointer");
  if (Co <= 0 || TV <= 0 || TV > lrf::kThreads * lrf::MAXPOS || n_pos <= 0) return fail(COSKAD_ERR_SHAPE, "lowrank_fold_fwd: Co=%d TV=%d", Co, TV);
  if ((rmean0 == nullptr) != (rvar0 == nullptr) || (rmean1 == nullptr) != (rvar1 == nullptr))
    return fail(COSKAD_ERR_ARG, "lowrank_fold_fwd: running mean and variance come together");
  lrf::BnPtrs bn;
  bn.gamma[0] = gamma0; bn.gamma[1] = gamma1; bn.beta[0] = beta0; bn.beta[1] = beta1; bn.cbias[0] = cbias0; bn.cbias[1] = cbias1;
  bn.rmean[0] = rmean0; bn.rmean[1] = rmean1; bn.rvar[0] = rvar0; bn.rvar[1] = rvar1; bn.nbt[0] = nbt0; bn.nbt[1] = nbt1;
  bn.momentum[0] = momentum0; bn.momentum[1] = momentum1; bn.eps[0] = eps0; bn.eps[1] = eps1;
  hipLaunchKernelGGL(lrf::k_fold_fwd, dim3(Co), dim3(lrf::kThreads), 0, stream, X, G, bn, n_pos, Mw, Mb, saved, xbar, xx, Co, TV);
  return check_launch("lowrank_fold_fwd");
}

/* dX [2, 9, Co, TV], dgamma [2][Co], dbeta [Co] (the same for both BatchNorms), dGc [Co][9][9] (sum over Co = dG) */
int coskad_lowrank_fold_bwd_f32(const float* X, const double* G, const float* dMw, const float* dMb, const double* saved,
                                const double* xbar, const double* xx, const float* gamma0, const float* gamma1, double n_pos, float* dX,
                                float* dgamma, float* dbeta, double* dGc, int Co, int TV, hipStream_t stream) {
  if (!X || !G || !dMw || !dMb || !saved || !xbar || !xx || !gamma0 || !gamma1 || !dX || !dgamma || !dbeta || !dGc)
    return fail(COSKAD_ERR_ARG, "lowrank_fold_bwd: null pointer");
  if (Co <= 0 || TV <= 0 || TV > lrf::kThreads * lrf::MAXPOS || n_pos <= 0) return fail(COSKAD_ERR_SHAPE, "lowrank_fold_bwd: Co=%d TV=%d", Co, TV);
  hipLaunchKernelGGL(lrf::k_fold_bwd, dim3(Co), dim3(lrf::kThreads), 0, stream, X, G, dMw, dMb, saved, xbar, xx, gamma0, gamma1, n_pos,
                     dX, dgamma, dbeta, dGc, Co, TV);
  return check_launch("lowrank_fold_bwd");
}

}  // extern "C"
