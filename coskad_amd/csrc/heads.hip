// One-class heads on the latent z [B, L] (L <= 16), their gradients, the centre statistics,
// the L2 regulariser and the optimiser step.
//
//   Euclidean : F.mse_loss(z, c)                 (reference euclidean_encoder_staticCenter.py:187,
//                                                  euclidean_encoder_dynamicCenter.py:116)
//   Poincare  : dist(c, project(expmap0(z))).mean()        (hyperbolic_encoder.py:147,157) with the
//               formulas and epsilons of utils/hyper_math.py (:13-29, 100-105, 173-179, 207-210,
//               302-306); geoopt itself is not in the tree (parity unpinned vs geoopt).
//   centre    : running sums for c = mean(z) (staticCenter.py:114-118,172-178) and for the
//               gyromidpoint  m = sum(gamma_i z_i) / sum(gamma_i - 1)  (hyperbolic_encoder.py:122,179)
//   reg       : utils/model_utils.py:90-105 (0.5 * sum ||p||^2 over non-bias tensors / #tensors)
//
// One thread per clip: a 16-float latent lives in registers; block partials are summed in a
// fixed order (deterministic).
#include "tile_ops.h"

namespace coskad {

constexpr int LMAX = 16;
constexpr float kMinNorm = 1e-5f;      // hyper_math.py:101,303
constexpr float kBallEps = 1e-3f;      // hyper_math.py:102
constexpr float kArtanhEps = 1e-5f;    // hyper_math.py:21
constexpr float kMobiusEps = 1e-5f;    // hyper_math.py:179
constexpr float kTanhClamp = 15.f;     // hyper_math.py:13

struct Vec {
  float v[LMAX];
};

__device__ __forceinline__ Vec load_vec(const float* p, int L) {
  Vec r;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) r.v[j] = j < L ? p[j] : 0.f;
  return r;
}
__device__ __forceinline__ void store_vec(float* p, const Vec& a, int L) {
#pragma unroll
  for (int j = 0; j < LMAX; ++j)
    if (j < L) p[j] = a.v[j];
}
__device__ __forceinline__ float dot(const Vec& a, const Vec& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) s = fmaf(a.v[j], b.v[j], s);
  return s;
}

// e = expmap0(u), p = project(e); returns p and the pieces the backward needs
struct Embed {
  Vec p;
  float un_raw, un, tn, en_raw;
  bool clamped;
};
__device__ __forceinline__ Embed hyp_embed(const Vec& u) {
  Embed r;
  r.un_raw = sqrtf(dot(u, u));
  r.un = fmaxf(r.un_raw, kMinNorm);
  r.tn = tanhf(fminf(fmaxf(r.un, -kTanhClamp), kTanhClamp));
  const float f = r.tn / r.un;
  Vec e;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) e.v[j] = f * u.v[j];
  r.en_raw = sqrtf(dot(e, e));
  const float en = fmaxf(r.en_raw, kMinNorm);
  const float maxnorm = 1.f - kBallEps;
  r.clamped = en > maxnorm;
  const float s = r.clamped ? maxnorm / en : 1.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) r.p.v[j] = r.clamped ? e.v[j] / en * maxnorm : e.v[j];
  (void)s;
  return r;
}

// d = dist(c, p) = 2 artanh(|(-c) (+) p|), and dd/dp
__device__ __forceinline__ float poincare_dist(const Vec& c, const Vec& p, Vec* gp) {
  const float x2 = dot(c, c), y2 = dot(p, p), xy = -dot(c, p);
  const float alpha = 1.f + 2.f * xy + y2, beta = 1.f - x2;
  const float den = 1.f + 2.f * xy + x2 * y2;
  const float D = den + kMobiusEps;
  Vec m;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) m.v[j] = (alpha * (-c.v[j]) + beta * p.v[j]) / D;
  const float mn = sqrtf(dot(m, m));
  const float mc = fminf(fmaxf(mn, -1.f + kArtanhEps), 1.f - kArtanhEps);
  const float d = (log1pf(mc) - log1pf(-mc));  // 2 * 0.5 * (log(1+x) - log(1-x))
  if (gp) {
    const float gmn = 2.f / (1.f - mc * mc);      // Artanh.backward on the clamped input, times 2
    const float inv = mn > 0.f ? gmn / mn : 0.f;  // d|m|/dm = m/|m| (0 at m = 0)
    Vec gm;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) gm.v[j] = inv * m.v[j];
    // m = num / D,  num = alpha x + beta y,  x = -c, y = p
    const float gD = -dot(gm, m) / D;
    float galpha = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) galpha = fmaf(gm.v[j] / D, -c.v[j], galpha);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
      const float x = -c.v[j], y = p.v[j];
      gp->v[j] = beta * gm.v[j] / D + galpha * (2.f * x + 2.f * y) + gD * (2.f * x + 2.f * x2 * y);
    }
  }
  return d;
}

// chain dL/dp back through project and expmap0 to dL/du
__device__ __forceinline__ Vec hyp_embed_bwd(const Vec& u, const Embed& em, const Vec& gp) {
  const float f = em.tn / em.un;
  Vec ge;
  if (em.clamped) {
    // p = maxnorm * e / |e|  ->  ge = maxnorm (gp/|e| - e (e.gp)/|e|^3)
    const float maxnorm = 1.f - kBallEps;
    const float en = em.en_raw;
    float egp = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) egp = fmaf(f * u.v[j], gp.v[j], egp);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) ge.v[j] = maxnorm * (gp.v[j] / en - f * u.v[j] * egp / (en * en * en));
  } else {
    ge = gp;
  }
  // e = f(un) u ; un = max(|u|, 1e-5): the norm carries gradient only when not clamped
  Vec gu;
  float gf = dot(ge, u);
  float coef = 0.f;
  if (em.un_raw > kMinNorm) {
    const float sech2 = em.un < kTanhClamp ? 1.f - em.tn * em.tn : 0.f;
    const float df = (sech2 * em.un - em.tn) / (em.un * em.un);
    coef = gf * df / em.un_raw;
  }
#pragma unroll
  for (int j = 0; j < LMAX; ++j) gu.v[j] = f * ge.v[j] + coef * u.v[j];
  return gu;
}

// Per-block partial row of kHeadSlots floats (fixed slots, summed in block order later):
//   [0] loss term   [1..16] vector sum   [17] scalar A   [18] scalar B
constexpr int kHeadSlots = LMAX + 3;

// A one-block launch (B <= kHeadSingleRows clips) writes the finished statistics itself -- no partial row, no k_head_finalize
// launch behind it.  Larger batches keep one row per thread: ONE block striding over 4096 rows measured 65 us against 12 + 5 us
// for eight blocks + k_head_finalize (eight dependent load rounds per thread).
constexpr int kHeadSingleRows = kFlatBlock;
__host__ __device__ inline int head_blocks(int B) { return B <= kHeadSingleRows ? 1 : ceil_div(B, kFlatBlock); }

__device__ __forceinline__ void block_partials(const float (&vals)[kHeadSlots], float* partials, float scale0, float* stats,
                                               float* acc) {
  __shared__ float sh[kFlatBlock / 64][kHeadSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < kHeadSlots; ++k) {
    const float s = wave_sum(vals[k]);
    if (lane == 0) sh[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < kHeadSlots) {
    float s = 0.f;
    for (int w = 0; w < kFlatBlock / 64; ++w) s += sh[w][threadIdx.x];
    if (gridDim.x == 1) {
      if (stats) stats[threadIdx.x] = threadIdx.x == 0 ? s * scale0 : s;
      if (acc) acc[threadIdx.x] += s;
    } else {
      partials[blockIdx.x * kHeadSlots + threadIdx.x] = s;
    }
  }
}

// Euclidean head.  slots: [0] sum (z-c)^2, [1..L] sum z, [17] #clips, [18] sum |z|
__global__ __launch_bounds__(kFlatBlock) void k_mse_head(const float* __restrict__ z,
                                                    const float* __restrict__ cvec,
                                                    float* __restrict__ dz, float* __restrict__ score,
                                                    float* __restrict__ partials, int B, int L,
                                                    float gscale, float scale0, float* __restrict__ stats,
                                                    float* __restrict__ acc) {
  float vals[kHeadSlots];
#pragma unroll
  for (int k = 0; k < kHeadSlots; ++k) vals[k] = 0.f;
  const Vec c = load_vec(cvec, L);
#pragma unroll 4
  for (int n = blockIdx.x * kFlatBlock + threadIdx.x; n < B; n += gridDim.x * kFlatBlock) {
    const Vec u = load_vec(z + (size_t)n * L, L);
    Vec g;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
      const float d = j < L ? u.v[j] - c.v[j] : 0.f;
      sq = fmaf(d, d, sq);
      g.v[j] = 2.f * d * gscale;  // d mean((z-c)^2)/dz, gscale = upstream / (B*L)
      vals[1 + j] += u.v[j];
    }
    vals[0] += sq;
    vals[LMAX + 1] += 1.f;
    vals[LMAX + 2] += sqrtf(dot(u, u));
    if (dz) store_vec(dz + (size_t)n * L, g, L);
    if (score) score[n] = sq / (float)L;  // MSELoss(reduction='none')(c, z).mean(-1), eval_utils.py:63-64
  }
  block_partials(vals, partials, scale0, stats, acc);
}

// Mahalanobis head (eval_utils.py:28-38; staticCenter.py:178-181).  slots: [0] sum dist, [1..L] sum z, [17] #clips,
// [18] sum |z|.  dist = sqrt(d^T VI d), d = z - c;  d dist / dz = (VI + VI^T) d / (2 dist).
__global__ __launch_bounds__(kFlatBlock) void k_mahalanobis_head(const float* __restrict__ z,
                                                            const float* __restrict__ cvec,
                                                            const float* __restrict__ VI,
                                                            float* __restrict__ dz, float* __restrict__ score,
                                                            float* __restrict__ partials, int B, int L,
                                                            float gscale, float scale0, float* __restrict__ stats,
                                                            float* __restrict__ acc) {
  __shared__ float vi[LMAX * LMAX];
  for (int e = threadIdx.x; e < LMAX * LMAX; e += kFlatBlock) {
    const int r = e / LMAX, q = e - r * LMAX;
    vi[e] = (r < L && q < L) ? VI[r * L + q] : 0.f;
  }
  __syncthreads();
  float vals[kHeadSlots];
#pragma unroll
  for (int k = 0; k < kHeadSlots; ++k) vals[k] = 0.f;
  const Vec c = load_vec(cvec, L);
  for (int n = blockIdx.x * kFlatBlock + threadIdx.x; n < B; n += gridDim.x * kFlatBlock) {
    const Vec u = load_vec(z + (size_t)n * L, L);
    Vec d, w, wt;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) { d.v[j] = j < L ? u.v[j] - c.v[j] : 0.f; w.v[j] = 0.f; wt.v[j] = 0.f; }
#pragma unroll
    for (int r = 0; r < LMAX; ++r)
#pragma unroll
      for (int q = 0; q < LMAX; ++q) {
        const float a = vi[r * LMAX + q];          // wave-uniform address: LDS broadcast
        w.v[r] = fmaf(a, d.v[q], w.v[r]);          // (VI d)[r]
        wt.v[q] = fmaf(a, d.v[r], wt.v[q]);        // (VI^T d)[q]
      }
    const float dist = sqrtf(dot(d, w));
    vals[0] += dist;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) vals[1 + j] += u.v[j];
    vals[LMAX + 1] += 1.f;
    vals[LMAX + 2] += sqrtf(dot(u, u));
    if (score) score[n] = dist;
    if (dz) {
      Vec g;
      const float inv = gscale / (2.f * dist);     // dist == 0: inf * 0 = NaN, as torch.sqrt's backward gives
#pragma unroll
      for (int j = 0; j < LMAX; ++j) g.v[j] = (w.v[j] + wt.v[j]) * inv;
      store_vec(dz + (size_t)n * L, g, L);
    }
  }
  block_partials(vals, partials, scale0, stats, acc);
}

// gram (+)= sum_n z_n z_n^T  [L x L]; one block, thread (i, j), clips staged through LDS in fixed order.
// Sigma (z-mu)(z-mu)^T for any mu follows from it and the [1..L], [17] slots (batch_cov_mat_step, staticCenter.py:40-46).
__global__ __launch_bounds__(LMAX * LMAX) void k_gram(const float* __restrict__ z, float* __restrict__ gram, int B,
                                                     int L, int accumulate) {
  constexpr int CH = 256;
  __shared__ float zs[CH * LMAX];
  const int i = threadIdx.x / LMAX, j = threadIdx.x % LMAX;
  float s = 0.f;
  for (int n0 = 0; n0 < B; n0 += CH) {
    const int nc = min(CH, B - n0);
    __syncthreads();
    for (int e = threadIdx.x; e < nc * L; e += LMAX * LMAX) {
      const int r = e / L, q = e - r * L;
      zs[r * LMAX + q] = z[(size_t)(n0 + r) * L + q];
    }
    __syncthreads();
    if (i < L && j < L)
      for (int r = 0; r < nc; ++r) s = fmaf(zs[r * LMAX + i], zs[r * LMAX + j], s);
  }
  if (i < L && j < L) gram[i * L + j] = accumulate ? gram[i * L + j] + s : s;
}

// Poincare head.  slots: [0] sum dist, [1..L] sum gamma*zh, [17] sum (gamma-1), [18] sum |zh|
__global__ __launch_bounds__(kFlatBlock) void k_poincare_head(const float* __restrict__ z,
                                                         const float* __restrict__ cvec,
                                                         float* __restrict__ dz, float* __restrict__ zh,
                                                         float* __restrict__ score,
                                                         float* __restrict__ partials, int B, int L,
                                                         float gscale, float scale0, float* __restrict__ stats,
                                                         float* __restrict__ acc) {
  float vals[kHeadSlots];
#pragma unroll
  for (int k = 0; k < kHeadSlots; ++k) vals[k] = 0.f;
  for (int n = blockIdx.x * kFlatBlock + threadIdx.x; n < B; n += gridDim.x * kFlatBlock) {
    const Vec u = load_vec(z + (size_t)n * L, L);
    const Embed em = hyp_embed(u);
    if (zh) store_vec(zh + (size_t)n * L, em.p, L);
    const float y2 = dot(em.p, em.p);
    const float gamma = 2.f / (1.f - y2);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) vals[1 + j] += gamma * em.p.v[j];
    vals[LMAX + 1] += gamma - 1.f;
    vals[LMAX + 2] += sqrtf(y2);
    if (cvec) {
      const Vec c = load_vec(cvec, L);
      Vec gp;
      const float d = poincare_dist(c, em.p, dz ? &gp : nullptr);
      vals[0] += d;
      if (score) score[n] = d;
      if (dz) {
#pragma unroll
        for (int j = 0; j < LMAX; ++j) gp.v[j] *= gscale;  // gscale = upstream / B
        store_vec(dz + (size_t)n * L, hyp_embed_bwd(u, em, gp), L);
      }
    }
  }
  block_partials(vals, partials, scale0, stats, acc);
}

// stats[k] = sum_p partials[p][k] (slot 0 additionally * scale0);  acc[k] += raw sums
__global__ __launch_bounds__(64) void k_head_finalize(const float* __restrict__ partials, int P,
                                                       float scale0, float* __restrict__ stats,
                                                       float* __restrict__ acc) {
  const int k = threadIdx.x;
  if (k >= kHeadSlots) return;
  double s = 0.0;
  for (int p = 0; p < P; ++p) s += (double)partials[p * kHeadSlots + k];
  if (stats) stats[k] = (float)(k == 0 ? s * (double)scale0 : s);
  if (acc) acc[k] += (float)s;
}

// dist(c, zh) per row for points already on the ball (eval scoring, eval_utils.py:66-67)
__global__ __launch_bounds__(kFlatBlock) void k_poincare_dist(const float* __restrict__ zh,
                                                         const float* __restrict__ cvec,
                                                         float* __restrict__ score, int B, int L) {
  const int n = blockIdx.x * kFlatBlock + threadIdx.x;
  if (n >= B) return;
  const Vec p = load_vec(zh + (size_t)n * L, L), c = load_vec(cvec, L);
  score[n] = poincare_dist(c, p, nullptr);
}

// logmap0(y) = y / |y| * artanh(|y|) at curvature -1 (hyper_math.py:367-370: norm clamp 1e-5, artanh on the clamped argument)
__global__ __launch_bounds__(kFlatBlock) void k_poincare_logmap0(const float* __restrict__ y, float* __restrict__ out, int B, int L) {
  const int n = blockIdx.x * kFlatBlock + threadIdx.x;
  if (n >= B) return;
  const Vec p = load_vec(y + (size_t)n * L, L);
  const float yn = fmaxf(sqrtf(dot(p, p)), kMinNorm);
  const float yc = fminf(fmaxf(yn, -1.f + kArtanhEps), 1.f - kArtanhEps);
  const float at = 0.5f * (log1pf(yc) - log1pf(-yc));
  Vec o;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) o.v[j] = p.v[j] / yn * at;
  store_vec(out + (size_t)n * L, o, L);
}

// Gyromidpoint from the running sums: m = S / s ; centre = (1/2) (x) m  (Mobius scalar mul)
//   acc: head slot layout, [1..L] = sum gamma*zh, [17] = sum (gamma-1)
__global__ void k_midpoint_finalize(const float* __restrict__ acc, float* __restrict__ cvec, int L) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double s = fmax((double)acc[LMAX + 1], 1e-10);
  double m[LMAX], mn2 = 0.0;
  for (int j = 0; j < L; ++j) { m[j] = (double)acc[1 + j] / s; mn2 += m[j] * m[j]; }
  const double mn = fmax(sqrt(mn2), 1e-15);
  const double mc = fmin(mn, 1.0 - 1e-7);
  const double at = 0.5 * log((1.0 + mc) / (1.0 - mc));
  const double sc = tanh(0.5 * at) / mn;
  for (int j = 0; j < L; ++j) cvec[j] = (float)(sc * m[j]);
}

// Euclidean centre: c = S / n, then |c| < eps -> +-eps  (staticCenter.py:118-121)
__global__ void k_center_finalize(const float* __restrict__ acc, float eps, float* __restrict__ cvec,
                                  int L) {
  const int j = threadIdx.x;
  if (j >= L || blockIdx.x != 0) return;
  float c = acc[1 + j] / acc[LMAX + 1];
  if (fabsf(c) < eps && c < 0.f) c = -eps;
  if (fabsf(c) < eps && c > 0.f) c = eps;
  cvec[j] = c;
}

// ---- regulariser + Adam on flat buffers -------------------------------------------------
// partials[b] = sum over this block's slice of mask[i] * p[i]^2
__global__ __launch_bounds__(256) void k_sqnorm_masked(const float* __restrict__ p,
                                                        const float* __restrict__ mask, size_t n,
                                                        float* __restrict__ partials) {
  __shared__ float sh[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = p[i];
    s = fmaf(mask ? mask[i] * v : v, v, s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// torch.optim.Adam (no amsgrad, weight_decay 0) on flat fp32 buffers, with the regulariser's
// gradient folded in: g = grad * gscale + reg_coef * mask * p.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                               float* __restrict__ m, float* __restrict__ v,
                                               const float* __restrict__ mask, size_t n, float lr,
                                               float beta1, float beta2, float eps, float bc1,
                                               float bc2_sqrt, float gscale, float reg_coef) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float pi = p[i];
  float gi = g[i] * gscale;
  if (reg_coef != 0.f) gi = fmaf(reg_coef * (mask ? mask[i] : 1.f), pi, gi);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

// hyper[0] = lr, hyper[1] = beta1^t, hyper[2] = beta2^t  (device-resident so a captured hipGraph can be
// replayed: the step count advances in memory, not in kernel arguments)
__global__ void k_adam_tick(float* __restrict__ hyper, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    hyper[1] *= beta1;
    hyper[2] *= beta2;
  }
}

__global__ __launch_bounds__(256) void k_adam_dev(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ mask, size_t n,
                                                   const float* __restrict__ hyper, float beta1, float beta2,
                                                   float eps, float gscale, float reg_coef) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lr = hyper[0], bc1 = 1.f - hyper[1], bc2s = sqrtf(1.f - hyper[2]);
  const float pi = p[i];
  float gi = g[i] * gscale;
  if (reg_coef != 0.f) gi = fmaf(reg_coef * (mask ? mask[i] : 1.f), pi, gi);
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - (lr / bc1) * (mi / (sqrtf(vi) / bc2s + eps));
}

__global__ __launch_bounds__(256) void k_scale_sum(const float* __restrict__ v, int n, float scale,
                                                    float* __restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(sh[0] * (double)scale);
}

}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_head_slots(void) { return kHeadSlots; }
size_t coskad_head_ws_floats(int B) { return (size_t)ceil_div(B, kFlatBlock) * kHeadSlots; }

/* Euclidean one-class head on z [B,L] (staticCenter.py:187, dynamicCenter.py:116, eval_utils.py:63-64).
 *   stats[19]: [0] = mean_{n,j} (z-c)^2 (the loss), [1..L] = sum_n z, [17] = B, [18] = sum_n |z_n|
 *   acc[19]  : += the raw sums (running centre accumulation, staticCenter.py:172-178); NULL to skip
 *   dz       : upstream * d loss / dz, NULL to skip;  score[n] = mean_j (c - z_n)^2, NULL to skip */
int coskad_mse_head_f32(const float* z, const float* c, float* dz, float* score, float* stats, float* acc,
                        float upstream, float* ws, int B, int L, hipStream_t stream) {
  if (!z || !c || !ws) return fail(COSKAD_ERR_ARG, "mse_head: null pointer");
  if (B <= 0 || L <= 0 || L > LMAX) return fail(COSKAD_ERR_SHAPE, "mse_head: B=%d latent=%d (max %d)", B, L, LMAX);
  const int P = head_blocks(B);
  hipLaunchKernelGGL(k_mse_head, dim3(P), dim3(kFlatBlock), 0, stream, z, c, dz, score, ws, B, L,
                     upstream / ((float)B * (float)L), 1.f / ((float)B * (float)L), stats, acc);
  if ((stats || acc) && P > 1)
    hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(64), 0, stream, ws, P, 1.f / ((float)B * (float)L), stats, acc);
  return check_launch("mse_head");
}

/* Mahalanobis one-class head: loss = mean_n sqrt((z_n-c)^T VI (z_n-c)) (utils/eval_utils.py:28-38 as called at
 * euclidean_encoder_staticCenter.py:181), gradient, per-window score (eval_utils.py:41-47), centre sums as in the
 * MSE head, and optionally gram (+)= sum_n z_n z_n^T (the second moments behind compute_inv_cov_mat, :133-142). */
int coskad_mahalanobis_head_f32(const float* z, const float* c, const float* VI, float* dz, float* score,
                                float* stats, float* acc, float* gram, int gram_accumulate, float upstream,
                                float* ws, int B, int L, hipStream_t stream) {
  if (!z || !c || !VI || !ws) return fail(COSKAD_ERR_ARG, "mahalanobis_head: null pointer");
  if (B <= 0 || L <= 0 || L > LMAX) return fail(COSKAD_ERR_SHAPE, "mahalanobis_head: B=%d latent=%d (max %d)", B, L, LMAX);
  const int P = head_blocks(B);
  hipLaunchKernelGGL(k_mahalanobis_head, dim3(P), dim3(kFlatBlock), 0, stream, z, c, VI, dz, score, ws, B, L,
                     upstream / (float)B, 1.f / (float)B, stats, acc);
  if ((stats || acc) && P > 1)
    hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(64), 0, stream, ws, P, 1.f / (float)B, stats, acc);
  if (gram) hipLaunchKernelGGL(k_gram, dim3(1), dim3(LMAX * LMAX), 0, stream, z, gram, B, L, gram_accumulate);
  return check_launch("mahalanobis_head");
}

/* Poincare one-class head (hyperbolic_encoder.py:147,157; utils/hyper_math.py formulas):
 *   zh = project(expmap0(z)) ; loss = mean_n dist(c, zh_n)
 *   stats[19]: [0] = loss, [1..L] = sum gamma*zh, [17] = sum (gamma-1), [18] = sum |zh|   (gamma = 2/(1-|zh|^2))
 *   acc[19]  : += raw sums (gyromidpoint accumulation replacing the reference's torch.cat, :148-153)
 *   c == NULL: embedding + midpoint sums only (centre initialisation, :110-122). */
int coskad_poincare_head_f32(const float* z, const float* c, float* dz, float* zh, float* score,
                             float* stats, float* acc, float upstream, float* ws, int B, int L,
                             hipStream_t stream) {
  if (!z || !ws) return fail(COSKAD_ERR_ARG, "poincare_head: null pointer");
  if (B <= 0 || L <= 0 || L > LMAX) return fail(COSKAD_ERR_SHAPE, "poincare_head: B=%d latent=%d (max %d)", B, L, LMAX);
  const int P = head_blocks(B);
  hipLaunchKernelGGL(k_poincare_head, dim3(P), dim3(kFlatBlock), 0, stream, z, c, dz, zh, score, ws, B, L,
                     upstream / (float)B, 1.f / (float)B, stats, acc);
  if ((stats || acc) && P > 1)
    hipLaunchKernelGGL(k_head_finalize, dim3(1), dim3(64), 0, stream, ws, P, 1.f / (float)B, stats, acc);
  return check_launch("poincare_head");
}

/* score[n] = dist(c, zh_n) for points already on the ball (eval_utils.py:66-67). */
int coskad_poincare_dist_f32(const float* zh, const float* c, float* score, int B, int L, hipStream_t stream) {
  if (!zh || !c || !score) return fail(COSKAD_ERR_ARG, "poincare_dist: null pointer");
  if (B <= 0 || L <= 0 || L > LMAX) return fail(COSKAD_ERR_SHAPE, "poincare_dist: B=%d latent=%d", B, L);
  hipLaunchKernelGGL(k_poincare_dist, dim3(ceil_div(B, kFlatBlock)), dim3(kFlatBlock), 0, stream, zh, c, score, B, L);
  return check_launch("poincare_dist");
}

int coskad_poincare_logmap0_f32(const float* y, float* out, int B, int L, hipStream_t stream) {
  if (!y || !out) return fail(COSKAD_ERR_ARG, "poincare_logmap0: null pointer");
  if (B <= 0 || L <= 0 || L > LMAX) return fail(COSKAD_ERR_SHAPE, "poincare_logmap0: B=%d latent=%d", B, L);
  hipLaunchKernelGGL(k_poincare_logmap0, dim3(ceil_div(B, kFlatBlock)), dim3(kFlatBlock), 0, stream, y, out, B, L);
  return check_launch("poincare_logmap0");
}

/* Euclidean centre from accumulated sums (acc layout of coskad_mse_head_f32): c = S/n with n = acc[17],
 * then |c| < eps -> +-eps (staticCenter.py:118-121). */
int coskad_center_finalize_f32(const float* acc, float* c, float eps, int L, hipStream_t stream) {
  if (!acc || !c || L <= 0 || L > LMAX) return fail(COSKAD_ERR_ARG, "center_finalize: bad argument");
  hipLaunchKernelGGL(k_center_finalize, dim3(1), dim3(64), 0, stream, acc, eps, c, L);
  return check_launch("center_finalize");
}

/* Gyromidpoint (geoopt weighted_midpoint with unit weights; hyperbolic_encoder.py:122,179) from the
 * sums accumulated by coskad_poincare_head_f32. */
int coskad_midpoint_finalize_f32(const float* acc, float* c, int L, hipStream_t stream) {
  if (!acc || !c || L <= 0 || L > LMAX) return fail(COSKAD_ERR_ARG, "midpoint_finalize: bad argument");
  hipLaunchKernelGGL(k_midpoint_finalize, dim3(1), dim3(64), 0, stream, acc, c, L);
  return check_launch("midpoint_finalize");
}

/* out[0] = scale * sum_i mask[i] * p[i]^2   (calc_reg_loss: scale = 0.5 / #non-bias tensors;
 * utils/model_utils.py:90-105).  mask NULL = all ones.  ws: >= 256 floats. */
int coskad_sqnorm_f32(const float* p, const float* mask, size_t n, float scale, float* out, float* ws,
                      hipStream_t stream) {
  if (!p || !out || !ws || n == 0) return fail(COSKAD_ERR_ARG, "sqnorm: bad argument");
  const int grid = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipLaunchKernelGGL(k_sqnorm_masked, dim3(grid), dim3(256), 0, stream, p, mask, n, ws);
  hipLaunchKernelGGL(k_scale_sum, dim3(1), dim3(256), 0, stream, ws, grid, scale, out);
  return check_launch("sqnorm");
}

/* torch.optim.Adam step (amsgrad off, weight_decay 0) on flat buffers, with the regulariser gradient
 * folded in: g = grad * gscale + reg_coef * mask * p.  `step` is the 1-based step count. */
int coskad_adam_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float lr,
                    float beta1, float beta2, float eps, int step, float gscale, float reg_coef,
                    hipStream_t stream) {
  if (!p || !g || !m || !v || n == 0 || step < 1) return fail(COSKAD_ERR_ARG, "adam: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, mask, n, lr,
                     beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), gscale, reg_coef);
  return check_launch("adam");
}

/* Same update with the caller's running products b1pow = beta1^t, b2pow = beta2^t (fp32, multiplied up step by step: the arithmetic
 * of coskad_adam_dev_f32's device-side tick, so that an eager step and a hipGraph-captured one agree bit for bit). */
int coskad_adam_pow_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float lr, float beta1,
                        float beta2, float eps, float b1pow, float b2pow, float gscale, float reg_coef, hipStream_t stream) {
  if (!p || !g || !m || !v || n == 0) return fail(COSKAD_ERR_ARG, "adam_pow: bad argument");
  if (!(b1pow < 1.f) || !(b2pow < 1.f)) return fail(COSKAD_ERR_ARG, "adam_pow: beta^t must be < 1 (t >= 1)");
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, mask, n, lr,
                     beta1, beta2, eps, 1.f - b1pow, sqrtf(1.f - b2pow), gscale, reg_coef);
  return check_launch("adam_pow");
}

/* Same update with lr / beta^t read from device memory: hyper = {lr, beta1^t, beta2^t} (initialise to
 * {lr, 1, 1}); every call first advances beta^t, so the call is replayable inside a hipGraph. */
int coskad_adam_dev_f32(float* p, const float* g, float* m, float* v, const float* mask, size_t n, float* hyper,
                        float beta1, float beta2, float eps, float gscale, float reg_coef, hipStream_t stream) {
  if (!p || !g || !m || !v || !hyper || n == 0) return fail(COSKAD_ERR_ARG, "adam_dev: bad argument");
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, stream, hyper, beta1, beta2);
  hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, mask, n, hyper,
                     beta1, beta2, eps, gscale, reg_coef);
  return check_launch("adam_dev");
}

}  // extern "C"
