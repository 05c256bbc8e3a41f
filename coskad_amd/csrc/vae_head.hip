// The spherical VAE's latent head (reference models/sts/vae.py:79-91,104-118 and models/spherical_vae.py:86-94 with the
// un-vendored `power_spherical` package -- De Cao & Aziz, "The Power Spherical distribution", 2020; restated in torch in
// coskad_amd/models/sts/vae.py, which is this file's specification): everything between the two small heads' raw outputs and the
// decoder's input, forward and backward, as three row-parallel launches instead of ~160 element-wise torch launches per step:
//
//   prep    mu = m / |m|;  kappa = softplus(v) + 1;  concentration = (alpha, beta) = ((d-1)/2 + kappa, (d-1)/2), total = alpha + beta
//           (torch draws x ~ Dirichlet(concentration) and eps ~ N(0, I_{d-1}) in between: the sampler's implicit reparameterisation
//            gradient, torch._dirichlet_grad, stays torch's)
//   sample  t = 2 x_0 - 1;  y = [t, sqrt(1 - t^2) eps / |eps|];  u = (e1 - mu) / |e1 - mu|;  z = y - 2 (y.u) u   (Householder)
//           kl  = -H(PowerSpherical(mu, kappa)) + H(Uniform(S^{d-1}))  per row;  1 / kappa per row
//   bwd     d m, d v from dz (the decoder's gradient) and the two scalar loss weights
// One thread per clip, the latent (d <= 16) in registers; digamma / trigamma by recurrence + asymptotic series in fp64.
#include "common.h"

namespace coskad {
namespace vh {

constexpr int LMAX = 16;
constexpr double kLog2 = 0.6931471805599453, kLogPi = 1.1447298858494002;

__device__ __forceinline__ double digamma_d(double x) {       // x > 0
  double r = 0.0;
  while (x < 6.0) { r -= 1.0 / x; x += 1.0; }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f * (1.0 / 132)))));
}
__device__ __forceinline__ double trigamma_d(double x) {      // x > 0
  double r = 0.0;
  while (x < 6.0) { r += 1.0 / (x * x); x += 1.0; }
  const double f = 1.0 / (x * x);
  return r + 1.0 / x + 0.5 * f + (1.0 / x) * f * (1.0 / 6 - f * (1.0 / 30 - f * (1.0 / 42 - f * (1.0 / 30 - f * (5.0 / 66)))));
}
__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }   // F.softplus(beta = 1, threshold = 20)

struct Row {
  float v[LMAX];
};
__device__ __forceinline__ Row load_row(const float* p, int L) {
  Row r;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) r.v[j] = j < L ? p[j] : 0.f;
  return r;
}
__device__ __forceinline__ void store_row(float* p, const Row& r, int L) {
#pragma unroll
  for (int j = 0; j < LMAX; ++j)
    if (j < L) p[j] = r.v[j];
}
__device__ __forceinline__ float dot(const Row& a, const Row& b) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) s = fmaf(a.v[j], b.v[j], s);
  return s;
}

__global__ __launch_bounds__(256) void k_ps_prep(const float* __restrict__ m, int ldm, const float* __restrict__ vr, int ldv,
                                                  float* __restrict__ mu, float* __restrict__ kappa, float* __restrict__ conc,
                                                  float* __restrict__ total, int B, int L) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  const Row a = load_row(m + (size_t)n * ldm, L);
  const float nm = sqrtf(dot(a, a));
  Row o;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) o.v[j] = a.v[j] / nm;
  store_row(mu + (size_t)n * L, o, L);
  const float k = softplus_f(vr[(size_t)n * ldv]) + 1.f;
  const float c = 0.5f * (float)(L - 1);
  kappa[n] = k;
  conc[2 * n] = c + k;
  conc[2 * n + 1] = c;
  total[n] = (c + k) + c;
}

// the Householder reflection's vector u = normalize(e1 - mu) (F.normalize: divide by max(|.|, 1e-12))
__device__ __forceinline__ Row reflect_vec(const Row& mu, int L, float* wn_out) {
  Row w;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) w.v[j] = j < L ? (j == 0 ? 1.f : 0.f) - mu.v[j] : 0.f;
  const float wn = fmaxf(sqrtf(dot(w, w)), 1e-12f);
#pragma unroll
  for (int j = 0; j < LMAX; ++j) w.v[j] = w.v[j] / wn;
  *wn_out = wn;
  return w;
}

__global__ __launch_bounds__(256) void k_ps_sample(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ mu,
                                                    const float* __restrict__ kappa, float* __restrict__ z, float* __restrict__ kl,
                                                    float* __restrict__ ikappa, int B, int L) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  const float t = 2.f * x[2 * n] - 1.f;
  Row e;                                                   // e.v[j] = eps[j - 1], j = 1 .. L - 1
#pragma unroll
  for (int j = 0; j < LMAX; ++j) e.v[j] = (j >= 1 && j < L) ? eps[(size_t)n * (L - 1) + j - 1] : 0.f;
  const float en = fmaxf(sqrtf(dot(e, e)), 1e-12f);
  const float s = sqrtf(fmaxf(1.f - t * t, 0.f));
  Row y;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) y.v[j] = j == 0 ? t : s * (e.v[j] / en);
  const Row m = load_row(mu + (size_t)n * L, L);
  float wn;
  const Row u = reflect_vec(m, L, &wn);
  const float yu = dot(y, u);
  Row o;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) o.v[j] = y.v[j] - 2.f * yu * u.v[j];
  store_row(z + (size_t)n * L, o, L);
  // KL(PowerSpherical || uniform) = -H(q) + H(p);  H(q) = (a + b) log 2 + lgamma(a) - lgamma(a + b) + b log pi - k (log 2 + psi(a) - psi(a + b))
  const double k = (double)kappa[n], b = 0.5 * (double)(L - 1), a = b + k;
  const double hq = (a + b) * kLog2 + lgamma(a) - lgamma(a + b) + b * kLogPi - k * (kLog2 + digamma_d(a) - digamma_d(a + b));
  const double hp = kLog2 + 0.5 * (double)L * kLogPi - lgamma(0.5 * (double)L);
  kl[n] = (float)(hp - hq);
  ikappa[n] = 1.f / (float)k;
}

// dz: the decoder's gradient w.r.t. the sampled latent; g: torch._dirichlet_grad(x, concentration, total) [B, 2];
// w_kl, w_exp: d loss / d (kl of a row), d loss / d (1 / kappa of a row) (the loss weights over the batch size)
__global__ __launch_bounds__(256) void k_ps_bwd(const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ g,
                                                 const float* __restrict__ eps, const float* __restrict__ mu, const float* __restrict__ kappa,
                                                 const float* __restrict__ m, int ldm, const float* __restrict__ vr, int ldv, float w_kl,
                                                 float w_exp, float* __restrict__ dm, int lddm, float* __restrict__ dv, int lddv, int B, int L) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= B) return;
  const Row gz = load_row(dz + (size_t)n * L, L);
  const Row mur = load_row(mu + (size_t)n * L, L);
  const float zb = x[2 * n], t = 2.f * zb - 1.f;
  Row e;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) e.v[j] = (j >= 1 && j < L) ? eps[(size_t)n * (L - 1) + j - 1] : 0.f;
  const float en = fmaxf(sqrtf(dot(e, e)), 1e-12f);
  const float om = 1.f - t * t, s = sqrtf(fmaxf(om, 0.f));
  Row vv, y;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) { vv.v[j] = e.v[j] / en; y.v[j] = j == 0 ? t : s * vv.v[j]; }
  float wn;
  const Row u = reflect_vec(mur, L, &wn);
  const float yu = dot(y, u), gu = dot(gz, u);
  // z = y - 2 (y.u) u:  dy = dz - 2 (dz.u) u;  du = -2 ((y.u) dz + (dz.u) y)
  Row dy, du;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) { dy.v[j] = gz.v[j] - 2.f * gu * u.v[j]; du.v[j] = -2.f * (yu * gz.v[j] + gu * y.v[j]); }
  // u = w / |w|, w = e1 - mu:  dw = (du - (du.u) u) / |w|;  dmu = -dw
  const float duu = dot(du, u);
  Row dmu;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) dmu.v[j] = j < L ? -(du.v[j] - duu * u.v[j]) / wn : 0.f;
  // mu = m / |m|:  dm = (dmu - (dmu.mu) mu) / |m|
  const Row mr = load_row(m + (size_t)n * ldm, L);
  const float nm = sqrtf(dot(mr, mr)), dmm = dot(dmu, mur);
  Row o;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) o.v[j] = (dmu.v[j] - dmm * mur.v[j]) / nm;
  store_row(dm + (size_t)n * lddm, o, L);
  // y = [t, s v], s = sqrt(clamp(1 - t^2, 0)):  dt = dy_0 + (dy_{1:}.v) ds/dt, ds/dt = -t / s where 1 - t^2 > 0
  float dyv = 0.f;
#pragma unroll
  for (int j = 1; j < LMAX; ++j) dyv = fmaf(dy.v[j], vv.v[j], dyv);
  const float dt = dy.v[0] + (om > 0.f ? dyv * (-t / s) : 0.f);
  // t = 2 x_0 - 1; x ~ Dirichlet(alpha, beta) reparameterised (torch/distributions/dirichlet.py, _Dirichlet_backward with the
  // upstream gradient (2 dt, 0)): d alpha = g_0 (2 dt - x_0 2 dt)
  const float dz0 = 2.f * dt;
  const float dalpha = g[2 * n] * (dz0 - zb * dz0);
  // kappa: alpha = (d-1)/2 + kappa;  d kl / d kappa = kappa (psi1(alpha) - psi1(alpha + beta));  d (1/kappa) = -1 / kappa^2
  const double k = (double)kappa[n], b = 0.5 * (double)(L - 1), a = b + k;
  const double dk = (double)dalpha + (double)w_kl * k * (trigamma_d(a) - trigamma_d(a + b)) - (double)w_exp / (k * k);
  // kappa = softplus(v) + 1:  dv = dk sigmoid(v)   (v > 20: softplus is the identity there)
  const float v = vr[(size_t)n * ldv];
  const float sg = v > 20.f ? 1.f : 1.f / (1.f + expf(-v));
  dv[(size_t)n * lddv] = (float)dk * sg;
}

}  // namespace vh
}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_ps_head_prep_f32(const float* mean_raw, int ld_mean, const float* var_raw, int ld_var, float* mu, float* kappa,
                            float* concentration, float* total, int B, int L, hipStream_t stream) {
  if (!mean_raw || !var_raw || !mu || !kappa || !concentration || !total) return fail(COSKAD_ERR_ARG, "ps_head_prep: null pointer");
  if (B <= 0 || L < 2 || L > vh::LMAX || ld_mean < L || ld_var < 1) return fail(COSKAD_ERR_SHAPE, "ps_head_prep: B=%d latent=%d", B, L);
  hipLaunchKernelGGL(vh::k_ps_prep, dim3(ceil_div(B, 256)), dim3(256), 0, stream, mean_raw, ld_mean, var_raw, ld_var, mu, kappa,
                     concentration, total, B, L);
  return check_launch("ps_head_prep");
}

int coskad_ps_head_sample_f32(const float* x, const float* eps, const float* mu, const float* kappa, float* z, float* kl,
                              float* inv_kappa, int B, int L, hipStream_t stream) {
  if (!x || !eps || !mu || !kappa || !z || !kl || !inv_kappa) return fail(COSKAD_ERR_ARG, "ps_head_sample: null pointer");
  if (B <= 0 || L < 2 || L > vh::LMAX) return fail(COSKAD_ERR_SHAPE, "ps_head_sample: B=%d latent=%d", B, L);
  hipLaunchKernelGGL(vh::k_ps_sample, dim3(ceil_div(B, 256)), dim3(256), 0, stream, x, eps, mu, kappa, z, kl, inv_kappa, B, L);
  return check_launch("ps_head_sample");
}

int coskad_ps_head_bwd_f32(const float* dz, const float* x, const float* dirichlet_grad, const float* eps, const float* mu,
                           const float* kappa, const float* mean_raw, int ld_mean, const float* var_raw, int ld_var, float w_kl,
                           float w_exp, float* d_mean_raw, int ld_dmean, float* d_var_raw, int ld_dvar, int B, int L,
                           hipStream_t stream) {
  if (!dz || !x || !dirichlet_grad || !eps || !mu || !kappa || !mean_raw || !var_raw || !d_mean_raw || !d_var_raw)
    return fail(COSKAD_ERR_ARG, "ps_head_bwd: null pointer");
  if (B <= 0 || L < 2 || L > vh::LMAX || ld_mean < L || ld_var < 1 || ld_dmean < L || ld_dvar < 1)
    return fail(COSKAD_ERR_SHAPE, "ps_head_bwd: B=%d latent=%d", B, L);
  hipLaunchKernelGGL(vh::k_ps_bwd, dim3(ceil_div(B, 256)), dim3(256), 0, stream, dz, x, dirichlet_grad, eps, mu, kappa, mean_raw,
                     ld_mean, var_raw, ld_var, w_kl, w_exp, d_mean_raw, ld_dmean, d_var_raw, ld_dvar, B, L);
  return check_launch("ps_head_bwd");
}

}  // extern "C"
