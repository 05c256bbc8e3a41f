// Forward kernels of the STS-GCN block (reference models/graph_layers/stsgcn.py:94-156).
//
//   out = PReLU( BN_t(W_t . gcn(X) + b_t) + BN_r(W_r . X + b_r) )
//
// BatchNorm is an affine map once its statistics are known (running stats in eval mode,
// batch stats from stsgcn_train.hip in train mode), so the block collapses to
//
//   U = Wz . gcn(X) + Wx . X + b          (Wz, Wx, b = "folded" weights, k_bn_fold below)
//
// and ONE kernel per layer reads X once and writes U once.  Layers exchange the
// PRE-activation U; the consumer applies PReLU while staging (1 VALU op / element), which
// hands the backward pass the PReLU mask and argument without storing anything extra.
#include "tile_ops.h"
#include <cstdlib>

namespace coskad {

// --------------------------------------------------------------------------------------
// k_layer_apply: U[n,:,p] = Wz . gcn(X)[n,:,p] + Wx . X[n,:,p] + b
// grid = ceil(B / NB) tiles, block = 256, dynamic LDS = NB*Ci*LD*4 bytes.
//   wfold : [2*Ci][CoP] row-major; rows 0..Ci-1 = Wz^T, rows Ci..2Ci-1 = Wx^T; CoP = Co
//           rounded up to 16, pad columns zero.
// --------------------------------------------------------------------------------------
template <int T, int V, int CB>
__global__ __launch_bounds__((Geo<T, V>::Block)) void k_layer_apply(
    const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ wfold, const float* __restrict__ bias,
    const float* __restrict__ in_slope, const float* __restrict__ out_slope, int B, int Ci, int Co,
    int CoP, int NB) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int clip0 = blockIdx.x * NB;
  const int nb = min(NB, B - clip0);
  const int rows = nb * Ci;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const bool post = out_slope != nullptr;
  const float a_out = post ? out_slope[0] : 0.f;

  const float* gin = in + (size_t)clip0 * Ci * TV;
  stage_rows<T, V>(gin, lds, rows * TV, pre, a_in);
  __syncthreads();
  gcn_rows<T, V, false>(lds, rows, Aw, Tw);
  __syncthreads();

  // position phase
  const int P = nb * TV;
  const int rounds = ceil_div(P, kBlock);
  for (int r = 0; r < rounds; ++r) {
    const int pos = r * kBlock + threadIdx.x;
    const bool act = pos < P;
    const int pc = act ? pos : 0;
    const int n = pc / TV;
    const int p = pc - n * TV;
    const float* zrow = lds + (n * Ci) * LD + p;
    const float* xg = gin + (size_t)n * Ci * TV + p;
    float* og = out + ((size_t)(clip0 + n) * Co) * TV + p;
    for (int o0 = 0; o0 < CoP; o0 += 16) {
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = bias[o0 + j];
      for (int c0 = 0; c0 < Ci; c0 += CB) {
        float zr[CB], xr[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
          zr[k] = zrow[(c0 + k) * LD];
          float x = xg[(c0 + k) * TV];
          xr[k] = pre ? prelu_f(x, a_in) : x;
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
          const float* wz = wfold + (c0 + k) * CoP + o0;
          const float* wx = wfold + (Ci + c0 + k) * CoP + o0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            acc[j] = fmaf(wz[j], zr[k], acc[j]);
            acc[j] = fmaf(wx[j], xr[k], acc[j]);
          }
        }
      }
      if (act) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (o0 + j < Co) og[(o0 + j) * TV] = post ? prelu_f(acc[j], a_out) : acc[j];
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// k_gcn: standalone ConvTemporalGraphical forward / adjoint (stsgcn.py:143-156), used by the
// module mirror of that class.  rows = N*C rows of TV floats, 64 rows per tile.
// --------------------------------------------------------------------------------------
template <int T, int V, bool ADJ>
__global__ __launch_bounds__((Geo<T, V>::Block)) void k_gcn(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ Aw,
                                                const float* __restrict__ Tw, int total_rows) {
  constexpr int TV = Geo<T, V>::TV;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row0 = blockIdx.x * 64;
  const int rows = min(64, total_rows - row0);
  stage_rows<T, V>(in + (size_t)row0 * TV, lds, rows * TV, false, 0.f);
  __syncthreads();
  gcn_rows<T, V, ADJ>(lds, rows, Aw, Tw);
  __syncthreads();
  unstage_rows<T, V>(out + (size_t)row0 * TV, lds, rows * TV);
}

// --------------------------------------------------------------------------------------
// k_bn_fold: BatchNorm statistics -> folded weights.  One block; tiny.
//   a_s = gamma_s / sqrt(var_s + eps),  a_r likewise
//   Wz[o][c] = a_s[o] * Wt[o][c]     Wx[o][c] = a_r[o] * Wr[o][c]   (identity residual: Wx = I)
//   b[o] = beta_s + a_s (bt - mean_s) + beta_r + a_r (br - mean_r)
// --------------------------------------------------------------------------------------
__global__ void k_bn_fold(const float* __restrict__ Wt, const float* __restrict__ bt,
                          const float* __restrict__ gs, const float* __restrict__ bs,
                          const float* __restrict__ mean_s, const float* __restrict__ var_s,
                          const float* __restrict__ Wr, const float* __restrict__ br,
                          const float* __restrict__ gr, const float* __restrict__ brr,
                          const float* __restrict__ mean_r, const float* __restrict__ var_r,
                          float* __restrict__ wfold, float* __restrict__ bias, int Ci, int Co, int CoP) {
  const bool ident = Wr == nullptr;
  for (int i = threadIdx.x; i < 2 * Ci * CoP; i += blockDim.x) {
    const int row = i / CoP, o = i - row * CoP;
    float w = 0.f;
    if (o < Co) {
      if (row < Ci) {
        w = gs[o] / sqrtf(var_s[o] + kBnEps) * Wt[o * Ci + row];
      } else {
        const int c = row - Ci;
        w = ident ? (c == o ? 1.f : 0.f) : gr[o] / sqrtf(var_r[o] + kBnEps) * Wr[o * Ci + c];
      }
    }
    wfold[i] = w;
  }
  for (int o = threadIdx.x; o < CoP; o += blockDim.x) {
    float b = 0.f;
    if (o < Co) {
      const float as = gs[o] / sqrtf(var_s[o] + kBnEps);
      b = bs[o] + as * ((bt ? bt[o] : 0.f) - mean_s[o]);
      if (!ident) {
        const float ar = gr[o] / sqrtf(var_r[o] + kBnEps);
        b += brr[o] + ar * ((br ? br[o] : 0.f) - mean_r[o]);
      }
    }
    bias[o] = b;
  }
}

// elementwise PReLU with one shared slope (nn.PReLU(), stsgcn.py:82,110) for the API paths that must
// materialise a post-activation tensor (the fused chain exchanges pre-activations instead).
__global__ __launch_bounds__(256) void k_prelu_fwd(const float* __restrict__ u, const float* __restrict__ slope,
                                                    float* __restrict__ out, size_t n4, size_t n) {
  const float a = slope[0];
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4 v = reinterpret_cast<const float4*>(u)[i];
    v.x = prelu_f(v.x, a); v.y = prelu_f(v.y, a); v.z = prelu_f(v.z, a); v.w = prelu_f(v.w, a);
    reinterpret_cast<float4*>(out)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const size_t j = 4 * n4 + threadIdx.x;
    out[j] = prelu_f(u[j], a);
  }
}

// dU = dOut * PReLU'(U);  partials[block] = sum dOut * U [U < 0]
__global__ __launch_bounds__(256) void k_prelu_bwd(const float* __restrict__ u, const float* __restrict__ dout,
                                                    const float* __restrict__ slope, float* __restrict__ du,
                                                    float* __restrict__ partials, size_t n) {
  __shared__ float sh[4];
  const float a = slope[0];
  float da = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = u[i], g = dout[i];
    if (x < 0.f) da = fmaf(g, x, da);
    du[i] = x > 0.f ? g : a * g;
  }
  da = wave_sum(da);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = da;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void k_sum_small(const float* __restrict__ v, int n, float* __restrict__ out,
                                                    int accumulate) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + (float)sh[0] : (float)sh[0];
}

// Reconstruction head of the decoder models (F.mse_loss(x_rec, x), euclidean_autoencoder.py:111, spherical_vae.py:90) fused
// with the last decoder layer's PReLU: x_rec = PReLU(U), loss = mean (x_rec - x)^2, dU = upstream * 2 (x_rec - x) / n * PReLU'(U),
// slope gradient.  partials[2 b] = block b's sum of squares, partials[2 b + 1] = its slope-gradient sum.
__global__ __launch_bounds__(256) void k_rec_head(const float* __restrict__ u, const float* __restrict__ x,
                                                   const float* __restrict__ slope, float* __restrict__ xrec,
                                                   float* __restrict__ du, float* __restrict__ partials, float gscale, size_t n) {
  __shared__ float sh[8];
  const float a = slope[0];
  float sq = 0.f, da = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = u[i];
    const float r = v > 0.f ? v : a * v;
    const float d = r - x[i];
    if (xrec) xrec[i] = r;
    sq = fmaf(d, d, sq);
    if (du) {
      const float g = gscale * d;
      if (v < 0.f) da = fmaf(g, v, da);
      du[i] = v > 0.f ? g : a * g;
    }
  }
  sq = wave_sum(sq);
  da = wave_sum(da);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = sq; sh[4 + (threadIdx.x >> 6)] = da; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    partials[2 * blockIdx.x + 1] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
  }
}

__global__ __launch_bounds__(256) void k_rec_head_final(const float* __restrict__ partials, int nblk, double inv_n,
                                                         float* __restrict__ loss, float* __restrict__ dslope, int accumulate) {
  __shared__ double sh[2][256];
  double s = 0.0, d = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { s += (double)partials[2 * i]; d += (double)partials[2 * i + 1]; }
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = d;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sh[0][threadIdx.x] += sh[0][threadIdx.x + w]; sh[1][threadIdx.x] += sh[1][threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = (float)(sh[0][0] * inv_n);
    if (dslope) dslope[0] = accumulate ? dslope[0] + (float)sh[1][0] : (float)sh[1][0];
  }
}

static int pick_nb(int Ci, int B, int LD) {
  // 64 rows per row-phase batch is the sweet spot (one lane per row); cap by LDS.
  int nb = Ci >= 64 ? 1 : 64 / Ci;
  if (nb < 1) nb = 1;
  while (nb > 1 && (size_t)nb * Ci * LD * 4 > 64 * 1024) --nb;
  if (nb > B) nb = B;
  return nb;
}

template <int T, int V>
int launch_layer_apply_m(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold,
                         const float* bias, const float* in_slope, const float* out_slope, int B, int Ci,
                         int Co, hipStream_t st, const float* Zg);  // stsgcn_fwd_mfma.hip

// eval_layer_bpc.hip: the 25-joint layout, 16 / 32 input channels, one clip per four-wave workgroup
bool eval_layer_bpc_ok(int T_, int V_, int Ci, int Co);
int launch_eval_layer_bpc(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold, const float* bias,
                          const float* in_slope, const float* out_slope, int B, int Ci, int Co, int T_, int V_, hipStream_t st);

static int use_mfma() {
#ifdef COSKAD_ABLATE   // A/B builds only: the product library always takes the MFMA kernel when the tile fits LDS
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("COSKAD_VALU_FWD");
    v = (e && e[0] == '1') ? 0 : 1;
  }
  return v;
#else
  return 1;
#endif
}

template <int T, int V>
static int launch_layer_apply(const float* in, float* out, const float* Aw, const float* Tw,
                              const float* wfold, const float* bias, const float* in_slope,
                              const float* out_slope, int B, int Ci, int Co, hipStream_t st) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = Geo<T, V>::LD;
  // COSKAD_EVAL_TILE=1: the round-1 LDS-table kernel at these shapes too (A/B)
  static const bool tile_only = [] { const char* e = getenv("COSKAD_EVAL_TILE"); return e && e[0] == '1'; }();
  if (!tile_only && eval_layer_bpc_ok(T, V, Ci, Co))
    return launch_eval_layer_bpc(in, out, Aw, Tw, wfold, bias, in_slope, out_slope, B, Ci, Co, T, V, st);
  if (use_mfma()) {
    const int rc = launch_layer_apply_m<T, V>(in, out, Aw, Tw, wfold, bias, in_slope, out_slope, B, Ci, Co, st, nullptr);
    if (rc <= 0) return rc;   // 1 = does not fit in LDS: VALU kernel below
  }
  const int CoP = round_up(Co, 16);
  const int NB = pick_nb(Ci, B, LD);
  const size_t lds = (size_t)NB * Ci * LD * sizeof(float);
  if (lds > (size_t)kMaxLdsBytes)
    return fail(COSKAD_ERR_SHAPE, "layer_apply: C_in=%d needs %zu B of LDS (> %d)", Ci, lds, kMaxLdsBytes);
  const int grid = ceil_div(B, NB);
#define LAUNCH_CB(CB)                                                                             \
  do {                                                                                            \
    auto k = k_layer_apply<T, V, CB>;                                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, out, Aw, Tw, wfold, bias,         \
                       in_slope, out_slope, B, Ci, Co, CoP, NB);                                  \
  } while (0)
  if (Ci % 8 == 0) LAUNCH_CB(8);
  else if (Ci % 4 == 0) LAUNCH_CB(4);
  else if (Ci % 2 == 0) LAUNCH_CB(2);
  else LAUNCH_CB(1);
#undef LAUNCH_CB
  return check_launch("layer_apply");
}

template <int T, int V>
static int launch_gcn(const float* in, float* out, const float* Aw, const float* Tw, int rows,
                      int adjoint, hipStream_t st) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = Geo<T, V>::LD;
  const size_t lds = (size_t)64 * LD * sizeof(float);
  const int grid = ceil_div(rows, 64);
  if (adjoint)
    hipLaunchKernelGGL((k_gcn<T, V, true>), dim3(grid), dim3(kBlock), lds, st, in, out, Aw, Tw, rows);
  else
    hipLaunchKernelGGL((k_gcn<T, V, false>), dim3(grid), dim3(kBlock), lds, st, in, out, Aw, Tw, rows);
  return check_launch("gcn");
}

}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_layer_apply_f32(const float* in, float* out, const float* A, const float* Tm,
                           const float* wfold, const float* bias, const float* in_slope,
                           const float* out_slope, int B, int Ci, int Co, int T, int V,
                           hipStream_t stream) {
  if (!in || !out || !A || !Tm || !wfold || !bias) return fail(COSKAD_ERR_ARG, "layer_apply: null pointer");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_apply: B=%d Ci=%d Co=%d", B, Ci, Co);
#define CALL(T_, V_) return launch_layer_apply<T_, V_>(in, out, A, Tm, wfold, bias, in_slope, out_slope, B, Ci, Co, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

int coskad_gcn_f32(const float* in, float* out, const float* A, const float* Tm, int rows, int T, int V,
                   int adjoint, hipStream_t stream) {
  if (!in || !out || !A || !Tm) return fail(COSKAD_ERR_ARG, "gcn: null pointer");
  if (rows <= 0) return fail(COSKAD_ERR_ARG, "gcn: rows=%d", rows);
#define CALL(T_, V_) return launch_gcn<T_, V_>(in, out, A, Tm, rows, adjoint, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

int coskad_prelu_fwd_f32(const float* u, const float* slope, float* out, size_t n, hipStream_t stream) {
  if (!u || !slope || !out || n == 0) return fail(COSKAD_ERR_ARG, "prelu_fwd: bad argument");
  const size_t n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 > 0 ? (n4 + 255) / 256 : 1);
  hipLaunchKernelGGL(k_prelu_fwd, dim3(grid), dim3(256), 0, stream, u, slope, out, n4, n);
  return check_launch("prelu_fwd");
}

/* ws: >= 1024 floats */
int coskad_prelu_bwd_f32(const float* u, const float* dout, const float* slope, float* du, float* dslope,
                         float* ws, int accumulate, size_t n, hipStream_t stream) {
  if (!u || !dout || !slope || !du || !ws || n == 0) return fail(COSKAD_ERR_ARG, "prelu_bwd: bad argument");
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_prelu_bwd, dim3(grid), dim3(256), 0, stream, u, dout, slope, du, ws, n);
  if (dslope) hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(256), 0, stream, ws, grid, dslope, accumulate);
  return check_launch("prelu_bwd");
}

int coskad_bn_fold_f32(const float* Wt, const float* bt, const float* gamma_t, const float* beta_t,
                       const float* mean_t, const float* var_t, const float* Wr, const float* br,
                       const float* gamma_r, const float* beta_r, const float* mean_r,
                       const float* var_r, float* wfold, float* bias, int Ci, int Co,
                       hipStream_t stream) {
  if (!Wt || !gamma_t || !beta_t || !mean_t || !var_t || !wfold || !bias)
    return fail(COSKAD_ERR_ARG, "bn_fold: null pointer");
  if (Wr && (!gamma_r || !beta_r || !mean_r || !var_r)) return fail(COSKAD_ERR_ARG, "bn_fold: residual BN missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "bn_fold: identity residual needs Ci == Co");
  const int CoP = round_up(Co, 16);
  hipLaunchKernelGGL(k_bn_fold, dim3(1), dim3(256), 0, stream, Wt, bt, gamma_t, beta_t, mean_t, var_t,
                     Wr, br, gamma_r, beta_r, mean_r, var_r, wfold, bias, Ci, Co, CoP);
  return check_launch("bn_fold");
}

/* Reconstruction head: x_rec = PReLU_slope(U) (the last decoder layer's activation), loss[0] = mean((x_rec - x)^2)
 * (F.mse_loss, euclidean_autoencoder.py:111 / spherical_vae.py:90); dU (optional) = upstream * dloss/dU, dslope (optional,
 * with dU) (+)= upstream * dloss/dslope; xrec (optional) receives the reconstruction.  ws: >= 2048 floats. */
int coskad_rec_head_f32(const float* U, const float* x, const float* slope, float* xrec, float* dU, float* loss, float* dslope,
                        float upstream, float* ws, int accumulate, size_t n, hipStream_t stream) {
  if (!U || !x || !slope || !loss || !ws || n == 0) return fail(COSKAD_ERR_ARG, "rec_head: bad argument");
  if (dslope && !dU) return fail(COSKAD_ERR_ARG, "rec_head: the slope gradient comes with dU");
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_rec_head, dim3(grid), dim3(256), 0, stream, U, x, slope, xrec, dU, ws, (float)(2.0 * (double)upstream / (double)n), n);
  int rc = check_launch("rec_head");
  if (rc) return rc;
  hipLaunchKernelGGL(k_rec_head_final, dim3(1), dim3(256), 0, stream, ws, grid, 1.0 / (double)n, loss, dslope, accumulate);
  return check_launch("rec_head_final");
}

}  // extern "C"
