// `rev_btlnk` of the decoder models (reference models/sts/ae.py:223-227: nn.Linear(latent_dim -> hidden * T * V) in front of the
// decoder) and its autograd, for latent_dim <= 16:
//     H[b][n]   = sum_l z[b][l] W[n][l] + bias[n]                       forward   (N = hidden T V = 13 056 / 19 200 outputs)
//     dz[b][l] (+)= sum_n dH[b][n] W[n][l]                              data gradient
//     dW[n][l]  = sum_b dH[b][n] z[b][l],   db[n] = sum_b dH[b][n]      parameter gradients
// With a contraction of 8..16 (forward, dW) or an output of 8..16 columns (dz) these are not GEMM-shaped work: the strided MFMA
// GEMM pads them to 64-wide tiles and ran each at ~1 ms for B = 4096 (rocprofv3: 4 of the spherical VAE step's 15 ms).  They are
// streaming kernels over the one large tensor (H / dH, B x N floats), so each is written as one coalesced pass: a thread owns four
// consecutive outputs n (float4 lines), the latent row of a clip is wave-uniform (scalar loads), reductions over clips or over n are
// two-stage in a fixed order (deterministic, no atomics).
#include "common.h"

namespace coskad {
namespace rb {


// H[b][n..n+3] for the clips of slice blockIdx.y; W rows of the thread's four outputs stay in registers
template <int L>
__global__ __launch_bounds__(256) void k_rev_fwd(const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ bias,
                                                 float* __restrict__ H, int B, int N, int chunk) {
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n >= N) return;
  float w[4][L], bv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    bv[u] = bias ? bias[n + u] : 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) w[u][l] = W[(size_t)(n + u) * L + l];
  }
  const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
  for (int b = b0; b < b1; ++b) {
    const float* zb = z + (size_t)b * L;          // wave-uniform: scalar loads
    float o[4] = {bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float zl = zb[l];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] = fmaf(zl, w[u][l], o[u]);
    }
    *reinterpret_cast<float4*>(H + (size_t)b * N + n) = float4{o[0], o[1], o[2], o[3]};
  }
}

// partial dW / db over the clips of slice blockIdx.y: part[slice][n][L + 1] (column L = the bias gradient)
template <int L>
__global__ __launch_bounds__(256) void k_rev_dw(const float* __restrict__ dH, const float* __restrict__ z, float* __restrict__ part, int B,
                                                int N, int chunk) {
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (n >= N) return;
  float acc[4][L + 1];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int l = 0; l <= L; ++l) acc[u][l] = 0.f;
  const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
  for (int b = b0; b < b1; ++b) {
    const float4 g = *reinterpret_cast<const float4*>(dH + (size_t)b * N + n);
    const float gv[4] = {g.x, g.y, g.z, g.w};
    const float* zb = z + (size_t)b * L;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float zl = zb[l];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u][l] = fmaf(gv[u], zl, acc[u][l]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u][L] += gv[u];
  }
  float* dst = part + ((size_t)blockIdx.y * N + n) * (L + 1);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int l = 0; l <= L; ++l) dst[u * (L + 1) + l] = acc[u][l];
}

// dW[n][l] = sum over slices (fp64, fixed order), db[n] likewise
__global__ __launch_bounds__(256) void k_rev_dw_sum(const float* __restrict__ part, int S, int N, int L, float* __restrict__ dW,
                                                    float* __restrict__ db, int accumulate) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t E = (size_t)N * (L + 1);
  if (e >= E) return;
  double s = 0.0;
  for (int k = 0; k < S; ++k) s += (double)part[(size_t)k * E + e];
  const int n = (int)(e / (L + 1)), l = (int)(e - (size_t)n * (L + 1));
  float* out = l < L ? dW + (size_t)n * L + l : (db ? db + n : nullptr);
  if (out) *out = accumulate ? *out + (float)s : (float)s;
}

// dz[b][l] (+)= sum_n dH[b][n] W[n][l]: a block = CB clips, its threads stride over n (float4 of dH per clip, the matching four W
// rows once for all CB clips), then a fixed-order tree over the block
template <int L, int CB>
__global__ __launch_bounds__(256) void k_rev_dz(const float* __restrict__ dH, const float* __restrict__ W, float* __restrict__ dz, int B,
                                                int N, int accumulate) {
  __shared__ float sh[4][CB * L];
  const int b0 = blockIdx.x * CB;
  float acc[CB][L];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int l = 0; l < L; ++l) acc[c][l] = 0.f;
  for (int n = threadIdx.x * 4; n < N; n += 1024) {
    float w[4][L];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int l = 0; l < L; ++l) w[u][l] = W[(size_t)(n + u) * L + l];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      if (b0 + c < B) {
        const float4 g = *reinterpret_cast<const float4*>(dH + (size_t)(b0 + c) * N + n);
#pragma unroll
        for (int l = 0; l < L; ++l) acc[c][l] = fmaf(g.x, w[0][l], fmaf(g.y, w[1][l], fmaf(g.z, w[2][l], fmaf(g.w, w[3][l], acc[c][l]))));
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float s = wave_sum(acc[c][l]);
      if (lane == 0) sh[wave][c * L + l] = s;
    }
  __syncthreads();
  if (threadIdx.x < CB * L) {
    const int c = threadIdx.x / L, l = threadIdx.x - c * L;
    if (b0 + c < B) {
      const float s = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
      float* out = dz + (size_t)(b0 + c) * L + l;
      *out = accumulate ? *out + s : s;
    }
  }
}

// clip slices of the (column blocks) x (slices) grids: enough of them for ~1024 workgroups (a narrow output -- the folded first decoder
// layer's 32 T V columns -- has 7..10 column blocks: at 16 slices the launch filled under half the CUs: k_rev_dw 125 us, k_rev_fwd 61 us
// for 107 MB at B = 4096), at most one slice per 16 clips, at most 64 (the partial dW table is slices x N x (L + 1) floats)
static int slices(int B, int N) {
  if (B < 16) return 1;
  const int gx = ceil_div(N / 4, 256);
  int s = ceil_div(1024, gx);
  if (s < 16) s = 16;
  if (s > 64) s = 64;
  if (s > B / 16) s = B / 16;
  return s < 1 ? 1 : s;
}

}  // namespace rb
}  // namespace coskad

using namespace coskad;

extern "C" {

/* floats of scratch coskad_rev_btlnk_bwd_f32 needs (partial dW / db per clip slice) */
size_t coskad_rev_btlnk_ws_floats(int B, int N, int L) { return (size_t)rb::slices(B, N) * (size_t)N * (L + 1); }

/* H = z W^T + bias  (ae.py:223-227): z [B, L], W [N, L], bias [N] or NULL, H [B, N]; L in {8, 16}, N % 4 == 0 */
int coskad_rev_btlnk_fwd_f32(const float* z, const float* W, const float* bias, float* H, int B, int N, int L, hipStream_t stream) {
  if (!z || !W || !H) return fail(COSKAD_ERR_ARG, "rev_btlnk_fwd: null pointer");
  if (B <= 0 || N <= 0 || N % 4 || (L != 8 && L != 16)) return fail(COSKAD_ERR_SHAPE, "rev_btlnk_fwd: B=%d N=%d L=%d (L in {8,16}, N %% 4 == 0)", B, N, L);
  if ((size_t)H & 15) return fail(COSKAD_ERR_ARG, "rev_btlnk_fwd: H must be 16-byte aligned");
  const int S = B < 64 ? 1 : (B < 1024 ? 4 : rb::slices(B, N));
  const int chunk = ceil_div(B, S);
  dim3 grid(ceil_div(N / 4, 256), S);
  if (L == 8) hipLaunchKernelGGL(rb::k_rev_fwd<8>, grid, dim3(256), 0, stream, z, W, bias, H, B, N, chunk);
  else hipLaunchKernelGGL(rb::k_rev_fwd<16>, grid, dim3(256), 0, stream, z, W, bias, H, B, N, chunk);
  return check_launch("rev_btlnk_fwd");
}

/* autograd of the above: dz [B, L] (+)= dH W (dz_accumulate: the latent already carries another gradient), dW [N, L] and db [N]
 * (+)= (accumulate) their batch sums; ws: coskad_rev_btlnk_ws_floats(B, N, L) floats */
int coskad_rev_btlnk_bwd_f32(const float* dH, const float* z, const float* W, float* dz, int dz_accumulate, float* dW, float* db,
                             int accumulate, float* ws, int B, int N, int L, hipStream_t stream) {
  if (!dH || !z || !W || !dz || !dW || !ws) return fail(COSKAD_ERR_ARG, "rev_btlnk_bwd: null pointer");
  if (B <= 0 || N <= 0 || N % 4 || (L != 8 && L != 16)) return fail(COSKAD_ERR_SHAPE, "rev_btlnk_bwd: B=%d N=%d L=%d (L in {8,16}, N %% 4 == 0)", B, N, L);
  if ((size_t)dH & 15) return fail(COSKAD_ERR_ARG, "rev_btlnk_bwd: dH must be 16-byte aligned");
  const int S = rb::slices(B, N), chunk = ceil_div(B, S);
  dim3 grid(ceil_div(N / 4, 256), S);
  int rc;
  if (L == 8) {
    hipLaunchKernelGGL(rb::k_rev_dw<8>, grid, dim3(256), 0, stream, dH, z, ws, B, N, chunk);
    if ((rc = check_launch("rev_btlnk_dw"))) return rc;
    hipLaunchKernelGGL((rb::k_rev_dz<8, 4>), dim3(ceil_div(B, 4)), dim3(256), 0, stream, dH, W, dz, B, N, dz_accumulate);
  } else {
    hipLaunchKernelGGL(rb::k_rev_dw<16>, grid, dim3(256), 0, stream, dH, z, ws, B, N, chunk);
    if ((rc = check_launch("rev_btlnk_dw"))) return rc;
    hipLaunchKernelGGL((rb::k_rev_dz<16, 2>), dim3(ceil_div(B, 2)), dim3(256), 0, stream, dH, W, dz, B, N, dz_accumulate);
  }
  if ((rc = check_launch("rev_btlnk_dz"))) return rc;
  const size_t E = (size_t)N * (L + 1);
  hipLaunchKernelGGL(rb::k_rev_dw_sum, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, ws, S, N, L, dW, db, accumulate);
  return check_launch("rev_btlnk_dw_sum");
}

}  // extern "C"
