// Bottleneck projection of STSE (reference models/sts/ae.py:97-101,157):
//     z[n][j] = b[j] + sum_k W[j][k] * PReLU(U[n][k]),   k = (c,t,v) flattened, K = hid*T*V
// and its backward.  The encoder's last layer hands over its PRE-activation U; PReLU is
// applied on load (forward) and its derivative on store (backward).
//
// Skinny GEMM (latent <= 16 columns) that must stream U at HBM rate:
//   forward : one block = 16 clips; the 4 waves split K; v_mfma_f32_16x16x4_f32 with
//             A = PReLU(U) [clip][k], B = W^T [k][latent]; every lane loads float4 along k
//             (the 4 components feed 4 successive MFMAs, so any k-permutation is consistent).
//   backward: one wave = a 64-column slab of K; per 16-clip block it loads U once (float4 per
//             lane) and uses it three ways: PReLU(U) as the B operand of dW += dz^T X, U's sign
//             as the mask of dU = (dz W) * PReLU'(U), and U itself for the slope gradient.
#include "tile_ops.h"

namespace coskad {

constexpr int kBtlBlock = 256;
constexpr int kBtlWaves = kBtlBlock / 64;
constexpr int kBtlFwdWaves = 8;   // forward: 8 waves split K (more loads in flight per CU)

__global__ __launch_bounds__(64 * kBtlFwdWaves) void k_btlnk_fwd(const float* __restrict__ U,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     const float* __restrict__ slope,
                                                     float* __restrict__ z, int B, int K, int L) {
  __shared__ float red[kBtlFwdWaves][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int n = blockIdx.x * 16 + i;
  const bool pre = slope != nullptr;
  const float a = pre ? slope[0] : 0.f;
  const bool rowok = n < B, colok = i < L;
  const float* up = U + (size_t)(rowok ? n : 0) * K;
  const float* wp = W + (size_t)(colok ? i : 0) * K;
  const int nsteps = ceil_div(K, 16);
  const int per = ceil_div(nsteps, kBtlFwdWaves);
  const int s0 = wave * per, s1 = min(nsteps, s0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // batches of UB k-steps: 2*UB 16-byte loads in flight per lane (the kernel must stream U at HBM rate)
  constexpr int UB = 4;
  for (int sb = s0; sb < s1; sb += UB) {
    float4 x[UB], w[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int k = (sb + u) * 16 + 4 * kk;
      const bool ok = sb + u < s1 && k < K;
      x[u] = (ok && rowok) ? *reinterpret_cast<const float4*>(up + k) : float4{0.f, 0.f, 0.f, 0.f};
      w[u] = (ok && colok) ? *reinterpret_cast<const float4*>(wp + k) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      float4 xv = x[u];
      if (pre) { xv.x = prelu_f(xv.x, a); xv.y = prelu_f(xv.y, a); xv.z = prelu_f(xv.z, a); xv.w = prelu_f(xv.w, a); }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, w[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, w[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, w[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, w[u].w, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
  __syncthreads();
  const int e = threadIdx.x, row = (e >> 4) & 15, col = e & 15;
  const int nn = blockIdx.x * 16 + row;
  if (e < 256 && nn < B && col < L) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kBtlFwdWaves; ++w) s += red[w][e];
    z[(size_t)nn * L + col] = s + (bias ? bias[col] : 0.f);
  }
}

// Split-K variant for large batches: a block = 64 clips (four MFMA row tiles per wave, which share every W operand:
// the L2 traffic of W drops to a quarter of the HBM traffic of U) x one of KS slices of K; the four waves split the
// slice again.  Partials [KS][B][16] are summed in a fixed order by k_btlnk_fwd_sum (deterministic, no atomics).
#ifndef BTL_KS
#define BTL_KS 4      // sweep at B = 4096 (tools/ab_fused.sh, -DBTL_KS / -DBTL_UB): 4 x 2 -> 40.6 us, 8 x 2 -> 44.5, 16 x 2 -> 51.5, 8 x 4 -> 50
#endif
#ifndef BTL_UB
#define BTL_UB 2
#endif
constexpr int kBtlKS = BTL_KS;
__global__ __launch_bounds__(256) void k_btlnk_fwd_t(const float* __restrict__ U, const float* __restrict__ W,
                                                     const float* __restrict__ slope, float* __restrict__ part,
                                                     int B, int K, int L) {
  __shared__ float red[4][4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int clip0 = blockIdx.x * 64;
  const bool pre = slope != nullptr;
  const float a = pre ? slope[0] : 0.f;
  const bool colok = i < L;
  const float* wp = W + (size_t)(colok ? i : 0) * K + 4 * kk;
  const float* up[4];
  bool rowok[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = clip0 + 16 * ct + i;
    rowok[ct] = n < B;
    up[ct] = U + (size_t)(rowok[ct] ? n : 0) * K + 4 * kk;
  }
  const int nsteps = K / 16;                                   // K % 16 == 0 (checked by the launcher)
  const int g = blockIdx.y * 4 + wave, G = gridDim.y * 4;      // this wave's slice of the k-steps
  const int s0 = (int)((long long)nsteps * g / G), s1 = (int)((long long)nsteps * (g + 1) / G);
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int UB = BTL_UB;
  for (int sb = s0; sb < s1; sb += UB) {
    float4 x[4][UB], w[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const bool ok = sb + u < s1;
      const int k = (sb + u) * 16;
      w[u] = (ok && colok) ? *reinterpret_cast<const float4*>(wp + k) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        x[ct][u] = (ok && rowok[ct]) ? *reinterpret_cast<const float4*>(up[ct] + k) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        float4 xv = x[ct][u];
        if (pre) { xv.x = prelu_f(xv.x, a); xv.y = prelu_f(xv.y, a); xv.z = prelu_f(xv.z, a); xv.w = prelu_f(xv.w, a); }
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, w[u].x, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, w[u].y, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, w[u].z, acc[ct], 0, 0, 0);
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, w[u].w, acc[ct], 0, 0, 0);
      }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][ct][(4 * kk + r) * 16 + i] = acc[ct][r];
  __syncthreads();
  // 64 clips x 16 latents = 1024 sums of four wave partials; thread e handles 4 of them
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int ct = e >> 8, rc = e & 255, row = rc >> 4, col = rc & 15;
    const int n = clip0 + 16 * ct + row;
    if (n < B) part[((size_t)blockIdx.y * B + n) * 16 + col] = (red[0][ct][rc] + red[1][ct][rc]) + (red[2][ct][rc] + red[3][ct][rc]);
  }
}

__global__ void k_btlnk_fwd_sum(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ z,
                                int B, int L, int KS) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * 16) return;
  const int n = e >> 4, col = e & 15;
  if (col >= L) return;
  float s = 0.f;
  for (int k = 0; k < KS; ++k) s += part[((size_t)k * B + n) * 16 + col];
  z[(size_t)n * L + col] = s + (bias ? bias[col] : 0.f);
}

// grid = (ceil(K/256), S clip-chunks).  dWp: [S][L][K] partials, dap: [gridDim.x*gridDim.y].
__global__ __launch_bounds__(kBtlBlock) void k_btlnk_bwd(const float* __restrict__ U,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ dz,
                                                     const float* __restrict__ slope,
                                                     float* __restrict__ dU, float* __restrict__ dWp,
                                                     float* __restrict__ dap, int B, int K, int L,
                                                     int chunk) {
  __shared__ float sred[kBtlWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int kc = (blockIdx.x * kBtlWaves + wave) * 64 + 4 * c;  // this lane's 4 columns
  const bool kok = kc < K;
  const bool pre = slope != nullptr;
  const float a = pre ? slope[0] : 0.f;
  // W fragments of the slab: B[kk = j][col] for j0 = 0,4,8,12 -> lane (c, q) holds W[j0+q][kc..kc+3]
  float4 wf[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int j = 4 * g + q;
    wf[g] = (kok && j < L) ? *reinterpret_cast<const float4*>(W + (size_t)j * K + kc)
                           : float4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 dw[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) dw[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float da = 0.f;
  const int nbeg = blockIdx.y * chunk, nend = min(B, nbeg + chunk);
  // U rows of the tile (lane group q owns clips n0 + 4q + s), fetched one tile ahead of their use
  auto load_u = [&](int n0, float4 (&u)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int nn = n0 + 4 * q + s;
      u[s] = (kok && nn < nend) ? *reinterpret_cast<const float4*>(U + (size_t)nn * K + kc) : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  float4 ucur[4], unext[4];
  load_u(nbeg, ucur);
  for (int n0 = nbeg; n0 < nend; n0 += 16) {
    load_u(n0 + 16, unext);
    // dX tile: A[i = clip][kk = j] = dz[n0+i][4g+kk]
    f32x4 dx[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) dx[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nn = n0 + c, j = 4 * g + q;
      const float av = (nn < nend && j < L) ? dz[(size_t)nn * L + j] : 0.f;
      dx[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[g].x, dx[0], 0, 0, 0);
      dx[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[g].y, dx[1], 0, 0, 0);
      dx[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[g].z, dx[2], 0, 0, 0);
      dx[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wf[g].w, dx[3], 0, 0, 0);
    }
    // step s: lane group q owns clip n0 + 4q + s  (K-order of the dW MFMA is ours to choose;
    // this choice makes the U load serve operand, mask and store address at once)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int nn = n0 + 4 * q + s;
      const bool ok = kok && nn < nend;
      const float4 u = ucur[s];
      float4 x = u;
      if (pre) { x.x = prelu_f(u.x, a); x.y = prelu_f(u.y, a); x.z = prelu_f(u.z, a); x.w = prelu_f(u.w, a); }
      // dW[j][k] += dz[n][j] * x[n][k]:  A[i = j][kk = q] = dz[n0+4q+s][j = c]
      const float dzv = (nn < nend && c < L) ? dz[(size_t)nn * L + c] : 0.f;
      dw[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x.x, dw[0], 0, 0, 0);
      dw[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x.y, dw[1], 0, 0, 0);
      dw[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x.z, dw[2], 0, 0, 0);
      dw[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzv, x.w, dw[3], 0, 0, 0);
      // dX row (4q + s) of the tile = register s of dx[m], column c, k = kc + m
      float4 g = {dx[0][s], dx[1][s], dx[2][s], dx[3][s]};
      if (pre) {
        da += (u.x < 0.f ? g.x * u.x : 0.f) + (u.y < 0.f ? g.y * u.y : 0.f) +
              (u.z < 0.f ? g.z * u.z : 0.f) + (u.w < 0.f ? g.w * u.w : 0.f);
        g.x = u.x > 0.f ? g.x : a * g.x; g.y = u.y > 0.f ? g.y : a * g.y;
        g.z = u.z > 0.f ? g.z : a * g.z; g.w = u.w > 0.f ? g.w : a * g.w;
      }
      if (ok) *reinterpret_cast<float4*>(dU + (size_t)nn * K + kc) = g;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) ucur[s] = unext[s];
  }
  // dW partial: tile m, reg r <-> j = 4q + r, k = kc + m
  if (kok) {
    float* dst = dWp + (size_t)blockIdx.y * L * K;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * q + r;
      if (j < L) *reinterpret_cast<float4*>(dst + (size_t)j * K + kc) = float4{dw[0][r], dw[1][r], dw[2][r], dw[3][r]};
    }
  }
  da = wave_sum(da);
  if (lane == 0) sred[wave] = da;
  __syncthreads();
  if (threadIdx.x == 0) dap[blockIdx.y * gridDim.x + blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// All reductions of the bottleneck backward in ONE launch (fp64 accumulate, fp32 store):
//   blocks [0, nE)       : out[e] (+)= sum_p partials[p][e]            (dW, 256 elements per block)
//   blocks [nE, nE + L)  : db[j]  (+)= sum_n dz[n][j]                  (one block per latent column; db may be NULL)
//   block  nE + L        : dslope (+)= sum of the nda block partials   (dslope may be NULL)
//   blocks beyond        : rsum[e] = sum_p rows[p][e] in fp64, 16 columns x 16 row slices per block (the top layer's backward chain
//                          buffer of btlnk_chain.hip: its partial rows are summed in this launch instead of one of their own)
__global__ __launch_bounds__(256) void k_btlnk_reduce(const float* __restrict__ partials, int P, size_t E,
                                                       float* __restrict__ out, const float* __restrict__ dz,
                                                       int B, int L, float* __restrict__ db,
                                                       const float* __restrict__ dap, int nda,
                                                       float* __restrict__ dslope, int accumulate,
                                                       const float* __restrict__ rows, int RP, int RE,
                                                       double* __restrict__ rsum) {
  __shared__ double sh[256];
  const unsigned nE = (unsigned)((E + 255) / 256);
  if (blockIdx.x > nE + L) {
    const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = (int)(blockIdx.x - (nE + L + 1)) * 16 + col;
    double s = 0.0;
    if (e < RE) {
      const float* base = rows + e;
      int p = slice;
      for (; p + 7 * 16 < RP; p += 8 * 16) {            // eight rows of the slice in flight, summed in row order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(p + 16 * u) * RE];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
      }
      for (; p < RP; p += 16) s += (double)base[(size_t)p * RE];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (slice == 0 && e < RE) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += sh[col + 16 * k];
      rsum[e] = t;
    }
    return;
  }
  if (blockIdx.x < nE) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double s = 0.0;
    int p = 0;
    for (; p + 8 <= P; p += 8) {                     // eight rows in flight, summed in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(p + u) * E + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; p < P; ++p) s += (double)partials[(size_t)p * E + e];
    out[e] = accumulate ? out[e] + (float)s : (float)s;
    return;
  }
  const int r = (int)(blockIdx.x - nE);        // 0..L-1: bias column, L: slope
  double s = 0.0;
  if (r < L) {
    if (!db) return;
    for (int n = threadIdx.x; n < B; n += 256) s += (double)dz[(size_t)n * L + r];
  } else {
    if (!dslope) return;
    for (int i = threadIdx.x; i < nda; i += 256) s += (double)dap[i];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* o = r < L ? db + r : dslope;
    o[0] = accumulate ? o[0] + (float)sh[0] : (float)sh[0];
  }
}

// Clip chunks of the backward grid: (column blocks) x (chunks) ~ 1024 = four 4-wave blocks per CU, all resident at once
// (sweep at B = 4096, K = 13 056: 51 x 16 blocks 140 us, 51 x 20 = 1020 blocks 123 us, 51 x 24 138 us, 51 x 32 135 us)
static int btl_chunks(int B, int K) {
  const int gx = ceil_div(K, 256);
  int s = (1024 + gx / 2) / gx;
  const int smax = ceil_div(B, 16);      // at least one 16-clip tile per chunk
  if (s > smax) s = smax;
  if (s > 64) s = 64;
  return s < 1 ? 1 : s;
}

// every reduction of the bottleneck backward in one launch (also used by btlnk_chain.hip)
int launch_btlnk_reduce(const float* partials, int P, size_t E, float* out, const float* dz, int B, int L, float* db,
                        const float* dap, int nda, float* dslope, int accumulate, hipStream_t stream, const float* rows,
                        int RP, int RE, double* rsum) {
  const unsigned extra = rows ? (unsigned)ceil_div(RE, 16) : 0u;
  hipLaunchKernelGGL(k_btlnk_reduce, dim3((unsigned)((E + 255) / 256) + L + 1 + extra), dim3(256), 0, stream, partials, P, E, out, dz,
                     B, L, db, dap, nda, dslope, accumulate, rows, RP, RE, rsum);
  return check_launch("btlnk_bwd_reduce");
}

}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_btlnk_fwd_f32(const float* U, const float* W, const float* bias, const float* slope, float* z,
                         int B, int K, int L, hipStream_t stream) {
  if (!U || !W || !z) return fail(COSKAD_ERR_ARG, "btlnk_fwd: null pointer");
  if (B <= 0 || K <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "btlnk_fwd: B=%d K=%d L=%d", B, K, L);
  if (L > 16) return fail(COSKAD_ERR_SHAPE, "btlnk_fwd: latent_dim=%d > 16 not supported", L);
  if (K % 4) return fail(COSKAD_ERR_SHAPE, "btlnk_fwd: K=%d must be a multiple of 4", K);
  hipLaunchKernelGGL(k_btlnk_fwd, dim3(ceil_div(B, 16)), dim3(64 * kBtlFwdWaves), 0, stream, U, W, bias, slope, z, B, K, L);
  return check_launch("btlnk_fwd");
}

size_t coskad_btlnk_fwd_ws_bytes(int B) { return (size_t)kBtlKS * (size_t)(B > 0 ? B : 0) * 16 * sizeof(float); }

/* z = W . PReLU(U) + b like coskad_btlnk_fwd_f32, as a split-K GEMM with a caller-provided workspace of
 * coskad_btlnk_fwd_ws_bytes(B) bytes: the fast path for large batches (K must be a multiple of 16). */
int coskad_btlnk_fwd_ws_f32(const float* U, const float* W, const float* bias, const float* slope, float* z, void* ws,
                            size_t ws_bytes, int B, int K, int L, hipStream_t stream) {
  if (!U || !W || !z || !ws) return fail(COSKAD_ERR_ARG, "btlnk_fwd_ws: null pointer");
  if (B <= 0 || K <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "btlnk_fwd_ws: B=%d K=%d L=%d", B, K, L);
  if (L > 16) return fail(COSKAD_ERR_SHAPE, "btlnk_fwd_ws: latent_dim=%d > 16 not supported", L);
  if (K % 16) return fail(COSKAD_ERR_SHAPE, "btlnk_fwd_ws: K=%d must be a multiple of 16", K);
  if (ws_bytes < coskad_btlnk_fwd_ws_bytes(B)) return fail(COSKAD_ERR_WORKSPACE, "btlnk_fwd_ws: workspace too small");
  float* part = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(k_btlnk_fwd_t, dim3(ceil_div(B, 64), kBtlKS), dim3(256), 0, stream, U, W, slope, part, B, K, L);
  int rc;
  if ((rc = check_launch("btlnk_fwd_t"))) return rc;
  hipLaunchKernelGGL(k_btlnk_fwd_sum, dim3(ceil_div(B * 16, 256)), dim3(256), 0, stream, part, bias, z, B, L, kBtlKS);
  return check_launch("btlnk_fwd_sum");
}

size_t coskad_btlnk_bwd_ws_bytes(int B, int K, int L) {
  const int S = btl_chunks(B, K);
  return ((size_t)S * L * K + (size_t)S * ceil_div(K, 256) + 64) * sizeof(float);
}

/* dU = (dz W) * PReLU'(U);  dW (+)= dz^T PReLU(U);  db (+)= sum dz;  dslope (+)= sum (dz W) U [U<0] */
int coskad_btlnk_bwd_f32(const float* U, const float* W, const float* dz, const float* slope, float* dU,
                         float* dW, float* db, float* dslope, void* ws, size_t ws_bytes, int accumulate,
                         int B, int K, int L, hipStream_t stream) {
  if (!U || !W || !dz || !dU || !dW || !ws) return fail(COSKAD_ERR_ARG, "btlnk_bwd: null pointer");
  if (B <= 0 || K <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "btlnk_bwd: B=%d K=%d L=%d", B, K, L);
  if (L > 16) return fail(COSKAD_ERR_SHAPE, "btlnk_bwd: latent_dim=%d > 16 not supported", L);
  if (K % 4) return fail(COSKAD_ERR_SHAPE, "btlnk_bwd: K=%d must be a multiple of 4", K);
  if (ws_bytes < coskad_btlnk_bwd_ws_bytes(B, K, L)) return fail(COSKAD_ERR_WORKSPACE, "btlnk_bwd: workspace too small");
  const int S = btl_chunks(B, K);
  const int chunk = round_up(ceil_div(B, S), 16);
  const int gx = ceil_div(K, 256);
  float* dWp = reinterpret_cast<float*>(ws);
  float* dap = dWp + (size_t)S * L * K;
  {
    ProbeScope probe(KID_BTLNK_BWD, 0, L, stream);
    hipLaunchKernelGGL(k_btlnk_bwd, dim3(gx, S), dim3(kBtlBlock), 0, stream, U, W, dz, slope, dU, dWp, dap, B, K, L, chunk);
  }
  int rc = check_launch("btlnk_bwd");
  if (rc) return rc;
  return launch_btlnk_reduce(dWp, S, (size_t)L * K, dW, dz, B, L, db, dap, gx * S, (dslope && slope) ? dslope : nullptr, accumulate, stream,
                             nullptr, 0, 0, nullptr);
}

}  // extern "C"
