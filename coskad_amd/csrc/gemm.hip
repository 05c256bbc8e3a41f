// Strided, batched fp32 GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains) with fused epilogues -- the GEMM-shaped
// work of the reference's plain-GCN encoders (models/graph_layers/learnable_gcn.py:65-72, gcn.py:48-54: X W, then the
// dense (T V x T V) adjacency) and of the wide ST-GCN layers' 1x1 convolutions (stsgcn.py:56-80 at C > 64), forward and
// both backward products, all in the tensors' native layouts (any strides: transposes are views, nothing is copied).
//
//   C[b][m][n] = act( sum_k A[b][m][k] * B[b][k][n] + bias )                      (reduce == 0)
//   P[c][m][n] = sum_{b in chunk c} sum_k A[b][m][k] * B[b][k][n]                 (reduce != 0: weight gradients;
//                partial sums per batch chunk, summed in a fixed order by coskad_gemm_sum_f32 -- deterministic)
//
// Block = 4 waves = a 64 x 64 tile of C, K in steps of 16 through double-buffered LDS (k-major images, so that an MFMA
// operand read is 16 consecutive floats per k: conflict-free); a wave owns 32 x 32 (2 x 2 MFMA tiles).  Global loads map
// consecutive threads onto whichever operand dimension is contiguous in memory.
#include "common.h"

namespace coskad {
namespace gemm {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 16;

struct Args {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long long sa_b, sa_m, sa_k, sb_b, sb_k, sb_n, sc_b, sc_m, sc_n;
  int M, N, K, batch;
  int bias_mode;      // 0 none, 1 bias[m % bias_mod], 2 bias[n]
  int bias_mod;
  int relu;
  int reduce;         // != 0: sum over the batches of a chunk, C = partials [chunks][M][N] (contiguous)
  int chunk;          // batches per chunk (reduce mode)
  long long ktotal;   // > 0: element (b, k) exists iff b * K + k < ktotal  (a long reduction axis cut into `batch` pieces)
  int accum;          // != 0 (non-reduce): C += result
};

// TM x TN MFMA tiles per wave: block tile = (32 TM) x (32 TN).  2 x 2 for small / skinny products, 4 x 4 (128 x 128 per
// block, 0.5 LDS operand reads per MFMA) when both M and N are large.
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm(Args a) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  constexpr int LDA = BM + 4, LDB = BN + 4;      // LDS row strides: + 4 keeps 16-float operand reads aligned and conflict-free
  constexpr int EA = BM / 16, EB = BN / 16;      // elements per thread and K-tile
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wm = (wave >> 1) * 16 * TM, wn = (wave & 1) * 16 * TN;
  const int j = lane & 15, q = lane >> 4;
  int b_first, b_last;
  if (a.reduce) {
    b_first = blockIdx.z * a.chunk;
    b_last = min(a.batch, b_first + a.chunk);
  } else {
    b_first = blockIdx.z;
    b_last = b_first + 1;
  }
  // global -> register mapping: 4 elements per thread and operand, consecutive threads along the contiguous dimension
  const bool a_kc = a.sa_k == 1;                 // A contiguous along k (else along m or generic)
  const bool b_nc = a.sb_n == 1;                 // B contiguous along n
  int am[EA], ak[EA], bk[EB], bn[EB];
#pragma unroll
  for (int i = 0; i < EA; ++i) {
    if (a_kc) { am[i] = (tid >> 4) + 16 * i; ak[i] = tid & 15; }
    else { am[i] = tid % BM; ak[i] = tid / BM + (256 / BM) * i; }
  }
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    if (b_nc) { bn[i] = tid % BN; bk[i] = tid / BN + (256 / BN) * i; }
    else { bn[i] = (tid >> 4) + 16 * i; bk[i] = tid & 15; }
  }
  f32x4 acc[TM][TN];
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TN; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ktiles = (a.K + BK - 1) / BK;
  const int total = (b_last - b_first) * ktiles;
  float ra[EA], rb[EB];
  auto gload = [&](int it) {
    const int b = b_first + it / ktiles, k0 = (it % ktiles) * BK;
    const float* Ab = a.A + (long long)b * a.sa_b;
    const float* Bb = a.B + (long long)b * a.sb_b;
    const long long kbase = (long long)b * a.K;
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int m = m0 + am[i], k = k0 + ak[i];
      const bool ok = m < a.M && k < a.K && (a.ktotal <= 0 || kbase + k < a.ktotal);
      ra[i] = ok ? Ab[(long long)m * a.sa_m + (long long)k * a.sa_k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int n = n0 + bn[i], kb = k0 + bk[i];
      const bool okb = n < a.N && kb < a.K && (a.ktotal <= 0 || kbase + kb < a.ktotal);
      rb[i] = okb ? Bb[(long long)kb * a.sb_k + (long long)n * a.sb_n] : 0.f;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < EA; ++i) As[buf][ak[i]][am[i]] = ra[i];
#pragma unroll
    for (int i = 0; i < EB; ++i) Bs[buf][bk[i]][bn[i]] = rb[i];
  };
  if (total > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    if (it + 1 < total) gload(it + 1);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      float av[TM], bv[TN];
#pragma unroll
      for (int x = 0; x < TM; ++x) av[x] = As[buf][4 * s + q][wm + 16 * x + j];
#pragma unroll
      for (int y = 0; y < TN; ++y) bv[y] = Bs[buf][4 * s + q][wn + 16 * y + j];
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], bv[y], acc[x][y], 0, 0, 0);
    }
    if (it + 1 < total) sstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: D[row = 4q + r][col = j] of tile (x, y)
  float* Cb;
  long long scm, scn;
  if (a.reduce) { Cb = a.C + (long long)blockIdx.z * a.M * a.N; scm = a.N; scn = 1; }
  else { Cb = a.C + (long long)b_first * a.sc_b; scm = a.sc_m; scn = a.sc_n; }
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TN; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + 16 * x + 4 * q + r, n = n0 + wn + 16 * y + j;
        if (m < a.M && n < a.N) {
          float v = acc[x][y][r];
          if (!a.reduce) {
            if (a.bias_mode == 1) v += a.bias[m % a.bias_mod];
            else if (a.bias_mode == 2) v += a.bias[n];
            if (a.relu) v = v > 0.f ? v : 0.f;
            if (a.accum) v += Cb[(long long)m * scm + (long long)n * scn];
          }
          Cb[(long long)m * scm + (long long)n * scn] = v;
        }
      }
}

// out[e] (+)= sum_c partials[c][e] (fp64, fixed order)
// block = 64 elements x 16 chunk slices, four loads in flight per thread, slices combined in a fixed order (one thread walking all
// chunks of an element took 62 us for 256 chunks of a 32 x 64 weight gradient: 8 blocks of dependent loads)
__global__ __launch_bounds__(1024) void k_sum(const float* __restrict__ part, int chunks, size_t E, float* __restrict__ out, int accumulate) {
  __shared__ double sh[1024];
  const size_t e = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < E) {
    int c = slice;
    for (; c + 48 < chunks; c += 64) {
      s0 += (double)part[(size_t)c * E + e];
      s1 += (double)part[(size_t)(c + 16) * E + e];
      s2 += (double)part[(size_t)(c + 32) * E + e];
      s3 += (double)part[(size_t)(c + 48) * E + e];
    }
    for (; c < chunks; c += 16) s0 += (double)part[(size_t)c * E + e];
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && e < E) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[threadIdx.x + 64 * k];
    out[e] = accumulate ? out[e] + (float)t : (float)t;
  }
}

// g = dout * (out > 0);  dbias[c] (+)= sum over (n, p) of g[n][c][p]: block = one (channel, slice) pair -> partials
__global__ __launch_bounds__(256) void k_relu_bwd(const float* __restrict__ out, const float* __restrict__ dout,
                                                  float* __restrict__ g, float* __restrict__ part, int Nb, int C, int P,
                                                  int slices) {
  __shared__ double sh[256];
  const int c = blockIdx.x, sl = blockIdx.y;
  double s = 0.0;
  for (int n = sl; n < Nb; n += slices) {
    const size_t base = ((size_t)n * C + c) * P;
    for (int p = threadIdx.x; p < P; p += 256) {
      const float v = out[base + p] > 0.f ? dout[base + p] : 0.f;
      g[base + p] = v;
      s += (double)v;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)sl * C + c] = (float)sh[0];
}

// row softmax of a small square matrix and its backward (learnable_gcn.py:36,66: nn.Softmax() on the 2-D Adj = dim 1)
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ x, float* __restrict__ y, int n) {
  __shared__ float sh[256];
  const int r = blockIdx.x;
  float mx = -3.4e38f;
  for (int c = threadIdx.x; c < n; c += 256) mx = fmaxf(mx, x[(size_t)r * n + c]);
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + w]); __syncthreads(); }
  mx = sh[0];
  __syncthreads();
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += expf(x[(size_t)r * n + c] - mx);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  const float inv = 1.f / sh[0];
  for (int c = threadIdx.x; c < n; c += 256) y[(size_t)r * n + c] = expf(x[(size_t)r * n + c] - mx) * inv;
}
__global__ __launch_bounds__(256) void k_softmax_rows_bwd(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int n) {
  __shared__ float sh[256];
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += y[(size_t)r * n + c] * dy[(size_t)r * n + c];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  const float dot = sh[0];
  for (int c = threadIdx.x; c < n; c += 256) dx[(size_t)r * n + c] = y[(size_t)r * n + c] * (dy[(size_t)r * n + c] - dot);
}

}  // namespace gemm
}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_gemm_f32(const float* A, const float* B, float* C, const float* bias, long long sa_b, long long sa_m,
                    long long sa_k, long long sb_b, long long sb_k, long long sb_n, long long sc_b, long long sc_m,
                    long long sc_n, int M, int N, int K, int batch, int bias_mode, int bias_mod, int relu, int reduce,
                    int chunk, long long ktotal, int accum, hipStream_t stream) {
  if (!A || !B || !C) return fail(COSKAD_ERR_ARG, "gemm: null pointer");
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return fail(COSKAD_ERR_ARG, "gemm: M=%d N=%d K=%d batch=%d", M, N, K, batch);
  if (bias_mode && !bias) return fail(COSKAD_ERR_ARG, "gemm: bias_mode %d without bias", bias_mode);
  if (bias_mode == 1 && bias_mod <= 0) return fail(COSKAD_ERR_ARG, "gemm: bias_mod=%d", bias_mod);
  if (reduce && chunk <= 0) return fail(COSKAD_ERR_ARG, "gemm: reduce mode needs chunk > 0");
  gemm::Args a{A, B, C, bias, sa_b, sa_m, sa_k, sb_b, sb_k, sb_n, sc_b, sc_m, sc_n, M, N, K, batch, bias_mode, bias_mod, relu,
               reduce, chunk, ktotal, accum};
  const int gz = reduce ? ceil_div(batch, chunk) : batch;
  if (gz > 65535) return fail(COSKAD_ERR_SHAPE, "gemm: %d batches/chunks exceed the grid limit", gz);
  if (M >= 128 && N >= 128)      // 128 x 128 block tiles: half the LDS operand traffic per MFMA
    hipLaunchKernelGGL((gemm::k_gemm<4, 4>), dim3(ceil_div(N, 128), ceil_div(M, 128), gz), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((gemm::k_gemm<2, 2>), dim3(ceil_div(N, 64), ceil_div(M, 64), gz), dim3(256), 0, stream, a);
  return check_launch("gemm");
}

/* out[e] (+)= sum_c partials[c][e], e < E (fp64 accumulation, fixed order) */
int coskad_gemm_sum_f32(const float* partials, int chunks, size_t E, float* out, int accumulate, hipStream_t stream) {
  if (!partials || !out || chunks <= 0 || E == 0) return fail(COSKAD_ERR_ARG, "gemm_sum: bad argument");
  hipLaunchKernelGGL(gemm::k_sum, dim3((unsigned)((E + 63) / 64)), dim3(1024), 0, stream, partials, chunks, E, out, accumulate);
  return check_launch("gemm_sum");
}

/* g = dout * (out > 0) on [Nb, C, P]; part [slices][C] receives per-slice channel sums of g (sum them with coskad_gemm_sum_f32) */
int coskad_relu_bwd_f32(const float* out, const float* dout, float* g, float* part, int Nb, int C, int P, int slices,
                        hipStream_t stream) {
  if (!out || !dout || !g || !part) return fail(COSKAD_ERR_ARG, "relu_bwd: null pointer");
  if (Nb <= 0 || C <= 0 || P <= 0 || slices <= 0 || C > 65535 || slices > 65535) return fail(COSKAD_ERR_ARG, "relu_bwd: bad sizes");
  hipLaunchKernelGGL(gemm::k_relu_bwd, dim3(C, slices), dim3(256), 0, stream, out, dout, g, part, Nb, C, P, slices);
  return check_launch("relu_bwd");
}

int coskad_softmax_rows_f32(const float* x, float* y, int n, hipStream_t stream) {
  if (!x || !y || n <= 0) return fail(COSKAD_ERR_ARG, "softmax_rows: bad argument");
  hipLaunchKernelGGL(gemm::k_softmax_rows, dim3(n), dim3(256), 0, stream, x, y, n);
  return check_launch("softmax_rows");
}
int coskad_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, int n, hipStream_t stream) {
  if (!y || !dy || !dx || n <= 0) return fail(COSKAD_ERR_ARG, "softmax_rows_bwd: bad argument");
  hipLaunchKernelGGL(gemm::k_softmax_rows_bwd, dim3(n), dim3(256), 0, stream, y, dy, dx, n);
  return check_launch("softmax_rows_bwd");
}

}  // extern "C"
