// 1x1 convolution of an ST_GCNN layer in NCHW (reference models/graph_layers/stsgcn.py:57-63,71-75: nn.Conv2d(C_in, C_out, 1)) for
// layers beyond the LDS-resident tile kernels (C > 64: the `C = 2 -> 256` stack of BASELINE.json's north_star; 64 input channels on
// the 25-joint layout), forward and data gradient, on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains):
//
//     Out[b][m][p] (+)= sum_k A(m, k) In[b][k][p] (+ bias[m])        In [batch][K][P], Out [batch][M][P] contiguous, P % 4 == 0
//     A(m, k) = Aw[m sa_m + k sa_k]:  forward A = W [M][K] (sa_k = 1);  data gradient A = W^T of W [K][M] (sa_m = 1)
//
// The generic strided GEMM (gemm.hip: scalar loads with 64-bit index arithmetic per element, 16-wide K tiles, scalar stores) runs
// these shapes at ~40 % of the fp32 MFMA peak.  This kernel is specialised to the layout instead:
//  * a block = WM x WC waves (8 for the large shapes: two per SIMD, <= 256 registers each) owns MB = 16 RT WM output channels of WC
//    clips; a wave RT x NT MFMA tiles (104 accumulator registers at P = 204), all P positions of its clip, so an activation row is
//    read from HBM once per call (M = 256: one block covers all channels);
//  * K in tiles of 16: the activation tile of a clip is ONE contiguous run of 16 P floats (float4 loads, float4 LDS stores), the
//    weight tile goes k-major; both double-buffered through registers (loads of tile i+1 fly while tile i multiplies), one block
//    barrier per tile, the pipeline runs across the block's clips (persistent blocks);
//  * the product is formed TRANSPOSED (A operand = activations, B operand = weights): an accumulator register quad is four
//    consecutive positions of one channel -> one 16-byte store per tile and lane, the channel on the lane -> bias, and the
//    per-channel sum / sum of squares of the output (the BatchNorm statistics of stsgcn.py:65,76 in training mode) in the epilogue
//    without another pass over the tensor (coskad_bn2_stats_parts_f32 finishes them).
#include "common.h"

namespace coskad {
namespace cv {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BK = 16;

struct Args {
  const float* A;
  const float* In;
  float* Out;
  const float* bias;
  double* stats;        // nullptr, or [gridDim.x][M][2]: per-block-column sum / sum of squares of Out per channel
  long long sa_m, sa_k;
  int M, K, P, batch, accumulate;
};

template <int RT, int NT, int WM, int WC>
__global__ __launch_bounds__(64 * WM * WC, (WM * WC) / 4) void k_conv1x1(Args a) {
  constexpr int NW = WM * WC, NTHR = 64 * NW;  // waves / threads per block: WM waves along the channels x WC clips
  constexpr int MB = 16 * RT * WM;             // output channels per block
  constexpr int PS = 16 * NT;                  // LDS row stride of an activation tile (P + 4 == PS for P = 204 / 300)
  constexpr int LA = MB + 16;                  // k-major weight tile: stride = 16 (mod 32) keeps the (k, channel) operand reads conflict-free
  constexpr int XF4 = (BK * PS / 4 + NTHR - 1) / NTHR;   // float4 per thread of one clip's activation tile (upper bound)
  constexpr int AE4 = MB * BK / 4;             // float4 of a weight tile
  constexpr int AF4 = (AE4 + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                              // [2][WC][BK][PS]
  float* As = lds + 2 * WC * BK * PS;           // [2][BK][LA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int wm = wave % WM, wc = wave / WM;
  const int m0 = blockIdx.y * MB;
  const int P = a.P, P4 = P / 4;
  const int ktiles = a.K / BK;
  const int ngroups = (a.batch + WC - 1) / WC;
  const bool a_kc = a.sa_k == 1;

  float4 rx[WC][XF4], ra[AF4];
  auto gload = [&](int g, int k0) {
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int b = g * WC + c;
      const float* src = a.In + ((size_t)(b < a.batch ? b : 0) * a.K + k0) * P;
#pragma unroll
      for (int i = 0; i < XF4; ++i) {
        const int e = tid + NTHR * i;
        rx[c][i] = (b < a.batch && e < BK * P4) ? *reinterpret_cast<const float4*>(src + 4 * e) : float4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      const int e = tid + NTHR * i;
      if (AE4 % NTHR == 0 || e < AE4) {
        if (a_kc) {                 // W [M][K]: a float4 = four consecutive k of one channel
          const int m = e >> 2, kq = e & 3;
          ra[i] = *reinterpret_cast<const float4*>(a.A + (size_t)(m0 + m) * a.sa_m + k0 + 4 * kq);
        } else {                    // W^T: a float4 = four consecutive channels of one k
          const int k = e / (MB / 4), mq = e - k * (MB / 4);
          ra[i] = *reinterpret_cast<const float4*>(a.A + (size_t)(k0 + k) * a.sa_k + m0 + 4 * mq);
        }
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int i = 0; i < XF4; ++i) {
        const int e = tid + NTHR * i;
        if (e < BK * P4) {
          const int row = e / P4, col = 4 * (e - row * P4);
          *reinterpret_cast<float4*>(Xs + ((buf * WC + c) * BK + row) * PS + col) = rx[c][i];
        }
      }
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      const int e = tid + NTHR * i;
      if (AE4 % NTHR == 0 || e < AE4) {
        float* dst = As + buf * BK * LA;
        if (a_kc) {
          const int m = e >> 2, kq = e & 3;
          dst[(4 * kq + 0) * LA + m] = ra[i].x; dst[(4 * kq + 1) * LA + m] = ra[i].y;
          dst[(4 * kq + 2) * LA + m] = ra[i].z; dst[(4 * kq + 3) * LA + m] = ra[i].w;
        } else {
          const int k = e / (MB / 4), mq = e - k * (MB / 4);
          *reinterpret_cast<float4*>(dst + k * LA + 4 * mq) = ra[i];
        }
      }
    }
  };

  float s1[RT], s2[RT], bq[RT];
#pragma unroll
  for (int x = 0; x < RT; ++x) {
    s1[x] = 0.f; s2[x] = 0.f;
    bq[x] = a.bias ? a.bias[m0 + 16 * (wm * RT + x) + j] : 0.f;
  }
  int buf = 0;
  if ((int)blockIdx.x < ngroups) {
    gload(blockIdx.x, 0);
    sstore(0);
  }
  __syncthreads();
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    f32x4 acc[RT][NT];
#pragma unroll
    for (int x = 0; x < RT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ktiles; ++kt) {
      // the next tile (of this clip group, or the first of the block's next group) flies while this one multiplies
      const bool last = kt + 1 == ktiles;
      const int gn = last ? g + gridDim.x : g, kn = last ? 0 : (kt + 1) * BK;
      const bool more = gn < ngroups;
      if (more) gload(gn, kn);
      const float* xb = Xs + ((buf * WC + wc) * BK) * PS;
      const float* ab = As + buf * BK * LA + 16 * wm * RT;
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float wv[RT], xv[NT];
#pragma unroll
        for (int x = 0; x < RT; ++x) wv[x] = ab[(4 * s + q) * LA + 16 * x + j];
#pragma unroll
        for (int y = 0; y < NT; ++y) xv[y] = xb[(4 * s + q) * PS + 16 * y + j];
        // transposed product: D[position 4 q + r of tile y][channel j of row tile x]
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
          for (int x = 0; x < RT; ++x) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[y], wv[x], acc[x][y], 0, 0, 0);
      }
      if (more) sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    // ---- the clip group is complete: bias, statistics, 16-byte stores (four consecutive positions of one channel per lane) ------
    const int b = g * WC + wc;
    if (b < a.batch) {
#pragma unroll
      for (int x = 0; x < RT; ++x) {
        float* orow = a.Out + ((size_t)b * a.M + m0 + 16 * (wm * RT + x) + j) * P;
#pragma unroll
        for (int y = 0; y < NT; ++y) {
          const int p0 = 16 * y + 4 * q;
          if (p0 < P) {
            f32x4 v = acc[x][y] + bq[x];
            if (a.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(orow + p0);
              v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
            }
            s1[x] += (v[0] + v[1]) + (v[2] + v[3]);
            s2[x] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            *reinterpret_cast<float4*>(orow + p0) = float4{v[0], v[1], v[2], v[3]};
          }
        }
      }
    }
  }
  if (a.stats) {
    // per channel: the four position-quad lanes (q) of a channel, then the WC clip waves of the block (through LDS), one row per block column
    __syncthreads();
    float* sh = lds;                              // [WC][MB][2]
#pragma unroll
    for (int x = 0; x < RT; ++x) {
      float t1 = s1[x], t2 = s2[x];
      t1 += __shfl_xor(t1, 16, 64); t2 += __shfl_xor(t2, 16, 64);
      t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) {
        const int ch = 16 * (wm * RT + x) + j;
        sh[(wc * MB + ch) * 2] = t1;
        sh[(wc * MB + ch) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    for (int ch = tid; ch < MB; ch += NTHR) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int c = 0; c < WC; ++c) { t1 += (double)sh[(c * MB + ch) * 2]; t2 += (double)sh[(c * MB + ch) * 2 + 1]; }
      double* dst = a.stats + ((size_t)blockIdx.x * a.M + m0 + ch) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

// ---- weight gradient: part[z][m][k] = sum over the clips of chunk z and all positions of G[b][m][p] X[b][k][p] ------------------------
// (autograd of nn.Conv2d(C_in, C_out, 1): G = the gradient of the conv output [batch][M][P], X = the conv input [batch][K][P]).
// A block owns an (16 TM WMR) x (16 TK WKR) tile of the M x K result; the contraction runs over positions, so both MFMA operands are
// (row, position) reads of LDS row images (stride PC + 2 = 2 (mod 4): conflict-free 8-byte reads, one ds_read_b64 serves two
// k-steps -- any assignment of positions to k slots is as good as any other as long as both operands use the same one).  A clip
// passes in P / PC position chunks, double-buffered through registers; chunks of clips give partial results that
// coskad_gemm_sum_f32 adds in a fixed order (deterministic).
template <int WMR, int WKR, int TM, int TK, int PC>
__global__ __launch_bounds__(64 * WMR * WKR, (WMR * WKR) / 4) void k_conv1x1_wgrad(const float* __restrict__ G,
                                                                                 const float* __restrict__ X,
                                                                                 float* __restrict__ part, int M, int K, int P,
                                                                                 int batch, int chunk) {
  constexpr int NTHR = 64 * WMR * WKR, MT = 16 * TM * WMR, KT = 16 * TK * WKR, ROWS = MT + KT;
  constexpr int LDP = PC + 2, PC4 = PC / 4;
  constexpr int NF4 = (ROWS * PC4 + NTHR - 1) / NTHR;
  constexpr int ND = PC / 8, REM = PC % 8;
  static_assert(PC % 4 == 0 && (REM == 0 || REM == 4), "position chunks are whole k-steps");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][ROWS][LDP]: G rows then X rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int wmr = wave % WMR, wkr = wave / WMR;
  const int m0 = blockIdx.y * MT, k0 = blockIdx.x * KT;
  const int b_first = blockIdx.z * chunk, b_last = min(batch, b_first + chunk);
  const int nchunk = P / PC;
  const int total = (b_last - b_first) * nchunk;
  float4 rg[NF4];
  auto gload = [&](int it) {
    const int b = b_first + it / nchunk, pc = (it % nchunk) * PC;
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int e = tid + NTHR * i;
      if ((ROWS * PC4) % NTHR == 0 || e < ROWS * PC4) {
        const int row = e / PC4, c4 = e - row * PC4;
        const float* src = row < MT ? G + ((size_t)b * M + m0 + row) * P : X + ((size_t)b * K + k0 + row - MT) * P;
        rg[i] = *reinterpret_cast<const float4*>(src + pc + 4 * c4);
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int e = tid + NTHR * i;
      if ((ROWS * PC4) % NTHR == 0 || e < ROWS * PC4) {
        const int row = e / PC4, c4 = e - row * PC4;
        float* dst = lds + (buf * ROWS + row) * LDP + 4 * c4;
        *reinterpret_cast<float2*>(dst) = float2{rg[i].x, rg[i].y};
        *reinterpret_cast<float2*>(dst + 2) = float2{rg[i].z, rg[i].w};
      }
    }
  };
  f32x4 acc[TM][TK];
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TK; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (total > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  for (int it = 0; it < total; ++it) {
    const int buf = it & 1;
    if (it + 1 < total) gload(it + 1);
    const float* gp = lds + (buf * ROWS + 16 * TM * wmr + j) * LDP + 2 * q;
    const float* xp = lds + (buf * ROWS + MT + 16 * TK * wkr + j) * LDP + 2 * q;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
      float2 av[TM], bv[TK];
#pragma unroll
      for (int x = 0; x < TM; ++x) av[x] = *reinterpret_cast<const float2*>(gp + 16 * x * LDP + 8 * d);
#pragma unroll
      for (int y = 0; y < TK; ++y) bv[y] = *reinterpret_cast<const float2*>(xp + 16 * y * LDP + 8 * d);
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TK; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x].x, bv[y].x, acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x].y, bv[y].y, acc[x][y], 0, 0, 0);
        }
    }
    if (REM) {                                   // the chunk's last four positions: one k-step, position 8 ND + q
      float av[TM], bv[TK];
#pragma unroll
      for (int x = 0; x < TM; ++x) av[x] = gp[16 * x * LDP + 8 * ND - q];          // (gp already carries + 2 q)
#pragma unroll
      for (int y = 0; y < TK; ++y) bv[y] = xp[16 * y * LDP + 8 * ND - q];
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TK; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], bv[y], acc[x][y], 0, 0, 0);
    }
    if (it + 1 < total) sstore(buf ^ 1);
    __syncthreads();
  }
  // D[m = 4 q + r][k = j] of tile (x, y)
  float* dst = part + (size_t)blockIdx.z * M * K;
#pragma unroll
  for (int x = 0; x < TM; ++x)
#pragma unroll
    for (int y = 0; y < TK; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dst[(size_t)(m0 + 16 * (TM * wmr + x) + 4 * q + r) * K + k0 + 16 * (TK * wkr + y) + j] = acc[x][y][r];
}

// stat[c] = mean, stat[C + c] = invstd from [S][C][2] partial sums; running statistics updated (wide.hip: k_stats_final's twin
// for partials that arrive from a GEMM epilogue)
__global__ __launch_bounds__(256) void k_stats_from_parts(const double* __restrict__ part, int S, float* __restrict__ stat,
                                                          float* __restrict__ rmean, float* __restrict__ rvar,
                                                          long long* __restrict__ nbt, float momentum, float eps, double count, int C) {
  // a block = 16 channels x 16 row slices (fixed-order tree over the slices)
  __shared__ double sh[2][16][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int k = sl; k < S; k += 16) {
      s += part[((size_t)k * C + c) * 2];
      q += part[((size_t)k * C + c) * 2 + 1];
    }
  sh[0][sl][cl] = s;
  sh[1][sl][cl] = q;
  __syncthreads();
  if (sl != 0 || c >= C) return;
  s = 0.0; q = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
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

struct Shape {
  int rt, nt, wm, wc;
};
// tile configuration for (M, P): RT x NT MFMA tiles per wave (104 accumulators at P = 204: two waves per SIMD), WM waves along
// the channels x WC clips per block
static bool pick(int M, int P, Shape* s) {
  const int nt = (P + 4 + 15) / 16;            // one padding quad behind the row: 204 -> 13 tiles, 300 -> 19
  if (P % 4 != 0 || (nt != 13 && nt != 19)) return false;
  if (nt == 13) {
    if (M % 256 == 0) { *s = {2, 13, 8, 1}; return true; }
    if (M % 128 == 0) { *s = {2, 13, 4, 2}; return true; }
    if (M % 64 == 0) { *s = {2, 13, 2, 4}; return true; }
    if (M % 32 == 0) { *s = {2, 13, 1, 4}; return true; }
  } else {
    if (M % 64 == 0) { *s = {2, 19, 2, 2}; return true; }
    if (M % 32 == 0) { *s = {2, 19, 1, 4}; return true; }
  }
  return false;
}

}  // namespace cv
}  // namespace coskad

using namespace coskad;

extern "C" {

/* 1 when coskad_conv1x1_f32 takes the shape: P in {204, 300} (T = 12, V = 17 / 25), K % 16 == 0, M % 32 == 0 (64 at P = 300 ...) */
int coskad_conv1x1_ok(int M, int K, int P) {
  cv::Shape s;
  return K > 0 && K % 16 == 0 && cv::pick(M, P, &s) ? 1 : 0;
}

/* rows of the statistics partials coskad_conv1x1_f32 writes for this shape (its block columns) */
int coskad_conv1x1_stat_rows(int M, int K, int P, int batch) {
  cv::Shape s;
  if (!(K % 16 == 0 && cv::pick(M, P, &s))) return 0;
  const int wc = s.wc, mb = 16 * s.rt * s.wm;
  const int groups = (batch + wc - 1) / wc;
  int gx = 512 / (M / mb);
  if (gx < 1) gx = 1;
  return groups < gx ? groups : gx;
}

/* Out[b][m][p] (+)= sum_k A(m,k) In[b][k][p] (+ bias[m]); A(m,k) = A[m sa_m + k sa_k] with sa_k == 1 or sa_m == 1.
 * stats (optional): [coskad_conv1x1_stat_rows()][M][2] doubles receive per-channel sum / sum of squares of Out. */
int coskad_conv1x1_f32(const float* A, long long sa_m, long long sa_k, const float* In, float* Out, const float* bias, double* stats,
                       int M, int K, int P, int batch, int accumulate, hipStream_t stream) {
  if (!A || !In || !Out) return fail(COSKAD_ERR_ARG, "conv1x1: null pointer");
  if (M <= 0 || K <= 0 || P <= 0 || batch <= 0) return fail(COSKAD_ERR_ARG, "conv1x1: M=%d K=%d P=%d batch=%d", M, K, P, batch);
  cv::Shape s;
  if (K % 16 != 0 || !cv::pick(M, P, &s)) return fail(COSKAD_ERR_SHAPE, "conv1x1: unsupported shape M=%d K=%d P=%d", M, K, P);
  if (sa_k != 1 && sa_m != 1) return fail(COSKAD_ERR_ARG, "conv1x1: the weight must be contiguous along m or along k");
  if ((sa_k == 1 && sa_m % 4 != 0) || (sa_m == 1 && sa_k % 4 != 0) || ((size_t)A & 15) || ((size_t)In & 15) || ((size_t)Out & 15))
    return fail(COSKAD_ERR_ARG, "conv1x1: operands must be 16-byte aligned (strides multiples of 4 floats)");
  cv::Args a{A, In, Out, bias, stats, sa_m, sa_k, M, K, P, batch, accumulate};
  const int wc = s.wc, mb = 16 * s.rt * s.wm, ps = 16 * s.nt;
  const int gx = coskad_conv1x1_stat_rows(M, K, P, batch);
  const size_t lds = ((size_t)2 * wc * cv::BK * ps + (size_t)2 * cv::BK * (mb + 16)) * sizeof(float);
  dim3 grid(gx, M / mb);
#define LAUNCH_CV(RT, NT, WM, WC)                                                                               \
  do {                                                                                                          \
    auto k = cv::k_conv1x1<RT, NT, WM, WC>;                                                                     \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, grid, dim3(64 * WM * WC), lds, stream, a);                                            \
  } while (0)
  if (s.nt == 13 && s.wm == 8) LAUNCH_CV(2, 13, 8, 1);
  else if (s.nt == 13 && s.wm == 4) LAUNCH_CV(2, 13, 4, 2);
  else if (s.nt == 13 && s.wm == 2) LAUNCH_CV(2, 13, 2, 4);
  else if (s.nt == 13 && s.wm == 1) LAUNCH_CV(2, 13, 1, 4);
  else if (s.nt == 19 && s.wm == 2) LAUNCH_CV(2, 19, 2, 2);
  else LAUNCH_CV(2, 19, 1, 4);
#undef LAUNCH_CV
  return check_launch("conv1x1");
}

/* BatchNorm statistics from the [rows][C][2] partials of coskad_conv1x1_f32: stat [2C] = (mean, 1 / sqrt(var + eps)), running
 * statistics updated (unbiased variance, momentum) and num_batches_tracked incremented as nn.BatchNorm2d does in training mode */
int coskad_bn2_stats_parts_f32(const double* parts, int rows, float* stat, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float momentum, float eps, double count, int C, hipStream_t stream) {
  if (!parts || !stat || rows <= 0 || C <= 0 || count <= 0) return fail(COSKAD_ERR_ARG, "bn2_stats_parts: bad argument");
  hipLaunchKernelGGL(cv::k_stats_from_parts, dim3(ceil_div(C, 16)), dim3(256), 0, stream, parts, rows, stat, running_mean, running_var,
                     num_batches_tracked, momentum, eps, count, C);
  return check_launch("bn2_stats_parts");
}

/* 1 when coskad_conv1x1_wgrad_f32 takes the shape: P in {204, 300}, M a multiple of 32, K of 64 */
int coskad_conv1x1_wgrad_ok(int M, int K, int P) { return (P == 204 || P == 300) && M > 0 && K > 0 && M % 32 == 0 && K % 64 == 0 ? 1 : 0; }

/* partials [chunks][M][K] of dW[m][k] = sum_b sum_p G[b][m][p] X[b][k][p] (G [batch][M][P], X [batch][K][P] contiguous); chunks =
 * ceil(batch / chunk); add them with coskad_gemm_sum_f32 (fp64, fixed order). */
int coskad_conv1x1_wgrad_f32(const float* G, const float* X, float* partials, int M, int K, int P, int batch, int chunk,
                             hipStream_t stream) {
  if (!G || !X || !partials) return fail(COSKAD_ERR_ARG, "conv1x1_wgrad: null pointer");
  if (batch <= 0 || chunk <= 0) return fail(COSKAD_ERR_ARG, "conv1x1_wgrad: batch=%d chunk=%d", batch, chunk);
  if (!coskad_conv1x1_wgrad_ok(M, K, P)) return fail(COSKAD_ERR_SHAPE, "conv1x1_wgrad: unsupported shape M=%d K=%d P=%d", M, K, P);
  if (((size_t)G & 15) || ((size_t)X & 15)) return fail(COSKAD_ERR_ARG, "conv1x1_wgrad: operands must be 16-byte aligned");
  const int chunks = ceil_div(batch, chunk);
  if (chunks > 65535) return fail(COSKAD_ERR_SHAPE, "conv1x1_wgrad: %d chunks exceed the grid limit", chunks);
  const bool big = M % 128 == 0 && K % 128 == 0;
#define LAUNCH_WG(WMR, WKR, TM, TK, PC)                                                                          \
  do {                                                                                                           \
    auto k = cv::k_conv1x1_wgrad<WMR, WKR, TM, TK, PC>;                                                          \
    const size_t lds = (size_t)2 * (16 * TM * WMR + 16 * TK * WKR) * (PC + 2) * sizeof(float);                   \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(K / (16 * TK * WKR), M / (16 * TM * WMR), chunks), dim3(64 * WMR * WKR), lds, stream, G, X,   \
                       partials, M, K, P, batch, chunk);                                                         \
  } while (0)
  const bool mid = M % 64 == 0;
  if (P == 204) {
    if (big) LAUNCH_WG(4, 2, 2, 4, 68);
    else if (mid) LAUNCH_WG(2, 2, 2, 2, 68);
    else LAUNCH_WG(2, 2, 1, 2, 68);
  } else {
    if (big) LAUNCH_WG(4, 2, 2, 4, 60);
    else if (mid) LAUNCH_WG(2, 2, 2, 2, 60);
    else LAUNCH_WG(2, 2, 1, 2, 60);
  }
#undef LAUNCH_WG
  return check_launch("conv1x1_wgrad");
}

}  // extern "C"
