// A 32 -> 16 ST_GCNN layer in training mode BY COMMUTATION (reference models/graph_layers/stsgcn.py:94-116 forward, 56-80 the mixing,
// autograd of both; BatchNorm2d in training mode).
//
// The mixing acts per channel on (frame, joint), a 1x1 convolution mixes channels at one position: Wt gcn(X) = gcn(Wt X).  With
//     Y = Wt X,  R = Wr X          X = PReLU(U_prev), 32 channels
// formed first, everything behind them sees 16 channels:
//     Zy = gcn(Y)                  (both in ONE kernel: fused_apply_flat.hip, XO form -- K-ring GEMM, then the clip's rows are mixed in the
//                                   flush image; it also leaves the per-channel sums of Zy, Zy^2, R, R^2)
//     U  = a_t Zy + a_r R + shift  (both BatchNorms are PER-CHANNEL affine maps of Zy and R: their batch statistics are row sums, not
//                                   two 32 x 32 Gram products: k_commute_fold, k_commute_combine)
// and the backward is ONE kernel per clip behind two row-sum reductions (k_commute_bsums -> k_commute_bfold):
//     dZy = c1 dU + c2 Zy + c3,  dR = e1 dU + e2 R + e3                   (BatchNorm backward, per channel)
//     dA[t] += Yt_t^T dZy_t      (Yt = temporal mix of Y)                 dT[v] += Y_v^T dYs_v   (dYs = spatial adjoint of dZy)
//     dY = gcn^T(dZy);  dX = [Wt; Wr]^T [dY; dR];  dU_prev = dX PReLU'(U_prev);  d[Wt; Wr] += [dY; dR] X^T;  dslope_prev
// By bytes the backward reads dU, Zy, R, Y (16 rows each) and U_prev (32) once and writes dU_prev -- against the statistics + data +
// dA / dT kernels of the 32-channel form (dU twice, Z, X three times, dZ out and back).  At B = 4096 on the 25-joint layout the layer
// costs 141 us forward (191 as statistics + apply kernels on 32 channels) and 297 us backward (494).  Built for 12 frames x 25 joints,
// the layout whose layers run unfused kernels; at 17 joints the chained kernels of fused_bwd.hip / fused_apply_next_bpc.hip already hold
// the layer on chip and the same rewrite costs the chain more than it saves.
#include "fused_ops.h"

namespace coskad {

// fused_apply_flat.hip: [Y; R] = [Wt; Wr] PReLU(in), Zy = gcn(Y), per-workgroup row sums of Zy, Zy^2, R, R^2
int launch_commute_apply_mix(const float* in, float* out, const float* wt, const float* wr, const float* in_slope, const float* Aw,
                             const float* Tw, float* zy, float* mixpart, int B, int Ci, int Jo, int TV_, hipStream_t st, int* rows_out);

// fwd_moments_bpc.hip: the next layer's statistics pass with this layer's combine formed on the way in
int launch_combine_moments_bpc(const float* Zy, const float* YR, const float* stat, float* U, const float* Aw, const float* Tw,
                               const float* slope, float* partials, int B, int T_, int V_, float* Zout, hipStream_t st, int* rows_out);

namespace cm {

using ff::f32x4;
using ff::Lane;
using ff::mfma;
using ff::prelu;

constexpr int T = 12, C = 16;
constexpr int kStat = 8 * C;         // floats of `stat`: a_t, a_r, shift, mean_z, istd_t, mean_r, istd_r, (unused)
constexpr int kCoef = 6 * C;         // c1 c2 c3 e1 e2 e3
constexpr int kMixCols = 4 * C;      // sum z, sum z^2, sum r, sum r^2
constexpr int kSumCols = 3 * C;      // sum dU, sum dU Zy, sum dU R
constexpr int kWCols = 32 * 32 + 16; // d[Wt; Wr] (32 x 32) + the slope partial (padded to a float4 multiple)
constexpr int kBelowCols = 2 * 32 * 2 + 32;   // [P 32 x 2][Q 32 x 2][sdU 32] of a 2-channel layer below

// sums the P partial rows of `cols` (<= 64) columns in fp64 (fixed order; common.h: 64 columns x 16 row slices, eight loads in flight)
__device__ __forceinline__ void sum_rows_f64(const float* __restrict__ rows, int P, int cols, double* out, double* sh) {
  const int col = threadIdx.x & 63;
  const double t = column_sum_f64<64>(rows, P, (size_t)cols, col, col < cols, sh);
  if (threadIdx.x < 64 && col < cols) out[col] = t;
  __syncthreads();
}

struct FoldArgs {
  const float* gamma_t; const float* beta_t; const float* gamma_r; const float* beta_r;
  const float* bias_t; const float* bias_r;            // conv biases (NULL: none): they only move the running means
  float* rm_t; float* rv_t; float* rm_r; float* rv_r;  // running statistics (NULL: not tracked)
  long long* nbt_t; long long* nbt_r;
  float momentum, eps;
};

// ---- forward 3: the batch statistics of Zy and R -> the two BatchNorms as one affine map per channel ------------------------------
__global__ __launch_bounds__(1024) void k_commute_fold(const float* __restrict__ partials, int P, double N, FoldArgs a,
                                                      float* __restrict__ stat) {
  __shared__ double sh[1024];
  __shared__ double tot[kMixCols];
  sum_rows_f64(partials, P, kMixCols, tot, sh);
  const int c = threadIdx.x;
  if (c < C) {
    const double mz = tot[c] / N, mr = tot[2 * C + c] / N;
    double vz = tot[C + c] / N - mz * mz, vr = tot[3 * C + c] / N - mr * mr;
    vz = vz > 0.0 ? vz : 0.0;
    vr = vr > 0.0 ? vr : 0.0;
    const double it = 1.0 / sqrt(vz + (double)a.eps), ir = 1.0 / sqrt(vr + (double)a.eps);
    const double at = (double)a.gamma_t[c] * it, ar = (double)a.gamma_r[c] * ir;
    stat[c] = (float)at;
    stat[C + c] = (float)ar;
    stat[2 * C + c] = (float)((double)a.beta_t[c] - at * mz + (double)a.beta_r[c] - ar * mr);
    stat[3 * C + c] = (float)mz;
    stat[4 * C + c] = (float)it;
    stat[5 * C + c] = (float)mr;
    stat[6 * C + c] = (float)ir;
    stat[kStat - C + c] = 0.f;
    const double unb = N > 1.0 ? N / (N - 1.0) : 1.0, m = (double)a.momentum;
    if (a.rm_t) {
      a.rm_t[c] = (float)((1.0 - m) * (double)a.rm_t[c] + m * (mz + (a.bias_t ? (double)a.bias_t[c] : 0.0)));
      a.rv_t[c] = (float)((1.0 - m) * (double)a.rv_t[c] + m * vz * unb);
    }
    if (a.rm_r) {
      a.rm_r[c] = (float)((1.0 - m) * (double)a.rm_r[c] + m * (mr + (a.bias_r ? (double)a.bias_r[c] : 0.0)));
      a.rv_r[c] = (float)((1.0 - m) * (double)a.rv_r[c] + m * vr * unb);
    }
  }
  if (threadIdx.x == 0) {
    if (a.nbt_t) a.nbt_t[0] += 1;
    if (a.nbt_r) a.nbt_r[0] += 1;
  }
}

// ---- forward 4: U = a_t Zy + a_r R + shift ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_commute_combine(const float* __restrict__ YR, const float* __restrict__ Zy,
                                                        const float* __restrict__ stat, float* __restrict__ U, int B, int R4) {
  __shared__ float st[3 * C];
  if (threadIdx.x < 3 * C) st[threadIdx.x] = stat[threadIdx.x];
  __syncthreads();
  const int n4 = C * R4;
  for (int clip = blockIdx.x; clip < B; clip += gridDim.x) {
    const float4* z4 = reinterpret_cast<const float4*>(Zy) + (size_t)clip * n4;
    const float4* r4 = reinterpret_cast<const float4*>(YR) + (size_t)clip * 2 * n4 + n4;
    float4* u4 = reinterpret_cast<float4*>(U) + (size_t)clip * n4;
    for (int e = threadIdx.x; e < n4; e += 256) {
      const int row = e / R4;
      const float at = st[row], ar = st[C + row], sh = st[2 * C + row];
      const float4 z = z4[e], r = r4[e];
      u4[e] = float4{fmaf(at, z.x, fmaf(ar, r.x, sh)), fmaf(at, z.y, fmaf(ar, r.y, sh)), fmaf(at, z.z, fmaf(ar, r.z, sh)),
                     fmaf(at, z.w, fmaf(ar, r.w, sh))};
    }
  }
}

// ---- backward 1: per-channel sums of dU, dU Zy, dU R ------------------------------------------------------------------------------
// thread <-> (row = tid / 16, 16 float4 columns apart): a row's sixteen threads read 256 B runs
__global__ __launch_bounds__(256) void k_commute_bsums(const float* __restrict__ YR, const float* __restrict__ Zy,
                                                      const float* __restrict__ dU, float* __restrict__ partials, int B, int R4) {
  const int row = threadIdx.x >> 4, sub = threadIdx.x & 15;
  const int n4 = C * R4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int clip = blockIdx.x; clip < B; clip += gridDim.x) {
    const float4* d4 = reinterpret_cast<const float4*>(dU) + (size_t)clip * n4 + row * R4;
    const float4* z4 = reinterpret_cast<const float4*>(Zy) + (size_t)clip * n4 + row * R4;
    const float4* r4 = reinterpret_cast<const float4*>(YR) + (size_t)clip * 2 * n4 + n4 + row * R4;
    for (int p = sub; p < R4; p += 16) {
      const float4 d = d4[p], z = z4[p], r = r4[p];
      s0 += (d.x + d.y) + (d.z + d.w);
      s1 = fmaf(d.x, z.x, fmaf(d.y, z.y, fmaf(d.z, z.z, fmaf(d.w, z.w, s1))));
      s2 = fmaf(d.x, r.x, fmaf(d.y, r.y, fmaf(d.z, r.z, fmaf(d.w, r.w, s2))));
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if (sub == 0) {
    float* dst = partials + (size_t)blockIdx.x * kSumCols;
    dst[row] = s0; dst[C + row] = s1; dst[2 * C + row] = s2;
  }
}

// ---- backward 2: BatchNorm backward of both branches as per-channel coefficients; dgamma / dbeta ---------------------------------
__global__ __launch_bounds__(1024) void k_commute_bfold(const float* __restrict__ partials, int P, double N,
                                                       const float* __restrict__ stat, float* __restrict__ coef,
                                                       float* __restrict__ dgamma_t, float* __restrict__ dbeta_t,
                                                       float* __restrict__ dgamma_r, float* __restrict__ dbeta_r) {
  __shared__ double sh[1024];
  __shared__ double tot[kSumCols];
  sum_rows_f64(partials, P, kSumCols, tot, sh);
  const int c = threadIdx.x;
  if (c < C) {
    const double sd = tot[c], sdz = tot[C + c], sdr = tot[2 * C + c];
    const double at = stat[c], ar = stat[C + c], mz = stat[3 * C + c], it = stat[4 * C + c], mr = stat[5 * C + c], ir = stat[6 * C + c];
    const double dgt = it * (sdz - mz * sd), dgr = ir * (sdr - mr * sd);
    dgamma_t[c] = (float)dgt; dbeta_t[c] = (float)sd;
    dgamma_r[c] = (float)dgr; dbeta_r[c] = (float)sd;
    coef[c] = (float)at;
    coef[C + c] = (float)(-at * it * dgt / N);
    coef[2 * C + c] = (float)(-at * sd / N + at * it * mz * dgt / N);
    coef[3 * C + c] = (float)ar;
    coef[4 * C + c] = (float)(-ar * ir * dgr / N);
    coef[5 * C + c] = (float)(-ar * sd / N + ar * ir * mr * dgr / N);
  }
}

#ifndef CMB_SKIP   // timing-only builds (wrong results): 2 temporal mix, 4 dA, 8 spatial adjoint, 16 dT, 32 temporal adjoint, 64 dX, 128 d[Wt; Wr], 256 row pass
#define CMB_SKIP 0
#endif
// ---- backward 3: everything per clip ----------------------------------------------------------------------------------------------
// NS: the batch reductions of the (2-channel) layer BELOW -- P = sum dU_prev Z_below^T, Q = sum dU_prev X_below^T, sdU (stage 1 of ITS
// backward: k_first_stats) -- are formed here from the dU_prev rows the row pass has just produced (the backward chain of fused_bwd.hip):
// a re-read of dU_prev and a launch fewer.  below_z / below_x [B, 2, TV] (X_below is the network input: no PReLU), `bpart` [grid][160]
// partial rows [P 32 x 2][Q 32 x 2][sdU 32] as k_bwd_fold reads their sums.
template <int V, bool NS>
__global__ __launch_bounds__(256, 2) void k_commute_bwd(const float* __restrict__ Uprev, const float* __restrict__ YR,
                                                       const float* __restrict__ Zy, const float* __restrict__ dU,
                                                       const float* __restrict__ coef, const float* __restrict__ Wt, const float* __restrict__ Wr,
                                                       const float* __restrict__ Aw, const float* __restrict__ Tw,
                                                       const float* __restrict__ in_slope, float* __restrict__ dIn,
                                                       float* __restrict__ gpart, float* __restrict__ wpart, int B,
                                                       const float* __restrict__ below_z, const float* __restrict__ below_x,
                                                       float* __restrict__ bpart) {
  constexpr int skip = CMB_SKIP;
  constexpr int TV = T * V, LD = TV + 2, R4 = TV / 4;
  static_assert(TV % 4 == 0, "rows are staged as float4");
  constexpr int NT = (TV + 15) / 16, MAXT = (NT + 1) / 2;
  constexpr int H4 = C * R4, HL = (H4 + 255) / 256;      // float4 of a 16-row half, per thread
  constexpr int NTV = (V + 15) / 16, KV = (V + 3) / 4, MAXF = T / 4, MAXJ = (V + 3) / 4;
  constexpr int NM = (TV + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;                  // 32 rows, stride LD: [dZy; dR] -> [dYs; dR] -> [dY; dR] -> dX
  float* win = lds + 32 * LD;        // 16 rows, stride LD (row-per-lane reads: 302 = 14 mod 32 keeps the 16 rows on 16 banks): Y -> Yt -> Y -> X halves
  __shared__ float cbs[kCoef];
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  // lane geometry / thread id behind an optimisation barrier: the address arithmetic of a phase is recomputed there instead of being
  // hoisted out of the clip loop and held (spilled) across every phase
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  auto tid_now = [&]() {
    int t = tid0;
    asm volatile("" : "+v"(t));
    return t;
  };
  Lane L = geo();
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  if (tid0 < kCoef) cbs[tid0] = coef[tid0];
  // dX = [Wt; Wr]^T D: this wave's 16 input channels (ot) x half of the position tiles; A[i = L.j][k = 4 s + L.q] = [Wt; Wr][k][16 ot + i]
  const int ot = wave & 1, t0 = (wave >> 1) * MAXT;
  const int nt = NT - t0 < MAXT ? NT - t0 : MAXT;
  float wa[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) wa[s] = (s < 4 ? Wt : Wr)[(4 * (s & 3) + L.q) * 32 + 16 * ot + L.j];
  f32x4 accA[MAXF][NTV][NTV], accT[MAXJ], accW[4];
#pragma unroll
  for (int a = 0; a < MAXF; ++a)
#pragma unroll
    for (int b = 0; b < NTV; ++b)
#pragma unroll
      for (int c = 0; c < NTV; ++c) accA[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) accT[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) accW[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float da = 0.f;
  f32x4 nsb[NS ? 2 : 1];             // (NS) [dU_prev rows 16 c ..][Z0 Z1 X0 X1]: this wave's share of the k-steps
  float nss[NS ? 2 : 1];
#pragma unroll
  for (int c = 0; c < (NS ? 2 : 1); ++c) { nsb[c] = f32x4{0.f, 0.f, 0.f, 0.f}; nss[c] = 0.f; }
  auto pos_of = [&](int t) {
    const int p = 16 * (t0 + (t < nt ? t : 0)) + L.j;
    return p < TV ? p : TV - 1;
  };
  // a clip's 16-row streams as the threads own them (float4 e = tid + 256 i of the half; beyond it / beyond the batch: zeros)
  auto hload = [&](const float* base, int clip, int rows, int row0, float4 (&dst)[HL]) {
    const int tid = tid_now();
    const float4* g4 = reinterpret_cast<const float4*>(base + ((size_t)(clip < B ? clip : 0) * rows + row0) * TV);
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int e = tid + 256 * i;
      dst[i] = (e < H4 && clip < B) ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto hstore = [&](float* dstimg, const float4 (&src)[HL], bool act) {   // -> 16 rows at stride LD
    const int tid = tid_now();
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int e = tid + 256 * i;
      if (e < H4) {
        const int row = e / R4, col = 4 * (e - row * R4);
        float4 v = src[i];
        if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        *reinterpret_cast<float2*>(dstimg + row * LD + col) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(dstimg + row * LD + col + 2) = float2{v.z, v.w};
      }
    }
  };
  // the next clip's dU, Zy, R, Y rows travel in registers (fetched behind the row pass of the clip before)
  float4 pd[HL], pz[HL], pr[HL], py[HL];
  int clip = blockIdx.x;
  hload(dU, clip, C, 0, pd);
  hload(Zy, clip, C, 0, pz);
  hload(YR, clip, 32, C, pr);
  hload(YR, clip, 32, 0, py);
  for (; clip < B; clip += gridDim.x) {
    // the mixing operands are fetched per clip (L2 / L1 hits) a phase ahead of their use: held for the launch (the compiler hoists
    // loop-invariant loads) they cost 84 registers this kernel does not have -- the pointers go through an optimisation barrier per clip
    const float* Twc = Tw;
    const float* Awc = Aw;
    asm volatile("" : "+s"(Twc), "+s"(Awc));
    //   temporal          Yt[q,v]  = sum_t Y[t,v] T[v][t][q]:       B[k = t][j = q]
    float tf[MAXJ][3];
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
#pragma unroll
      for (int s = 0; s < 3; ++s) tf[k][s] = (v < V && L.j < T) ? Twc[(v * T + 4 * s + L.q) * T + L.j] : 0.f;
    }
    __syncthreads();                                     // the previous clip's readers of the image are done (first clip: cbs is written)
    {
      const int tid = tid_now();
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const int e = tid + 256 * i;
        if (e < H4) {
          const int row = e / R4, col = 4 * (e - row * R4);
          const float4 d = pd[i], z = pz[i], r = pr[i];
          const float c1 = cbs[row], c2 = cbs[C + row], c3 = cbs[2 * C + row];
          const float e1 = cbs[3 * C + row], e2 = cbs[4 * C + row], e3 = cbs[5 * C + row];
          float* qz = img + row * LD + col;
          float* qr = qz + C * LD;
          *reinterpret_cast<float2*>(qz) = float2{fmaf(c1, d.x, fmaf(c2, z.x, c3)), fmaf(c1, d.y, fmaf(c2, z.y, c3))};
          *reinterpret_cast<float2*>(qz + 2) = float2{fmaf(c1, d.z, fmaf(c2, z.z, c3)), fmaf(c1, d.w, fmaf(c2, z.w, c3))};
          *reinterpret_cast<float2*>(qr) = float2{fmaf(e1, d.x, fmaf(e2, r.x, e3)), fmaf(e1, d.y, fmaf(e2, r.y, e3))};
          *reinterpret_cast<float2*>(qr + 2) = float2{fmaf(e1, d.z, fmaf(e2, r.z, e3)), fmaf(e1, d.w, fmaf(e2, r.w, e3))};
        }
      }
      hstore(win, py, false);
    }
    __syncthreads();                                     // the image holds [dZy; dR], the window Y
    L = geo();
    //   spatial adjoint   dYs[t,v] = sum_w dZy[t,w] A[t][v][w]:     B[k = w][j = v]   (this wave's frames; in flight behind the temporal mix)
    float sb[MAXF][NTV][KV];
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int c = 0; c < NTV; ++c)
#pragma unroll
        for (int s = 0; s < KV; ++s)
          sb[tt][c][s] = (16 * c + L.j < V && 4 * s + L.q < V) ? Awc[(t * V + 16 * c + L.j) * V + 4 * s + L.q] : 0.f;
    }
    // ---- Yt = temporal mix of Y in the window: joints v = wave, wave + 4, .. ------------------------------------------------------
    if (!(skip & 2))
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) d = mfma(win[L.j * LD + (4 * s + L.q) * V + v], tf[k][s], d);
        if (L.j < T) {
#pragma unroll
          for (int r = 0; r < 4; ++r) win[(4 * L.q + r) * LD + L.j * V + v] = d[r];
        }
      }
    }
    __syncthreads();                                     // the window holds Yt
    L = geo();
    //   temporal adjoint  dY[t,v]  = sum_q dYs[q,v] T[v][t][q]:     B[k = q][j = t]   (in flight behind dA and the spatial adjoint)
    float tb[MAXJ][3];
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
#pragma unroll
      for (int s = 0; s < 3; ++s) tb[k][s] = (v < V && L.j < T) ? Twc[(v * T + L.j) * T + 4 * s + L.q] : 0.f;
    }
    // ---- this wave's frames: dA[t] += Yt_t^T dZy_t (K = the 16 rows), then the spatial adjoint of dZy in place ----------------------
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
      if (!(skip & 4))
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int row = 4 * s + L.q;
        float a[NTV], b[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          const bool ok = 16 * c + L.j < V;
          a[c] = ok ? win[row * LD + t * V + 16 * c + L.j] : 0.f;
          b[c] = ok ? img[row * LD + t * V + 16 * c + L.j] : 0.f;
        }
#pragma unroll
        for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
          for (int tb2 = 0; tb2 < NTV; ++tb2) accA[tt][ta][tb2] = mfma(a[ta], b[tb2], accA[tt][ta][tb2]);
      }
      if (skip & 8) continue;
      float a[KV];
#pragma unroll
      for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? img[L.j * LD + t * V + 4 * s + L.q] : 0.f;
      f32x4 d[NTV];
#pragma unroll
      for (int c = 0; c < NTV; ++c) {
        d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], sb[tt][c][s], d[c]);
      }
#pragma unroll
      for (int c = 0; c < NTV; ++c)
        if (16 * c + L.j < V) {
#pragma unroll
          for (int r = 0; r < 4; ++r) img[(4 * L.q + r) * LD + t * V + 16 * c + L.j] = d[c][r];
        }
    }
    __syncthreads();                                     // every wave has read Yt; rows 0 .. 15 of the image hold dYs
    hstore(win, py, false);                              // Y again (still in its registers)
    __syncthreads();
    L = geo();
    // the pre-activations of the layer input take off: X halves for d[Wt; Wr], the PReLU mask of the row pass
    float4 u[2][HL];
    hload(Uprev, clip, 32, 0, u[0]);
    hload(Uprev, clip, 32, C, u[1]);
    // ---- this wave's joints: dT[v] += Y_v^T dYs_v, then the temporal adjoint in place ---------------------------------------------
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
        if (!(skip & 16))
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int row = 4 * s + L.q;
          const float a = L.j < T ? win[row * LD + L.j * V + v] : 0.f;
          const float b = L.j < T ? img[row * LD + L.j * V + v] : 0.f;
          accT[k] = mfma(a, b, accT[k]);
        }
        if (skip & 32) continue;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) d = mfma(img[L.j * LD + (4 * s + L.q) * V + v], tb[k][s], d);
        if (L.j < T) {
#pragma unroll
          for (int r = 0; r < 4; ++r) img[(4 * L.q + r) * LD + L.j * V + v] = d[r];
        }
      }
    }
    __syncthreads();                                     // the image holds D = [dY; dR]; the window is free
    L = geo();
    // ---- dX = W4^T D: this wave's tiles stay in registers until D's last readers are done ---------------------------------------
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    hstore(win, u[0], pre);                              // X rows 0 .. 15 -> window behind the products
    if (!(skip & 64))
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) acc[t] = mfma(wa[s], img[(4 * s + L.q) * LD + pos_of(t)], acc[t]);
    }
    // ---- d[Wt; Wr] += D X^T: X = PReLU(U_prev) through the window 16 rows at a time, (row, position) products dealt to the waves ---
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) {
        __syncthreads();                                 // the first half's readers are done
        hstore(win, u[1], pre);
      }
      __syncthreads();
      L = geo();
      const float* pb = win + L.j * LD + 2 * L.q;
      const float* pa = img + L.j * LD + 2 * L.q;
      if (!(skip & 128))
      for (int m = wave; m < NM; m += 4) {
        const bool ok = 8 * m + 2 * L.q < TV;
        float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
        b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          float2 a = *reinterpret_cast<const float2*>(pa + 16 * dh * LD + 8 * m);
          a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f;
          accW[2 * dh + h] = mfma(a.x, b.x, accW[2 * dh + h]);
          accW[2 * dh + h] = mfma(a.y, b.y, accW[2 * dh + h]);
        }
      }
    }
    __syncthreads();                                     // D's last readers are done: dX over it
    L = geo();
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int p = 16 * (t0 + t) + L.j;
      if (t < nt) {
        float* dst = img + (16 * ot + 4 * L.q) * LD + (p < TV ? p : TV);
        dst[0] = acc[t][0]; dst[LD] = acc[t][1]; dst[2 * LD] = acc[t][2]; dst[3 * LD] = acc[t][3];
      }
    }
    // the next clip's rows take off behind the row pass
    hload(dU, clip + gridDim.x, C, 0, pd);
    hload(Zy, clip + gridDim.x, C, 0, pz);
    hload(YR, clip + gridDim.x, 32, C, pr);
    hload(YR, clip + gridDim.x, 32, 0, py);
    __syncthreads();                                     // the image holds dX
    // ---- dU_prev = dX PReLU'(U_prev), slope gradient: row-wise, full lines --------------------------------------------------------
    if (!(skip & 256)) {
      const int tid = tid_now();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4* g4 = reinterpret_cast<float4*>(dIn + (size_t)clip * 32 * TV) + h * H4;
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          const int e = tid + 256 * i;
          if (e < H4) {
            const int row = e / R4, col = 4 * (e - row * R4);
            const float* p = img + (16 * h + row) * LD + col;
            const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(p + 2);
            float g[4] = {g0.x, g0.y, g1.x, g1.y};
            if (pre) {
              const float uu[4] = {u[h][i].x, u[h][i].y, u[h][i].z, u[h][i].w};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (uu[c] < 0.f) da = fmaf(g[c], uu[c], da);
                g[c] = uu[c] > 0.f ? g[c] : a_in * g[c];
              }
            }
            g4[e] = float4{g[0], g[1], g[2], g[3]};
            if constexpr (NS) {                          // the image keeps dU_prev for the reductions below
              float* pw = img + (16 * h + row) * LD + col;
              *reinterpret_cast<float2*>(pw) = float2{g[0], g[1]};
              *reinterpret_cast<float2*>(pw + 2) = float2{g[2], g[3]};
            }
          }
        }
      }
      if constexpr (NS) {
        // ---- the layer below: [P | Q] += dU_prev (image rows) x (Z0 Z1 X0 X1)^T (window rows 0 .. 3), k-steps dealt to the waves ------
        {
          const bool fst = tid < 2 * R4;
          const float4 vz = fst ? reinterpret_cast<const float4*>(below_z + (size_t)clip * 2 * TV)[tid] : float4{0.f, 0.f, 0.f, 0.f};
          const float4 vx = fst ? reinterpret_cast<const float4*>(below_x + (size_t)clip * 2 * TV)[tid] : float4{0.f, 0.f, 0.f, 0.f};
          if (fst) {
            const int frow = tid / R4, fcol = 4 * (tid - frow * R4);
            *reinterpret_cast<float2*>(win + frow * LD + fcol) = float2{vz.x, vz.y};
            *reinterpret_cast<float2*>(win + frow * LD + fcol + 2) = float2{vz.z, vz.w};
            *reinterpret_cast<float2*>(win + (2 + frow) * LD + fcol) = float2{vx.x, vx.y};
            *reinterpret_cast<float2*>(win + (2 + frow) * LD + fcol + 2) = float2{vx.z, vx.w};
          }
        }
        __syncthreads();                                 // the image holds dU_prev, the window the four rows
        L = geo();
        const float* pb = win + (L.j & 3) * LD + 2 * L.q;
        const float* pa = img + L.j * LD + 2 * L.q;
        for (int m = wave; m < NM; m += 4) {
          float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
          const bool aok = 8 * m + 2 * L.q < TV, ok = aok && L.j < 4;
          b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float2 a = *reinterpret_cast<const float2*>(pa + 16 * c * LD + 8 * m);
            a.x = aok ? a.x : 0.f; a.y = aok ? a.y : 0.f;
            nsb[c] = mfma(a.x, b.x, nsb[c]);
            nsb[c] = mfma(a.y, b.y, nsb[c]);
            nss[c] += a.x + a.y;
          }
        }
      }
    }
  }
  // ---- every wave owns its frames of dA and its joints of dT: its part of the workgroup's partial row [dA | dT] --------------------
  L = geo();
  const int tid = tid_now();
  float* dstA = gpart + (size_t)blockIdx.x * (T * V * V + V * T * T);
  float* dstT = dstA + T * V * V;
#pragma unroll
  for (int tt = 0; tt < MAXF; ++tt) {
    const int t = wave + 4 * tt;
#pragma unroll
    for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
      for (int tb2 = 0; tb2 < NTV; ++tb2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = 16 * ta + 4 * L.q + r, w = 16 * tb2 + L.j;
          if (v < V && w < V) dstA[(t * V + v) * V + w] = accA[tt][ta][tb2][r];
        }
  }
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) {
    const int v = wave + 4 * k;
    if (v < V) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t1 = 4 * L.q + r, t2 = L.j;
        if (t1 < T && t2 < T) dstT[(v * T + t1) * T + t2] = accT[k][r];
      }
    }
  }
  // ---- d[Wt; Wr]: the waves add their position shares into one row in LDS one after another (fixed order) --------------------------
  __syncthreads();
  float* row = lds;
  da = wave_sum(da);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = row + (16 * dh + 4 * L.q + r) * 32 + 16 * h + L.j;
            p[0] = (w ? p[0] : 0.f) + accW[2 * dh + h][r];
          }
      if (lane == 0) row[1024] = (w ? row[1024] : 0.f) + da;
      if constexpr (NS) {                                // [o][Z0 Z1 X0 X1] -> [P 32 x 2][Q 32 x 2][sdU 32] behind the weight sums
        float* ex = row + kWCols;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * c + 4 * L.q + r;
            if (L.j < 4) {
              float* p = ex + (L.j < 2 ? o * 2 + L.j : 64 + o * 2 + (L.j - 2));
              p[0] = (w ? p[0] : 0.f) + nsb[c][r];
            }
          }
          const float t = ff::quad_sum(nss[c]);
          if (L.q == 0) {
            float* p = ex + 128 + 16 * c + L.j;
            p[0] = (w ? p[0] : 0.f) + t;
          }
        }
      }
    }
    __syncthreads();
  }
  float* dstW = wpart + (size_t)blockIdx.x * kWCols;
  for (int e = tid; e < 1025; e += 256) dstW[e] = row[e];
  if constexpr (NS) {
    float* dstB = bpart + (size_t)blockIdx.x * kBelowCols;
    for (int e = tid; e < kBelowCols; e += 256) dstB[e] = row[kWCols + e];
  }
}

// dA, dT, d[Wt; Wr] and the producer's slope gradient from the workgroups' partial rows, in ONE launch (fp64, fixed order)
constexpr int kRedCols = 32;
__global__ __launch_bounds__(1024) void k_commute_reduce(const float* __restrict__ gpart, const float* __restrict__ wpart, int P, int nA,
                                                        int nT, float* __restrict__ dA, float* __restrict__ dT,
                                                        float* __restrict__ dWt, float* __restrict__ dWr, float* __restrict__ dslope,
                                                        const float* __restrict__ bpart, double* __restrict__ bout) {
  __shared__ double sh[1024];
  const int E = nA + nT, nblk = (E + kRedCols - 1) / kRedCols;
  constexpr int nwb = (1025 + kRedCols - 1) / kRedCols;
  if ((int)blockIdx.x >= nblk + nwb) {                   // the layer below's chain rows -> fp64 sums (k_bwd_fold reads them)
    const int e = ((int)blockIdx.x - nblk - nwb) * kRedCols + (threadIdx.x % kRedCols);
    const bool ok = e < kBelowCols;
    const double t = column_sum_f64<kRedCols>(bpart, P, (size_t)kBelowCols, e, ok, sh);
    if ((int)threadIdx.x < kRedCols && ok) bout[e] = t;
    return;
  }
  if ((int)blockIdx.x < nblk) {
    const int e = blockIdx.x * kRedCols + (threadIdx.x % kRedCols);
    const double t = column_sum_f64<kRedCols>(gpart, P, (size_t)E, e, e < E, sh);
    if ((int)threadIdx.x < kRedCols && e < E) (e < nA ? dA + e : dT + (e - nA))[0] = (float)t;
    return;
  }
  const int e = ((int)blockIdx.x - nblk) * kRedCols + (threadIdx.x % kRedCols);
  const bool ok = e < 1025;
  const double t = column_sum_f64<kRedCols>(wpart, P, (size_t)kWCols, e, ok, sh);
  if ((int)threadIdx.x < kRedCols && ok) {
    if (e < 512) dWt[e] = (float)t;
    else if (e < 1024) dWr[e - 512] = (float)t;
    else if (dslope) dslope[0] = (float)t;
  }
}

template <int V>
struct Geo {
  static constexpr int TV = T * V;
  static constexpr size_t bwd_lds = (size_t)48 * (TV + 2) * sizeof(float);
};

inline int mix_rows(int B) { return B < 512 ? B : 512; }
inline int sum_rows(int B) { return B < 1024 ? B : 1024; }
inline int bwd_rows(int B) { return B < 512 ? B : 512; }

}  // namespace cm

extern "C" {

/* 1: coskad_commute_fwd_f32 / _bwd_f32 take this layer (n_frames x n_joints, C_in -> C_out) */
int coskad_commute_ok(int T_, int V_, int Ci, int Co) { return T_ == 12 && (V_ == 25 || V_ == 17) && Ci == 32 && Co == 16; }

/* floats of the scratch both entries need (partial rows of every kernel of the call) */
size_t coskad_commute_ws_floats(int B, int T_, int V_) {
  if (B <= 0 || !coskad_commute_ok(T_, V_, 32, 16)) return 0;
  const size_t g = (size_t)T_ * V_ * V_ + (size_t)V_ * T_ * T_;
  return (size_t)cm::bwd_rows(B) * (g + cm::kWCols) + (size_t)cm::sum_rows(B) * cm::kSumCols + (size_t)cm::mix_rows(B) * cm::kMixCols +
         cm::kCoef + 64;
}

/* Forward of a (32 -> 16) layer in training mode.  u_prev [B, 32, T, V] (in_slope NULL: already activated), wt / wr [16, 32] the two
 * convolutions' weights; YR [B, 32, TV] = [Wt X; Wr X], Zy [B, 16, TV] = gcn(Wt X), U [B, 16, TV] (pre-activation output) and stat [128]
 * are written and kept for the backward.  Running statistics are updated as torch.nn.BatchNorm2d does (NULL: not tracked); momentum
 * must be a number (cumulative averaging is not built here).
 * A_next != NULL: the NEXT layer's (16 input channels) statistics pass rides on the combine -- Z_next [B, 16, TV] = gcn_next(PReLU(U))
 * with slope_out = this layer's PReLU weight, partials_next: *rows_next (<= 768) rows of 2 (16^2 + 16) floats for
 * coskad_layer_train_fold_f32. */
int coskad_commute_fwd_f32(const float* u_prev, const float* in_slope, const float* wt, const float* wr, const float* A, const float* Tm,
                           const float* gamma_t, const float* beta_t, const float* gamma_r, const float* beta_r, const float* bias_t,
                           const float* bias_r, float* rm_t, float* rv_t, float* rm_r, float* rv_r, long long* nbt_t, long long* nbt_r,
                           float momentum, float eps, float* YR, float* Zy, float* U, float* stat, float* ws, size_t ws_floats,
                           const float* A_next, const float* T_next, const float* slope_out, float* Z_next, float* partials_next,
                           int* rows_next, int B, int T_, int V_, hipStream_t stream) {
  if (A_next && (!T_next || !slope_out || !Z_next || !partials_next || !rows_next))
    return fail(COSKAD_ERR_ARG, "commute_fwd: the next layer's statistics need T_next, slope_out, Z_next, partials_next, rows_next");
  if (!u_prev || !wt || !wr || !A || !Tm || !gamma_t || !beta_t || !gamma_r || !beta_r || !YR || !Zy || !U || !stat || !ws)
    return fail(COSKAD_ERR_ARG, "commute_fwd: null pointer");
  if (B <= 0 || !coskad_commute_ok(T_, V_, 32, 16)) return fail(COSKAD_ERR_SHAPE, "commute_fwd: built for 12 x 17 / 25, 32 -> 16");
  if (ws_floats < coskad_commute_ws_floats(B, T_, V_)) return fail(COSKAD_ERR_WORKSPACE, "commute_fwd: scratch too small");
  const int TV = T_ * V_;
  int rows = 0;
  int rc = launch_commute_apply_mix(u_prev, YR, wt, wr, in_slope, A, Tm, Zy, ws, B, 32, 32, TV, stream, &rows);
  if (rc) return rc;
  cm::FoldArgs fa{gamma_t, beta_t, gamma_r, beta_r, bias_t, bias_r, rm_t, rv_t, rm_r, rv_r, nbt_t, nbt_r, momentum, eps};
  hipLaunchKernelGGL(cm::k_commute_fold, dim3(1), dim3(1024), 0, stream, ws, rows, (double)B * TV, fa, stat);
  if ((rc = check_launch("commute_fold"))) return rc;
  if (A_next) return launch_combine_moments_bpc(Zy, YR, stat, U, A_next, T_next, slope_out, partials_next, B, T_, V_, Z_next, stream, rows_next);
  hipLaunchKernelGGL(cm::k_commute_combine, dim3(B < 2048 ? B : 2048), dim3(256), 0, stream, YR, Zy, stat, U, B, TV / 4);
  return check_launch("commute_combine");
}

/* Backward: dU [B, 16, TV] -> d_in [B, 32, TV] (gradient of u_prev, PReLU mask applied), dA [T, V, V], dT [V, T, T], dWt / dWr [16, 32],
 * dgamma / dbeta [16] of both BatchNorms, dslope [1] (NULL with in_slope NULL): all OVERWRITTEN.
 * below_stats != NULL: the layer below has 2 input channels (the first layer; below_x [B, 2, TV] its input, below_z its stored Z) and its
 * batch reductions are formed here: below_stats = a chain buffer of coskad_commute_below_floats(B) floats (8-byte aligned) that
 * coskad_layer_bwd_chain_f32 takes as `stats_in` with stats_in_rows = coskad_commute_below_rows(B). */
int coskad_commute_below_rows(int B) { return cm::bwd_rows(B); }
size_t coskad_commute_below_floats(int B) { return ((size_t)cm::bwd_rows(B) * cm::kBelowCols + 1) / 2 * 2 + 2 * (size_t)cm::kBelowCols; }
int coskad_commute_bwd_f32(const float* u_prev, const float* in_slope, const float* wt, const float* wr, const float* A, const float* Tm, const float* YR,
                           const float* Zy, const float* stat, const float* dU, float* d_in, float* dA, float* dT, float* dWt, float* dWr,
                           float* dgamma_t, float* dbeta_t, float* dgamma_r, float* dbeta_r, float* dslope, float* ws, size_t ws_floats,
                           const float* below_x, const float* below_z, float* below_stats, size_t below_stats_floats, int B, int T_,
                           int V_, hipStream_t stream) {
  if (below_stats && (!below_x || !below_z || ((size_t)below_stats & 7) || below_stats_floats < coskad_commute_below_floats(B)))
    return fail(COSKAD_ERR_ARG, "commute_bwd: below_stats needs below_x, below_z, 8-byte alignment and coskad_commute_below_floats(B) floats");
  if (!u_prev || !wt || !wr || !A || !Tm || !YR || !Zy || !stat || !dU || !d_in || !dA || !dT || !dWt || !dWr || !dgamma_t || !dbeta_t ||
      !dgamma_r || !dbeta_r || !ws)
    return fail(COSKAD_ERR_ARG, "commute_bwd: null pointer");
  if (B <= 0 || !coskad_commute_ok(T_, V_, 32, 16)) return fail(COSKAD_ERR_SHAPE, "commute_bwd: built for 12 x 17 / 25, 32 -> 16");
  if (ws_floats < coskad_commute_ws_floats(B, T_, V_)) return fail(COSKAD_ERR_WORKSPACE, "commute_bwd: scratch too small");
  if ((in_slope != nullptr) != (dslope != nullptr)) return fail(COSKAD_ERR_ARG, "commute_bwd: dslope goes with in_slope");
  const int V = V_;
  const int TV = 12 * V, nA = 12 * V * V, nT = V * 12 * 12;
  const int prow = cm::bwd_rows(B), srow = cm::sum_rows(B);
  float* gpart = ws;
  float* wpart = gpart + (size_t)prow * (nA + nT);
  float* spart = wpart + (size_t)prow * cm::kWCols;
  float* coef = spart + (size_t)srow * cm::kSumCols + (size_t)cm::mix_rows(B) * cm::kMixCols;
  int rc;
  hipLaunchKernelGGL(cm::k_commute_bsums, dim3(srow), dim3(256), 0, stream, YR, Zy, dU, spart, B, TV / 4);
  if ((rc = check_launch("commute_bsums"))) return rc;
  hipLaunchKernelGGL(cm::k_commute_bfold, dim3(1), dim3(1024), 0, stream, spart, srow, (double)B * TV, stat, coef, dgamma_t, dbeta_t,
                     dgamma_r, dbeta_r);
  if ((rc = check_launch("commute_bfold"))) return rc;
#define LAUNCH_CB(V__)                                                                                                          \
  do {                                                                                                                          \
    auto k = below_stats ? cm::k_commute_bwd<V__, true> : cm::k_commute_bwd<V__, false>;                                        \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cm::Geo<V__>::bwd_lds);          \
    hipLaunchKernelGGL(k, dim3(prow), dim3(256), cm::Geo<V__>::bwd_lds, stream, u_prev, YR, Zy, dU, coef, wt, wr, A, Tm, in_slope, d_in, \
                       gpart, wpart, B, below_z, below_x, below_stats);                                                         \
  } while (0)
  if (V == 17) LAUNCH_CB(17); else LAUNCH_CB(25);
#undef LAUNCH_CB
  if ((rc = check_launch("commute_bwd"))) return rc;
  const int nblk = ceil_div(nA + nT, cm::kRedCols) + ceil_div(1025, cm::kRedCols) + (below_stats ? ceil_div(cm::kBelowCols, cm::kRedCols) : 0);
  double* bout = below_stats ? reinterpret_cast<double*>(below_stats + ((size_t)prow * cm::kBelowCols + 1) / 2 * 2) : nullptr;
  hipLaunchKernelGGL(cm::k_commute_reduce, dim3(nblk), dim3(1024), 0, stream, gpart, wpart, prow, nA, nT, dA, dT, dWt, dWr, dslope,
                     below_stats, bout);
  return check_launch("commute_reduce");
}

}  // extern "C"
}  // namespace coskad
