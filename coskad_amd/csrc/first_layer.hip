// The backward of a layer with a handful of input channels (the first layer of every stack: 2 or 3 pose channels;
// autograd of models/graph_layers/stsgcn.py:94-116 without a dIn) on the stored-Z path.
//
// With C_in <= 4 the matrix-core kernels pad the channel axis to 16 and spend their time in staging, tables and
// barriers (round-2 profile, B = 4096: statistics 41 us + data 37 us + dA/dT 16 us for a layer whose only real traffic is
// TWO reads of dU, 2 x 107 MB = 2 x 17 us at the HBM rate).  Here the channel products are plain FMAs on
// full-line loads:
//
//   k_first_stats : P[o][c] = sum dU[o] . Z[c],  Q[o][c] = sum dU[o] . X[c],  s[o] = sum dU[o]      (stage 1)
//                   one clip per block round, the block's four waves split the dU rows, a lane owns four positions of every row;
//                   per-lane accumulators, summed over the lanes once at the very end; rows [P][Q][s] as k_bwd_fold reads them
//   k_first_bwd   : dZ = Bt.dU + Kt.Z + kt (C_in channels), Y = temporal mix of X, dY = spatial adjoint of dZ,
//                   dA[t][v][w] += sum_c Y[c][t,v] dZ[c][t,w],  dT[v][t][q] += sum_c X[c][t,v] dY[c][q,v]   (stages 3 + 4)
//                   the four waves split the dU rows of the clip, partial dZ meets in LDS; every thread keeps its share of
//                   the dA / dT sums in registers for the whole launch; rows [dA][dT] as k_reduce_gcn reads them
#include "mfma_ops.h"

namespace coskad {
namespace fl {

__device__ __forceinline__ float act(float x, bool pre, float a) { return (pre && x < 0.f) ? a * x : x; }
__device__ __forceinline__ float4 act4(float4 v, bool pre, float a) {
  return float4{act(v.x, pre, a), act(v.y, pre, a), act(v.z, pre, a), act(v.w, pre, a)};
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// NW waves per block, RW = rows of dU per wave (C_out <= NW RW).  Measured at B = 4096, 2 -> 32 channels: 32 us as is (4 waves
// per SIMD); prefetching the next clip's rows at half the occupancy 39 us, 8 waves per SIMD (64 registers, spills) 49 us
constexpr int NW = 4;
template <int CI, int RW>
__global__ __launch_bounds__(64 * NW, 4) void k_first_stats(const float* __restrict__ in, const float* __restrict__ Zg,
                                                           const float* __restrict__ dU, const float* __restrict__ in_slope,
                                                           float* __restrict__ partials, int B, int Co, int TVr, int need_q) {
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int n4 = TVr >> 2;                      // float4 per row
  float pP[RW][CI], pQ[RW][CI], pS[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    pS[k] = 0.f;
#pragma unroll
    for (int c = 0; c < CI; ++c) { pP[k][c] = 0.f; pQ[k][c] = 0.f; }
  }
  for (int clip = blockIdx.x; clip < B; clip += gridDim.x) {
    const float* gz = Zg + (size_t)clip * CI * TVr;
    const float* gx = in + (size_t)clip * CI * TVr;
    const float* gd = dU + ((size_t)clip * Co + (size_t)wave * RW) * TVr;
    for (int l = lane; l < n4; l += 64) {       // (one trip for T V <= 256)
      float4 z[CI], x[CI];
#pragma unroll
      for (int c = 0; c < CI; ++c) {
        z[c] = *reinterpret_cast<const float4*>(gz + c * TVr + 4 * l);
        x[c] = need_q ? act4(*reinterpret_cast<const float4*>(gx + c * TVr + 4 * l), pre, a_in) : float4{0.f, 0.f, 0.f, 0.f};
      }
      float4 d[RW];
#pragma unroll
      for (int k = 0; k < RW; ++k)
        d[k] = (wave * RW + k < Co) ? *reinterpret_cast<const float4*>(gd + (size_t)k * TVr + 4 * l) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        pS[k] += (d[k].x + d[k].y) + (d[k].z + d[k].w);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          pP[k][c] += dot4(d[k], z[c]);
          pQ[k][c] += dot4(d[k], x[c]);
        }
      }
    }
  }
  // lanes -> one value per (row, channel); every wave owns its own rows of the block's partial row
  const int Ci = CI;
  float* dst = partials + (size_t)blockIdx.x * (2 * Co * Ci + Co);
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int o = wave * RW + k;
    const float s = wave_sum(pS[k]);
    if (lane == 0 && o < Co) dst[2 * Co * Ci + o] = s;
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const float p = wave_sum(pP[k][c]), q = wave_sum(pQ[k][c]);
      if (lane == 0 && o < Co) {
        dst[o * Ci + c] = p;
        dst[Co * Ci + o * Ci + c] = q;
      }
    }
  }
}

// NWB waves per block.  Measured on the whole train step (B = 4096, 17 joints; tools/ab_engine_flag.py on one box): 4 waves x 1024
// blocks 1.618 ms, 8 x 512 1.601, 8 x 384 1.606, 16 x 256 1.614: half the partial rows (24 -> 12 MB through k_reduce_gcn), half the
// dA / dT outputs per thread.  The 25-joint layout (44 KB of mixing tables: two blocks per CU either way) gained 176 -> 90 us
template <int T, int V>
constexpr int first_bwd_waves() { return 8; }
template <int T, int V, int CI, int RW>
__global__ __launch_bounds__((64 * first_bwd_waves<T, V>()), 4) void k_first_bwd(const float* __restrict__ in, const float* __restrict__ Zg,
                                                      const float* __restrict__ dU, const float* __restrict__ Aw,
                                                      const float* __restrict__ Tw, const float* __restrict__ coef,
                                                      const float* __restrict__ in_slope, float* __restrict__ partials, int B,
                                                      int Co) {
  constexpr int TV = T * V, NA = T * V * V, NT = V * T * T, N4 = TV / 4;
  constexpr int CiP = 16;                       // round_up(C_in, 16): row stride of the coefficient block
  constexpr int NW = first_bwd_waves<T, V>();   // (shadows the 4 of k_first_stats)
  static_assert(TV % 4 == 0 && CI <= 4, "few-channel layer: T V a multiple of 4");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* AwL = lds;                             // [T][V][V]
  float* TwL = AwL + NA;                        // [V][T][T]
  float* Xl = TwL + NT;                         // [CI][TV]
  float* Yl = Xl + CI * TV;
  float* dZl = Yl + CI * TV;
  float* dYl = dZl + CI * TV;
  float* part = dYl + CI * TV;                  // [NW waves][CI][TV]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  constexpr int NTH = 64 * NW;
  for (int e = tid; e < NA; e += NTH) AwL[e] = Aw[e];
  for (int e = tid; e < NT; e += NTH) {         // [t][q][v]: consecutive lanes (joints) read consecutive words
    const int v = e / (T * T), tq = e - v * T * T;
    TwL[tq * V + v] = Tw[e];
  }
  const float* Bt = coef;                       // rows o < Co: Bt[o][c];  rows Co + c2: Kt[c2][c];  then kt[c]
  const float* Kt = coef + (size_t)Co * CiP;
  const float* kt = coef + (size_t)(Co + CI) * CiP;
  // this thread's share of the sums: dA outputs e = tid + NTH k, dT outputs likewise
  constexpr int KA = (NA + NTH - 1) / NTH, KT = (NT + NTH - 1) / NTH;
  float accA[KA], accT[KT];
  int ya[KA], za[KA], xt[KT], yt[KT];            // LDS offsets of the two factors (channel 0)
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    const int e = tid + NTH * k, ec = e < NA ? e : 0;
    const int t = ec / (V * V), v = (ec / V) % V, w = ec % V;
    ya[k] = t * V + v;
    za[k] = t * V + w;
    accA[k] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const int e = tid + NTH * k, ec = e < NT ? e : 0;
    const int v = ec / (T * T), t = (ec / T) % T, q = ec % T;
    xt[k] = t * V + v;
    yt[k] = q * V + v;
    accT[k] = 0.f;
  }

  for (int clip = blockIdx.x; clip < B; clip += gridDim.x) {
    // ---- partial dZ over this wave's rows of dU (a lane owns four positions) ----------------------------------------------
    const float* gd = dU + ((size_t)clip * Co + (size_t)wave * RW) * TV;
    for (int l = lane; l < N4; l += 64) {
      float4 dz[CI];
#pragma unroll
      for (int c = 0; c < CI; ++c) dz[c] = float4{0.f, 0.f, 0.f, 0.f};
      float4 d[RW];
#pragma unroll
      for (int k = 0; k < RW; ++k)
        d[k] = (wave * RW + k < Co) ? *reinterpret_cast<const float4*>(gd + (size_t)k * TV + 4 * l) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        const int o = wave * RW + k < Co ? wave * RW + k : 0;
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          const float b = Bt[o * CiP + c];      // wave-uniform
          dz[c].x = fmaf(b, d[k].x, dz[c].x); dz[c].y = fmaf(b, d[k].y, dz[c].y);
          dz[c].z = fmaf(b, d[k].z, dz[c].z); dz[c].w = fmaf(b, d[k].w, dz[c].w);
        }
      }
#pragma unroll
      for (int c = 0; c < CI; ++c) *reinterpret_cast<float4*>(part + (wave * CI + c) * TV + 4 * l) = dz[c];
    }
    __syncthreads();                            // (also: the previous clip's dA / dT reads of Xl .. dYl are done)
    // ---- dZ = kt + Kt.Z + the four partials;  X ------------------------------------------------------------------------------
    for (int e = tid; e < CI * TV; e += NTH) {
      const int c = e / TV, p = e - c * TV;
      float v = kt[c];
#pragma unroll
      for (int c2 = 0; c2 < CI; ++c2) v = fmaf(Kt[c2 * CiP + c], Zg[((size_t)clip * CI + c2) * TV + p], v);
#pragma unroll
      for (int w = 0; w < NW; ++w) v += part[(w * CI + c) * TV + p];
      dZl[e] = v;
      Xl[e] = act(in[(size_t)clip * CI * TV + e], pre, a_in);
    }
    __syncthreads();
    // ---- Y[c][q,v] = sum_t X[c][t,v] T[v][t][q];  dY[c][t,v] = sum_w A[t][v][w] dZ[c][t,w] ----------------------------------------
    for (int e = tid; e < CI * TV; e += NTH) {
      const int c = e / TV, p = e - c * TV, q = p / V, v = p - q * V;
      float y = 0.f, dy = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) y = fmaf(Xl[c * TV + t * V + v], TwL[(t * T + q) * V + v], y);
#pragma unroll
      for (int w = 0; w < V; ++w) dy = fmaf(AwL[q * V * V + v * V + w], dZl[c * TV + q * V + w], dy);
      Yl[e] = y;
      dYl[e] = dy;
    }
    __syncthreads();
    // ---- dA, dT ---------------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < KA; ++k) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CI; ++c) s = fmaf(Yl[c * TV + ya[k]], dZl[c * TV + za[k]], s);
      accA[k] += s;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CI; ++c) s = fmaf(Xl[c * TV + xt[k]], dYl[c * TV + yt[k]], s);
      accT[k] += s;
    }
  }
  float* dst = partials + (size_t)blockIdx.x * (NA + NT);
#pragma unroll
  for (int k = 0; k < KA; ++k)
    if (tid + NTH * k < NA) dst[tid + NTH * k] = accA[k];
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (tid + NTH * k < NT) dst[NA + tid + NTH * k] = accT[k];
}


// ---- forward (training) of the few-channel layer ------------------------------------------------------------------------------------
// k_first_moments: per clip X -> sum x x^T, sum x;  Z = gcn(X) (stsgcn.py:154-155) -> stored;  sum z z^T, sum z.  One clip per
//   wave round, a lane owns positions lane + 64 k; the mixing is 12 + 17 FMAs per element on LDS operands.  Partial row per
//   block: [MX C^2][sumX C][MZ C^2][sumZ C] (what k_reduce_partials / k_train_fold consume).
constexpr int NWM = 8;   // waves per block of k_first_moments: 512 blocks x 8 waves take B = 4096 in ONE round (4 x 768: two, the second a third full)
template <int T, int V, int CI>
__global__ __launch_bounds__(64 * NWM, 4) void k_first_moments(const float* __restrict__ in, const float* __restrict__ Aw,
                                                          const float* __restrict__ Tw, const float* __restrict__ in_slope,
                                                          float* __restrict__ partials, int B, float* __restrict__ Zout) {
  constexpr int TV = T * V, NA = T * V * V, NT = V * T * T, KP = (TV + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* AwL = lds;
  float* TwL = AwL + NA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  float* Xl = TwL + NT + wave * 2 * CI * TV;     // this wave's X and Y
  float* Yl = Xl + CI * TV;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  for (int e = tid; e < NA; e += 64 * NWM) AwL[e] = Aw[e];
  for (int e = tid; e < NT; e += 64 * NWM) {    // [t][q][v]: consecutive lanes (joints) read consecutive words
    const int v = e / (T * T), tq = e - v * T * T;
    TwL[tq * V + v] = Tw[e];
  }
  float mx[CI][CI], mz[CI][CI], sx[CI], sz[CI];
#pragma unroll
  for (int a = 0; a < CI; ++a) {
    sx[a] = 0.f; sz[a] = 0.f;
#pragma unroll
    for (int b = 0; b < CI; ++b) { mx[a][b] = 0.f; mz[a][b] = 0.f; }
  }
  __syncthreads();
  const int rounds = (B + gridDim.x * NWM - 1) / (gridDim.x * NWM);
  for (int r = 0; r < rounds; ++r) {              // every wave runs every round (the block barriers below are uniform)
    const int clip = (r * gridDim.x + blockIdx.x) * NWM + wave;
    const bool live = clip < B;
    const float* gx = in + (size_t)(live ? clip : 0) * CI * TV;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + 64 * k;
      if (p < TV) {
        float x[CI];
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          x[c] = live ? act(gx[c * TV + p], pre, a_in) : 0.f;
          Xl[c * TV + p] = x[c];
          sx[c] += x[c];
        }
#pragma unroll
        for (int a = 0; a < CI; ++a)
#pragma unroll
          for (int b = 0; b < CI; ++b) mx[a][b] = fmaf(x[a], x[b], mx[a][b]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + 64 * k;
      if (p < TV) {
        const int q = p / V, v = p - q * V;
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          float y = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) y = fmaf(Xl[c * TV + t * V + v], TwL[(t * T + q) * V + v], y);
          Yl[c * TV + p] = y;
        }
      }
    }
    __syncthreads();
    float* gz = Zout + (size_t)(live ? clip : 0) * CI * TV;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + 64 * k;
      if (p < TV) {
        const int t = p / V, w = p - t * V;
        float z[CI];
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          float acc = 0.f;
#pragma unroll
          for (int v = 0; v < V; ++v) acc = fmaf(Yl[c * TV + t * V + v], AwL[t * V * V + v * V + w], acc);
          z[c] = acc;
          if (live) gz[c * TV + p] = acc;
          sz[c] += acc;
        }
#pragma unroll
        for (int a = 0; a < CI; ++a)
#pragma unroll
          for (int b = 0; b < CI; ++b) mz[a][b] = fmaf(z[a], z[b], mz[a][b]);
      }
    }
    __syncthreads();                              // Xl / Yl are rewritten next round
  }
  // lanes, then waves (fixed order) -> the block's partial row
  constexpr int E = 2 * (CI * CI + CI);
  float* row = TwL + NT;                          // E floats per wave, over the (now idle) X images
  __syncthreads();
#pragma unroll
  for (int a = 0; a < CI; ++a) {
    const float s0 = wave_sum(sx[a]), s1 = wave_sum(sz[a]);
    if (lane == 0) { row[wave * E + CI * CI + a] = s0; row[wave * E + 2 * CI * CI + CI + a] = s1; }
#pragma unroll
    for (int b = 0; b < CI; ++b) {
      const float m0 = wave_sum(mx[a][b]), m1 = wave_sum(mz[a][b]);
      if (lane == 0) { row[wave * E + a * CI + b] = m0; row[wave * E + CI * CI + CI + a * CI + b] = m1; }
    }
  }
  __syncthreads();
  if (tid < E) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWM; ++w) t += row[w * E + tid];   // fixed order
    partials[(size_t)blockIdx.x * E + tid] = t;
  }
}

// k_first_apply: U[o][p] = (sum_c Wz[c][o] Z[c][p] + Wx[c][o] X[c][p]) + b[o]  (wfold rows: Z channels, then X channels; the
//   bias joins LAST: a folded BatchNorm bias can dwarf the result, and added first it costs a per-channel offset of an ulp of
//   the BIAS -- tools/dbg_sens.py shows what a 1e-6 per-channel offset does to the golden model's gradients).  One clip per
//   block round; the 2 C_in rows sit in LDS, every thread writes full float4 lines of the output.
template <int CI>
__global__ __launch_bounds__(256, 4) void k_first_apply(const float* __restrict__ Zg, const float* __restrict__ in,
                                                        float* __restrict__ out, const float* __restrict__ wfold,
                                                        const float* __restrict__ bias, const float* __restrict__ in_slope, int B,
                                                        int Co, int CoP, int TVr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [2 CI][TVr]
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int n4 = TVr >> 2;
  for (int clip = blockIdx.x; clip < B; clip += gridDim.x) {
    __syncthreads();
    for (int e = threadIdx.x; e < CI * n4; e += 256) {
      reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(Zg + (size_t)clip * CI * TVr)[e];
      reinterpret_cast<float4*>(lds)[CI * n4 + e] = act4(reinterpret_cast<const float4*>(in + (size_t)clip * CI * TVr)[e], pre, a_in);
    }
    __syncthreads();
    float4* go = reinterpret_cast<float4*>(out + (size_t)clip * Co * TVr);
    for (int e = threadIdx.x; e < Co * n4; e += 256) {
      const int o = e / n4, l = e - o * n4;
      float4 u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 2 * CI; ++c) {
        const float w = wfold[c * CoP + o];
        const float4 v = reinterpret_cast<const float4*>(lds)[c * n4 + l];
        u.x = fmaf(w, v.x, u.x); u.y = fmaf(w, v.y, u.y); u.z = fmaf(w, v.z, u.z); u.w = fmaf(w, v.w, u.w);
      }
      const float b = bias[o];
      go[e] = float4{u.x + b, u.y + b, u.z + b, u.w + b};
    }
  }
}

}  // namespace fl

bool first_layer_ok(int T_, int V_, int Ci, int Co) { return Ci <= 4 && Co <= 64 && (T_ * V_) % 4 == 0; }

// stage 1 for a few-channel layer; partial rows written: *rows_out (each 2 Co Ci + Co floats)
int launch_first_stats(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B, int Ci,
                       int Co, int TVr, int need_q, int max_rows, hipStream_t st, int* rows_out) {
  int grid = B < 1024 ? B : 1024;
  if (grid > max_rows) grid = max_rows;
  *rows_out = grid;
  const int rw = ceil_div(Co, fl::NW);
#define LAUNCH_FS(CI, RW) \
  hipLaunchKernelGGL((fl::k_first_stats<CI, RW>), dim3(grid), dim3(64 * fl::NW), 0, st, in, Zg, dU, in_slope, partials, B, Co, TVr, need_q)
#define LAUNCH_FS_C(CI)                         \
  do {                                          \
    if (rw <= 4) LAUNCH_FS(CI, 4);              \
    else if (rw <= 8) LAUNCH_FS(CI, 8);         \
    else LAUNCH_FS(CI, 16);                     \
  } while (0)
  {
    ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
    if (Ci == 1) LAUNCH_FS_C(1);
    else if (Ci == 2) LAUNCH_FS_C(2);
    else if (Ci == 3) LAUNCH_FS_C(3);
    else LAUNCH_FS_C(4);
  }
#undef LAUNCH_FS_C
#undef LAUNCH_FS
  return check_launch("first_stats");
}

// stages 3 + 4 for a few-channel layer without dIn; partial rows written: *rows_out (each T V V + V T T floats)
template <int T, int V>
static int launch_first_bwd_tv(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw, const float* coef,
                               const float* in_slope, float* partials, int B, int Ci, int Co, int max_rows, hipStream_t st,
                               int* rows_out) {
  constexpr int NWB = fl::first_bwd_waves<T, V>();
  int grid = B < 512 ? B : 512;                 // two 8-wave blocks per CU, one round (sweep below)
  if (grid > max_rows) grid = max_rows;
  *rows_out = grid;
  const int rw = ceil_div(Co, NWB);
  const size_t lds = ((size_t)T * V * V + (size_t)V * T * T + (4 + NWB) * (size_t)Ci * T * V) * sizeof(float);
#define LAUNCH_FB1(CI, RW)                                                                                              \
  do {                                                                                                                  \
    auto k = fl::k_first_bwd<T, V, CI, RW>;                                                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NWB), lds, st, in, Zg, dU, Aw, Tw, coef, in_slope, partials, B, Co);             \
  } while (0)
#define LAUNCH_FB1_C(CI)                        \
  do {                                          \
    if (rw <= 4) LAUNCH_FB1(CI, 4);             \
    else if (rw <= 8) LAUNCH_FB1(CI, 8);        \
    else LAUNCH_FB1(CI, 16);                    \
  } while (0)
  {
    ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
    if (Ci == 1) LAUNCH_FB1_C(1);
    else if (Ci == 2) LAUNCH_FB1_C(2);
    else if (Ci == 3) LAUNCH_FB1_C(3);
    else LAUNCH_FB1_C(4);
  }
#undef LAUNCH_FB1_C
#undef LAUNCH_FB1
  return check_launch("first_bwd");
}

int launch_first_bwd(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw, const float* coef,
                     const float* in_slope, float* partials, int B, int Ci, int Co, int T, int V, int max_rows, hipStream_t st,
                     int* rows_out) {
#define CALL(T_, V_) return launch_first_bwd_tv<T_, V_>(in, Zg, dU, Aw, Tw, coef, in_slope, partials, B, Ci, Co, max_rows, st, rows_out)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

// training forward of a few-channel layer: moments + stored Z (partial rows: *rows_out, each 2 (Ci^2 + Ci) floats)
template <int T, int V>
static int launch_first_moments_tv(const float* in, const float* Aw, const float* Tw, const float* in_slope, float* partials, int B,
                                   int Ci, float* Zout, int max_rows, hipStream_t st, int* rows_out) {
  int grid = (B + fl::NWM - 1) / fl::NWM;
  if (grid > 512) grid = 512;
  if (grid > max_rows) grid = max_rows;
  *rows_out = grid;
  const size_t lds = ((size_t)T * V * V + (size_t)V * T * T + 2 * fl::NWM * (size_t)Ci * T * V) * sizeof(float);
#define LAUNCH_FM(CI)                                                                                                   \
  do {                                                                                                                  \
    auto k = fl::k_first_moments<T, V, CI>;                                                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * fl::NWM), lds, st, in, Aw, Tw, in_slope, partials, B, Zout);                    \
  } while (0)
  {
    ProbeScope probe(KID_FWD_MOMENTS, Ci, 0, st);
    if (Ci == 1) LAUNCH_FM(1);
    else if (Ci == 2) LAUNCH_FM(2);
    else if (Ci == 3) LAUNCH_FM(3);
    else LAUNCH_FM(4);
  }
#undef LAUNCH_FM
  return check_launch("first_moments");
}

int launch_first_moments(const float* in, const float* Aw, const float* Tw, const float* in_slope, float* partials, int B, int Ci,
                         int T, int V, float* Zout, int max_rows, hipStream_t st, int* rows_out) {
#define CALL(T_, V_) return launch_first_moments_tv<T_, V_>(in, Aw, Tw, in_slope, partials, B, Ci, Zout, max_rows, st, rows_out)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

int launch_first_apply(const float* Z, const float* in, float* out, const float* wfold, const float* bias, const float* in_slope,
                       int B, int Ci, int Co, int TVr, hipStream_t st) {
  const int grid = B < 2048 ? B : 2048;
  const size_t lds = 2 * (size_t)Ci * TVr * sizeof(float);
  const int CoP = round_up(Co, 16);
#define LAUNCH_FAP(CI) hipLaunchKernelGGL((fl::k_first_apply<CI>), dim3(grid), dim3(256), lds, st, Z, in, out, wfold, bias, in_slope, B, Co, CoP, TVr)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 1) LAUNCH_FAP(1);
    else if (Ci == 2) LAUNCH_FAP(2);
    else if (Ci == 3) LAUNCH_FAP(3);
    else LAUNCH_FAP(4);
  }
#undef LAUNCH_FAP
  return check_launch("first_apply");
}

}  // namespace coskad
