// Backward of one ST_GCNN_layer in training mode (reference stsgcn.py:94-116 under autograd).  Saved from the forward:
// the layer input (pre-activation of the producer), the small stat block and -- stored-Z path -- Z = gcn(X); without
// Z the kernels recompute the mixing (k_bwd_reduce / gcn phases below).
//
// Forward (per position):  Z = gcn(X),  S = Wt Z + bt,  R = Wr X + br,
//                          U = BN_s(S) + BN_r(R),       next layer's input = PReLU(U).
// Given dU (gradient w.r.t. the pre-activation U):
//
//  1. k_bwd_reduce(_z) : batch reductions  sdU = sum dU,  P = sum dU Z^T,  Q = sum dU X^T
//                    (GEMMs with K = positions -> MFMA f32).  Everything BatchNorm's backward
//                    needs is linear in these: e.g. sum dU*S = rowdot(Wt, P) + bt*sdU.
//                    _z: X and the stored Z both in LDS, ONE pass over dU.
//  2. k_bwd_fold   : one block.  Parameter gradients (dWt, dgamma, dbeta, ...; conv biases in
//                    front of a train-mode BN get exactly 0) and the coefficient matrices of
//                    the data path:
//                        dZ      = Bt dU + Kt Z + kt          (BN_s and conv_t transposed)
//                        dX_res  = Br dU + Kr X + kr          (BN_r and conv_r transposed)
//  3. k_bwd_data(_f) : dZ per position, adjoint gcn, + dX_res, then the PReLU derivative of the
//                    producer layer -> dU_prev; also the slope gradient of that PReLU.
//                    Stores dZ for step 4.  _f: single pass over dU (tiles of 32 rows = 1 clip of >= 32
//                    channels or 2 clips of 16), the default wherever it applies.
//  4. k_bwd_gcn_params : dA[t,v,w] = sum_rows Y[t,v] dZ[t,w],  dT[v,t,q] = sum_rows X[t,v] dY[q,v]
//                    (GEMMs with K = rows (clip,channel) -> MFMA f32).
#include "mfma_ops.h"
#include "fused_ops.h"
#include <cstdlib>

namespace coskad {

constexpr int kMaxGridBwd = 1024;  // partial rows: up to three 512-thread blocks per CU, or the 768 workgroups of the fused kernel
constexpr int kBtabFloats = (2 * 17 * 64 + 12 * 3 * 64) * 4;   // operand streams of fused_bwd.hip (T = 12, V = 17)

// ---------------------------------------------------------------------------------------
// 1. reductions.  LDS: X image (nb*Ci rows) then dU image (nb*Co rows) then 1024 scratch.
//    partial row: [P Co*Ci][Q Co*Ci][sdU Co]
// ---------------------------------------------------------------------------------------
// positions are processed in NCH chunks so that only a [Co x chunk] slab of dU sits in LDS
template <int T, int V>
struct RedGeo {
  static constexpr int TV = T * V;
  static constexpr int NCH = 3;
  static constexpr int CH = ((TV + NCH - 1) / NCH + 3) / 4 * 4;   // 68 for 12x17
  static constexpr int LDC = CH + 1;
  // strides of the stored-Z kernel, whose only LDS access pattern is the MFMA outer-product operand
  // "lane (i, k) -> row i, position p0 + k": a row stride == 2 (mod 4) spreads the 32 lanes of a half-wave over all
  // 32 banks (i * S takes 16 distinct even residues, + k in {0, 1} / {2, 3}); the odd strides above leave ~45 % of the
  // LDS cycles of that pattern in bank conflicts (rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
  static constexpr int LDZ = TV + 2;         // 206 for 12x17 (TV % 4 == 0)
  static constexpr int LDCZ = CH + 2;        // 70
};

// Out[a][b] += sum_{q < npos} A[a][offA + q] * B[b][offB + q]   (MFMA f32, waves split q)
template <int NTA, int NTB, bool SUMS>
__device__ __forceinline__ void outer_accum2(const float* ldsA, int ldA, int offA, int va, const float* ldsB,
                                             int ldB, int offB, int vb, int npos, f32x4 (&acc)[NTA][NTB],
                                             float (&rs)[NTA]) {
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  for (int p0 = 4 * wave; p0 < npos; p0 += 4 * (int)(blockDim.x >> 6)) {
    const bool pok = p0 + k < npos;
    float a[NTA], b[NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
      const int row = 16 * ta + i;
      a[ta] = (pok && row < va) ? ldsA[row * ldA + offA + p0 + k] : 0.f;
    }
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const int row = 16 * tb + i;
      b[tb] = (pok && row < vb) ? ldsB[row * ldB + offB + p0 + k] : 0.f;
    }
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb)
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
      if constexpr (SUMS) rs[ta] += a[ta];   // a is 0 outside the valid rows / positions
    }
  }
}

// stage rows x [pbeg, pbeg+npos) of a [rows][TV] global tile into a chunk image with stride LDC
template <int T, int V, int LDCX = 0>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ g, float* lds, int rows, int pbeg, int npos) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = T * V, LDC = LDCX ? LDCX : RedGeo<T, V>::LDC;
  const int n4 = npos >> 2;               // pbeg and npos are multiples of 4, TV % 4 == 0
  constexpr int UB = 3;   // HBM loads in flight per thread (64 rows x 17 float4 = 2.1 per thread)
  const int n = rows * n4;
  for (int e0 = threadIdx.x; e0 < n; e0 += UB * kBlock) {
    float4 v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * kBlock;
      const int row = e / n4, q4 = e - row * n4;
      v[u] = e < n ? *reinterpret_cast<const float4*>(g + (size_t)row * TV + pbeg + 4 * q4) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * kBlock;
      if (e < n) {
        const int row = e / n4, q4 = e - row * n4;
        float* d = lds + row * LDC + 4 * q4;
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
      }
    }
  }
}

template <int T, int V, int NTO, int NTC>
__global__ __launch_bounds__((Geo<T, V>::Block), (Geo<T, V>::Block <= 512 && NTO * NTC <= 2 ? 6 : 4)) void k_bwd_reduce(const float* __restrict__ in,
                                                      const float* __restrict__ dU,
                                                      const float* __restrict__ Aw,
                                                      const float* __restrict__ Tw,
                                                      const float* __restrict__ in_slope,
                                                      float* __restrict__ partials, int B, int Ci,
                                                      int Co, int NB, int need_q, const float* __restrict__ Zg) {
  constexpr int kScratchFloats = Geo<T, V>::Scratch;
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  constexpr int NCH = RedGeo<T, V>::NCH, CH = RedGeo<T, V>::CH, LDC = RedGeo<T, V>::LDC;
  static_assert(TV % 4 == 0, "chunked staging uses float4");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldx = lds;                         // X / Z image: NB*Ci rows x LD
  float* ldu = lds + NB * Ci * LD;          // dU slab: NB*Co rows x LDC
  float* scratch = lds;                     // kScratchFloats, aliased onto the images (only used after the tile loop)
  float* AwL = lds + max(NB * Ci * LD + NB * Co * LDC, kScratchFloats);
  float* TwL = AwL + T * V * V;
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  f32x4 pacc[NTO][NTC], qacc[NTO][NTC];
  float srow[NTO], sdummy[NTO];
  zero_acc(pacc); zero_acc(qacc);
#pragma unroll
  for (int t = 0; t < NTO; ++t) { srow[t] = 0.f; sdummy[t] = 0.f; }

  const int ntiles = ceil_div(B, NB);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const float* gdu = dU + (size_t)clip0 * Co * TV;
    lds_barrier();
    stage_rows<T, V>(in + (size_t)clip0 * Ci * TV, ldx, nb * Ci * TV, pre, a_in);
    if (need_q) {
      for (int ch = 0; ch < NCH; ++ch) {
        const int pbeg = ch * CH, npos = min(CH, TV - pbeg);
        lds_barrier();
        stage_chunk<T, V>(gdu, ldu, nb * Co, pbeg, npos);
        lds_barrier();
        for (int n = 0; n < nb; ++n)
          outer_accum2<NTO, NTC, false>(ldu + n * Co * LDC, LDC, 0, Co, ldx + n * Ci * LD, LD, pbeg, Ci, npos, qacc, sdummy);
      }
    }
    lds_barrier();
    if (Zg) stage_rows<T, V>(Zg + (size_t)clip0 * Ci * TV, ldx, nb * Ci * TV, false, 0.f);   // stored gcn(X)
    else gcn_mfma<T, V, false>(ldx, nb * Ci, AwL, TwL);
    for (int ch = 0; ch < NCH; ++ch) {
      const int pbeg = ch * CH, npos = min(CH, TV - pbeg);
      lds_barrier();
      stage_chunk<T, V>(gdu, ldu, nb * Co, pbeg, npos);
      lds_barrier();
      for (int n = 0; n < nb; ++n)
        outer_accum2<NTO, NTC, true>(ldu + n * Co * LDC, LDC, 0, Co, ldx + n * Ci * LD, LD, pbeg, Ci, npos, pacc, srow);
    }
  }
  float* dst = partials + (size_t)blockIdx.x * (2 * Co * Ci + Co);
  store_outer<NTO, NTC>(pacc, scratch, dst, Ci, Co, Ci);
  store_outer<NTO, NTC>(qacc, scratch, dst + Co * Ci, Ci, Co, Ci);
  store_rowsums<NTO>(srow, scratch, dst + 2 * Co * Ci, Co);
}

// Stored-Z variant: X and Z = gcn(X) both sit in LDS (no mixing tables needed), so ONE pass over dU feeds both
// P = sum dU Z^T and Q = sum dU X^T -- half the dU staging and barriers of the recompute kernel above.
template <int NTA, int NTB>
__device__ __forceinline__ void outer_accum_pq(const float* ldsA, int ldA, int va, const float* ldsX, const float* ldsZ,
                                               int ldB, int offB, int vb, int npos, bool need_q,
                                               f32x4 (&pacc)[NTA][NTB], f32x4 (&qacc)[NTA][NTB], float (&rs)[NTA]) {
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  for (int p0 = 4 * wave; p0 < npos; p0 += 4 * (int)(blockDim.x >> 6)) {
    const bool pok = p0 + k < npos;
    float a[NTA], bz[NTB], bx[NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
      const int row = 16 * ta + i;
      a[ta] = (pok && row < va) ? ldsA[row * ldA + p0 + k] : 0.f;
    }
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const int row = 16 * tb + i;
      const bool ok = pok && row < vb;
      bz[tb] = ok ? ldsZ[row * ldB + offB + p0 + k] : 0.f;
      bx[tb] = (ok && need_q) ? ldsX[row * ldB + offB + p0 + k] : 0.f;
    }
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb) {
        pacc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], bz[tb], pacc[ta][tb], 0, 0, 0);
        if (need_q) qacc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], bx[tb], qacc[ta][tb], 0, 0, 0);
      }
      rs[ta] += a[ta];
    }
  }
}

template <int T, int V, int NTO, int NTC>
__global__ __launch_bounds__((Geo<T, V>::Block), (Geo<T, V>::Block <= 512 && NTO * NTC <= 2 ? 6 : 4)) void k_bwd_reduce_z(
    const float* __restrict__ in, const float* __restrict__ Zg, const float* __restrict__ dU,
    const float* __restrict__ in_slope, float* __restrict__ partials, int B, int Ci, int Co, int NB, int need_q) {
  constexpr int TV = Geo<T, V>::TV, LD = RedGeo<T, V>::LDZ;
  constexpr int NCH = RedGeo<T, V>::NCH, CH = RedGeo<T, V>::CH, LDC = RedGeo<T, V>::LDCZ;
  static_assert(TV % 4 == 0, "chunked staging uses float4");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldz = lds;                         // Z image: NB*Ci rows x LD
  float* ldx = ldz + NB * Ci * LD;          // X image (only when the residual branch has a conv)
  float* ldu = ldx + (need_q ? NB * Ci * LD : 0);   // dU slab: NB*Co rows x LDC
  float* scratch = lds;                     // aliased: used after the tile loop only
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  f32x4 pacc[NTO][NTC], qacc[NTO][NTC];
  float srow[NTO];
  zero_acc(pacc); zero_acc(qacc);
#pragma unroll
  for (int t = 0; t < NTO; ++t) srow[t] = 0.f;
  const int ntiles = ceil_div(B, NB);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const float* gdu = dU + (size_t)clip0 * Co * TV;
    lds_barrier();
    stage_rows<T, V, LD>(Zg + (size_t)clip0 * Ci * TV, ldz, nb * Ci * TV, false, 0.f);
    if (need_q) stage_rows<T, V, LD>(in + (size_t)clip0 * Ci * TV, ldx, nb * Ci * TV, pre, a_in);
    for (int ch = 0; ch < NCH; ++ch) {
      const int pbeg = ch * CH, npos = min(CH, TV - pbeg);
      if (ch > 0) lds_barrier();          // the previous slab has been consumed
      stage_chunk<T, V, LDC>(gdu, ldu, nb * Co, pbeg, npos);
      lds_barrier();
      for (int n = 0; n < nb; ++n)
        outer_accum_pq<NTO, NTC>(ldu + n * Co * LDC, LDC, Co, ldx + n * Ci * LD, ldz + n * Ci * LD, LD, pbeg, Ci, npos,
                                 need_q != 0, pacc, qacc, srow);
    }
  }
  float* dst = partials + (size_t)blockIdx.x * (2 * Co * Ci + Co);
  store_outer<NTO, NTC>(pacc, scratch, dst, Ci, Co, Ci);
  store_outer<NTO, NTC>(qacc, scratch, dst + Co * Ci, Ci, Co, Ci);
  store_rowsums<NTO>(srow, scratch, dst + 2 * Co * Ci, Co);
}

constexpr int kRedCols = 16;   // columns per block of the partial-row sums (common.h: column_sum_f64)
__global__ __launch_bounds__(1024) void k_reduce_partials_d(const float* __restrict__ partials, int P, int E,
                                                             double* __restrict__ out) {
  __shared__ double sh[1024];
  const int e = blockIdx.x * kRedCols + (threadIdx.x % kRedCols);
  const double t = column_sum_f64<kRedCols>(partials, P, (size_t)E, e, e < E, sh);
  if ((int)threadIdx.x < kRedCols && e < E) out[e] = t;
}

// ---------------------------------------------------------------------------------------
// 2. fold.  stat block layout: see stsgcn_train.hip (muX, muZ, WCs, WCr, mean_s, istd_s, mean_r, istd_r)
//    coef block (floats): [wDZ (Co+Ci)*CiP][kt CiP][wDX (Co+Ci)*CiP][kr CiP]
// ---------------------------------------------------------------------------------------
__host__ __device__ inline int coef_floats(int Ci, int Co) {
  const int CiP = round_up(Ci, 16);
  return 2 * ((Co + Ci) * CiP + CiP);
}

__device__ __forceinline__ void put(float* dst, float v, int accumulate) { *dst = accumulate ? *dst + v : v; }

#ifdef COSKAD_FOLD_TIMING   // timing-only build (tools/time_fold.py): wall-clock stamps (100 MHz) of thread 0 at the fold's phase boundaries
__device__ long long g_fold_stamps[16];
#define FOLD_STAMP(k) do { if (threadIdx.x == 0) g_fold_stamps[k] = wall_clock64(); } while (0)
#else
#define FOLD_STAMP(k) do {} while (0)
#endif

__global__ __launch_bounds__(1024) void k_bwd_fold(
    const double* __restrict__ red, double npos, const float* __restrict__ stat,
    const float* __restrict__ Wt, const float* __restrict__ gs, const float* __restrict__ Wr,
    const float* __restrict__ gr, float* __restrict__ dWt, float* __restrict__ dbt,
    float* __restrict__ dgs, float* __restrict__ dbs, float* __restrict__ dWr, float* __restrict__ dbr,
    float* __restrict__ dgr, float* __restrict__ dbr2, float* __restrict__ coef, int Ci, int Co,
    int accumulate, const float* __restrict__ Aw, const float* __restrict__ Tw, float* __restrict__ btab, int NF) {
  // blocks NF..: the operand tables of the fused data kernel (fused_bwd.hip) from this layer's A / T -- parameter-only work
  // that rides in this launch instead of standing between the fold and the data kernel as a launch of its own
  if ((int)blockIdx.x >= NF) {
    const int e0 = (blockIdx.x - NF) * blockDim.x + threadIdx.x;
    for (int e = e0; e < ff::BTAB_F4 * 4; e += (gridDim.x - NF) * blockDim.x) btab[e] = ff::btab_value(Aw, Tw, e);
    return;
  }
  // Blocks 0..NF-1: the fold.  One global round trip at the start (every input goes to LDS), then LDS-only phases.  Every
  // block forms the per-channel quantities of ALL channels (the K matrices and constants sum over them), then writes its
  // slice of the outputs: output channels [o_lo, o_hi) of the parameter gradients and coefficient rows, a slice of the
  // (c, c') pairs of the K matrices (formed from W k2 kept as doubles: one DFMA per term; symmetric: c' >= c only).
  // LDS: doubles k2[2][Co], g1[2][Co], h[2][Co], red[E] (its P / Q part becomes W k2 once the weight gradients are out);
  //      floats W[2][Co*Ci], WC[2][Co*Ci], mu[2][Ci], istd[2][Co], gamma[2][Co]
  extern __shared__ double shd[];
  const bool ident = Wr == nullptr;
  const int CiP = round_up(Ci, 16);
  const int E = 2 * Co * Ci + Co;
  const int CC = Co * Ci;
  double* k2 = shd;            // [2][Co]
  double* g1 = shd + 2 * Co;   // [2][Co]
  double* hh = shd + 4 * Co;   // [2][Co]
  double* redl = shd + 6 * Co; // [E]
  double* P = redl;
  double* Q = redl + CC;
  const double* sdU = redl + 2 * CC;
  float* Wl = reinterpret_cast<float*>(redl + E);      // [2][Co*Ci]
  float* WCl = Wl + 2 * CC;                            // [2][Co*Ci]: Wt C_Z, Wr C_X (saved by the forward fold)
  float* mul = WCl + 2 * CC;                           // [2][Ci]: muZ then muX
  float* isl = mul + 2 * Ci;                           // [2][Co]: istd_s, istd_r
  float* gml = isl + 2 * Co;                           // [2][Co]: gamma_s, gamma_r
  float* wDZ = coef;
  float* kt = wDZ + (Co + Ci) * CiP;
  float* wDX = kt + CiP;
  float* kr = wDX + (Co + Ci) * CiP;
  const float* WCs = stat + 2 * Ci;
  const float* istd_s = WCs + 2 * CC + Co;
  const float* istd_r = istd_s + 2 * Co;
  const int chunk = (Co + NF - 1) / NF;
  const int o_lo = blockIdx.x * chunk, o_hi = min(Co, o_lo + chunk);

  FOLD_STAMP(0);
  for (int i = threadIdx.x; i < E; i += blockDim.x) redl[i] = red[i];
  for (int i = threadIdx.x; i < CC; i += blockDim.x) {
    Wl[i] = Wt[i];
    Wl[CC + i] = ident ? 0.f : Wr[i];
    WCl[i] = WCs[i];
    WCl[CC + i] = WCs[CC + i];
  }
  for (int i = threadIdx.x; i < Ci; i += blockDim.x) {
    mul[i] = stat[Ci + i];   // muZ (branch 0)
    mul[Ci + i] = stat[i];   // muX (branch 1)
  }
  for (int i = threadIdx.x; i < Co; i += blockDim.x) {
    isl[i] = istd_s[i];
    isl[Co + i] = ident ? 1.f : istd_r[i];
    gml[i] = gs[i];
    gml[Co + i] = ident ? 1.f : gr[i];
  }
  __syncthreads();
  FOLD_STAMP(1);
  // per-channel BN gradients: 8 lanes per (branch, channel), each sums every 8th input channel, then a fixed-order tree
  for (int i0 = 0; i0 < 2 * Co; i0 += blockDim.x / 8) {
    const int i = i0 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    const bool live = i < 2 * Co;
    const int b = live ? i / Co : 0, o = live ? i - b * Co : 0;
    double wp = 0.0, wmu = 0.0;
    if (live && !(b == 1 && ident)) {
      const float* W = Wl + b * CC + o * Ci;
      const double* M = (b ? Q : P) + o * Ci;
      const float* mu = mul + b * Ci;
      for (int c = l8; c < Ci; c += 8) {
        const double w = (double)W[c];
        wp += w * M[c];
        wmu += w * (double)mu[c];
      }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      wp += __shfl_xor(wp, off, 8);
      wmu += __shfl_xor(wmu, off, 8);
    }
    if (live && l8 == 0) {
      if (b == 1 && ident) { k2[i] = 0.0; g1[i] = 1.0; hh[i] = 0.0; }
      else {
        const double istd = (double)isl[i];
        const double gamma = (double)gml[i];
        const double sdus = wp - wmu * sdU[o];       // sum dU * (conv_out - mean)
        const double dgamma = istd * sdus;
        if (o >= o_lo && o < o_hi) {                 // (this block's output channels)
          put((b ? dgr : dgs) + o, (float)dgamma, accumulate);
          put((b ? dbr2 : dbs) + o, (float)sdU[o], accumulate);
          float* dbias = b ? dbr : dbt;
          if (dbias) put(dbias + o, 0.f, accumulate);  // conv bias in front of a train-mode BN: exactly 0
        }
        const double g = gamma * istd;
        const double kk = g * dgamma * istd / npos;
        g1[i] = g;
        k2[i] = kk;
        hh[i] = g * sdU[o] / npos - kk * wmu;
      }
    }
  }
  __syncthreads();
  FOLD_STAMP(2);
  // conv weight gradients: dW[o,c] = g1 (M[o,c] - sdU[o] mu[c]) - k2 * n * WC[o,c]
  const int no = max(0, o_hi - o_lo);
  for (int ii = threadIdx.x; ii < 2 * no * Ci; ii += blockDim.x) {
    const int b = ii / (no * Ci);
    if (b == 1 && ident) break;
    const int j = o_lo * Ci + ii - b * no * Ci, o = j / Ci, c = j - o * Ci;
    const double* M = b ? Q : P;
    const double v = g1[b * Co + o] * (M[j] - sdU[o] * (double)mul[b * Ci + c]) - k2[b * Co + o] * npos * (double)WCl[b * CC + j];
    put((b ? dWr : dWt) + j, (float)v, accumulate);
  }
  // data-path coefficients, rows 0..Co-1: B[c][o] = g1[o] W[o,c]
  for (int ii = threadIdx.x; ii < 2 * no * CiP; ii += blockDim.x) {
    const int b = ii / (no * CiP);
    const int j = o_lo * CiP + ii - b * no * CiP;
    const int row = j / CiP, c = j - row * CiP;
    float v = 0.f;
    if (c < Ci) {
      if (b == 1 && ident) v = row == c ? 1.f : 0.f;
      else v = (float)(g1[b * Co + row] * (double)Wl[b * CC + row * Ci + c]);
    }
    (b ? wDX : wDZ)[j] = v;
  }
  __syncthreads();
  FOLD_STAMP(3);
  // W k2 as doubles over the dead P / Q sums
  for (int i = threadIdx.x; i < 2 * CC; i += blockDim.x) {
    const int b = i / CC, j = i - b * CC, o = j / Ci;
    redl[i] = (double)Wl[i] * k2[b * Co + o];
  }
  __syncthreads();
  // rows Co..: K[c][c'] = -sum_o W[o,c] k2[o] W[o,c'] (symmetric: formed for c' >= c, written to both places); padding columns 0
  const int tri = Ci * (Ci + 1) / 2;
  const int tchunk = (2 * tri + NF - 1) / NF;
  const int t_hi = min(2 * tri, ((int)blockIdx.x + 1) * tchunk);
  for (int i = blockIdx.x * tchunk + threadIdx.x; i < t_hi; i += blockDim.x) {
    const int b = i / tri;
    int r = i - b * tri, c = 0;
    while (r >= Ci - c) { r -= Ci - c; ++c; }
    const int c2 = c + r;
    float v = 0.f;
    if (!(b == 1 && ident)) {
      const double* Wk = redl + b * CC;
      const float* W = Wl + b * CC;
      double acc = 0.0;
      for (int o = 0; o < Co; ++o) acc += Wk[o * Ci + c] * (double)W[o * Ci + c2];
      v = (float)(-acc);
    }
    float* dst = b ? wDX : wDZ;
    dst[(Co + c2) * CiP + c] = v;
    dst[(Co + c) * CiP + c2] = v;
  }
  if (CiP > Ci && blockIdx.x == 0) {
    for (int i = threadIdx.x; i < 2 * Ci * (CiP - Ci); i += blockDim.x) {
      const int b = i / (Ci * (CiP - Ci)), j = i - b * Ci * (CiP - Ci);
      const int row = j / (CiP - Ci), c = Ci + j - row * (CiP - Ci);
      (b ? wDX : wDZ)[(Co + row) * CiP + c] = 0.f;
    }
  }
  FOLD_STAMP(4);
  // constants: k[c] = -sum_o W[o,c] (g1[o] sdU[o]/n - k2[o] (W[o] . mu)) = -sum_o W[o,c] h[o]   (the last fold block)
  if ((int)blockIdx.x == NF - 1) {
    for (int i = threadIdx.x; i < 2 * CiP; i += blockDim.x) {
      const int b = i / CiP, c = i - b * CiP;
      float v = 0.f;
      if (c < Ci && !(b == 1 && ident)) {
        const float* W = Wl + b * CC;
        double acc = 0.0;
        for (int o = 0; o < Co; ++o) acc += (double)W[o * Ci + c] * hh[b * Co + o];
        v = (float)(-acc);
      }
      (b ? kr : kt)[c] = v;
    }
  }
  FOLD_STAMP(5);
}

// ---------------------------------------------------------------------------------------
// 3. data path.  LDS: one Ci-row image per clip (+ 64 floats).  CIP = Ci rounded up to a
//    power of two (register arrays need static bounds).
// ---------------------------------------------------------------------------------------
template <int T, int V, int OTI>
__global__ __launch_bounds__((Geo<T, V>::Block), kMinWaves) void k_bwd_data(
    const float* __restrict__ in, const float* __restrict__ dU, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ coef, const float* __restrict__ in_slope,
    float* __restrict__ dIn, float* __restrict__ dZout, float* __restrict__ da_partials, int B, int Ci,
    int Co, int NB, const float* __restrict__ Zg) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sred[kBlock / 64];
  const int CiP = round_up(Ci, 16);
  const int KZ = round_up(Ci, 4), K1 = round_up(Co, 4);
  const float* wDZ = coef;
  const float* kt = wDZ + (Co + Ci) * CiP;
  const float* wDX = kt + CiP;
  const float* kr = wDX + (Co + Ci) * CiP;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  float* AwL = lds + NB * Ci * LD;
  float* TwL = AwL + T * V * V;
  float* WlA = TwL + V * T * T;              // [(KZ + K1)][CiP]: Kt rows (LDS source Z), then Bt rows (dU)
  float* WlB = WlA + (KZ + K1) * CiP;        // [(K1 + KZ)][CiP]: Br rows (dU), then Kr rows (X)
  float* ktl = WlB + (K1 + KZ) * CiP;        // [CiP]
  float* krl = ktl + CiP;                    // [CiP]
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  for (int e = threadIdx.x; e < (KZ + K1) * CiP; e += kBlock) {
    const int k = e / CiP, c = e - k * CiP;
    float va = 0.f, vb = 0.f;
    if (k < KZ) { if (k < Ci) va = wDZ[(Co + k) * CiP + c]; }
    else if (k - KZ < Co) va = wDZ[(k - KZ) * CiP + c];
    if (k < K1) { if (k < Co) vb = wDX[k * CiP + c]; }
    else if (k - K1 < Ci) vb = wDX[(Co + k - K1) * CiP + c];
    WlA[e] = va;
    WlB[e] = vb;
  }
  copy_to_lds(ktl, kt, CiP);
  copy_to_lds(krl, kr, CiP);
  const int wave = uniform(threadIdx.x >> 6);
  float da = 0.f;
  const int ntiles = ceil_div(B, NB);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const int rows = nb * Ci;
    const float* gin = in + (size_t)clip0 * Ci * TV;
    const float* gdu = dU + (size_t)clip0 * Co * TV;

    lds_barrier();
    if (Zg) {
      stage_rows<T, V>(Zg + (size_t)clip0 * Ci * TV, lds, rows * TV, false, 0.f);   // stored gcn(X): X itself is only read in phase B
    } else {
      stage_rows<T, V>(gin, lds, rows * TV, pre, a_in);
      lds_barrier();
      gcn_mfma<T, V, false>(lds, rows, AwL, TwL);
    }
    lds_barrier();
    // phase A (MFMA): dZ[:,p] = kt + Kt Z[:,p] + Bt dU[:,p], in place over Z.  One item = all channels of a
    // strip, so every Z column is fully read before it is overwritten.
    for (int n = 0; n < nb; ++n) {
      float* img = lds + n * Ci * LD;
      auto epiA = [&](int o, int p, bool pok, float v0, float v1) {
        if (o < Ci && pok) {
          img[o * LD + p] = v0 + ktl[o];
          img[o * LD + p + 1] = v1 + ktl[o];
        }
      };
      conv_mfma_s<T, V, OTI>(img, KZ, Ci, gdu + (size_t)n * Co * TV, K1, Co, nullptr, 0, 1, false, 0.f, WlA, CiP, 0,
                             (wave + n) % (kBlock / 64), kBlock / 64, epiA);
    }
    lds_barrier();
    if (dZout) {
      unstage_rows<T, V>(dZout + (size_t)clip0 * Ci * TV, lds, rows * TV);
      lds_barrier();   // the adjoint mixing below overwrites dZ in place: every wave must have copied it out
    }
    if (dIn) {
      gcn_mfma<T, V, true>(lds, rows, AwL, TwL);
      lds_barrier();
      // phase B (MFMA): dX = gcn^T(dZ) + kr + Br dU + Kr X ; dU_prev = dX * PReLU'(U_prev)
      for (int n = 0; n < nb; ++n) {
        const float* img = lds + n * Ci * LD;
        const float* ug = gin + (size_t)n * Ci * TV;
        float* dg = dIn + (size_t)(clip0 + n) * Ci * TV;
        auto epiB = [&](int o, int p, bool pok, float v0, float v1) {
          if (o < Ci && pok) {
            float g0 = v0 + krl[o] + img[o * LD + p];
            float g1 = v1 + krl[o] + img[o * LD + p + 1];
            if (pre) {
              const float2 u = *reinterpret_cast<const float2*>(ug + (size_t)o * TV + p);
              if (u.x < 0.f) da = fmaf(g0, u.x, da);
              if (u.y < 0.f) da = fmaf(g1, u.y, da);
              g0 = u.x > 0.f ? g0 : a_in * g0;
              g1 = u.y > 0.f ? g1 : a_in * g1;
            }
            *reinterpret_cast<float2*>(dg + (size_t)o * TV + p) = float2{g0, g1};
          }
        };
        conv_mfma_s<T, V, OTI>(nullptr, 0, 1, gdu + (size_t)n * Co * TV, K1, Co, ug, KZ, Ci, pre, a_in, WlB, CiP, 0,
                               (wave + n) % (kBlock / 64), kBlock / 64, epiB);
      }
    }
  }  // tile loop
  if (da_partials) {
    da = wave_sum(da);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = da;
    lds_barrier();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < kBlock / 64; ++w) t += sred[w];
      da_partials[blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------
// 3b. data path, single-read variant for C_in >= 32 (one clip per tile, one strip per wave):
//     * Kr.X is taken from the LDS image BEFORE the mixing overwrites it,
//     * dU is read ONCE: each 8-byte operand feeds both Bt.dU (-> dZ) and Br.dU (-> dX_res),
//     * the dX_res accumulators stay in registers across the two mixing phases.
// ---------------------------------------------------------------------------------------
// NBF clips per tile (NBF * C_in = 32 rows: 1 clip of 32 channels or 2 clips of 16); a wave owns its strip in every clip.
template <int T, int V, int OTI, int NBF>
__global__ __launch_bounds__((Geo<T, V>::Block), kMinWaves) void k_bwd_data_f(
    const float* __restrict__ in, const float* __restrict__ dU, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ coef, const float* __restrict__ in_slope,
    float* __restrict__ dIn, float* __restrict__ dZout, float* __restrict__ da_partials, int B, int Ci,
    int Co, const float* __restrict__ Zg
#ifdef COSKAD_ABLATE
    , int abl
#endif
    ) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
#ifndef COSKAD_ABLATE
  constexpr int abl = 0;
#endif
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  constexpr int PS = (TV + 31) / 32;
#ifndef COSKAD_XBF
#define COSKAD_XBF 2
#endif
  constexpr int XB = COSKAD_XBF;
  static_assert(PS <= kBlock / 64, "one strip per wave");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float sred[kBlock / 64];
  const int CiP = round_up(Ci, 16);
  const int KZ = round_up(Ci, 4), K1 = round_up(Co, 4);
  const float* wDZ = coef;
  const float* kt = wDZ + (Co + Ci) * CiP;
  const float* wDX = kt + CiP;
  const float* kr = wDX + (Co + Ci) * CiP;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  float* imgt = lds;                         // NBF * C_in rows
  float* AwL = lds + NBF * Ci * LD;
  float* TwL = AwL + T * V * V;
  float* WlA = TwL + V * T * T;              // [(KZ + K1)][CiP]: Kt rows (Z), then Bt rows (dU)
  float* WlB = WlA + (KZ + K1) * CiP;        // [(K1 + KZ)][CiP]: Br rows (dU), then Kr rows (X)
  float* ktl = WlB + (K1 + KZ) * CiP;
  float* krl = ktl + CiP;
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  for (int e = threadIdx.x; e < (KZ + K1) * CiP; e += kBlock) {
    const int k = e / CiP, c = e - k * CiP;
    float va = 0.f, vb = 0.f;
    if (k < KZ) { if (k < Ci) va = wDZ[(Co + k) * CiP + c]; }
    else if (k - KZ < Co) va = wDZ[(k - KZ) * CiP + c];
    if (k < K1) { if (k < Co) vb = wDX[k * CiP + c]; }
    else if (k - K1 < Ci) vb = wDX[(Co + k - K1) * CiP + c];
    WlA[e] = va;
    WlB[e] = vb;
  }
  copy_to_lds(ktl, kt, CiP);
  copy_to_lds(krl, kr, CiP);
  const int wave = uniform(threadIdx.x >> 6);
  const bool mine = wave < PS;                 // this wave's strip
  const int KZS = KZ / 4, K1S = K1 / 4;
  float da = 0.f;
  int tid = 0, lane = 0, j = 0, kk = 0, p = 0, pc = 0;
  bool pok = false;
  // lane geometry behind an optimisation barrier, refreshed per clip: the address arithmetic of every phase is
  // then recomputed where it is used instead of being hoisted out of the clip loop and parked in VGPRs
  auto refresh = [&]() {
    tid = tid_here();
    lane = tid & 63;
    j = lane & 15; kk = lane >> 4;
    p = 32 * wave + 2 * j;
    pok = mine && p < TV;
    pc = p < TV ? p : TV - 2;
  };
  refresh();
  // acc += W[0:Ci] . img over this wave's strip; 4 k-steps of LDS operands in flight before their MFMAs
  auto lds_conv = [&](const float* Wk, f32x4 (&acc)[OTI][2], const float* img) {
    for (int s0 = 0; s0 < KZS; s0 += 4) {
      float b0[4], b1[4], a[4][OTI];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = 4 * (s0 + u) + kk;
        const bool ok = s0 + u < KZS;                  // beyond KZ: zero data, any (valid) weight
        const int cc = (ok && c < Ci) ? c : Ci - 1;
        b0[u] = ok ? img[cc * LD + pc] : 0.f;
        b1[u] = ok ? img[cc * LD + pc + 1] : 0.f;
        const float* w = Wk + (ok ? c : 0) * CiP + j;
#pragma unroll
        for (int t = 0; t < OTI; ++t) a[u][t] = w[16 * t];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < OTI; ++t) {
          acc[t][0] = mfma4(a[u][t], b0[u], acc[t][0]);
          acc[t][1] = mfma4(a[u][t], b1[u], acc[t][1]);
        }
    }
  };

  const int ntiles = ceil_div(B, NBF);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip = tile * NBF;                       // first clip of the tile
    const int nb = min(NBF, B - clip);
    const int rows = nb * Ci;
    const float* gin = in + (size_t)clip * Ci * TV;
    f32x4 accA[NBF][OTI][2], accB[NBF][OTI][2];
#pragma unroll
    for (int n = 0; n < NBF; ++n)
#pragma unroll
      for (int t = 0; t < OTI; ++t) {
        accA[n][t][0] = accA[n][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        accB[n][t][0] = accB[n][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    refresh();
    lds_barrier();
    if (!(abl & 1)) stage_rows<T, V>(gin, imgt, rows * TV, pre, a_in, tid);
    lds_barrier();
    // phase 0: accB = Kr . X  (LDS, before the mixing overwrites X)
    if (mine && !(abl & 2)) {
#pragma unroll
      for (int n = 0; n < NBF; ++n)
        if (n < nb) lds_conv(WlB + K1 * CiP, accB[n], imgt + n * Ci * LD);
    }
    lds_barrier();
    if (Zg) stage_rows<T, V>(Zg + (size_t)clip * Ci * TV, imgt, rows * TV, false, 0.f, tid);   // stored gcn(X) replaces X
    else if (!(abl & 4)) gcn_mfma<T, V, false>(imgt, rows, AwL, TwL, tid);
    lds_barrier();
    // phase A: accA = Kt.Z (LDS) + Bt.dU ; accB += Br.dU -- ONE pass over dU
    if (mine) {
#pragma unroll
     for (int n = 0; n < NBF; ++n) {
      if (n >= nb) break;
      float* img = imgt + n * Ci * LD;
      const float* gdu = dU + (size_t)(clip + n) * Co * TV;
      float2 cur[XB], nxt[XB];
      auto gload = [&](int g) -> float2 {
        if (g >= K1S) return float2{0.f, 0.f};
        const int c = 4 * g + kk;
        const int cc = c < Co ? c : Co - 1;
        return *reinterpret_cast<const float2*>(gdu + (size_t)cc * TV + pc);
      };
#pragma unroll
      for (int u = 0; u < XB; ++u) cur[u] = gload(u);
      if (!(abl & 16)) lds_conv(WlA, accA[n], img);
      for (int g0 = 0; g0 < ((abl & 8) ? 0 : K1S); g0 += XB) {
#pragma unroll
        for (int u = 0; u < XB; ++u) nxt[u] = gload(g0 + XB + u);
#pragma unroll
        for (int u = 0; u < XB; ++u) {
          if (g0 + u < K1S) {
            const float* wa = WlA + (KZ + 4 * (g0 + u) + kk) * CiP + j;
            const float* wb = WlB + (4 * (g0 + u) + kk) * CiP + j;
#pragma unroll
            for (int t = 0; t < OTI; ++t) {
              const float a1 = wa[16 * t], a2 = wb[16 * t];
              accA[n][t][0] = mfma4(a1, cur[u].x, accA[n][t][0]);
              accA[n][t][1] = mfma4(a1, cur[u].y, accA[n][t][1]);
              if (dIn) {
                accB[n][t][0] = mfma4(a2, cur[u].x, accB[n][t][0]);
                accB[n][t][1] = mfma4(a2, cur[u].y, accB[n][t][1]);
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < XB; ++u) cur[u] = nxt[u];
      }
      // dZ in place (all of this strip's Z columns were read above)
      if (pok) {
#pragma unroll
        for (int t = 0; t < OTI; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * t + 4 * kk + r;
            if (o < Ci) {
              img[o * LD + p] = accA[n][t][0][r] + ktl[o];
              img[o * LD + p + 1] = accA[n][t][1][r] + ktl[o];
            }
          }
      }
     }
    }
    lds_barrier();
    if (dZout && !(abl & 32)) {
      unstage_rows<T, V>(dZout + (size_t)clip * Ci * TV, imgt, rows * TV, tid);
      lds_barrier();   // the adjoint mixing below overwrites dZ in place: every wave must have copied it out
    }
    if (dIn) {
      // the PReLU masks of the epilogue (pre-activations of the layer input) are fetched before the adjoint mixing,
      // so their latency is covered by it
      float2 um[NBF][OTI][4];
      if (pre && pok) {
#pragma unroll
        for (int n = 0; n < NBF; ++n)
#pragma unroll
          for (int t = 0; t < OTI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int o = 16 * t + 4 * kk + r;
              const int nn = n < nb ? n : 0;
              um[n][t][r] = *reinterpret_cast<const float2*>(gin + ((size_t)nn * Ci + (o < Ci ? o : Ci - 1)) * TV + p);
            }
      }
      if (!(abl & 64)) gcn_mfma<T, V, true>(imgt, rows, AwL, TwL, tid);
      lds_barrier();
      if (pok && !(abl & 128)) {
#pragma unroll
       for (int n = 0; n < NBF; ++n) {
        if (n >= nb) break;
        const float* img = imgt + n * Ci * LD;
        float* dg = dIn + (size_t)(clip + n) * Ci * TV;
#pragma unroll
        for (int t = 0; t < OTI; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * t + 4 * kk + r;
            if (o < Ci) {
              float g0 = accB[n][t][0][r] + krl[o] + img[o * LD + p];
              float g1 = accB[n][t][1][r] + krl[o] + img[o * LD + p + 1];
              if (pre) {
                const float2 u = um[n][t][r];
                if (u.x < 0.f) da = fmaf(g0, u.x, da);
                if (u.y < 0.f) da = fmaf(g1, u.y, da);
                g0 = u.x > 0.f ? g0 : a_in * g0;
                g1 = u.y > 0.f ? g1 : a_in * g1;
              }
              *reinterpret_cast<float2*>(dg + (size_t)o * TV + p) = float2{g0, g1};
            }
          }
       }
      }
    }
  }
  if (da_partials) {
    da = wave_sum(da);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = da;
    lds_barrier();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < kBlock / 64; ++w) t += sred[w];
      da_partials[blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------
// 4. gcn parameter gradients.  LDS: image Y/X (nb*Ci rows) + image dZ/dY (nb*Ci rows).
//    MFMA with K = rows: A[i][k] = img1[row k][col a0+i], B[k][j] = img2[row k][col b0+j].
//    waves own disjoint frames (dA) / joints (dT): no cross-wave reduction.
//    partial row: [dA T*V*V][dT V*T*T]
// ---------------------------------------------------------------------------------------
template <int T, int V, int NTA, int NTB>
__device__ __forceinline__ void rowk_accum(const float* img1, int a0, int na, const float* img2, int b0,
                                           int nbcols, int rows, f32x4 (&acc)[NTA][NTB]) {
  constexpr int LD = Geo<T, V>::LD;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, k = lane >> 4;
  for (int r0 = 0; r0 < rows; r0 += 4) {
    const bool rok = r0 + k < rows;
    const int row = rok ? r0 + k : 0;
    float a[NTA], b[NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
      const int col = 16 * ta + i;
      a[ta] = (rok && col < na) ? img1[row * LD + a0 + col] : 0.f;
    }
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const int col = 16 * tb + i;
      b[tb] = (rok && col < nbcols) ? img2[row * LD + b0 + col] : 0.f;
    }
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb)
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
  }
}

template <int T, int V>
__global__ __launch_bounds__((Geo<T, V>::Block), (V <= 17 ? 6 : 4)) void k_bwd_gcn_params(const float* __restrict__ in,
                                                          const float* __restrict__ dZ,
                                                          const float* __restrict__ Aw,
                                                          const float* __restrict__ Tw,
                                                          const float* __restrict__ in_slope,
                                                          float* __restrict__ partials, int B, int Ci,
                                                          int NB, float* __restrict__ dX_out,
                                                          const float* __restrict__ dX_add
#ifdef COSKAD_ABLATE
                                                          , int abl
#endif
                                                          ) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
#ifndef COSKAD_ABLATE
  constexpr int abl = 0;
#endif
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  constexpr int NW = kBlock / 64;
  // dA is V x V per frame: MFMA tiles of 16 for the bulk; up to 2 leftover joints (V = 17, 18) on the VALU
  constexpr int VX = (V > 16 && V - 16 <= 2) ? V - 16 : 0;
  constexpr int VM = V - VX;                 // joints covered by MFMA tiles
  constexpr int NTV = (VM + 15) / 16, NTT = (T + 15) / 16;
  constexpr int TPW = (T + NW - 1) / NW, VPW = (V + NW - 1) / NW;
  constexpr int VXA = VX > 0 ? VX : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img1 = lds;
  float* img2 = lds + NB * Ci * LD;
  float* AwL = img2 + NB * Ci * LD;
  float* TwL = AwL + T * V * V;
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int wave = uniform(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, k = lane >> 4;
  const int tbeg = (T * wave) / NW, tend = (T * (wave + 1)) / NW;
  const int vbeg = (V * wave) / NW, vend = (V * (wave + 1)) / NW;
  f32x4 accA[TPW][NTV][NTV], accT[VPW][NTT][NTT];
  float pcol[TPW][VXA][NTV], prow[TPW][VXA][NTV], pcor[TPW][VXA][VXA];
#pragma unroll
  for (int a = 0; a < TPW; ++a) {
    zero_acc(accA[a]);
#pragma unroll
    for (int x = 0; x < VXA; ++x) {
#pragma unroll
      for (int t2 = 0; t2 < NTV; ++t2) { pcol[a][x][t2] = 0.f; prow[a][x][t2] = 0.f; }
#pragma unroll
      for (int y = 0; y < VXA; ++y) pcor[a][x][y] = 0.f;
    }
  }
#pragma unroll
  for (int a = 0; a < VPW; ++a) zero_acc(accT[a]);

  const int ntiles = ceil_div(B, NB);
  // Wide joint layouts (V > 18: the tables + two 32-row images leave ONE block per CU, so nothing else runs while a tile's rows
  // arrive): the NEXT tile's rows travel in registers while this tile multiplies -- dZ's from the first staging on, X's from the
  // second (X is staged twice: the registers keep it for that) -- 24 registers for <= 32 rows per tile.
  constexpr bool PFB = V > 18 && TV % 4 == 0;
  constexpr int PF4 = PFB ? (32 * (TV / 4) + kBlock - 1) / kBlock : 1;
  const bool pf = PFB && NB * Ci <= 32;
  float4 px[PF4], pz[PF4];
  auto pf_load = [&](const float* g, int rows_, float4 (&r)[PF4]) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int n4 = (rows_ * TV) >> 2;
#pragma unroll
    for (int u = 0; u < PF4; ++u) {
      const int i = threadIdx.x + u * kBlock;
      r[u] = i < n4 ? g4[i] : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto pf_store = [&](const float4 (&r)[PF4], float* img, int rows_, bool act) {
    const int n4 = (rows_ * TV) >> 2;
#pragma unroll
    for (int u = 0; u < PF4; ++u) {
      const int i = threadIdx.x + u * kBlock;
      if (i < n4) {
        const int e = i << 2;
        const int row = e / TV, col = e - row * TV;
        float4 v = r[u];
        if (act) { v.x = prelu_f(v.x, a_in); v.y = prelu_f(v.y, a_in); v.z = prelu_f(v.z, a_in); v.w = prelu_f(v.w, a_in); }
        float* d = img + row * LD + col;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
  };
  auto tile_rows = [&](int tile_) { return min(NB, B - tile_ * NB) * Ci; };
  if (pf && (int)blockIdx.x < ntiles) {
    pf_load(in + (size_t)blockIdx.x * NB * Ci * TV, tile_rows(blockIdx.x), px);
    pf_load(dZ + (size_t)blockIdx.x * NB * Ci * TV, tile_rows(blockIdx.x), pz);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const int rows = nb * Ci;
    const float* gin = in + (size_t)clip0 * Ci * TV;
    const int tnext = tile + gridDim.x;
    lds_barrier();
    if (pf) {
      pf_store(px, img1, rows, pre);
      pf_store(pz, img2, rows, false);
      if (tnext < ntiles) pf_load(dZ + (size_t)tnext * NB * Ci * TV, tile_rows(tnext), pz);
    } else {
      if (!(abl & 64)) stage_rows<T, V>(gin, img1, rows * TV, pre, a_in);
      if (!(abl & 1)) stage_rows<T, V>(dZ + (size_t)clip0 * Ci * TV, img2, rows * TV, false, 0.f);
    }
    lds_barrier();
    if (!(abl & 2)) temporal_mfma<T, V, false>(img1, rows, TwL);  // Y = temporal(X)
    lds_barrier();
    // dA[t] += Y[:, t, :]^T dZ[:, t, :]   (K = rows)
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      const int t = tbeg + tt;
      if (t < tend && !(abl & 4)) {
        // K = rows in batches of KB k-steps: all LDS operands of a batch are in flight before its MFMAs
        constexpr int KB = 4;
        for (int r0 = 0; r0 < rows; r0 += 4 * KB) {
          float a[KB][NTV], b[KB][NTV], yx[KB][VXA], dx[KB][VXA];
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            const int r = r0 + 4 * u + k;
            const bool rok = r < rows;
            const float* y = img1 + (rok ? r : 0) * LD + t * V;
            const float* d = img2 + (rok ? r : 0) * LD + t * V;
#pragma unroll
            for (int q = 0; q < NTV; ++q) {
              const int col = 16 * q + i;
              const bool ok = rok && col < VM;
              a[u][q] = ok ? y[col] : 0.f;
              b[u][q] = ok ? d[col] : 0.f;
            }
#pragma unroll
            for (int x = 0; x < VXA; ++x) {
              yx[u][x] = (VX > 0 && rok) ? y[VM + x] : 0.f;   // same address for the 16 lanes of a k group: broadcast
              dx[u][x] = (VX > 0 && rok) ? d[VM + x] : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < KB; ++u) {
#pragma unroll
            for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
              for (int tb = 0; tb < NTV; ++tb)
                accA[tt][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ta], b[u][tb], accA[tt][ta][tb], 0, 0, 0);
            if constexpr (VX > 0) {
#pragma unroll
              for (int x = 0; x < VX; ++x) {
#pragma unroll
                for (int q = 0; q < NTV; ++q) {
                  pcol[tt][x][q] = fmaf(a[u][q], dx[u][x], pcol[tt][x][q]);   // dA[t][v = 16q+i][VM+x]
                  prow[tt][x][q] = fmaf(yx[u][x], b[u][q], prow[tt][x][q]);   // dA[t][VM+x][w = 16q+i]
                }
#pragma unroll
                for (int y2 = 0; y2 < VX; ++y2) pcor[tt][x][y2] = fmaf(yx[u][x], dx[u][y2], pcor[tt][x][y2]);
              }
            }
          }
        }
      }
    }
    lds_barrier();
    if (!(abl & 8)) spatial_mfma<T, V, true>(img2, rows, AwL);  // dY = spatial^T(dZ)
    if (pf) {                                                          // X again (img1 is free: dA is done)
      pf_store(px, img1, rows, pre);
      if (tnext < ntiles) pf_load(in + (size_t)tnext * NB * Ci * TV, tile_rows(tnext), px);
    } else if (!(abl & 16)) stage_rows<T, V>(gin, img1, rows * TV, pre, a_in);
    lds_barrier();
    // dT[v][t][q] += sum_rows X[t*V+v] dY[q*V+v]
#pragma unroll
    for (int vv = 0; vv < VPW; ++vv) {
      const int v = vbeg + vv;
      if (v < vend && !(abl & 32)) {
        constexpr int KB = 4;
        for (int r0 = 0; r0 < rows; r0 += 4 * KB) {
          float a[KB][NTT], b[KB][NTT];
#pragma unroll
          for (int u = 0; u < KB; ++u) {
            const int r = r0 + 4 * u + k;
            const bool rok = r < rows;
            const int row = rok ? r : 0;
#pragma unroll
            for (int ta = 0; ta < NTT; ++ta) {
              const int t = 16 * ta + i;
              a[u][ta] = (rok && t < T) ? img1[row * LD + t * V + v] : 0.f;
              b[u][ta] = (rok && t < T) ? img2[row * LD + t * V + v] : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < KB; ++u)
#pragma unroll
            for (int ta = 0; ta < NTT; ++ta)
#pragma unroll
              for (int tb = 0; tb < NTT; ++tb)
                accT[vv][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ta], b[u][tb], accT[vv][ta][tb], 0, 0, 0);
        }
      }
    }
    if (dX_out) {
      // the adjoint mix's other half rides here: the second image holds dY = spatial^T(dZ) -- temporal^T of it is the input
      // gradient of ConvTemporalGraphical (+ an addend: the identity residual's gradient), written in place of a k_gcn launch
      // that would read dZ a second time
      lds_barrier();                                 // every wave has read dY
      temporal_mfma<T, V, true>(img2, rows, TwL);
      lds_barrier();
      if constexpr (TV % 4 == 0) {
        float4* g4 = reinterpret_cast<float4*>(dX_out + (size_t)clip0 * Ci * TV);
        const float4* a4 = dX_add ? reinterpret_cast<const float4*>(dX_add + (size_t)clip0 * Ci * TV) : nullptr;
        const int n4 = (rows * TV) >> 2;
        for (int i = threadIdx.x; i < n4; i += kBlock) {
          const int e = i << 2;
          const int row = e / TV, col = e - row * TV;
          const float* sp = img2 + row * LD + col;
          float4 v = {sp[0], sp[1], sp[2], sp[3]};
          if (a4) { const float4 w = a4[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
          g4[i] = v;
        }
      } else {
        float* gp = dX_out + (size_t)clip0 * Ci * TV;
        for (int e = threadIdx.x; e < rows * TV; e += kBlock) {
          const int row = e / TV, col = e - row * TV;
          gp[e] = img2[row * LD + col] + (dX_add ? dX_add[(size_t)clip0 * Ci * TV + e] : 0.f);
        }
      }
    }
  }
  // store: D[row = 4*(lane>>4)+r][col = lane&15]
  float* dstA = partials + (size_t)blockIdx.x * (T * V * V + V * T * T);
  float* dstT = dstA + T * V * V;
  const int col = lane & 15, rg = lane >> 4;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int t = tbeg + tt;
    if (t < tend) {
#pragma unroll
      for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTV; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int v = 16 * ta + 4 * rg + r, w = 16 * tb + col;
            if (v < VM && w < VM) dstA[(t * V + v) * V + w] = accA[tt][ta][tb][r];
          }
      if constexpr (VX > 0) {
#pragma unroll
        for (int x = 0; x < VX; ++x) {
#pragma unroll
          for (int q = 0; q < NTV; ++q) {
            float pc = pcol[tt][x][q], pr = prow[tt][x][q];
            pc += __shfl_xor(pc, 16, 64); pc += __shfl_xor(pc, 32, 64);
            pr += __shfl_xor(pr, 16, 64); pr += __shfl_xor(pr, 32, 64);
            const int idx = 16 * q + i;
            if (k == 0 && idx < VM) {
              dstA[(t * V + idx) * V + VM + x] = pc;
              dstA[(t * V + VM + x) * V + idx] = pr;
            }
          }
#pragma unroll
          for (int y2 = 0; y2 < VX; ++y2) {
            float c = pcor[tt][x][y2];   // identical in the 16 lanes of a k group
            c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
            if (lane == 0) dstA[(t * V + VM + x) * V + VM + y2] = c;
          }
        }
      }
    }
  }
#pragma unroll
  for (int vv = 0; vv < VPW; ++vv) {
    const int v = vbeg + vv;
    if (v < vend) {
#pragma unroll
      for (int ta = 0; ta < NTT; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTT; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = 16 * ta + 4 * rg + r, q = 16 * tb + col;
            if (t < T && q < T) dstT[(v * T + t) * T + q] = accT[vv][ta][tb][r];
          }
    }
  }
}

// out[e] (+)= sum_p partials[p*Erow + offset + e],  e < count.  block = 64 elements x 16 partial-slices.
__global__ __launch_bounds__(1024) void k_reduce_to_f32(const float* __restrict__ partials, int P, int Erow,
                                                         int offset, int count, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ double sh[1024];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < count) {
    const float* base = partials + offset + e;
    int p = slice;
    for (; p + 48 < P; p += 64) {
      s0 += (double)base[(size_t)p * Erow];
      s1 += (double)base[(size_t)(p + 16) * Erow];
      s2 += (double)base[(size_t)(p + 32) * Erow];
      s3 += (double)base[(size_t)(p + 48) * Erow];
    }
    for (; p < P; p += 16) s0 += (double)base[(size_t)p * Erow];
  }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0 && e < count) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[threadIdx.x + 64 * k];
    out[e] = accumulate ? out[e] + (float)t : (float)t;
  }
}

__global__ __launch_bounds__(256) void k_sum_to(const float* __restrict__ v, int n, float* __restrict__ out,
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

// dA, dT (and optionally the producer's slope gradient) from the block partials in ONE launch:
// blocks [0, ceil(E / kGcnCols)) sum kGcnCols columns x 32 row slices of [dA | dT] (common.h: column_sum_f64); one extra block sums `dap`.
constexpr int kGcnCols = 32;
__global__ __launch_bounds__(1024) void k_reduce_gcn(const float* __restrict__ partials, int P, int nA, int nT,
                                                      float* __restrict__ dA, float* __restrict__ dT,
                                                      const float* __restrict__ dap, int ndap,
                                                      float* __restrict__ dslope, int accumulate) {
  __shared__ double sh[1024];
  const int E = nA + nT;
  const int nblk = (E + kGcnCols - 1) / kGcnCols;
  if ((int)blockIdx.x == nblk) {     // slope-gradient block (launched only when dap != NULL)
    double s = 0.0;
    for (int i = threadIdx.x; i < ndap; i += 1024) s += (double)dap[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) dslope[0] = accumulate ? dslope[0] + (float)sh[0] : (float)sh[0];
    return;
  }
  const int e = blockIdx.x * kGcnCols + (threadIdx.x % kGcnCols);
  const double t = column_sum_f64<kGcnCols>(partials, P, (size_t)E, e, e < E, sh);
  if ((int)threadIdx.x < kGcnCols && e < E) {
    float* out = e < nA ? dA + e : dT + (e - nA);
    out[0] = accumulate ? out[0] + (float)t : (float)t;
  }
}

static int nb_for(int rows_per_clip, int B, int LD, int budget) {
  int nb = rows_per_clip >= 64 ? 1 : 64 / rows_per_clip;
  if (nb < 1) nb = 1;
  while (nb > 1 && (size_t)nb * rows_per_clip * LD * 4 > (size_t)budget) --nb;
  if (nb > B) nb = B;
  return nb;
}

// workspace layout (bytes): [partials: kMaxGridBwd * Emax floats][red: Emax doubles][coef floats][da partials]
struct BwdWs {
  float* partials;
  double* red;
  float* coef;
  float* dap;
  float* btab;
  float* dz;
};

// fused_bwd.hip
int layer_bwd_below_rows(int T_, int V_, int B, int Ci, int Co, int below_Ci);
int launch_layer_bwd_bpc(const float* in, const float* Zg, const float* dU, const float* coef, const float* in_slope, float* dIn,
                         float* btab, float* partials, float* dap, int B, int Ci, int Co, hipStream_t st, int* rows_out,
                         const float* below_z, const float* below_x, const float* below_slope, int below_Ci, float* below_stats);
bool layer_bwd_fused_ok(int T_, int V_, int Ci, int Co);
// fused_stats.hip
bool bwd_stats_ring_ok(int T_, int V_, int Ci, int Co);
int launch_bwd_stats_ring(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                          int Ci, int Co, hipStream_t st, int* rows_out);
bool bwd_stats_bpc_ok(int T_, int V_, int Ci, int Co);
bool bwd_stats_flat_ok(int TV_, int Ci, int Co);
int launch_bwd_stats_flat(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                          int Ci, int Co, int TV_, hipStream_t st, int* rows_out);
int launch_bwd_stats_bpc(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                         int Ci, int Co, hipStream_t st, int* rows_out);
// bwd_data_bpc.hip
bool bwd_data_bpc_ok(int T_, int V_, int Ci, int Co);
int launch_bwd_data_bpc(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw, const float* coef,
                        const float* in_slope, float* dIn, float* dZout, float* dap, int B, int Ci, int Co, int T_, int V_,
                        hipStream_t st, int* rows_out, float* gpart);
// gcn_params_bpc.hip
bool gcn_params_bpc_ok(int T_, int V_);
int launch_gcn_params_bpc(const float* in, const float* in_slope, const float* dz, const float* Aw, const float* Tw, float* partials,
                          int rows_total, int T_, int V_, hipStream_t st, int* rows_out);
// first_layer.hip
bool first_layer_ok(int T_, int V_, int Ci, int Co);
int launch_first_stats(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B, int Ci,
                       int Co, int TVr, int need_q, int max_rows, hipStream_t st, int* rows_out);
int launch_first_bwd(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw, const float* coef,
                     const float* in_slope, float* partials, int B, int Ci, int Co, int T, int V, int max_rows, hipStream_t st,
                     int* rows_out);
int launch_reduce_fused(const float* partials, int rows, float* dA, float* dT, const float* dap, float* dslope, int accumulate,
                        hipStream_t st, const float* brows, int bE, double* bout);
// chain buffer of a layer's stage-1 sums: [rows][E] partial rows, then (8-byte aligned) their E fp64 sums
static inline size_t chain_sums_offset(int rows, int E) { return ((size_t)rows * E + 1) / 2 * 2; }
constexpr size_t kFusedRowFloats = 32 * 256;   // lane-major partial row of fused_bwd.hip (fb::EROW)

static size_t bwd_emax(int Ci, int Co, int T, int V) {
  const size_t e1 = 2 * (size_t)Co * Ci + Co;
  size_t e2 = (size_t)T * V * V + (size_t)V * T * T;
  if (e2 < kFusedRowFloats) e2 = kFusedRowFloats;
  return e1 > e2 ? e1 : e2;
}

size_t layer_bwd_ws_bytes(int B, int Ci, int Co, int T, int V) {
  const size_t E = bwd_emax(Ci, Co, T, V);
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  return al(kMaxGridBwd * E * sizeof(float)) + al(E * sizeof(double)) +
         al(coef_floats(Ci, Co) * sizeof(float)) + al((size_t)((B > kMaxGridBwd ? B : kMaxGridBwd) + 1) * sizeof(float)) +
         al((size_t)kBtabFloats * sizeof(float)) + al((size_t)B * Ci * T * V * sizeof(float));
}

static BwdWs carve(void* ws, int B, int Ci, int Co, int T, int V) {
  const size_t E = bwd_emax(Ci, Co, T, V);
  char* p = reinterpret_cast<char*>(ws);
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  BwdWs w;
  w.partials = reinterpret_cast<float*>(p); p += al(kMaxGridBwd * E * sizeof(float));
  w.red = reinterpret_cast<double*>(p);     p += al(E * sizeof(double));
  w.coef = reinterpret_cast<float*>(p);     p += al(coef_floats(Ci, Co) * sizeof(float));
  w.dap = reinterpret_cast<float*>(p);      p += al((size_t)((B > kMaxGridBwd ? B : kMaxGridBwd) + 1) * sizeof(float));
  w.btab = reinterpret_cast<float*>(p);     p += al((size_t)kBtabFloats * sizeof(float));
  w.dz = reinterpret_cast<float*>(p);
  return w;
}

// Stage 4 of the layer backward: dA, dT from the layer input (pre-activation + producer slope) and the stored dZ.
// partials: >= min(512, tiles) * (T*V*V + V*T*T) floats.
template <int T, int V>
static int launch_layer_gcn_params(const float* in, const float* in_slope, const float* dz, const float* Aw,
                                   const float* Tw, float* dA, float* dT, float* partials, int accumulate, int B,
                                   int Ci, int Co_tag, hipStream_t st, const float* dap = nullptr, int ndap = 0,
                                   float* dslope = nullptr) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = Geo<T, V>::LD;
  const int E = T * V * V + V * T * T;
  // Both products sum over rows = (clip, channel) and the mixing is per row, so a tile is ANY run of consecutive rows
  // of the [B*C_in, T*V] matrix: 16-row tiles (one MFMA row tile) keep two images + tables under a third of the LDS
  // -> three blocks per CU.  The kernel is told "C_in = 1, B = rows".
  const int rows_total = B * Ci;
  if (gcn_params_bpc_ok(T, V)) {
    // the 25-joint layout: one 32-row tile per four-wave workgroup, the tables as operands from L2 (gcn_params_bpc.hip)
    int rows_p = 0, rc_;
    {
      ProbeScope probe(KID_GCN_PARAMS, Ci, Co_tag, st);
      if ((rc_ = launch_gcn_params_bpc(in, in_slope, dz, Aw, Tw, partials, rows_total, T, V, st, &rows_p))) return rc_;
    }
    hipLaunchKernelGGL(k_reduce_gcn, dim3(ceil_div(E, kGcnCols) + (dap ? 1 : 0)), dim3(1024), 0, st, partials, rows_p, T * V * V,
                       V * T * T, dA, dT, dap, ndap, dslope, accumulate);
    return check_launch("bwd_gcn_reduce");
  }
  const int rt = kBlock > 512 ? 32 : 16;       // (16-wave blocks, one per CU: twice the rows per barrier round; 3.20 -> 3.16 ms on the 25-joint step)
  const int RTILE = rows_total < rt ? rows_total : rt;
  const size_t lds = ((size_t)2 * RTILE * LD + (size_t)T * V * V + (size_t)V * T * T) * sizeof(float);
  if (lds > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "layer_bwd: LDS %zu too large", lds);
  const int ntiles = ceil_div(rows_total, RTILE);
  const int per_cu = (lds <= (size_t)52 * 1024 && V <= 17) ? 3 : (lds <= (size_t)80 * 1024 ? 2 : 1);
  const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
  const int NB = RTILE;
  auto k = k_bwd_gcn_params<T, V>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  {
    ProbeScope probe(KID_GCN_PARAMS, Ci, Co_tag, st);
#ifdef COSKAD_ABLATE
    static int ablg = -1;
    if (ablg < 0) { const char* e = getenv("COSKAD_ABLG"); ablg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, dz, Aw, Tw, in_slope, partials, rows_total, 1, NB, (float*)nullptr,
                       (const float*)nullptr, ablg);
#else
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, dz, Aw, Tw, in_slope, partials, rows_total, 1, NB, (float*)nullptr,
                       (const float*)nullptr);
#endif
  }
  int rc;
  if ((rc = check_launch("bwd_gcn_params"))) return rc;
  hipLaunchKernelGGL(k_reduce_gcn, dim3(ceil_div(E, kGcnCols) + (dap ? 1 : 0)), dim3(1024), 0, st, partials, grid, T * V * V,
                     V * T * T, dA, dT, dap, ndap, dslope, accumulate);
  return check_launch("bwd_gcn_reduce");
}

template <int T, int V>
static int launch_layer_bwd(const float* in, const float* dU, const float* Aw, const float* Tw,
                            const float* in_slope, const float* stat, const float* Wt, const float* gs,
                            const float* Wr, const float* gr, float* dIn, float* dA, float* dT, float* dWt,
                            float* dbt, float* dgs, float* dbs, float* dWr, float* dbr, float* dgr,
                            float* dbr2, float* dslope_in, void* ws, size_t ws_bytes, int accumulate,
                            int B, int Ci, int Co, hipStream_t st, float* dz_ext = nullptr,
                            const float* Zg = nullptr, const float* stats_in = nullptr, int stats_in_rows = 0,
                            const float* below_in = nullptr, const float* below_z = nullptr, const float* below_slope = nullptr,
                            int below_Ci = 0, float* below_stats = nullptr, double stats_count = 0.0, float* stats_out = nullptr,
                            int* stats_rows_out = nullptr) {
  // stats_count: positions the stage-1 sums cover (0: this batch, B T V; SyncBN: all ranks' batches);
  // stats_out: run stage 1 ONLY, into that chain buffer (partial rows, then their fp64 sums); *stats_rows_out = rows written
  // stats_in: this layer's stage-1 partial rows, written by the call for the layer above (stage 1 is then skipped);
  // below_*: the layer below's input / stored Z and the buffer its partial rows go to (fused data kernel only)
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int kScratchFloats = Geo<T, V>::Scratch;
  // dz_ext != NULL: dZ goes to the caller's buffer and stage 4 (dA, dT) is left to coskad_layer_gcn_params_f32
  constexpr int LD = Geo<T, V>::LD, TV = Geo<T, V>::TV;
  if (Ci > 64 || Co > 64) return fail(COSKAD_ERR_SHAPE, "layer_bwd: channels (%d,%d) > 64 not supported", Ci, Co);
  if (ws_bytes < layer_bwd_ws_bytes(B, Ci, Co, T, V))
    return fail(COSKAD_ERR_WORKSPACE, "layer_bwd: workspace %zu < %zu bytes", ws_bytes, layer_bwd_ws_bytes(B, Ci, Co, T, V));
  BwdWs w = carve(ws, B, Ci, Co, T, V);
  if (dz_ext) w.dz = dz_ext;
  if (stats_out) w.partials = stats_out;
  auto red_of = [&](int rows_, int E_) { return stats_out ? reinterpret_cast<double*>(stats_out + chain_sums_offset(rows_, E_)) : w.red; };
  int rc;
  // 1. reductions
  {
    const int E = 2 * Co * Ci + Co;
    int NB = Ci >= 32 ? 1 : 32 / Ci;
    if (NB > B) NB = B;
    auto red_lds = [&](int nb_) {
      size_t img = (size_t)nb_ * Ci * LD + (size_t)nb_ * Co * RedGeo<T, V>::LDC;
      if (img < (size_t)kScratchFloats) img = kScratchFloats;
      return (img + (size_t)T * V * V + (size_t)V * T * T) * sizeof(float);
    };
    const int nto = ceil_div(Co, 16), ntc = ceil_div(Ci, 16);
    const int need_q = Wr != nullptr;
    auto zlds = [&](int nb_) {
      size_t img = (size_t)(need_q ? 2 : 1) * nb_ * Ci * RedGeo<T, V>::LDZ + (size_t)nb_ * Co * RedGeo<T, V>::LDCZ;
      if (img < (size_t)kScratchFloats) img = kScratchFloats;
      return img * sizeof(float);
    };
    // stored Z: one dU pass with X and Z both resident, no mixing tables in LDS.  When one clip's X + Z images exceed
    // the LDS (64 input channels at 25 joints: the default-width decoder on the NTU layout) the two-pass kernel below
    // runs instead; it takes the stored Z as well.
    if (stats_in) {
      // summed already: the partial rows rode in the reduction launch of the layer above (k_reduce_fused)
    } else if (Zg && first_layer_ok(T, V, Ci, Co)) {
      // a handful of input channels (the first layer): plain FMAs on full-line loads (first_layer.hip)
      int rows = 0;
      if ((rc = launch_first_stats(in, Zg, dU, in_slope, w.partials, B, Ci, Co, TV, need_q, kMaxGridBwd, st, &rows))) return rc;
      hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, rows, E, red_of(rows, E));
      if ((rc = check_launch("bwd_reduce_partials"))) return rc;
      if (stats_rows_out) *stats_rows_out = rows;
    } else if (Zg && bwd_stats_bpc_ok(T, V, Ci, Co)) {
      // default geometry, 32 input channels and a wide output (the top layer): one clip per workgroup (fused_stats.hip)
      int rows = 0;
      if ((rc = launch_bwd_stats_bpc(in, Zg, dU, in_slope, w.partials, B, Ci, Co, st, &rows))) return rc;
      hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, rows, E, red_of(rows, E));
      if ((rc = check_launch("bwd_reduce_partials"))) return rc;
      if (stats_rows_out) *stats_rows_out = rows;
    } else if (Zg && bwd_stats_flat_ok(T * V, Ci, Co)) {
      // the 25-joint layout: the same scheme over flat positions (fused_stats.hip)
      int rows = 0;
      if ((rc = launch_bwd_stats_flat(in, Zg, dU, in_slope, w.partials, B, Ci, Co, T * V, st, &rows))) return rc;
      hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, rows, E, red_of(rows, E));
      if ((rc = check_launch("bwd_reduce_partials"))) return rc;
      if (stats_rows_out) *stats_rows_out = rows;
    } else if (Zg && bwd_stats_ring_ok(T, V, Ci, Co)) {
      // default geometry, 16 / 32 input channels: wave-per-clip reductions (fused_stats.hip)
      int rows = 0;
      if ((rc = launch_bwd_stats_ring(in, Zg, dU, in_slope, w.partials, B, Ci, Co, st, &rows))) return rc;
      hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, rows, E, red_of(rows, E));
      if ((rc = check_launch("bwd_reduce_partials"))) return rc;
      if (stats_rows_out) *stats_rows_out = rows;
    } else if (Zg && zlds(1) <= (size_t)kMaxLdsBytes) {
      const bool three_z = nto * ntc <= 2 && kBlock <= 512;   // (16-wave blocks: at most two per CU)
      const size_t cap = three_z ? (size_t)52 * 1024 : (size_t)76 * 1024;
      int NBz = NB;
      while (NBz > 1 && zlds(NBz) > cap) --NBz;
      const size_t ldsz = zlds(NBz);
      if (ldsz > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "layer_bwd: LDS %zu too large", ldsz);
      const int ntl = ceil_div(B, NBz);
      int gridz = 256 * ((three_z && ldsz <= cap) ? 3 : (ldsz <= (size_t)80 * 1024 ? 2 : 1));
      if (gridz > ntl) gridz = ntl;
#define LAUNCH_RZ(NTO, NTC)                                                                              \
  do {                                                                                                  \
    auto k = k_bwd_reduce_z<T, V, NTO, NTC>;                                                            \
    if (ldsz > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz); \
    hipLaunchKernelGGL(k, dim3(gridz), dim3(kBlock), ldsz, st, in, Zg, dU, in_slope, w.partials, B, Ci, Co, NBz, need_q); \
  } while (0)
#define LAUNCH_RZ_O(NTO)                                         \
  do {                                                           \
    if (ntc == 1) LAUNCH_RZ(NTO, 1);                             \
    else if (ntc == 2) LAUNCH_RZ(NTO, 2);                        \
    else if (ntc == 3) LAUNCH_RZ(NTO, 3);                        \
    else LAUNCH_RZ(NTO, 4);                                      \
  } while (0)
      {
        ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
        if (nto == 1) LAUNCH_RZ_O(1);
        else if (nto == 2) LAUNCH_RZ_O(2);
        else if (nto == 3) LAUNCH_RZ_O(3);
        else LAUNCH_RZ_O(4);
      }
#undef LAUNCH_RZ_O
#undef LAUNCH_RZ
      if ((rc = check_launch("bwd_reduce_z"))) return rc;
      hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, gridz, E, red_of(gridz, E));
      if ((rc = check_launch("bwd_reduce_partials"))) return rc;
      if (stats_rows_out) *stats_rows_out = gridz;
    } else {
    // blocks per CU: three when the accumulators are small enough for 6 waves/SIMD and the images fit a third of
    // the LDS (fewer clips per tile if need be), else two
    const bool three = nto * ntc <= 2 && kBlock <= 512;
    const size_t lds_cap = three ? (size_t)52 * 1024 : (size_t)76 * 1024;   // (LDS is allocated in coarse granules)
    while (NB > 1 && red_lds(NB) > lds_cap) --NB;
    const size_t lds = red_lds(NB);
    if (lds > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "layer_bwd: LDS %zu too large", lds);
    const int ntiles = ceil_div(B, NB);
    int grid = 256 * ((three && lds <= lds_cap) ? 3 : (lds <= (size_t)80 * 1024 ? 2 : 1));
    if (grid > ntiles) grid = ntiles;
#define LAUNCH_R(NTO, NTC)                                                                              \
  do {                                                                                                  \
    auto k = k_bwd_reduce<T, V, NTO, NTC>;                                                              \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, dU, Aw, Tw, in_slope, w.partials, B,    \
                       Ci, Co, NB, need_q, Zg);                                                         \
  } while (0)
#define LAUNCH_R_O(NTO)                                          \
  do {                                                           \
    if (ntc == 1) LAUNCH_R(NTO, 1);                              \
    else if (ntc == 2) LAUNCH_R(NTO, 2);                         \
    else if (ntc == 3) LAUNCH_R(NTO, 3);                         \
    else LAUNCH_R(NTO, 4);                                       \
  } while (0)
    {
    ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
    if (nto == 1) LAUNCH_R_O(1);
    else if (nto == 2) LAUNCH_R_O(2);
    else if (nto == 3) LAUNCH_R_O(3);
    else LAUNCH_R_O(4);
    }
#undef LAUNCH_R_O
#undef LAUNCH_R
    if ((rc = check_launch("bwd_reduce"))) return rc;
    hipLaunchKernelGGL(k_reduce_partials_d, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, w.partials, grid, E, red_of(grid, E));
    if ((rc = check_launch("bwd_reduce_partials"))) return rc;
    if (stats_rows_out) *stats_rows_out = grid;
    }
  }
  if (stats_out) return 0;
  // 2. fold
  const size_t fold_lds = (size_t)(6 * Co + 2 * Co * Ci + Co) * sizeof(double) + (size_t)(4 * Co * Ci + 2 * Ci + 4 * Co) * sizeof(float);
  if (fold_lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_bwd_fold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_lds);
  const bool fused = Zg && dIn && in_slope && !dz_ext && layer_bwd_fused_ok(T, V, Ci, Co);
  if (below_stats && !fused) return fail(COSKAD_ERR_SHAPE, "layer_bwd_chain: the layer below's reductions need the fused data kernel");
  // the fused data kernel's operand tables are built by extra blocks of the fold launch (parameter-only work)
  const int tab_blocks = fused ? ceil_div(ff::BTAB_F4 * 4, 1024) : 0;
  const int NF = Co >= 32 ? 8 : (Co >= 16 ? 4 : 1);          // fold blocks (slices of the output channels / K pairs)
  const double* red = stats_in ? reinterpret_cast<const double*>(stats_in + chain_sums_offset(stats_in_rows, 2 * Co * Ci + Co)) : w.red;
  hipLaunchKernelGGL(k_bwd_fold, dim3(NF + tab_blocks), dim3(1024), fold_lds, st, red, stats_count > 0.0 ? stats_count : (double)B * TV, stat,
                     Wt, gs, Wr, gr, dWt, dbt, dgs, dbs, dWr, dbr, dgr, dbr2, w.coef, Ci, Co, accumulate, Aw, Tw, w.btab, NF);
  if ((rc = check_launch("bwd_fold"))) return rc;
  // 3 + 4 in one kernel (fused_bwd.hip) for the stored-Z path at the shapes it is built for: dZ never leaves the CU
  if (fused) {
    int rows = 0;
    float* dap = (dslope_in && in_slope) ? w.dap : nullptr;
    if ((rc = launch_layer_bwd_bpc(in, Zg, dU, w.coef, in_slope, dIn, w.btab, w.partials, dap, B, Ci, Co, st, &rows, below_z, below_in,
                                   below_slope, below_Ci, below_stats)))
      return rc;
    const int bE = 2 * Ci * below_Ci + Ci;
    return launch_reduce_fused(w.partials, rows, dA, dT, dap, dslope_in, accumulate, st, below_stats, bE,
                               below_stats ? reinterpret_cast<double*>(below_stats + chain_sums_offset(rows, bE)) : nullptr);
  }
  if (Zg && !dIn && !dz_ext && first_layer_ok(T, V, Ci, Co)) {
    int rows = 0;
    if ((rc = launch_first_bwd(in, Zg, dU, Aw, Tw, w.coef, in_slope, w.partials, B, Ci, Co, T, V, kMaxGridBwd, st, &rows))) return rc;
    hipLaunchKernelGGL(k_reduce_gcn, dim3(ceil_div(T * V * V + V * T * T, kGcnCols)), dim3(1024), 0, st, w.partials, rows, T * V * V,
                       V * T * T, dA, dT, (const float*)nullptr, 0, (float*)nullptr, accumulate);
    return check_launch("bwd_gcn_reduce");
  }
  // 3. data path
  int grid_d;
  const float* dap_sum = nullptr;   // block partials of the producer's slope gradient (summed by stage 4's reduce)
  if (Zg && dIn && !dz_ext && bwd_data_bpc_ok(T, V, Ci, Co)) {
    // the 25-joint layout, stored Z: one clip per four-wave workgroup (bwd_data_bpc.hip)
    // dA / dT in the same kernel (dZ never leaves the CU): 16 -> 32 226 -> 207 us, 32 -> 64 471 -> 457 us at B = 4096 (32 channels: the dT
    // sums in LDS, the next clip's first group no longer carried -- with all 76 sum registers beside the two K-pass accumulator sets the
    // kernel spilled 116 B per lane and ran 480 us).  COSKAD_V25_SPLIT=1: the two-kernel form with the dZ round trip (A/B)
    static const int force = [] { const char* e = getenv("COSKAD_V25_SPLIT"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    const bool split = force == 1;
    float* dap = (dslope_in && in_slope) ? w.dap : nullptr;
    {
      ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
      if ((rc = launch_bwd_data_bpc(in, Zg, dU, Aw, Tw, w.coef, in_slope, dIn, w.dz, dap, B, Ci, Co, T, V, st, &grid_d,
                                    split ? nullptr : w.partials)))
        return rc;
    }
    if (!split) {
      hipLaunchKernelGGL(k_reduce_gcn, dim3(ceil_div(T * V * V + V * T * T, kGcnCols) + (dap ? 1 : 0)), dim3(1024), 0, st, w.partials, grid_d,
                         T * V * V, V * T * T, dA, dT, dap, grid_d, dslope_in, accumulate);
      return check_launch("bwd_gcn_reduce");
    }
    dap_sum = dap;
  } else {
    int NB = Ci >= 32 ? 1 : 32 / Ci;
    if (NB > B && !(Ci == 16 && dIn != nullptr)) NB = B;   // (the two-clip single-read kernel keeps its 32-row image)
    const int CiP = round_up(Ci, 16), KZ = round_up(Ci, 4), K1 = round_up(Co, 4);
    const size_t lds = ((size_t)NB * Ci * LD + (size_t)T * V * V + (size_t)V * T * T + 2 * (size_t)(KZ + K1) * CiP +
                        2 * CiP) * sizeof(float);
    if (lds > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "layer_bwd: LDS %zu too large", lds);
    const int ntl = ceil_div(B, NB);
    const int per_cu = (int)((size_t)kMaxLdsBytes / lds);
    grid_d = 256 * (per_cu < 1 ? 1 : (per_cu > 2048 / kBlock ? 2048 / kBlock : per_cu));   // (a CU holds 32 waves)
    if (grid_d > ntl) grid_d = ntl;
    float* dap = (dslope_in && in_slope) ? w.dap : nullptr;
#define LAUNCH_D(OTI)                                                                                   \
  do {                                                                                                  \
    auto k = k_bwd_data<T, V, OTI>;                                                                     \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid_d), dim3(kBlock), lds, st, in, dU, Aw, Tw, w.coef, in_slope, dIn,    \
                       w.dz, dap, B, Ci, Co, NB, Zg);                                                   \
  } while (0)
#ifdef COSKAD_ABLATE
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("COSKAD_ABL"); abl = e ? atoi(e) : 0; }
#define ABL_ARG , abl
#else
#define ABL_ARG
#endif
#define LAUNCH_DF(OTI, NBF_)                                                                                  \
  do {                                                                                                  \
    auto k = k_bwd_data_f<T, V, OTI, NBF_>;                                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid_d), dim3(kBlock), lds, st, in, dU, Aw, Tw, w.coef, in_slope, dIn,    \
                       w.dz, dap, B, Ci, Co, Zg ABL_ARG);                                               \
  } while (0)
#ifdef COSKAD_ABLATE
    static int fused_ok = -1;
    if (fused_ok < 0) { const char* e = getenv("COSKAD_BWD_UNFUSED"); fused_ok = (e && e[0] == '1') ? 0 : 1; }
#else
    constexpr int fused_ok = 1;
#endif
    constexpr bool strips_fit = (Geo<T, V>::TV + 31) / 32 <= kBlock / 64;
    // single-read variant: tiles of 32 rows = one clip of >= 32 channels, or two clips of exactly 16
    const bool two_clip = Ci == 16;            // NB was kept at 2 above
    if (fused_ok && strips_fit && (NB == 1 || two_clip) && dIn != nullptr && CiP <= 64) {
      ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
      if constexpr (strips_fit) {
        if (two_clip) LAUNCH_DF(1, 2);
        else if (CiP == 16) LAUNCH_DF(1, 1);   // (one clip of < 16 channels: only when the batch is a single clip)
        else if (CiP == 32) LAUNCH_DF(2, 1);
        else if (CiP == 48) LAUNCH_DF(3, 1);
        else LAUNCH_DF(4, 1);
      }
    } else
    {
    ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
    if (CiP == 16) LAUNCH_D(1);
    else if (CiP == 32) LAUNCH_D(2);
    else if (CiP == 48) LAUNCH_D(3);
    else LAUNCH_D(4);
    }
#undef LAUNCH_D
    if ((rc = check_launch("bwd_data"))) return rc;
    if (dap && dz_ext) {   // split call: stage 4 is not ours, finish the slope gradient here
      hipLaunchKernelGGL(k_sum_to, dim3(1), dim3(256), 0, st, dap, grid_d, dslope_in, accumulate);
      if ((rc = check_launch("bwd_dslope"))) return rc;
    }
    dap_sum = dap;
  }
  // 4. gcn parameter gradients
  if (!dz_ext)
    return launch_layer_gcn_params<T, V>(in, in_slope, w.dz, Aw, Tw, dA, dT, w.partials, accumulate, B, Ci, Co, st,
                                         dap_sum, grid_d, dslope_in);
  return COSKAD_OK;
}

}  // namespace coskad

namespace coskad {
template <int T, int V>
static int launch_gcn_bwd_params(const float* x, const float* dZ, const float* Aw, const float* Tw, float* dA,
                                 float* dT, void* ws, int accumulate, int rows, hipStream_t st, float* dX = nullptr,
                                 const float* dX_add = nullptr) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = Geo<T, V>::LD;
  const int E = T * V * V + V * T * T;
  const int NB = rows < 32 ? rows : 32;
  const size_t lds = ((size_t)2 * NB * LD + (size_t)T * V * V + (size_t)V * T * T) * sizeof(float);
  const int ntiles = ceil_div(rows, NB);
  const int grid = ntiles < 512 ? ntiles : 512;
  float* partials = reinterpret_cast<float*>(ws);
  auto k = k_bwd_gcn_params<T, V>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef COSKAD_ABLATE
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, x, dZ, Aw, Tw, (const float*)nullptr, partials, rows, 1, NB, dX, dX_add, 0);
#else
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, x, dZ, Aw, Tw, (const float*)nullptr, partials, rows, 1, NB, dX, dX_add);
#endif
  hipLaunchKernelGGL(k_reduce_to_f32, dim3(ceil_div(T * V * V, 64)), dim3(1024), 0, st, partials, grid, E, 0, T * V * V, dA, accumulate);
  hipLaunchKernelGGL(k_reduce_to_f32, dim3(ceil_div(V * T * T, 64)), dim3(1024), 0, st, partials, grid, E, T * V * V, V * T * T, dT, accumulate);
  return check_launch("gcn_bwd_params");
}
}  // namespace coskad

using namespace coskad;

extern "C" {

size_t coskad_layer_bwd_ws_bytes(int B, int Ci, int Co, int T, int V) { return layer_bwd_ws_bytes(B, Ci, Co, T, V); }

/* 1 when one clip of a (Ci -> Co) layer fits the LDS-resident tile kernels of this library (forward, statistics,
 * backward), 0 when it does not (more than 64 channels; 64 input channels on the 25-joint layout): the module mirror
 * then runs that layer through its composed path (mixing kernels + library GEMMs). */
int coskad_layer_fits(int Ci, int Co, int T, int V) {
  if (Ci <= 0 || Co <= 0 || Ci > 64 || Co > 64) return 0;
  const int TV = T * V, LD = TV % 2 == 0 ? TV + 1 : TV;
  const int NB = Ci >= 32 ? 1 : 32 / Ci;
  const int CiP = round_up(Ci, 16), KZ = round_up(Ci, 4), K1 = round_up(Co, 4);
  const size_t tables = (size_t)T * V * V + (size_t)V * T * T;
  // the backward data path holds the input image, both mixing tables and the four coefficient matrices
  const size_t data = ((size_t)NB * Ci * LD + tables + 2 * (size_t)(KZ + K1) * CiP + 2 * CiP) * sizeof(float);
  // the two-pass batch reduction: input image, a dU chunk image, both tables
  const int CH = ((TV + 2) / 3 + 3) / 4 * 4;
  const size_t red = ((size_t)Ci * LD + (size_t)Co * (CH + 1) + tables) * sizeof(float);
  const size_t fwd = ((size_t)NB * Ci * LD + tables + 2 * (size_t)KZ * round_up(Co, 16) + round_up(Co, 16)) * sizeof(float);
  const size_t cap = (size_t)kMaxLdsBytes;
  return data <= cap && red <= cap && fwd <= cap;
}

size_t coskad_gcn_bwd_params_ws_bytes(int T, int V) {
  return (size_t)kMaxGridBwd * ((size_t)T * V * V + (size_t)V * T * T) * sizeof(float);
}


/* Parameter gradients of ConvTemporalGraphical (stsgcn.py:154-155): dA[t,v,w] = sum Y[.,t,v] dZ[.,t,w],
 * dT[v,t,q] = sum X[.,t,v] dY[.,q,v] over rows = N*C, given X and dZ. */
int coskad_gcn_bwd_params_f32(const float* x, const float* dZ, const float* A, const float* Tm, float* dA,
                              float* dT, void* ws, size_t ws_bytes, int accumulate, int rows, int T, int V,
                              hipStream_t stream) {
  if (!x || !dZ || !A || !Tm || !dA || !dT || !ws) return fail(COSKAD_ERR_ARG, "gcn_bwd_params: null pointer");
  if (rows <= 0) return fail(COSKAD_ERR_ARG, "gcn_bwd_params: rows=%d", rows);
  if (ws_bytes < coskad_gcn_bwd_params_ws_bytes(T, V)) return fail(COSKAD_ERR_WORKSPACE, "gcn_bwd_params: workspace too small");
#define CALL(T_, V_) return launch_gcn_bwd_params<T_, V_>(x, dZ, A, Tm, dA, dT, ws, accumulate, rows, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* coskad_gcn_bwd_params_f32 AND the input gradient of ConvTemporalGraphical in one pass over dZ:
 * dX = gcn^T(dZ) (+ dX_add, optional: same shape -- the gradient an identity residual carries beside the mix), what
 * coskad_gcn_f32(adjoint) would compute from a second read of dZ.  The wide layers' backward (stsgcn.py:94-116 under autograd). */
int coskad_gcn_bwd_params_dx_f32(const float* x, const float* dZ, const float* A, const float* Tm, float* dA, float* dT,
                                 float* dX, const float* dX_add, void* ws, size_t ws_bytes, int accumulate, int rows, int T,
                                 int V, hipStream_t stream) {
  if (!x || !dZ || !A || !Tm || !dA || !dT || !dX || !ws) return fail(COSKAD_ERR_ARG, "gcn_bwd_params_dx: null pointer");
  if (rows <= 0) return fail(COSKAD_ERR_ARG, "gcn_bwd_params_dx: rows=%d", rows);
  if (ws_bytes < coskad_gcn_bwd_params_ws_bytes(T, V)) return fail(COSKAD_ERR_WORKSPACE, "gcn_bwd_params_dx: workspace too small");
#define CALL(T_, V_) return launch_gcn_bwd_params<T_, V_>(x, dZ, A, Tm, dA, dT, ws, accumulate, rows, stream, dX, dX_add)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

int coskad_layer_bwd_f32(const float* in, const float* dU, const float* A, const float* Tm,
                         const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                         const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                         float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                         float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                         int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream) {
  if (!in || !dU || !A || !Tm || !stat || !Wt || !gamma_t || !dA || !dT || !dWt || !dgamma_t || !dbeta_t || !ws)
    return fail(COSKAD_ERR_ARG, "layer_bwd: null pointer");
  if (Wr && (!gamma_r || !dWr || !dgamma_r || !dbeta_r)) return fail(COSKAD_ERR_ARG, "layer_bwd: residual grads missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_bwd: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd: B=%d Ci=%d Co=%d", B, Ci, Co);
  ProbeScope layer_probe(KID_LAYER_BWD, Ci, Co, stream);   // bench.py: the whole layer backward (all its launches)
#define CALL(T_, V_)                                                                                       \
  return launch_layer_bwd<T_, V_>(in, dU, A, Tm, in_slope, stat, Wt, gamma_t, Wr, gamma_r, dIn, dA, dT, dWt, \
                                  dbt, dgamma_t, dbeta_t, dWr, dbr, dgamma_r, dbeta_r, dslope_in, ws,      \
                                  ws_bytes, accumulate, B, Ci, Co, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* The same with Z = gcn(PReLU(in)) as stored by coskad_layer_train_stats_z_f32 (NULL: recompute it). */
int coskad_layer_bwd_z_f32(const float* in, const float* dU, const float* A, const float* Tm,
                         const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                         const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                         float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                         float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                         int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z) {
  if (!in || !dU || !A || !Tm || !stat || !Wt || !gamma_t || !dA || !dT || !dWt || !dgamma_t || !dbeta_t || !ws)
    return fail(COSKAD_ERR_ARG, "layer_bwd: null pointer");
  if (Wr && (!gamma_r || !dWr || !dgamma_r || !dbeta_r)) return fail(COSKAD_ERR_ARG, "layer_bwd: residual grads missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_bwd: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd: B=%d Ci=%d Co=%d", B, Ci, Co);
  ProbeScope layer_probe(KID_LAYER_BWD, Ci, Co, stream);   // bench.py: the whole layer backward (all its launches)
#define CALL(T_, V_)                                                                                       \
  return launch_layer_bwd<T_, V_>(in, dU, A, Tm, in_slope, stat, Wt, gamma_t, Wr, gamma_r, dIn, dA, dT, dWt, \
                                  dbt, dgamma_t, dbeta_t, dWr, dbr, dgamma_r, dbeta_r, dslope_in, ws,      \
                                  ws_bytes, accumulate, B, Ci, Co, stream, nullptr, Z)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* coskad_layer_bwd_z_f32 inside a chain of layers.  The batch reductions of a layer's backward (stage 1: P = sum dU.Z^T,
 * Q = sum dU.X^T, sdU) read the dU the layer ABOVE has just produced; where that layer's data kernel can, it forms them itself:
 *   stats_in [stats_in_rows][2 Co Ci + Co] + sums : this layer's chain buffer from the call for the layer above (NULL: stage 1 runs here)
 *   below_in / below_Z [B, below_Ci, T, V], below_in_slope, below_stats [coskad_layer_bwd_below_rows(...)][2 Ci below_Ci + Ci] :
 *     the layer below's input as stored (pre-activation + its producer's PReLU weight, NULL for the raw network input), its
 *     stored Z and ITS chain buffer (NULL: not formed): the partial rows, then (8-byte aligned) their fp64 sums, which this
 *     call's partial-sum launch forms as well: coskad_layer_bwd_below_floats(...) floats in all */
int coskad_layer_bwd_below_rows(int B, int Ci, int Co, int below_Ci, int T, int V) {
  return layer_bwd_below_rows(T, V, B, Ci, Co, below_Ci);
}
size_t coskad_layer_bwd_below_floats(int B, int Ci, int Co, int below_Ci, int T, int V) {
  const int rows = layer_bwd_below_rows(T, V, B, Ci, Co, below_Ci), bE = 2 * Ci * below_Ci + Ci;
  return rows ? chain_sums_offset(rows, bE) + 2 * (size_t)bE : 0;
}

int coskad_layer_bwd_chain_f32(const float* in, const float* dU, const float* A, const float* Tm,
                               const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                               const float* Wr, const float* gamma_r, float* dIn, float* dA, float* dT, float* dWt,
                               float* dbt, float* dgamma_t, float* dbeta_t, float* dWr, float* dbr,
                               float* dgamma_r, float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes,
                               int accumulate, int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z,
                               const float* stats_in, int stats_in_rows, size_t stats_in_bytes, const float* below_in,
                               const float* below_Z, const float* below_in_slope, int below_Ci, float* below_stats,
                               size_t below_stats_bytes, double stats_count) {
  if (!in || !dU || !A || !Tm || !stat || !Wt || !gamma_t || !dA || !dT || !dWt || !dgamma_t || !dbeta_t || !ws || !Z)
    return fail(COSKAD_ERR_ARG, "layer_bwd_chain: null pointer");
  if (Wr && (!gamma_r || !dWr || !dgamma_r || !dbeta_r)) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: residual grads missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: B=%d Ci=%d Co=%d", B, Ci, Co);
  if (stats_in && stats_in_rows <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: stats_in_rows=%d", stats_in_rows);
  if (stats_in && ((size_t)stats_in & 7)) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: stats_in must be 8-byte aligned");
  if (stats_in) {   // the fold reads the fp64 sums behind the partial rows: a short buffer must not become an out-of-bounds read
    const size_t E = 2 * (size_t)Co * Ci + Co;
    const size_t need = chain_sums_offset(stats_in_rows, (int)E) * sizeof(float) + E * sizeof(double);
    if (stats_in_bytes < need) return fail(COSKAD_ERR_WORKSPACE, "layer_bwd_chain: stats_in %zu < %zu bytes", stats_in_bytes, need);
  }
  if (below_stats && ((size_t)below_stats & 7)) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: below_stats must be 8-byte aligned");
  if (below_stats) {
    if (!below_in || !below_Z) return fail(COSKAD_ERR_ARG, "layer_bwd_chain: below_in / below_Z missing");
    const int rows = layer_bwd_below_rows(T, V, B, Ci, Co, below_Ci);
    if (rows == 0) return fail(COSKAD_ERR_SHAPE, "layer_bwd_chain: (%d -> %d) cannot form the reductions of a layer with %d input channels", Ci, Co, below_Ci);
    const int bE = 2 * Ci * below_Ci + Ci;
    const size_t need = chain_sums_offset(rows, bE) * sizeof(float) + (size_t)bE * sizeof(double);
    if (below_stats_bytes < need) return fail(COSKAD_ERR_WORKSPACE, "layer_bwd_chain: below_stats %zu < %zu bytes", below_stats_bytes, need);
  }
  ProbeScope layer_probe(KID_LAYER_BWD, Ci, Co, stream);
#define CALL(T_, V_)                                                                                       \
  return launch_layer_bwd<T_, V_>(in, dU, A, Tm, in_slope, stat, Wt, gamma_t, Wr, gamma_r, dIn, dA, dT, dWt, \
                                  dbt, dgamma_t, dbeta_t, dWr, dbr, dgamma_r, dbeta_r, dslope_in, ws,      \
                                  ws_bytes, accumulate, B, Ci, Co, stream, nullptr, Z, stats_in, stats_in_rows, \
                                  below_stats ? below_in : nullptr, below_stats ? below_Z : nullptr,                  \
                                  below_stats ? below_in_slope : nullptr, below_Ci, below_stats, stats_count)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* Stage 1 of coskad_layer_bwd_z_f32 ALONE, into a chain buffer (SyncBN: the caller adds the other ranks' fp64 sums in place, then
 * passes the buffer as `stats_in` with `stats_count` = global clips x T x V).  stats_out: coskad_layer_bwd_stats_floats(...) floats,
 * 8-byte aligned; *rows_out = partial rows written (the sums sit behind them: coskad_layer_bwd_sums_offset(rows, Ci, Co)). */
size_t coskad_layer_bwd_stats_floats(int B, int Ci, int Co, int T, int V) {
  (void)B; (void)T; (void)V;
  const size_t E = 2 * (size_t)Co * Ci + Co;
  return ((size_t)kMaxGridBwd * E + 1) / 2 * 2 + 2 * E;
}
size_t coskad_layer_bwd_sums_offset(int rows, int Ci, int Co) { return chain_sums_offset(rows, 2 * Co * Ci + Co); }

int coskad_layer_bwd_stats_f32(const float* in, const float* dU, const float* A, const float* Tm, const float* in_slope,
                               int has_residual, float* stats_out, size_t stats_out_bytes, int* rows_out, void* ws, size_t ws_bytes,
                               int B, int Ci, int Co, int T, int V, hipStream_t stream, const float* Z) {
  if (!in || !dU || !A || !Tm || !stats_out || !rows_out || !ws) return fail(COSKAD_ERR_ARG, "layer_bwd_stats: null pointer");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd_stats: B=%d Ci=%d Co=%d", B, Ci, Co);
  if ((size_t)stats_out & 7) return fail(COSKAD_ERR_ARG, "layer_bwd_stats: stats_out must be 8-byte aligned");
  if (stats_out_bytes < coskad_layer_bwd_stats_floats(B, Ci, Co, T, V) * sizeof(float))
    return fail(COSKAD_ERR_WORKSPACE, "layer_bwd_stats: stats_out %zu bytes too small", stats_out_bytes);
  const float* wr_tag = has_residual ? in : nullptr;   // stage 1 only asks whether the residual branch exists
#define CALL(T_, V_)                                                                                                      \
  return launch_layer_bwd<T_, V_>(in, dU, A, Tm, in_slope, nullptr, nullptr, nullptr, wr_tag, nullptr, nullptr, nullptr, nullptr, \
                                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, \
                                  0, B, Ci, Co, stream, nullptr, Z, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, 0.0,  \
                                  stats_out, rows_out)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* Stages 1-3 of coskad_layer_bwd_f32 with dZ [B,Ci,T,V] written to the caller's buffer; stage 4 is
 * coskad_layer_gcn_params_f32, which a caller may enqueue on a second stream while the chain continues. */
int coskad_layer_bwd_data_f32(const float* in, const float* dU, const float* A, const float* Tm,
                              const float* in_slope, const float* stat, const float* Wt, const float* gamma_t,
                              const float* Wr, const float* gamma_r, float* dIn, float* dZ, float* dWt, float* dbt,
                              float* dgamma_t, float* dbeta_t, float* dWr, float* dbr, float* dgamma_r,
                              float* dbeta_r, float* dslope_in, void* ws, size_t ws_bytes, int accumulate, int B,
                              int Ci, int Co, int T, int V, hipStream_t stream, const float* Z) {
  if (!in || !dU || !A || !Tm || !stat || !Wt || !gamma_t || !dZ || !dWt || !dgamma_t || !dbeta_t || !ws)
    return fail(COSKAD_ERR_ARG, "layer_bwd_data: null pointer");
  if (Wr && (!gamma_r || !dWr || !dgamma_r || !dbeta_r)) return fail(COSKAD_ERR_ARG, "layer_bwd_data: residual grads missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_bwd_data: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_bwd_data: B=%d Ci=%d Co=%d", B, Ci, Co);
#define CALL(T_, V_)                                                                                              \
  return launch_layer_bwd<T_, V_>(in, dU, A, Tm, in_slope, stat, Wt, gamma_t, Wr, gamma_r, dIn, nullptr, nullptr, dWt, \
                                  dbt, dgamma_t, dbeta_t, dWr, dbr, dgamma_r, dbeta_r, dslope_in, ws, ws_bytes,   \
                                  accumulate, B, Ci, Co, stream, dZ, Z)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

size_t coskad_layer_gcn_params_ws_bytes(int T, int V) { return coskad_gcn_bwd_params_ws_bytes(T, V); }

/* dA, dT of one layer from its input `in` (pre-activation of the producer, PReLU(in_slope) applied on load; NULL for
 * the raw network input) and the dZ stored by coskad_layer_bwd_data_f32. */
int coskad_layer_gcn_params_f32(const float* in, const float* in_slope, const float* dZ, const float* A,
                                const float* Tm, float* dA, float* dT, void* ws, size_t ws_bytes, int accumulate,
                                int B, int Ci, int T, int V, hipStream_t stream) {
  if (!in || !dZ || !A || !Tm || !dA || !dT || !ws) return fail(COSKAD_ERR_ARG, "layer_gcn_params: null pointer");
  if (B <= 0 || Ci <= 0 || Ci > 64) return fail(COSKAD_ERR_ARG, "layer_gcn_params: B=%d Ci=%d", B, Ci);
  if (ws_bytes < coskad_layer_gcn_params_ws_bytes(T, V)) return fail(COSKAD_ERR_WORKSPACE, "layer_gcn_params: workspace too small");
#define CALL(T_, V_) return launch_layer_gcn_params<T_, V_>(in, in_slope, dZ, A, Tm, dA, dT, reinterpret_cast<float*>(ws), accumulate, B, Ci, 0, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

#ifdef COSKAD_FOLD_TIMING
int coskad_debug_fold_stamps(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(coskad::g_fold_stamps), 16 * sizeof(long long)); }
#endif

}  // extern "C"
