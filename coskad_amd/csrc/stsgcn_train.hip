// Train-mode BatchNorm statistics of one ST_GCNN_layer (reference stsgcn.py:56-80, 94-110)
// WITHOUT materialising the conv outputs.
//
// Both BatchNorms see a 1x1 conv of a tensor we hold per tile (Z = gcn(X) for `tcn`, X for
// `residual`), so their batch statistics follow from the first and second moments of the
// conv INPUT:
//     mean(W z + b) = W mu_z + b          var(W z + b) = diag(W C_z W^T)
// with mu_z, C_z the mean / (biased) covariance of z over (N,T,V).  Moments are C_in x C_in
// GEMMs with K = positions -> v_mfma_f32_16x16x4_f32 (exact fp32) per tile, fp64 across
// tiles.  The layer kernel (stsgcn_fwd.hip) then runs once with the folded weights.
//
//   k_fwd_moments   : per-block partials of  sum x x^T, sum x, sum z z^T, sum z; optionally stores Z = gcn(X) for
//                     the rest of the step (coskad_layer_train_stats_z_f32)
//   k_reduce_partials: fp64 sum over blocks (deterministic: fixed order, no atomics)
//   k_train_fold    : stats -> folded weights, saved stats for backward, running-stat update
#include "mfma_ops.h"

namespace coskad {

constexpr int kMaxGrid = 768;  // persistent blocks of the reduction kernels (= partials to sum)

template <int T, int V, int NTC>
__global__ __launch_bounds__((Geo<T, V>::Block), (Geo<T, V>::Block <= 512 && NTC <= 2 ? 6 : 4)) void k_fwd_moments(const float* __restrict__ in,
                                                       const float* __restrict__ Aw,
                                                       const float* __restrict__ Tw,
                                                       const float* __restrict__ in_slope,
                                                       float* __restrict__ partials, int B, int Ci,
                                                       int NB, int need_x, float* __restrict__ Zout) {
  constexpr int kScratchFloats = Geo<T, V>::Scratch;
  constexpr int TV = Geo<T, V>::TV, LD = TV + 2;   // even stride == 2 (mod 4): no strip phase here, conflict-free (row, k) reads
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* scratch = lds;                 // kScratchFloats, aliased onto the row image (only used after the tile loop)
  float* AwL = lds + max(NB * Ci * LD, kScratchFloats);
  float* TwL = AwL + T * V * V;
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  f32x4 mx[NTC][NTC], mz[NTC][NTC];
  float sx[NTC], sz[NTC];
  zero_acc(mx); zero_acc(mz);
#pragma unroll
  for (int t = 0; t < NTC; ++t) { sx[t] = 0.f; sz[t] = 0.f; }

  const int ntiles = ceil_div(B, NB);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const int rows = nb * Ci;
    lds_barrier();  // previous tile's MFMA reads are done
    stage_rows<T, V, LD>(in + (size_t)clip0 * Ci * TV, lds, rows * TV, pre, a_in);
    lds_barrier();
    if (need_x) {
      for (int n = 0; n < nb; ++n) moment_accum<T, V, NTC, LD>(lds + n * Ci * LD, Ci, mx, sx);
      lds_barrier();
    }
    gcn_mfma<T, V, false, LD>(lds, rows, AwL, TwL);
    lds_barrier();
    // Z = gcn(X) is kept for the rest of the step (apply and both backward kernels read it instead of recomputing)
    if (Zout) unstage_rows<T, V, LD>(Zout + (size_t)clip0 * Ci * TV, lds, rows * TV);
    for (int n = 0; n < nb; ++n) moment_accum<T, V, NTC, LD>(lds + n * Ci * LD, Ci, mz, sz);
  }
  // partial layout: [MX Ci*Ci][sumX Ci][MZ Ci*Ci][sumZ Ci]
  float* dst = partials + (size_t)blockIdx.x * (2 * (Ci * Ci + Ci));
  store_moments<NTC>(mx, sx, scratch, dst, Ci, dst + Ci * Ci, Ci);
  store_moments<NTC>(mz, sz, scratch, dst + Ci * Ci + Ci, Ci, dst + 2 * Ci * Ci + Ci, Ci);
}

// out[e] = sum_p partials[p][e] in fp64 (fixed order).  block = 64 elements x 16 partial-slices, 4 loads in
// flight per thread: the table is tiny (<= 512 x E floats) and the kernel is latency-bound.
constexpr int kRedCols = 16;   // columns per block (132 blocks at E = 2112; a thread's <= 16 rows all in flight)
__global__ __launch_bounds__(1024) void k_reduce_partials(const float* __restrict__ partials, int P, int E,
                                                           double* __restrict__ out) {
  __shared__ double sh[1024];
  const int e = blockIdx.x * kRedCols + (threadIdx.x % kRedCols);
  const double t = column_sum_f64<kRedCols>(partials, P, (size_t)E, e, e < E, sh);
  if ((int)threadIdx.x < kRedCols && e < E) out[e] = t;
}

// stat block (floats), saved for the backward pass:
//   [muX Ci][muZ Ci][WCs Co*Ci = Wt C_Z][WCr Co*Ci = Wr C_X][mean_s Co][istd_s Co][mean_r Co][istd_r Co]
__host__ __device__ inline int stat_floats(int Ci, int Co) { return 2 * Ci + 2 * Co * Ci + 4 * Co; }

#ifdef COSKAD_FOLD_TIMING   // timing-only build (tools/time_fold.py)
__device__ long long g_tfold_stamps[16];
#define TFOLD_STAMP(k) do { if (threadIdx.x == 0) g_tfold_stamps[k] = wall_clock64(); } while (0)
#else
#define TFOLD_STAMP(k) do {} while (0)
#endif

constexpr int kFoldSelfSum = 48;   // E = 2 (Ci^2 + Ci) up to Ci = 4
__global__ __launch_bounds__(1024) void k_train_fold(
    const double* __restrict__ red, double npos, const float* __restrict__ Wt,
    const float* __restrict__ bt, const float* __restrict__ gs, const float* __restrict__ bs,
    float* __restrict__ rm_s, float* __restrict__ rv_s, long long* __restrict__ nbt_s,
    const float* __restrict__ Wr, const float* __restrict__ br, const float* __restrict__ gr,
    const float* __restrict__ brr, float* __restrict__ rm_r, float* __restrict__ rv_r,
    long long* __restrict__ nbt_r, float momentum, float* __restrict__ wfold,
    float* __restrict__ bias, float* __restrict__ stat, int Ci, int Co, int CoP,
    const float* __restrict__ parts, int P) {
  // One block, VALU-bound on fp64 and latency-bound on its global round trips: every input is read once at the start, the
  // phases exchange through LDS only (the stat block is written along the way, never read back), W C is register-blocked
  // (4 columns per thread: one conversion of W[o][k] per four DFMAs), per-channel sums use 8 lanes per channel.
  // LDS: doubles C[2][Ci*Ci] (centred covariances: Z then X), mu[2][Ci]; floats W[2][Co*Ci], WC[2][Co*Ci], mean[2][Co], istd[2][Co]
  extern __shared__ double shd[];
  const bool ident = Wr == nullptr;
  const int CC = Co * Ci;
  double* Cl = shd;                       // [2][Ci*Ci]
  double* mul = shd + 2 * Ci * Ci;        // [2][Ci]
  float* Wl = reinterpret_cast<float*>(mul + 2 * Ci);   // [2][Co*Ci]
  float* WCl = Wl + 2 * CC;                             // [2][Co*Ci]
  float* meanl = WCl + 2 * CC;                          // [2][Co]
  float* istdl = meanl + 2 * Co;                        // [2][Co]
  // parts != NULL (a handful of input channels: E <= kFoldSelfSum): every block sums the moment partials itself, in
  // k_reduce_partials' order -- the table is a few thousand floats and its own launch cost more than the sums
  __shared__ double red_l[kFoldSelfSum];
  if (parts) {
    __shared__ double shs[1024];
    const int E = 2 * (Ci * Ci + Ci);
    for (int e0 = 0; e0 < E; e0 += kRedCols) {
      const int e = e0 + (int)(threadIdx.x % kRedCols);
      const double t = column_sum_f64<kRedCols>(parts, P, (size_t)E, e, e < E, shs);
      if ((int)threadIdx.x < kRedCols && e < E) red_l[e] = t;
      __syncthreads();
    }
  }
  const double* redp = parts ? red_l : red;
  const double* MX = redp;
  const double* SX = redp + Ci * Ci;
  const double* MZ = redp + Ci * Ci + Ci;
  const double* SZ = redp + 2 * Ci * Ci + Ci;
  float* muX = stat;
  float* muZ = stat + Ci;
  float* WCs = stat + 2 * Ci;
  float* WCr = WCs + CC;
  float* mean_s = WCr + CC;
  float* istd_s = mean_s + Co;
  float* mean_r = istd_s + Co;
  float* istd_r = mean_r + Co;
  const double inv_n = 1.0 / npos;
  // every block folds a contiguous slice of the output channels (everything the fold produces is per output channel once
  // the Ci x Ci covariances are known, which each block forms for itself): the fp64 work spreads over several CUs
  const int chunk = (Co + gridDim.x - 1) / gridDim.x;
  const int o_lo = blockIdx.x * chunk, o_hi = min(Co, o_lo + chunk), no = max(0, o_hi - o_lo);
  const bool first = blockIdx.x == 0;

  TFOLD_STAMP(0);
  for (int c = threadIdx.x; c < Ci; c += blockDim.x) {
    const double mz = SZ[c] * inv_n, mx = SX[c] * inv_n;
    mul[c] = mz;
    mul[Ci + c] = mx;
    if (first) { muZ[c] = (float)mz; muX[c] = (float)mx; }
  }
  for (int i = threadIdx.x; i < CC; i += blockDim.x) {
    Wl[i] = Wt[i];
    Wl[CC + i] = ident ? 0.f : Wr[i];
  }
  for (int i = threadIdx.x; i < Ci * Ci; i += blockDim.x) {
    const int k = i / Ci, c = i - k * Ci;
    Cl[i] = MZ[i] * inv_n - (SZ[k] * inv_n) * (SZ[c] * inv_n);
    Cl[Ci * Ci + i] = ident ? 0.0 : MX[i] * inv_n - (SX[k] * inv_n) * (SX[c] * inv_n);
  }
  __syncthreads();
  TFOLD_STAMP(1);
  // WC[o][c] = sum_k W[o][k] C[k][c]   (fp64 accumulate), four consecutive columns per thread
  const int Cq = (Ci + 3) / 4;
  for (int i = threadIdx.x; i < 2 * no * Cq; i += blockDim.x) {
    const int b = i / (no * Cq);
    const int j = i - b * no * Cq;
    const int o = o_lo + j / Cq, c0 = 4 * (j % Cq);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (!(b == 1 && ident)) {
      const float* W = Wl + b * CC + o * Ci;
      const double* C = Cl + b * Ci * Ci + c0;
      if (c0 + 3 < Ci) {
        for (int k = 0; k < Ci; ++k) {
          const double w = (double)W[k];
          const double* r = C + k * Ci;
          a0 += w * r[0]; a1 += w * r[1]; a2 += w * r[2]; a3 += w * r[3];
        }
      } else {
        for (int k = 0; k < Ci; ++k) {
          const double w = (double)W[k];
          const double* r = C + k * Ci;
          a0 += w * r[0];
          if (c0 + 1 < Ci) a1 += w * r[1];
          if (c0 + 2 < Ci) a2 += w * r[2];
        }
      }
    }
    const double acc[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + u < Ci) {
        const int e = b * CC + o * Ci + c0 + u;
        WCl[e] = (float)acc[u];
        (b ? WCr : WCs)[o * Ci + c0 + u] = (float)acc[u];
      }
  }
  __syncthreads();
  TFOLD_STAMP(2);
  const double unbias = npos > 1.0 ? npos / (npos - 1.0) : 1.0;
  // per-channel statistics: 8 lanes per (branch, channel), fixed-order tree
  for (int i0 = 0; i0 < 2 * no; i0 += blockDim.x / 8) {
    const int ii = i0 + (threadIdx.x >> 3), l8 = threadIdx.x & 7;
    const bool live = ii < 2 * no;
    const int b = live ? ii / no : 0, o = live ? o_lo + ii - b * no : 0;
    const int i = b * Co + o;
    double m = 0.0, var = 0.0;
    if (live && !(b == 1 && ident)) {
      const float* W = Wl + b * CC + o * Ci;
      const float* WC = WCl + b * CC + o * Ci;
      for (int k = l8; k < Ci; k += 8) {
        const double w = (double)W[k];
        m += w * mul[b * Ci + k];
        var += w * (double)WC[k];
      }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      m += __shfl_xor(m, off, 8);
      var += __shfl_xor(var, off, 8);
    }
    if (live && l8 == 0) {
      if (b == 1 && ident) { mean_r[o] = 0.f; istd_r[o] = 1.f; meanl[i] = 0.f; istdl[i] = 1.f; }
      else {
        const float* bb = b ? br : bt;
        m += bb ? (double)bb[o] : 0.0;
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)kBnEps));
        (b ? mean_r : mean_s)[o] = mf;
        (b ? istd_r : istd_s)[o] = isf;
        meanl[i] = mf;
        istdl[i] = isf;
        float* rm = b ? rm_r : rm_s;
        float* rv = b ? rv_r : rv_s;
        if (rm) {  // nn.BatchNorm2d: running = (1-m) running + m batch ; unbiased variance
          rm[o] = (1.f - momentum) * rm[o] + momentum * (float)m;
          rv[o] = (1.f - momentum) * rv[o] + momentum * (float)(var * unbias);
        }
      }
    }
  }
  if (threadIdx.x == 0 && first) {
    if (nbt_s) nbt_s[0] += 1;
    if (nbt_r && !ident) nbt_r[0] += 1;
  }
  __syncthreads();
  TFOLD_STAMP(3);
  // folded weights: U = Wz Z + Wx X + b   (this block's columns; the last block also zeroes the padding columns)
  const bool last = blockIdx.x == gridDim.x - 1;
  const int w_hi = last ? CoP : min(o_lo + chunk, CoP), nw = max(0, w_hi - o_lo);
  for (int i = threadIdx.x; i < 2 * Ci * nw; i += blockDim.x) {
    const int row = i / nw, o = o_lo + i - row * nw;
    float w = 0.f;
    if (o < Co) {
      if (row < Ci) w = gs[o] * istdl[o] * Wl[o * Ci + row];
      else {
        const int c = row - Ci;
        w = ident ? (c == o ? 1.f : 0.f) : gr[o] * istdl[Co + o] * Wl[CC + o * Ci + c];
      }
    }
    wfold[row * CoP + o] = w;
  }
  for (int o = o_lo + threadIdx.x; o < w_hi; o += blockDim.x) {
    float b = 0.f;
    if (o < Co) {
      b = bs[o] + gs[o] * istdl[o] * ((bt ? bt[o] : 0.f) - meanl[o]);
      if (!ident) b += brr[o] + gr[o] * istdl[Co + o] * ((br ? br[o] : 0.f) - meanl[Co + o]);
    }
    bias[o] = b;
  }
  TFOLD_STAMP(4);
}

// rows of the tile kernels that keep ONE Ci-row image (+1024 floats scratch) in LDS
static int pick_nb_rows(int rows_per_clip, int B, int LD, int budget_bytes) {
  int nb = rows_per_clip >= 64 ? 1 : 64 / rows_per_clip;
  if (nb < 1) nb = 1;
  while (nb > 1 && (size_t)nb * rows_per_clip * LD * 4 > (size_t)budget_bytes) --nb;
  if (nb > B) nb = B;
  return nb;
}

// first_layer.hip
int launch_first_moments(const float* in, const float* Aw, const float* Tw, const float* in_slope, float* partials, int B, int Ci,
                         int T, int V, float* Zout, int max_rows, hipStream_t st, int* rows_out);
// fwd_moments_bpc.hip
bool fwd_moments_bpc_ok(int T_, int V_, int Ci);
int launch_fwd_moments_bpc(const float* in, const float* Aw, const float* Tw, const float* in_slope, float* partials, int B, int Ci,
                           int T_, int V_, int need_x, float* Zout, hipStream_t st, int* rows_out);

size_t train_stats_ws_bytes(int Ci) {
  const size_t E = 2 * ((size_t)Ci * Ci + Ci);
  return kMaxGrid * E * sizeof(float) + round_up((int)(E * sizeof(double)), 256) + 256;
}

// moment partials [rows][2 (Ci^2 + Ci)] -> fp64 sums (fixed order) -> statistics, folded weights, running-stat update
static int launch_reduce_fold(const float* partials, int rows, double* red, double npos, const float* Wt, const float* bt,
                              const float* gs, const float* bs, float* rm_s, float* rv_s, long long* nbt_s, const float* Wr,
                              const float* br, const float* gr, const float* brr, float* rm_r, float* rv_r, long long* nbt_r,
                              float momentum, float* wfold, float* bias, float* stat, int Ci, int Co, hipStream_t st) {
  const int E = 2 * (Ci * Ci + Ci);
  int rc = 0;
  const bool self_sum = partials && Wt && E <= kFoldSelfSum;   // the fold sums the (tiny) table itself
  if (partials && !self_sum) {                           // (NULL: `red` holds the sums already -- SyncBN, summed over the ranks)
    hipLaunchKernelGGL(k_reduce_partials, dim3(ceil_div(E, kRedCols)), dim3(1024), 0, st, partials, rows, E, red);
    if ((rc = check_launch("reduce_partials"))) return rc;
  }
  if (!Wt) return 0;                                     // sums only
  const size_t fold_lds = (2 * (size_t)Ci * Ci + 2 * Ci) * sizeof(double) + (4 * (size_t)Co * Ci + 4 * Co) * sizeof(float);
  if (fold_lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_train_fold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_lds);
  const int fold_blocks = Co >= 32 ? 8 : (Co >= 16 ? 4 : 1);
  hipLaunchKernelGGL(k_train_fold, dim3(fold_blocks), dim3(1024), fold_lds, st, red, npos, Wt, bt, gs, bs, rm_s,
                     rv_s, nbt_s, Wr, br, gr, brr, rm_r, rv_r, nbt_r, momentum, wfold, bias, stat, Ci,
                     Co, round_up(Co, 16), self_sum ? partials : (const float*)nullptr, rows);
  return check_launch("train_fold");
}

template <int T, int V>
static int launch_train_stats(const float* in, const float* Aw, const float* Tw, const float* in_slope,
                              const float* Wt, const float* bt, const float* gs, const float* bs,
                              float* rm_s, float* rv_s, long long* nbt_s, const float* Wr,
                              const float* br, const float* gr, const float* brr, float* rm_r,
                              float* rv_r, long long* nbt_r, float momentum, float* wfold, float* bias,
                              float* stat, void* ws, size_t ws_bytes, int B, int Ci, int Co,
                              hipStream_t st, float* Zout = nullptr, double* sums_out = nullptr, int need_x_sums = 1) {
  // sums_out != NULL: stop behind the fp64 moment sums (written there; Wt .. stat unused) -- the caller adds the other ranks'
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int kScratchFloats = Geo<T, V>::Scratch;
  constexpr int TV = Geo<T, V>::TV, LD = TV + 2;   // k_fwd_moments' row stride
  static_assert(TV % 4 == 0, "the moments kernel stages with float4 and needs TV + 2 == 2 (mod 4)");
  if (Ci > 64) return fail(COSKAD_ERR_SHAPE, "train_stats: C_in=%d > 64 not supported", Ci);
  if (ws_bytes < train_stats_ws_bytes(Ci))
    return fail(COSKAD_ERR_WORKSPACE, "train_stats: workspace %zu < %zu bytes", ws_bytes, train_stats_ws_bytes(Ci));
  const int E = 2 * (Ci * Ci + Ci);
  int NB = Ci >= 32 ? 1 : 32 / Ci;   // 32 rows per tile (2 MFMA row tiles)
  if (NB > B) NB = B;
  const size_t img = (size_t)NB * Ci * LD > (size_t)kScratchFloats ? (size_t)NB * Ci * LD : (size_t)kScratchFloats;
  const size_t lds = (img + (size_t)T * V * V + (size_t)V * T * T) * sizeof(float);
  if (lds > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "train_stats: LDS %zu too large", lds);
  const int ntiles = ceil_div(B, NB);
  // persistent blocks: as many 512-thread blocks per CU as LDS (coarse granules: keep a margin) and VGPRs allow
  const int per_cu = lds <= (size_t)52 * 1024 && Ci <= 32 && kBlock <= 512 ? 3 : (lds <= (size_t)80 * 1024 ? 2 : 1);
  const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
  float* partials = reinterpret_cast<float*>(ws);
  double* red = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + round_up((int)(kMaxGrid * (size_t)E * sizeof(float)), 256));
  const int need_x = sums_out ? need_x_sums : (Wr != nullptr);
  const int ntc = ceil_div(Ci, 16);
  int rows = grid;
  if (Zout && Ci <= 4 && TV % 4 == 0) {
    // a handful of input channels (the first layer): plain FMAs, one clip per wave (first_layer.hip)
    int rc1 = launch_first_moments(in, Aw, Tw, in_slope, partials, B, Ci, T, V, Zout, kMaxGrid, st, &rows);
    if (rc1) return rc1;
  } else if (Zout && fwd_moments_bpc_ok(T, V, Ci)) {
    // the 25-joint layout, 16 / 32 channels: one clip per four-wave workgroup, mixing operands in registers (fwd_moments_bpc.hip)
    ProbeScope probe(KID_FWD_MOMENTS, Ci, Co, st);
    int rc1 = launch_fwd_moments_bpc(in, Aw, Tw, in_slope, partials, B, Ci, T, V, need_x, Zout, st, &rows);
    if (rc1) return rc1;
  } else {
#define LAUNCH_M(NTC)                                                                           \
  do {                                                                                          \
    auto k = k_fwd_moments<T, V, NTC>;                                                          \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, Aw, Tw, in_slope, partials, B, Ci, NB, need_x, Zout); \
  } while (0)
  {
  ProbeScope probe(KID_FWD_MOMENTS, Ci, Co, st);
  if (ntc == 1) LAUNCH_M(1);
  else if (ntc == 2) LAUNCH_M(2);
  else if (ntc == 3) LAUNCH_M(3);
  else LAUNCH_M(4);
  }
#undef LAUNCH_M
  }
  int rc = check_launch("fwd_moments");
  if (rc) return rc;
  if (sums_out)
    return launch_reduce_fold(partials, rows, sums_out, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, Ci, Co, st);
  return launch_reduce_fold(partials, rows, red, (double)B * TV, Wt, bt, gs, bs, rm_s, rv_s, nbt_s, Wr, br, gr, brr, rm_r, rv_r,
                            nbt_r, momentum, wfold, bias, stat, Ci, Co, st);
}

}  // namespace coskad

using namespace coskad;

extern "C" {

size_t coskad_train_stats_ws_bytes(int Ci) { return train_stats_ws_bytes(Ci); }
int coskad_stat_floats(int Ci, int Co) { return stat_floats(Ci, Co); }

int coskad_layer_train_stats_f32(const float* in, const float* A, const float* Tm, const float* in_slope,
                                 const float* Wt, const float* bt, const float* gamma_t,
                                 const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                 const float* Wr, const float* br, const float* gamma_r,
                                 const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                 float momentum, float* wfold, float* bias, float* stat, void* ws,
                                 size_t ws_bytes, int B, int Ci, int Co, int T, int V,
                                 hipStream_t stream) {
  if (!in || !A || !Tm || !Wt || !gamma_t || !beta_t || !wfold || !bias || !stat || !ws)
    return fail(COSKAD_ERR_ARG, "layer_train_stats: null pointer");
  if (Wr && (!gamma_r || !beta_r)) return fail(COSKAD_ERR_ARG, "layer_train_stats: residual BN missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_train_stats: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_train_stats: B=%d Ci=%d Co=%d", B, Ci, Co);
#define CALL(T_, V_)                                                                                   \
  return launch_train_stats<T_, V_>(in, A, Tm, in_slope, Wt, bt, gamma_t, beta_t, rmean_t, rvar_t, nbt_t, \
                                    Wr, br, gamma_r, beta_r, rmean_r, rvar_r, nbt_r, momentum, wfold, \
                                    bias, stat, ws, ws_bytes, B, Ci, Co, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* The same, additionally storing Z = gcn(PReLU(in)) [B,Ci,T,V] for coskad_layer_apply_z_f32 and the *_z backward. */
int coskad_layer_train_stats_z_f32(const float* in, const float* A, const float* Tm, const float* in_slope,
                                 const float* Wt, const float* bt, const float* gamma_t,
                                 const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                 const float* Wr, const float* br, const float* gamma_r,
                                 const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                 float momentum, float* wfold, float* bias, float* stat, void* ws,
                                 size_t ws_bytes, int B, int Ci, int Co, int T, int V,
                                 hipStream_t stream, float* Z) {
  if (!in || !A || !Tm || !Wt || !gamma_t || !beta_t || !wfold || !bias || !stat || !ws)
    return fail(COSKAD_ERR_ARG, "layer_train_stats: null pointer");
  if (Wr && (!gamma_r || !beta_r)) return fail(COSKAD_ERR_ARG, "layer_train_stats: residual BN missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_train_stats: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0) return fail(COSKAD_ERR_ARG, "layer_train_stats: B=%d Ci=%d Co=%d", B, Ci, Co);
#define CALL(T_, V_)                                                                                   \
  return launch_train_stats<T_, V_>(in, A, Tm, in_slope, Wt, bt, gamma_t, beta_t, rmean_t, rvar_t, nbt_t, \
                                    Wr, br, gamma_r, beta_r, rmean_r, rvar_r, nbt_r, momentum, wfold, \
                                    bias, stat, ws, ws_bytes, B, Ci, Co, stream, Z)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* Statistics of a layer whose moment partials were produced by the PREVIOUS layer's coskad_layer_apply_next_f32:
 * partials [rows][2 (Ci^2 + Ci)] -> fold.  ws: >= coskad_train_stats_ws_bytes(Ci) (the fp64 sums live behind the
 * partial area, as in coskad_layer_train_stats_f32). */
int coskad_layer_train_fold_f32(const float* partials, int rows, const float* Wt, const float* bt, const float* gamma_t,
                                const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                const float* Wr, const float* br, const float* gamma_r,
                                const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                float momentum, float* wfold, float* bias, float* stat, void* ws,
                                size_t ws_bytes, int B, int Ci, int Co, int T, int V, hipStream_t stream) {
  if (!partials || !Wt || !gamma_t || !beta_t || !wfold || !bias || !stat || !ws)
    return fail(COSKAD_ERR_ARG, "layer_train_fold: null pointer");
  if (Wr && (!gamma_r || !beta_r)) return fail(COSKAD_ERR_ARG, "layer_train_fold: residual BN missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_train_fold: identity residual needs Ci == Co");
  if (B <= 0 || Ci <= 0 || Co <= 0 || Ci > 64 || rows <= 0) return fail(COSKAD_ERR_ARG, "layer_train_fold: B=%d Ci=%d Co=%d rows=%d", B, Ci, Co, rows);
  const size_t E = 2 * ((size_t)Ci * Ci + Ci);
  if (ws_bytes < round_up((int)(E * sizeof(double)), 256)) return fail(COSKAD_ERR_WORKSPACE, "layer_train_fold: workspace %zu too small", ws_bytes);
  return launch_reduce_fold(partials, rows, reinterpret_cast<double*>(ws), (double)B * T * V, Wt, bt, gamma_t, beta_t, rmean_t,
                            rvar_t, nbt_t, Wr, br, gamma_r, beta_r, rmean_r, rvar_r, nbt_r, momentum, wfold, bias, stat, Ci, Co,
                            stream);
}

/* ---- SyncBN (optional; the reference trains with per-rank statistics, train_COSKAD.py:75-78): the three steps of
 * coskad_layer_train_stats_z_f32 / coskad_layer_train_fold_f32 as separate calls, so that the caller can add the other ranks'
 * moment sums (an all-reduce of 2 (Ci^2 + Ci) doubles) between the batch reduction and the fold ------------------------------ */

/* step 1 (a layer with its own statistics pass): Z = gcn(PReLU(in)) [B,Ci,T,V] (NULL: not stored) and this rank's fp64 moment
 * sums [sum xx^T Ci^2][sum x Ci][sum zz^T Ci^2][sum z Ci] -> sums.  ws: coskad_train_stats_ws_bytes(Ci). */
int coskad_layer_train_moments_f32(const float* in, const float* A, const float* Tm, const float* in_slope, float* Z, double* sums,
                                   void* ws, size_t ws_bytes, int B, int Ci, int T, int V, hipStream_t stream) {
  if (!in || !A || !Tm || !sums || !ws) return fail(COSKAD_ERR_ARG, "layer_train_moments: null pointer");
  if (B <= 0 || Ci <= 0) return fail(COSKAD_ERR_ARG, "layer_train_moments: B=%d Ci=%d", B, Ci);
  if ((size_t)sums & 7) return fail(COSKAD_ERR_ARG, "layer_train_moments: sums must be 8-byte aligned");
#define CALL(T_, V_)                                                                                                       \
  return launch_train_stats<T_, V_>(in, A, Tm, in_slope, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, \
                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, ws,     \
                                    ws_bytes, B, Ci, Ci, stream, Z, sums, 1)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}

/* step 1 (moment partials written by the previous layer's coskad_layer_apply_next_f32): partials [rows][2 (Ci^2 + Ci)] -> sums */
int coskad_layer_moment_sums_f32(const float* partials, int rows, int Ci, double* sums, hipStream_t stream) {
  if (!partials || !sums || rows <= 0 || Ci <= 0 || Ci > 64) return fail(COSKAD_ERR_ARG, "layer_moment_sums: bad argument");
  if ((size_t)sums & 7) return fail(COSKAD_ERR_ARG, "layer_moment_sums: sums must be 8-byte aligned");
  return launch_reduce_fold(partials, rows, sums, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, Ci, Ci, stream);
}

/* step 2: the fold (statistics, folded weights, running-statistics update) from moment sums over `count` positions
 * (= global clips x T x V once the ranks' sums are added). */
int coskad_layer_train_fold_sums_f32(const double* sums, double count, const float* Wt, const float* bt, const float* gamma_t,
                                     const float* beta_t, float* rmean_t, float* rvar_t, long long* nbt_t,
                                     const float* Wr, const float* br, const float* gamma_r,
                                     const float* beta_r, float* rmean_r, float* rvar_r, long long* nbt_r,
                                     float momentum, float* wfold, float* bias, float* stat, int Ci, int Co, hipStream_t stream) {
  if (!sums || !Wt || !gamma_t || !beta_t || !wfold || !bias || !stat) return fail(COSKAD_ERR_ARG, "layer_train_fold_sums: null pointer");
  if (Wr && (!gamma_r || !beta_r)) return fail(COSKAD_ERR_ARG, "layer_train_fold_sums: residual BN missing");
  if (!Wr && Ci != Co) return fail(COSKAD_ERR_ARG, "layer_train_fold_sums: identity residual needs Ci == Co");
  if (count <= 0.0 || Ci <= 0 || Co <= 0 || Ci > 64) return fail(COSKAD_ERR_ARG, "layer_train_fold_sums: count=%g Ci=%d Co=%d", count, Ci, Co);
  return launch_reduce_fold(nullptr, 0, const_cast<double*>(sums), count, Wt, bt, gamma_t, beta_t, rmean_t, rvar_t, nbt_t, Wr, br,
                            gamma_r, beta_r, rmean_r, rvar_r, nbt_r, momentum, wfold, bias, stat, Ci, Co, stream);
}

#ifdef COSKAD_FOLD_TIMING
int coskad_debug_tfold_stamps(long long* out16) { return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(coskad::g_tfold_stamps), 16 * sizeof(long long)); }
#endif

}  // extern "C"
