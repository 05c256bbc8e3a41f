// Backward of one ST_GCNN layer (autograd of models/graph_layers/stsgcn.py:94-116 in training mode) behind the batch
// reductions and the fp64 fold (stsgcn_bwd.hip stages 1-2): the data path AND the mixing-parameter gradients in ONE
// kernel, one clip per wavefront, for the stored-Z training path at n_frames 12 / n_joints 17, 16 or 32 input channels.
//
//   dZ      = Bt.dU + Kt.Z + kt                          (coefficient matrices from k_bwd_fold)
//   dX      = gcn^T(dZ) + Br.dU + Kr.X + kr ;  dU_prev = dX * PReLU'(U_prev) ;  dslope_prev = sum dX * U_prev [U_prev < 0]
//   dA[t]   = Y_t^T dZ_t   (Y = temporal mix of X)     dT[v] = X_v^T dY_v   (dY = spatial adjoint of dZ)
//
// Replaces k_bwd_data_f + k_bwd_gcn_params (round 1: 229 + 110 us at layer 4, B = 4096) and their HBM round trips: dZ
// is never written (107 MB + re-read), X is staged once for both, the PReLU mask is the only re-read.  Same toolkit as
// fused_fwd.hip: no workgroup barrier, 39.6 KB of LDS per wave, accumulator tiles as the next product's operand
// (dZ tile -> B operand of dA), coefficient matrices as A operands in registers, buffer-addressed streams, hand-written
// software pipeline.  One 32-row LDS image carries X -> Y -> dZ (frame by frame, as soon as dA has consumed Y's frame)
// -> dY -> gcn^T(dZ) in place; Br.dU + Kr.X waits in 104 registers; X is re-staged 16 rows at a time for dT.
// Per-wave partial sums of dA / dT live in the workspace (summed in a fixed order by k_reduce_gcn: deterministic).
#include "fused_ops.h"

namespace coskad {
namespace fb {

using namespace ff;


// A wave's partial sums of dA / dT in the workspace, lane-major (one float4 per lane and record: 1 KB per load / store):
//   records [0, 12)   dA[t][4q + r][j]            (t = record)
//           12        dA[t = 4q + r][16][j]       (t < 12)
//           13        dA[t = j][v = 4q + r][16]   (j < 12)
//           14        dA[t = j][16][16]           (r == 0, q == 0, j < 12)
//           [15, 32)  dT[v][4q + r][j]            (v = record - 15; 4q + r < 12, j < 12)
constexpr int PR_A = 0, PR_XA = 12, PR_XB = 13, PR_C = 14, PR_T = 15, PR_N = 32, EROW = PR_N * 256;
// dA, dT (+)= sum over the P lane-major partial rows (fp64, fixed order); one extra block sums the slope partials; blocks beyond
// that one sum the partial rows the data kernel wrote for the layer below (backward chain: brows [bP][bE] -> bout [bE], the
// k_reduce_partials_d of that layer's call riding in this launch)
__global__ __launch_bounds__(1024) void k_reduce_fused(const float* __restrict__ partials, int P, float* __restrict__ dA,
                                                       float* __restrict__ dT, const float* __restrict__ dap, int ndap,
                                                       float* __restrict__ dslope, int accumulate, const float* __restrict__ brows,
                                                       int bP, int bE, double* __restrict__ bout) {
  __shared__ double sh[1024];
  constexpr int NB = EROW / 64;
  if ((int)blockIdx.x > NB) {
    const int e = ((int)blockIdx.x - NB - 1) * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double s = 0.0;
    if (e < bE)
      for (int p = slice; p < bP; p += 16) s += (double)brows[(size_t)p * bE + e];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (slice == 0 && e < bE) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += sh[threadIdx.x + 64 * k];
      bout[e] = t;
    }
    return;
  }
  if ((int)blockIdx.x == NB) {
    if (!dap) return;
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
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  double s = 0.0;
  for (int p = slice; p < P; p += 16) s += (double)partials[(size_t)p * EROW + e];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (slice == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[threadIdx.x + 64 * k];
    const int rec = e >> 8, l = (e >> 2) & 63, r = e & 3, j = l & 15, q = l >> 4;
    float* out = nullptr;
    if (rec < PR_XA) out = dA + rec * V * V + (4 * q + r) * V + j;
    else if (rec == PR_XA) { if (4 * q + r < T) out = dA + (4 * q + r) * V * V + 16 * V + j; }
    else if (rec == PR_XB) { if (j < T) out = dA + j * V * V + (4 * q + r) * V + 16; }
    else if (rec == PR_C) { if (r == 0 && q == 0 && j < T) out = dA + j * V * V + 16 * V + 16; }
    else if (4 * q + r < T && j < T) out = dT + (rec - PR_T) * T * T + (4 * q + r) * T + j;
    if (out) *out = accumulate ? *out + (float)t : (float)t;
  }
}

#ifdef FB_TIMING   // timing-only builds: per-phase wall-clock (100 MHz) sums of wave 0, written over dIn[block * 16 + phase]
#define FB_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = wall_clock64(); tacc[k] += (float)(now_ - tlast); tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FB_STAMP(k) do {} while (0)
#endif

// NS != 0: the batch reductions of the layer BELOW (stage 1 of ITS backward: P = sum dU.Z^T, Q = sum dU.X^T, sdU -- k_first_stats /
// k_bwd_stats_ring) are formed here, from the dU rows this kernel has just produced in its image: a re-read of dU_prev and a
// launch fewer.  `below_z` / `below_x` [B, Cb, T, V] (X = PReLU(below_x) with `below_slope`, NULL: raw input), `below_stats`
// [grid][2 Ci Cb + Ci] partial rows as k_bwd_fold reads them.
//   NS = 1: Cb = 2 (a first layer): Z0 Z1 X0 X1 are ONE 4-row operand group
//   NS = 2: Cb = 16 CB: 2 CB groups of 16 rows through the K window, (row, position) operands on both sides
template <int CT, int OT, int NS = 0, int CB = 0>
__global__ __launch_bounds__(256, 1) void k_layer_bwd_fused(const float* __restrict__ in, const float* __restrict__ Zg,
                                                           const float* __restrict__ dU, const float* __restrict__ coef,
                                                           const float* __restrict__ btab, const float* __restrict__ in_slope,
                                                           float* __restrict__ dIn, float* __restrict__ partials,
                                                           float* __restrict__ dap, int B, const float* __restrict__ below_z,
                                                           const float* __restrict__ below_x, const float* __restrict__ below_slope,
                                                           float* __restrict__ below_stats) {
  constexpr int Ci = 16 * CT, Co = 16 * OT, CiP = Ci, NG = OT + CT;
  constexpr int KT0 = (Co + Ci) * CiP, DX0 = KT0 + CiP, KR0 = DX0 + (Co + Ci) * CiP;
  static_assert(CT <= 2, "dT stages one 16-row half of X at a time");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = lds_all + wave * WAVE_LDS_W;
  float* r1 = lds + R1;
  float* r2 = lds + R2;
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  // the same for row-wise staging addresses: recomputed where they are used, never carried across the clip loop
  auto olane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  Lane L = geo();
  constexpr bool pre = true;                             // every layer on this path reads PReLU(U_prev) (launcher checks the slope pointer)
  const float a_in = in_slope[0];
  const int l16 = lane * 16;
  const BufRes tabres = make_res(btab, BTAB_F4 * 16u);
  const BufRes cres = make_res(coef, (KR0 + CiP) * 4u);
  const BufRes pres = make_res(partials + (size_t)blockIdx.x * EROW, EROW * 4u);
  const int nwaves = gridDim.x * 4;
  float da = 0.f;
  int clip = blockIdx.x * 4 + wave;
  // dA / dT sums of ALL this wave's clips stay in accumulator registers (124 of them) and leave once, at the end.  The
  // 17th row / column of dA[t] ride on two more tiles: row t of exA collects dA[t][16][0..15] (A operand masked to row t),
  // column t of exB collects dA[t][0..15][16] (B operand masked to column t); the corner element is a plain sum.
  f32x4 dAacc[T], dTacc[V], exA = {0.f, 0.f, 0.f, 0.f}, exB = {0.f, 0.f, 0.f, 0.f};
  float corner = 0.f;
  // (NS) the layer below: rows of [P | Q] for its output channels 16 ct + 4q + r against (Z0 Z1 X0 X1) in columns j < 4, row sums
  f32x4 nsacc[CT][2];
  float nss[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { nsacc[ct][0] = nsacc[ct][1] = f32x4{0.f, 0.f, 0.f, 0.f}; nss[ct] = 0.f; }
  // (NS = 2) group g of the layer below (Z groups, then X groups) against row tile ct: NCH chains per tile keep two MFMAs apart
  constexpr int NGB = NS == 2 ? 2 * CB : 1, NCH = CT == 1 ? 2 : 1, NBUF = (NS == 2 && CB == 2) ? 2 : 1;
  f32x4 nsb[NGB][CT][NCH];
#pragma unroll
  for (int g = 0; g < NGB; ++g)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int h = 0; h < NCH; ++h) nsb[g][ct][h] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool bpre = NS != 0 && below_slope != nullptr;
  const float a_b = bpre ? below_slope[0] : 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) dAacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int v = 0; v < V; ++v) dTacc[v] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Streams are buffer resources per clip; beyond the batch they are empty (loads return 0 without traffic)
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
#ifdef COSKAD_HOT   // timing-only: every stream from 64 L2-resident clips
    return make_res(base + (size_t)(in_range ? (c & 63) : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#else
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#endif
  };
  // X of the NEXT clip is fetched while this clip's epilogue runs and staged at the loop head
  constexpr int XL = (Ci * (TV / 4) + 63) / 64;          // float4 per lane of a clip's input (26 / 13)
  float4 xs[XL];
  auto xload = [&](float4 (&dst)[XL], const BufRes& r) {
#pragma unroll
    for (int i = 0; i < XL; ++i) dst[i] = buf_load4(r, l16, 64 * i * 16);
  };
  // A 16-row group of dU / Z / X travels as four quarters of 4 rows (204 float4 = 3 full 64-lane pieces + one of 12 lanes)
  // through ONE set of 16 registers per lane: a quarter is stored to R2 behind the k-step that consumed those rows, and
  // its registers are refilled at once with the same quarter of the following group -- staging never stands between two
  // groups' MFMAs and every load has a whole group of MFMAs to arrive.
  constexpr int QTAIL = 4 * (TV / 4) - 192;              // lanes of the 4th piece (12)
  const int l16t = lane < QTAIL ? l16 : 0x7ffffff0;      // lanes beyond the quarter: out of range
  float4 gb[16];
  auto qload = [&](const BufRes& res, int row0, int q) {
#pragma unroll
    for (int c = 0; c < 4; ++c) gb[4 * q + c] = buf_load4(res, c < 3 ? l16 : l16t, ((row0 + 4 * q) * (TV / 4) + 64 * c) * 16);
  };
  auto qstore = [&](int q, bool act) {                   // rows 4q .. 4q+3 of R2
    const int ln = olane();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ln + 64 * c;
      float4 v = gb[4 * q + c];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
      if (c < 3 || lane < QTAIL) *reinterpret_cast<float4*>(r2 + (4 * q + row) * LDW + col) = v;
    }
  };
  auto gload = [&](const BufRes& res, int row0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) qload(res, row0, q);
  };
  // (NS = 2, two buffers) the same through a second set of registers
  float4 gc[NBUF == 2 ? 16 : 1];
  auto qload2 = [&](const BufRes& res, int row0, int q) {
#pragma unroll
    for (int c = 0; c < 4; ++c) gc[(4 * q + c) % (NBUF == 2 ? 16 : 1)] = buf_load4(res, c < 3 ? l16 : l16t, ((row0 + 4 * q) * (TV / 4) + 64 * c) * 16);
  };
  auto qstore2 = [&](int q, bool act, float slope) {
    const int ln = olane();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ln + 64 * c;
      float4 v = gc[(4 * q + c) % (NBUF == 2 ? 16 : 1)];
      if (act) { v.x = prelu(v.x, slope); v.y = prelu(v.y, slope); v.z = prelu(v.z, slope); v.w = prelu(v.w, slope); }
      const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
      float* p = r2 + ((c < 3 || lane < QTAIL) ? (4 * q + row) * LD + col : 15 * LD + PADCOL);
      *reinterpret_cast<float2*>(p) = float2{v.x, v.y};
      *reinterpret_cast<float2*>(p + 2) = float2{v.z, v.w};
    }
  };
  // (the statistics phase reads the window as (row, position) operands: its rows then lie LD apart like the image's -- the K
  // window's own 208-float stride puts every other row on the same banks for that pattern -- at two 8-byte stores per float4)
  auto qstore_b = [&](int q, bool act, float slope) {     // gb with another layer's slope
    const int ln = olane();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ln + 64 * c;
      float4 v = gb[4 * q + c];
      if (act) { v.x = prelu(v.x, slope); v.y = prelu(v.y, slope); v.z = prelu(v.z, slope); v.w = prelu(v.w, slope); }
      const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
      float* p = r2 + ((c < 3 || lane < QTAIL) ? (4 * q + row) * LD + col : 15 * LD + PADCOL);
      *reinterpret_cast<float2*>(p) = float2{v.x, v.y};
      *reinterpret_cast<float2*>(p + 2) = float2{v.z, v.w};
    }
  };
  {
    const BufRes x0 = clip_res(in, clip, Ci), du0 = clip_res(dU, clip, Co);
    xload(xs, x0);
    gload(du0, 0);                                       // first K pass, group 0
  }
#ifdef FB_TIMING
  float tacc[16] = {};
  long long tlast = wall_clock64();
#endif

  for (; clip < B; clip += nwaves) {
    FB_STAMP(15);
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), dures = clip_res(dU, clip, Co);
    const BufRes ores = clip_res(dIn, clip, Ci);
    // quarter q of group g of a K pass: dU rows first (OT groups), then the pass's second source
    auto kq = [&](int g, int q, const BufRes& res2) {
      if (g < OT) qload(dures, 16 * g, q);
      else qload(res2, 16 * (g - OT), q);
    };
    // group 0 (fetched a phase ago) goes to R2, group 1 takes its registers: a phase before the pass itself
    auto kprime = [&](const BufRes& res2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { qstore(q, false); kq(1, q, res2); }
    };

    // ---- stage X = PReLU(U_prev) into the image (fetched during the previous clip) ----------------------------------------
    // (register budget: at most 256 of a lane's registers can hold operands; X, the K group and the tables take turns)
    L = geo();
    kprime(zres);
    {
      constexpr int n4 = Ci * (TV / 4);
      const int ln = olane();
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e4 = ln + 64 * i;
        float4 v = xs[i];
        if (pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        const bool ok = e4 < n4;
        *reinterpret_cast<float2*>(r1 + (ok ? row * LD + col : PADCOL)) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(r1 + (ok ? row * LD + col + 2 : PADCOL)) = float2{v.z, v.w};
      }
    }
    TTab tt;
    load_ttab(tt, tabres, 0, l16);
    f32x4 ktq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float4 a = buf_load4(cres, L.q * 16, (KT0 + 16 * ct) * 4);
      ktq[ct] = f32x4{a.x, a.y, a.z, a.w};
    }
    FB_STAMP(0);

    // ---- Y = temporal mix of X, in place -------------------------------------------------------------------------------------
    L = geo();
    temporal_phase<16, CT>(r1, tt, L);
    FB_STAMP(1);

    // ---- K passes:  dZ = Bt.dU + Kt.Z + kt  now,  + dXres = Br.dU + Kr.X + kr  at the end ------------------------------------
    // acc[tile][ct] += coefficient rows [c0 ..) x dU (OT groups) + rows [c1 ..) x the second source (CT groups), all 13
    // position tiles at once: 4 k-steps x 13 x CT independent MFMA chains per group, B operands from R2 (fetched a k-step
    // ahead), coefficient A operands a group ahead.  Two passes (104 accumulators each) instead of one with 208: that one
    // left the register allocator no room to keep operand fetches ahead of the MFMAs; dU is read twice for it.
    L = geo();
    const int lq = (L.q * CiP + L.j) * 4;
    const int jc = L.j < T ? L.j : T - 1;
    auto kpass = [&](f32x4 (&acc)[NTILE][CT], const BufRes& res2, bool act2, int c0, int c1, auto&& last_group_hook) {
      float wc[2][4][CT];
      auto cload = [&](int buf, int g) {
        const int krow = g < OT ? c0 + 16 * g : c1 + 16 * (g - OT);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) wc[buf][s][ct] = buf_load1(cres, lq, ((krow + 4 * s) * CiP + 16 * ct) * 4);
      };
      // on entry group 0 is staged in R2 and group 1 is in the registers (kprime)
      cload(0, 0);
      float b[2][NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) b[0][t] = r2[L.q * LDW + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) cload((g + 1) & 1, g + 1);
        else last_group_hook();                          // the group registers are free from here on
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s + 1 < 4 || g + 1 < NG) {                 // operands of the next k-step (the next group's first: stored at s = 0)
            const int sn = (s + 1) & 3;
#pragma unroll
            for (int t = 0; t < NTILE; ++t) b[(s + 1) & 1][t] = r2[(4 * sn + L.q) * LDW + (t < T ? t * V + L.j : jc * V + 16)];
          }
          if (g + 1 < NG) {                              // this k-step's rows are free: the next group's quarter moves in
            qstore(s, g + 1 >= OT && act2);
            if (g + 2 < NG) kq(g + 2, s, res2);
          }
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[t][ct] = mfma(wc[g & 1][s][ct], b[s & 1][t], acc[t][ct]);
        }
      }
    };
    f32x4 az[NTILE][CT];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) az[t][ct] = ktq[ct];
    kpass(az, zres, false, 0, Co, [] {});                // Bt rows [0, Co), Kt rows [Co, Co + Ci)
    FB_STAMP(2);
    gload(xres, 0);                                      // dT's first X half takes off behind the dA products

    // ---- dA += Y^T dZ per frame, dZ over Y ---------------------------------------------------------------------------------
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {                    // dA[t = j][16][16]
      const f32x4 y16 = tile_load(r1, 16 * ct, jc * V + 16, L);
#pragma unroll
      for (int r = 0; r < 4; ++r) corner = fmaf(y16[r], az[T][ct][r], corner);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const f32x4 y = tile_load(r1, 16 * ct, t * V + L.j, L);          // A operand: Y[16 ct + 4q + r][t, v = j]
        const f32x4 y16 = tile_load(r1, 16 * ct, t * V + 16, L);        // Y[..][t, 16] (same address in every column)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dAacc[t] = mfma(y[r], az[t][ct][r], dAacc[t]);
          exA = mfma(L.j == t ? y16[r] : 0.f, az[t][ct][r], exA);
          exB = mfma(y[r], L.j == t ? az[T][ct][r] : 0.f, exB);
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) tile_store(r1, 16 * ct, t * V + L.j, true, az[t][ct], L);   // dZ over Y's frame t
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) tile_store(r1, 16 * ct, jc * V + 16, L.j < T, az[T][ct], L);
    FB_STAMP(3);

    // ---- dY = spatial adjoint of dZ, in place: the hand-pipelined frame loop of fused_ops.h (MFMA chains of frame t, joint 16
    // on the VALU, operand reads of frame t+1, stores of frame t-1, interleaved by sched_group_barriers) on the adjoint tables ------
    L = geo();
#ifndef FB_OLD_SPATIAL
    spatial_phase<CT>(r1, tabres, 0, l16, L);
#else
    {
      SpatRec rec = load_spat(tabres, 0, 0, l16);
      SOp op[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) op[ct] = spatial_read<16>(r1, ct, 0, L);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : T - 1;
        const SpatRec nxt = load_spat(tabres, 0, tn, l16);
        SOp opn[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) opn[ct] = spatial_read<16>(r1, ct, tn, L);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const f32x4 d = spatial_mm(op[ct], rec);
          spatial_extra<16>(r1, ct, t, op[ct], rec, L);
          tile_store(r1, 16 * ct, t * V + L.j, true, d, L);
        }
        rec = nxt;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) op[ct] = opn[ct];
      }
    }
#endif
    FB_STAMP(5);

    // ---- dT[v] += X_v^T dY_v: X re-staged 16 rows at a time beside the image; the second K pass's group 0 follows ------------
    L = geo();
    load_ttab(tt, tabres, TEMP_F4 + SPAT_F4, l16);       // adjoint temporal operands, ahead of the far loads
    {
      const int ic = L.j < T ? L.j : T - 1;
#pragma unroll
      for (int h = 0; h < CT; ++h) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // X rows 16h .. 16h+15 -> R2; the next half / dU group 0 -> registers
          qstore(q, pre);
          if (h + 1 < CT) qload(xres, 16 * (h + 1), q);
          else qload(dures, 0, q);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {                    // 17 independent chains per k-step
          float a[V], b[V];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            a[v] = r2[(4 * s + L.q) * LDW + ic * V + v];
            b[v] = r1[(16 * h + 4 * s + L.q) * LD + ic * V + v];
          }
#pragma unroll
          for (int v = 0; v < V; ++v) dTacc[v] = mfma(L.j < T ? a[v] : 0.f, L.j < T ? b[v] : 0.f, dTacc[v]);
        }
      }
    }
    kprime(xres);                                        // second pass: dU group 0 -> R2, group 1 on its way across the adjoint
    FB_STAMP(6);

    // ---- gcn^T: temporal adjoint in place ----------------------------------------------------------------------------------
    L = geo();
    temporal_phase<16, CT>(r1, tt, L);
    FB_STAMP(7);

    // ---- + dXres: the second K pass starts from the image's own tiles and returns them in place -------------------------------
    L = geo();
    float4 u[XL];
    constexpr int UH = XL <= 13 ? XL : 13;               // pre-activation pieces fetched inside the pass (the rest: behind it)
    {
      f32x4 krq[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float4 b = buf_load4(cres, L.q * 16, (KR0 + 16 * ct) * 4);
        krq[ct] = f32x4{b.x, b.y, b.z, b.w};
      }
      f32x4 xr[NTILE][CT];
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) xr[t][ct] = tile_load(r1, 16 * ct, t < T ? t * V + L.j : jc * V + 16, L) + krq[ct];
      kpass(xr, xres, pre, DX0 / CiP, DX0 / CiP + Co, [&] {   // Br / Kr rows; the pre-activations come back (from L2) meanwhile
        __builtin_amdgcn_sched_barrier(0);               // not earlier: the registers are taken until here
        if (pre) {
#pragma unroll
          for (int i = 0; i < UH; ++i) u[i] = buf_load4(xres, l16, 64 * i * 16);
        }
        if constexpr (NS == 2) gload(clip_res(below_z, clip, 16 * CB), 0);   // the layer below: its first Z group
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) tile_store(r1, 16 * ct, t < T ? t * V + L.j : jc * V + 16, t < T || L.j < T, xr[t][ct], L);
    }
    FB_STAMP(4);

    // ---- dU_prev = image * PReLU'(U_prev), slope gradient: row-wise, full lines both ways.  The next clip's first dU group
    // takes off first; its X follows piece by piece into the registers the pre-activations leave (loads return in order:
    // nothing this pass waits for is queued behind a far fetch)
    {
      const BufRes xn = clip_res(in, clip + nwaves, Ci), dun = clip_res(dU, clip + nwaves, Co);
      constexpr int N4 = Ci * (TV / 4);
      __builtin_amdgcn_sched_barrier(0);
      if (pre) {
#pragma unroll
        for (int i = UH; i < XL; ++i) u[i] = buf_load4(xres, l16, 64 * i * 16);
      }
      float4 pg[4];                                      // (NS) the layer below: Z rows 0, 1 and X rows 0, 1, two pieces each
      if constexpr (NS == 1) {
        const BufRes zp = clip_res(below_z, clip, 2), xp = clip_res(below_x, clip, 2);
        pg[0] = buf_load4(zp, l16, 0); pg[1] = buf_load4(zp, l16, 1024);
        pg[2] = buf_load4(xp, l16, 0); pg[3] = buf_load4(xp, l16, 1024);
      }
      // group g of the layer below: Z rows 16 g .. (g < CB), then X rows
      auto bres = [&](int g) { return g < CB ? clip_res(below_z, clip, 16 * CB) : clip_res(below_x, clip, 16 * CB); };
      auto brow = [&](int g) { return 16 * (g < CB ? g : g - CB); };
      if constexpr (NS == 2) {
        if constexpr (NBUF == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) qload2(bres(1), brow(1), q);
        }
      } else {
        gload(dun, 0);
      }
      const int ln = olane();
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e4 = ln + 64 * i;
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        float* p = r1 + (e4 < N4 ? row * LD + col : PADCOL);
        const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(p + 2);
        float g[4] = {g0.x, g0.y, g1.x, g1.y};
        if (pre && e4 < N4) {
          const float uu[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (uu[c] < 0.f) da = fmaf(g[c], uu[c], da);
            g[c] = uu[c] > 0.f ? g[c] : a_in * g[c];
          }
        }
        buf_store4(ores, l16, 64 * i * 16, float4{g[0], g[1], g[2], g[3]});    // beyond the clip: dropped (bounds check)
        if constexpr (NS != 0) {                         // the image keeps dU_prev for the reductions below
          // (masked lanes: BOTH halves to the padding columns -- p + 2 there is row 1's first two positions)
          *reinterpret_cast<float2*>(p) = float2{g[0], g[1]};
          *reinterpret_cast<float2*>(e4 < N4 ? p + 2 : p) = float2{g[2], g[3]};
        }
        xs[i] = buf_load4(xn, l16, 64 * i * 16);
      }
      FB_STAMP(9);
      if constexpr (NS == 2) {
        // ---- the layer below: [P | Q] += dU_prev (image rows) x group^T (16 window rows), (row, position) operands on both sides.
        // Group g is stored from its registers, which take group g + NBUF (at the end: the next clip's first dU group) at once ----
        L = geo();
        constexpr int NM = (TV + 7) / 8;
#pragma unroll
        for (int g = 0; g < NGB; ++g) {
          const bool second = NBUF == 2 && (g & 1);
          const bool act = g >= CB && bpre;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (second) {
              qstore2(q, act, a_b);
              if (g + NBUF < NGB) qload2(bres(g + NBUF), brow(g + NBUF), q);
            } else {
              qstore_b(q, act, a_b);
              if (g + NBUF < NGB) qload(bres(g + NBUF), brow(g + NBUF), q);
              else qload(dun, 0, q);                     // this buffer's last group: the next clip's first dU group takes off
            }
          }
          // operands of step m + PF are read before step m multiplies: with two MFMAs per pair of LDS reads the loop is
          // LDS-latency bound otherwise (measured: 9.5 us per clip for 208 MFMAs)
          const float* pb = r2 + L.j * LD + 2 * L.q;
          const float* pa = r1 + L.j * LD + 2 * L.q;
          // (the 32 -> 64 kernel has no registers to spare: there the pipelined form costs spills in the row pass, 355 vs 325 us)
          constexpr int PF = CT == 1 ? 6 : 0;
          float2 bq[PF ? PF : 1], aq[PF ? PF : 1][CT];
          if constexpr (PF == 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
              float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
              const bool tail = 8 * (m + 1) > TV;
              const bool ok = 8 * m + 2 * L.q < TV;
              if (tail) { b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f; }
              float2 a[CT];
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) {
                a[ct] = *reinterpret_cast<const float2*>(pa + 16 * ct * LD + 8 * m);
                if (tail) { a[ct].x = ok ? a[ct].x : 0.f; a[ct].y = ok ? a[ct].y : 0.f; }
                if (g == 0) nss[ct] += a[ct].x + a[ct].y;
              }
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) nsb[g][ct][0] = mfma(a[ct].x, b.x, nsb[g][ct][0]);
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) nsb[g][ct][NCH - 1] = mfma(a[ct].y, b.y, nsb[g][ct][NCH - 1]);
            }
          }
#pragma unroll
          for (int m = 0; m < PF; ++m) {
            bq[m] = *reinterpret_cast<const float2*>(pb + 8 * m);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) aq[m][ct] = *reinterpret_cast<const float2*>(pa + 16 * ct * LD + 8 * m);
          }
#pragma unroll
          for (int m = 0; m < (PF ? NM : 0); ++m) {
            constexpr int PFD = PF ? PF : 1;
            float2 b = bq[m % PFD];
            float2 a[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) a[ct] = aq[m % PFD][ct];
            if (m + PF < NM) {
              bq[m % PFD] = *reinterpret_cast<const float2*>(pb + 8 * (m + PF));
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) aq[m % PFD][ct] = *reinterpret_cast<const float2*>(pa + 16 * ct * LD + 8 * (m + PF));
            }
            if (8 * (m + 1) > TV) {                      // the last step's tail lies in the rows' padding
              const bool ok = 8 * m + 2 * L.q < TV;
              b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) { a[ct].x = ok ? a[ct].x : 0.f; a[ct].y = ok ? a[ct].y : 0.f; }
            }
            if (g == 0) {
#pragma unroll
              for (int ct = 0; ct < CT; ++ct) nss[ct] += a[ct].x + a[ct].y;
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) nsb[g][ct][0] = mfma(a[ct].x, b.x, nsb[g][ct][0]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) nsb[g][ct][NCH - 1] = mfma(a[ct].y, b.y, nsb[g][ct][NCH - 1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + CT, 0);   // this step's reads (of step m + PF) ...
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * CT, 0);   // ... then its MFMAs
          }
        }
      }
      if constexpr (NS == 1) {
        // ---- the layer below: [P | Q] += dU_prev (image rows) x (Z0 Z1 X0 X1)^T (window rows 0..3), (row, position) operands -----
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = ln + 64 * (c & 1);
          const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
          float4 v = pg[c];
          if (c >= 2 && bpre) { v.x = prelu(v.x, a_b); v.y = prelu(v.y, a_b); v.z = prelu(v.z, a_b); v.w = prelu(v.w, a_b); }
          float* pw = r2 + (e < 2 * (TV / 4) ? (2 * (c >> 1) + row) * LD + col : 3 * LD + PADCOL);
          *reinterpret_cast<float2*>(pw) = float2{v.x, v.y};
          *reinterpret_cast<float2*>(pw + 2) = float2{v.z, v.w};
        }
        L = geo();
        const float* pb = r2 + (L.j & 3) * LD + 2 * L.q;
        const float* pa = r1 + L.j * LD + 2 * L.q;
        const bool bok = L.j < 4;
        constexpr int NM = (TV + 7) / 8, PF = 6;
        float2 bq[PF], aq[PF][CT];
#pragma unroll
        for (int m = 0; m < PF; ++m) {
          bq[m] = *reinterpret_cast<const float2*>(pb + 8 * m);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) aq[m][ct] = *reinterpret_cast<const float2*>(pa + 16 * ct * LD + 8 * m);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          float2 b = bq[m % PF];
          float2 a[CT];
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) a[ct] = aq[m % PF][ct];
          if (m + PF < NM) {
            bq[m % PF] = *reinterpret_cast<const float2*>(pb + 8 * (m + PF));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) aq[m % PF][ct] = *reinterpret_cast<const float2*>(pa + 16 * ct * LD + 8 * (m + PF));
          }
          bool ok = bok;
          if (8 * (m + 1) > TV) {                        // the last step's tail lies in the rows' padding
            const bool aok = 8 * m + 2 * L.q < TV;
            ok = ok && aok;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) { a[ct].x = aok ? a[ct].x : 0.f; a[ct].y = aok ? a[ct].y : 0.f; }
          }
          b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            nsacc[ct][0] = mfma(a[ct].x, b.x, nsacc[ct][0]);
            nsacc[ct][1] = mfma(a[ct].y, b.y, nsacc[ct][1]);
            nss[ct] += a[ct].x + a[ct].y;
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 1 + CT, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2 * CT, 0);
        }
      }
      FB_STAMP(10);
    }
  }
  // ---- the block's dA / dT sums: the four waves park their 32 records in their own LDS images, then each wave adds a
  // quarter of them across the four (fixed order) and writes that quarter of the block's lane-major row -----------------------
  {
    float4* mine = reinterpret_cast<float4*>(lds) + lane;
#pragma unroll
    for (int t = 0; t < T; ++t) mine[(PR_A + t) * 64] = float4{dAacc[t][0], dAacc[t][1], dAacc[t][2], dAacc[t][3]};
    mine[PR_XA * 64] = float4{exA[0], exA[1], exA[2], exA[3]};
    mine[PR_XB * 64] = float4{exB[0], exB[1], exB[2], exB[3]};
    mine[PR_C * 64] = float4{quad_sum(corner), 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < V; ++v) mine[(PR_T + v) * 64] = float4{dTacc[v][0], dTacc[v][1], dTacc[v][2], dTacc[v][3]};
    da = wave_sum(da);
    if (lane == 0) lds[PR_N * 256] = da;                 // behind the records
    if constexpr (NS == 1) {                             // the layer below's sums: [o][Z0 Z1 X0 X1] and the row sums, behind that
      float* S = lds + PR_N * 256 + 64;
      L = geo();
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (L.j < 4) S[(16 * ct + 4 * L.q + r) * 4 + L.j] = nsacc[ct][0][r] + nsacc[ct][1][r];
        const float t = quad_sum(nss[ct]);
        if (L.q == 0) S[64 * CT + 16 * ct + L.j] = t;
      }
    }
    if constexpr (NS == 2) {                             // the layer below's sums in the partial row's own layout, behind that
      constexpr int Cb = 16 * CB;
      float* S = lds + PR_N * 256 + 64;
      L = geo();
#pragma unroll
      for (int g = 0; g < NGB; ++g)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * ct + 4 * L.q + r;
            const float v = NCH == 2 ? nsb[g][ct][0][r] + nsb[g][ct][NCH - 1][r] : nsb[g][ct][0][r];
            S[(g < CB ? 0 : Ci * Cb) + o * Cb + 16 * (g < CB ? g : g - CB) + L.j] = v;
          }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float t = quad_sum(nss[ct]);
        if (L.q == 0) S[2 * Ci * Cb + 16 * ct + L.j] = t;
      }
    }
    __syncthreads();
    if constexpr (NS == 2) {                             // one partial row per block: [P Ci x Cb][Q Ci x Cb][sdU Ci], waves in fixed order
      constexpr int NSE = 2 * Ci * 16 * CB + Ci;
      for (int e = threadIdx.x; e < NSE; e += 256) {
        const int off = PR_N * 256 + 64 + e;
        below_stats[(size_t)blockIdx.x * NSE + e] =
            ((lds_all[off] + lds_all[WAVE_LDS_W + off]) + lds_all[2 * WAVE_LDS_W + off]) + lds_all[3 * WAVE_LDS_W + off];
      }
    }
    if constexpr (NS == 1) {                             // one partial row per block: [P Ci x 2][Q Ci x 2][sdU Ci], waves in fixed order
      constexpr int NSE = 5 * Ci;
      const int e = threadIdx.x;
      if (e < NSE) {
        const int off = PR_N * 256 + 64 + e;
        const float v = ((lds_all[off] + lds_all[WAVE_LDS_W + off]) + lds_all[2 * WAVE_LDS_W + off]) + lds_all[3 * WAVE_LDS_W + off];
        int dst = e;                                     // sums: 4 Ci + o
        if (e < 4 * Ci) { const int o = e >> 2, jj = e & 3; dst = jj < 2 ? o * 2 + jj : 2 * Ci + o * 2 + (jj - 2); }
        below_stats[(size_t)blockIdx.x * NSE + dst] = v;
      }
    }
    constexpr int RPW = PR_N / 4;                        // records per wave (8)
    static_assert(PR_N % 4 == 0, "records split evenly over the four waves");
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int rec = wave * RPW + k;
      float4 s = reinterpret_cast<const float4*>(lds_all)[rec * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float4 o = reinterpret_cast<const float4*>(lds_all + w * WAVE_LDS_W)[rec * 64 + lane];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      // one record at a time, drained: with all 32 b128 reads of the unrolled loop in flight (more than the 4-bit LGKM
      // counter can count) the first sums consumed registers before their data had landed (tools/dbg_ragged.py showed it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      buf_store4(pres, l16, rec * 1024, s);
    }
  }
#ifdef FB_TIMING
  FB_STAMP(8);
  __syncthreads();
  if (wave == 0) {
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) mine = lane == k ? tacc[k] : mine;
    if (lane < 16) dIn[blockIdx.x * 16 + lane] = mine;
  }
#endif
  if (threadIdx.x == 0 && dap)
    dap[blockIdx.x] = ((lds_all[PR_N * 256] + lds_all[WAVE_LDS_W + PR_N * 256]) + lds_all[2 * WAVE_LDS_W + PR_N * 256]) +
                      lds_all[3 * WAVE_LDS_W + PR_N * 256];
}

}  // namespace fb

// stage 3 + 4 of launch_layer_bwd for the shapes this kernel is built for; partials: >= grid rows of EROW floats, dap: >= grid floats
int launch_layer_bwd_fused(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw,
                           const float* coef, const float* in_slope, float* dIn, float* btab, float* partials, float* dap,
                           float* xscr, int B, int Ci, int Co, hipStream_t st, int* rows_out, const float* below_z,
                           const float* below_x, const float* below_slope, int below_Ci, float* below_stats) {
  // (btab: built from Aw / Tw by the extra blocks of the fold launch, stsgcn_bwd.hip)
  const size_t lds = (size_t)4 * ff::WAVE_LDS_W * sizeof(float);
  const int nblk = (B + 3) / 4;
#ifndef FB_GRID
#define FB_GRID 256   // one 4-wave block per CU (LDS-bound)
#endif
  const int grid = nblk < FB_GRID ? nblk : FB_GRID;
  *rows_out = grid;
#define LAUNCH_FB(CT, OT)                                                                                              \
  do {                                                                                                                 \
    auto k = fb::k_layer_bwd_fused<CT, OT>;                                                                            \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, coef, btab, in_slope, dIn, partials, dap, B, \
                       below_z, below_x, below_slope, below_stats);                                                    \
  } while (0)
#define LAUNCH_FB_NS(CT, OT, NS, CB)                                                                                   \
  do {                                                                                                                 \
    auto k = fb::k_layer_bwd_fused<CT, OT, NS, CB>;                                                                    \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, coef, btab, in_slope, dIn, partials, dap, B, \
                       below_z, below_x, below_slope, below_stats);                                                    \
  } while (0)
  {
    ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
    if (below_stats && Ci == 32 && Co == 16 && below_Ci == 2) LAUNCH_FB_NS(2, 1, 1, 0);
    else if (below_stats && Ci == 16 && Co == 32 && below_Ci == 32) LAUNCH_FB_NS(1, 2, 2, 2);
    else if (below_stats && Ci == 32 && Co == 64 && below_Ci == 16) LAUNCH_FB_NS(2, 4, 2, 1);
    else if (below_stats) return fail(COSKAD_ERR_SHAPE, "bwd_fused: no kernel forms the reductions of a %d-channel layer below (%d -> %d)", below_Ci, Ci, Co);
    else if (Ci == 16 && Co == 16) LAUNCH_FB(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FB(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_FB(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_FB(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FB(2, 2);
    else if (Ci == 32 && Co == 64) LAUNCH_FB(2, 4);
    else return fail(COSKAD_ERR_SHAPE, "bwd_fused: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FB
#undef LAUNCH_FB_NS
  return check_launch("bwd_fused");
}

int launch_reduce_fused(const float* partials, int rows, float* dA, float* dT, const float* dap, float* dslope, int accumulate,
                        hipStream_t st, const float* brows, int bE, double* bout) {
  const int extra = brows ? 1 + ceil_div(bE, 64) : (dap ? 1 : 0);
  hipLaunchKernelGGL(fb::k_reduce_fused, dim3(fb::EROW / 64 + extra), dim3(1024), 0, st, partials, rows, dA, dT, dap, rows,
                     dslope, accumulate, brows, rows, bE, bout);
  return check_launch("bwd_reduce_fused");
}

int bwd_bpc_rows(int B);   // fused_bwd_bpc.hip
bool bwd_bpc_on(int Ci, int Co) {
  static const bool on = [] { const char* e = getenv("COSKAD_BWD_BPC"); return e && e[0] == '1'; }();   // EXPERIMENT
  return on && ((Ci == 32 && Co == 16) || (Ci == 16 && Co == 32) || (Ci == 32 && Co == 64));
}

// rows of [2 Ci below_Ci + Ci] floats the data kernel of a (Ci -> Co) layer writes for the layer below it (0: it cannot)
int layer_bwd_below_rows(int T_, int V_, int B, int Ci, int Co, int below_Ci) {
  const bool built = (Ci == 32 && Co == 16 && below_Ci == 2) || (Ci == 16 && Co == 32 && below_Ci == 32) ||
                     (Ci == 32 && Co == 64 && below_Ci == 16);
  if (!(T_ == ff::T && V_ == ff::V && built) || B <= 0) return 0;
  if (bwd_bpc_on(Ci, Co)) return bwd_bpc_rows(B);
  const int nblk = (B + 3) / 4;
  return nblk < FB_GRID ? nblk : FB_GRID;
}

bool layer_bwd_fused_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

}  // namespace coskad
