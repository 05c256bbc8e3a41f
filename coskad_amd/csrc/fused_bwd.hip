// Backward of one ST_GCNN layer (autograd of models/graph_layers/stsgcn.py:94-116 in training mode) behind the batch reductions
// and the fp64 fold (stsgcn_bwd.hip stages 1-2): the data path AND the mixing-parameter gradients in ONE kernel, ONE CLIP PER
// WORKGROUP, for the stored-Z training path at n_frames 12 / n_joints 17, 16 or 32 input channels.
//
//   dZ      = Bt.dU + Kt.Z + kt                          (coefficient matrices from k_bwd_fold)
//   dX      = gcn^T(dZ) + Br.dU + Kr.X + kr ;  dU_prev = dX * PReLU'(U_prev) ;  dslope_prev = sum dX * U_prev [U_prev < 0]
//   dA[t]   = Y_t^T dZ_t   (Y = temporal mix of X)     dT[v] = X_v^T dY_v   (dY = spatial adjoint of dZ)
//   (+ the batch reductions of the layer BELOW from the dU_prev rows, NS != 0: the backward chain)
//
// dZ never leaves the CU and the rows of dU are read ONCE: the K pass feeds both dZ = Bt.dU + .. and Br.dU (two accumulator
// sets on the same window rows), Kr.X rides on the X halves that dT stages anyway -- round 2's second pass over dU (from the
// Infinity Cache) and its 24 barriers are gone, for 28 more accumulator registers: two waves per SIMD instead of three.  One
// 32-row LDS image carries X -> Y -> dZ -> dY -> gcn^T(dZ) -> dX -> dU_prev in place; the rows of dU and Z stream through a
// 16-row K window quarter by quarter (full-line buffer loads by all 256 threads, a quarter per k-step, one group in flight in
// registers).  The four waves of a workgroup share the image and the window (40.7 KB per workgroup) and split the work:
//   K pass        a wave owns one channel tile x 6 frames (3 at 16 channels) + the joint-16 tile, which every wave computes for
//                 itself for dZ (dA's 17th column needs it beside every frame): 7 + 6.25 (4 + 3.25) accumulator tiles
//   dA            from the wave's own dZ tiles (accumulator tile -> B operand): its frames' dA[t] only; the waves' sums meet at
//                 the very end
//   mixing        joints (temporal) / frames (spatial) round-robin;  dT + Kr.X: joints round-robin / the wave's tiles, X staged
//                 in the window 16 rows at a time by all threads
//   row pass      all 256 threads
// with a workgroup barrier between the phases and one per k-step of the K pass.  Round 2's kernel kept all of this in ONE
// wave per clip (one wave per SIMD: 104 pass accumulators + 124 sums, 37 % of its cycles issuing MFMAs; 290 / 160 / 240 us at
// B = 4096 for 32->64 / 16->32 / 32->16 with the chain); four waves per clip with two passes: 262 / 143 / 217 us at three waves
// per SIMD; one pass: 32->64 -18 us, the train step 1.401 -> 1.376 ms.  Phase costs at 32->64 before the merge (timing-only
// builds, FBB_SKIP): K passes 150 us, temporal mixes 45, dA 45, dT 45, spatial 20, row pass + statistics 35; with every stream
// L2-resident (COSKAD_HOT) the kernel loses 45 us: the fp32 MFMA issue of its ~550 products per wave and clip is half its time.
// Per-workgroup partial sums of dA / dT live in the workspace (summed in a fixed order by k_reduce_fused: deterministic).
#include "fused_ops.h"

namespace coskad {
namespace fb {

using namespace ff;

// A workgroup's partial sums of dA / dT in the workspace, lane-major (one float4 per lane and record: 1 KB per load / store):
//   records [0, 12)   dA[t][4q + r][j]            (t = record)
//           12        dA[t = 4q + r][16][j]       (t < 12)
//           13        dA[t = 4q + r][v = j][16]   (t < 12)
//           14        dA[t = j][16][16]           (r == 0, q == 0, j < 12)
//           [15, 32)  dT[v][4q + r][j]            (v = record - 15; 4q + r < 12, j < 12)
constexpr int PR_A = 0, PR_XA = 12, PR_XB = 13, PR_C = 14, PR_T = 15, PR_N = 32, EROW = PR_N * 256;
// dA, dT (+)= sum over the P lane-major partial rows (fp64, fixed order); one extra block sums the slope partials; blocks beyond
// that one sum the partial rows the data kernel wrote for the layer below (backward chain: brows [bP][bE] -> bout [bE], the
// k_reduce_partials_d of that layer's call riding in this launch)
constexpr int RE = 32;   // columns per block of k_reduce_fused: 1024 / RE row slices (256 + 66 blocks at the default stack's widths)
__global__ __launch_bounds__(1024) void k_reduce_fused(const float* __restrict__ partials, int P, float* __restrict__ dA,
                                                       float* __restrict__ dT, const float* __restrict__ dap, int ndap,
                                                       float* __restrict__ dslope, int accumulate, const float* __restrict__ brows,
                                                       int bP, int bE, double* __restrict__ bout) {
  __shared__ double sh[1024];
  constexpr int NB = EROW / RE;
  const int col = threadIdx.x % RE;
  if ((int)blockIdx.x > NB) {
    const int e = ((int)blockIdx.x - NB - 1) * RE + col;
    const double t = column_sum_f64<RE>(brows, bP, (size_t)bE, e, e < bE, sh);
    if ((int)threadIdx.x < RE && e < bE) bout[e] = t;
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
  const int e = blockIdx.x * RE + col;
  const double t = column_sum_f64<RE>(partials, P, (size_t)EROW, e, true, sh);
  if ((int)threadIdx.x < RE) {
    const int rec = e >> 8, l = (e >> 2) & 63, r = e & 3, j = l & 15, q = l >> 4;
    float* out = nullptr;
    if (rec < PR_XA) out = dA + rec * V * V + (4 * q + r) * V + j;
    else if (rec == PR_XA) { if (4 * q + r < T) out = dA + (4 * q + r) * V * V + 16 * V + j; }
    else if (rec == PR_XB) { if (4 * q + r < T) out = dA + (4 * q + r) * V * V + j * V + 16; }
    else if (rec == PR_C) { if (r == 0 && q == 0 && j < T) out = dA + j * V * V + 16 * V + 16; }
    else if (4 * q + r < T && j < T) out = dT + (rec - PR_T) * T * T + (4 * q + r) * T + j;
    if (out) *out = accumulate ? *out + (float)t : (float)t;
  }
}

// NS != 0: the batch reductions of the layer BELOW (stage 1 of ITS backward: P = sum dU.Z^T, Q = sum dU.X^T, sdU -- k_first_stats /
// k_bwd_stats_ring) are formed here, from the dU rows this kernel has just produced in its image: a re-read of dU_prev and a
// launch fewer.  `below_z` / `below_x` [B, Cb, T, V] (X = PReLU(below_x) with `below_slope`, NULL: raw input), `below_stats`
// [grid][2 Ci Cb + Ci] partial rows as k_bwd_fold reads them.
//   NS = 1: Cb = 2 (a first layer): Z0 Z1 X0 X1 are ONE 4-row operand group
//   NS = 2: Cb = 16 CB: 2 CB groups of 16 rows through the K window, (row, position) operands on both sides
// waves per SIMD: the one-pass kernel needs up to 256 registers (two); 16 input channels without the chain fit three in 168
// (B = 4096, 16 -> 32: 163 vs 172 us; WITH the chain's sums three waves spill 30 registers and the step loses 7 us)
constexpr int bpc_occ(int CT, int NS) { return CT == 1 && NS == 0 ? 3 : 2; }
#ifndef FBB_BDBL   // K passes: operands of k-step s+1 read in front of step s's MFMAs (two register sets) / behind them (one): same speed
#define FBB_BDBL 0
#endif
#ifndef FBB_SKIP   // timing-only builds (wrong results): 1 temporal mixes, 2 K passes, 4 dA, 8 spatial, 16 dT, 32 row pass + statistics
#define FBB_SKIP 0
#endif

template <int CT, int OT, int NS, int CB>
__global__ __launch_bounds__(256, bpc_occ(CT, NS)) void k_layer_bwd_bpc(const float* __restrict__ in, const float* __restrict__ Zg,
                                                               const float* __restrict__ dU, const float* __restrict__ coef,
                                                               const float* __restrict__ btab, const float* __restrict__ in_slope,
                                                               float* __restrict__ dIn, float* __restrict__ partials,
                                                               float* __restrict__ dap, int B, const float* __restrict__ below_z,
                                                               const float* __restrict__ below_x, const float* __restrict__ below_slope,
                                                               float* __restrict__ below_stats) {
  constexpr int Ci = 16 * CT, Co = 16 * OT, CiP = Ci, NG = OT + CT;
  constexpr int KT0 = (Co + Ci) * CiP, DX0 = KT0 + CiP, KR0 = DX0 + (Co + Ci) * CiP;
  constexpr int MAXF = CT == 2 ? 6 : 3;                  // frames per wave (its channel tile); + the joint-16 tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r1 = lds;                 // 32-row image (stride LD)
  float* r2 = lds + 32 * LD;       // 16-row K window (stride LDW; LD in the statistics phase)
  float* zs = lds + WAVE_LDS_W + 256 * (threadIdx.x >> 6);   // 1 KB per wave: the joint-16 dZ tile, lane-major
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  Lane L = geo();
  const float a_in = in_slope[0];
  const int l16 = lane * 16;
  const BufRes tabres = make_res(btab, BTAB_F4 * 16u);
  const BufRes cres = make_res(coef, (KR0 + CiP) * 4u);
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
#ifdef COSKAD_HOT   // timing-only: every stream from 64 L2-resident clips
    return make_res(base + (size_t)(in_range ? (c & 63) : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#else
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#endif
  };
  const int ct = CT == 2 ? (wave & 1) : 0;               // this wave's channel tile
  const int f0 = CT == 2 ? 6 * (wave >> 1) : 3 * wave;   // its frames f0 .. f0 + MAXF - 1
  const bool owns16 = CT == 2 ? (wave >> 1) == 1 : wave == 3;   // stores the joint-16 tile / adds the corner sum
  // staging by all threads: thread t < 204 owns float4 `t` of a quarter (4 rows x 51 float4)
  constexpr int Q4 = 4 * (TV / 4);
  const bool stg = tid < Q4;
  const int srow = tid / (TV / 4), scol = 4 * (tid - srow * (TV / 4));
  const int svoff = stg ? tid * 16 : 0x7ffffff0;
  auto qload = [&](const BufRes& res, int row0, int q) { return buf_load4(res, svoff, (row0 + 4 * q) * (TV / 4) * 16); };
  auto qstore = [&](int q, float4 v, bool act, float slope, int stride) {
    if (act) { v.x = prelu(v.x, slope); v.y = prelu(v.y, slope); v.z = prelu(v.z, slope); v.w = prelu(v.w, slope); }
    *reinterpret_cast<float2*>(r2 + (stg ? (4 * q + srow) * stride + scol : 15 * stride + PADCOL)) = float2{v.x, v.y};
    *reinterpret_cast<float2*>(r2 + (stg ? (4 * q + srow) * stride + scol + 2 : 15 * stride + PADCOL)) = float2{v.z, v.w};
  };
  // the clip's input rows as the threads own them: Ci x 51 float4 = XL per thread
  constexpr int N4 = Ci * (TV / 4), XL = (N4 + 255) / 256;
  float4 xs[XL];
  auto xload = [&](float4 (&dst)[XL], const BufRes& r) {
#pragma unroll
    for (int i = 0; i < XL; ++i) dst[i] = buf_load4(r, (tid + 256 * i) < N4 ? (tid + 256 * i) * 16 : 0x7ffffff0, 0);
  };
  // sums over all the workgroup's clips: this wave's frames of dA, its share of the 17th row / column, its joints of dT
  f32x4 dAacc[MAXF], exA = {0.f, 0.f, 0.f, 0.f}, exB = {0.f, 0.f, 0.f, 0.f};
  constexpr int MAXJ = (V + 3) / 4;                      // joints per wave: v = wave, wave + 4, ..
  f32x4 dTacc[MAXJ];
  float corner = 0.f, da = 0.f;
#pragma unroll
  for (int t = 0; t < MAXF; ++t) dAacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) dTacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (NS) the layer below: this wave's share of the k-steps of [P | Q], row sums
  constexpr int NGB = NS == 2 ? 2 * CB : 1;
  f32x4 nsb[NGB][CT];
  float nss[CT];
#pragma unroll
  for (int g = 0; g < NGB; ++g)
#pragma unroll
    for (int c = 0; c < CT; ++c) nsb[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < CT; ++c) nss[c] = 0.f;
  const bool bpre = NS != 0 && below_slope != nullptr;
  const float a_b = bpre ? below_slope[0] : 0.f;

  // mixing phases of the whole image, dealt to the waves: temporal by joint, spatial by frame (fused_apply_next_bpc.hip)
  auto temporal_rr = [&](int base4) {                    // base4: float4 index of the temporal table's first record
    const Lane Lt = geo();
    TOp cur[CT], nxt[CT];
    f32x4 dprev[CT];
    int vprev = -1;
#pragma unroll
    for (int rt = 0; rt < CT; ++rt) cur[rt] = temporal_read<16>(r1, rt, wave, Lt);
    float4 rec = buf_load4(tabres, l16, (base4 + wave * 64) * 16);
    for (int v = wave; v < V; v += 4) {
      const int vn = v + 4 < V ? v + 4 : v;
      const float4 recn = buf_load4(tabres, l16, (base4 + vn * 64) * 16);
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) nxt[rt] = temporal_read<16>(r1, rt, vn, Lt);
      f32x4 d[CT];
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) d[rt] = temporal_mm(cur[rt], rec);
      if (vprev >= 0) {
#pragma unroll
        for (int rt = 0; rt < CT; ++rt) temporal_store<16>(r1, rt, vprev, dprev[rt], Lt);
      }
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) { dprev[rt] = d[rt]; cur[rt] = nxt[rt]; }
      vprev = v;
      rec = recn;
    }
#pragma unroll
    for (int rt = 0; rt < CT; ++rt) temporal_store<16>(r1, rt, vprev, dprev[rt], Lt);
  };
  auto spatial_rr = [&]() {                              // the adjoint spatial table: btab section behind the forward temporal one
    const Lane Ls = geo();
    SpatRec srec = load_spat(tabres, 0, wave, l16);
    SOp op[CT];
#pragma unroll
    for (int rt = 0; rt < CT; ++rt) op[rt] = spatial_read<16>(r1, rt, wave, Ls);
#pragma unroll
    for (int k = 0; k < T / 4; ++k) {
      const int t = wave + 4 * k, tn = k + 1 < T / 4 ? t + 4 : t;
      const SpatRec nrec = load_spat(tabres, 0, tn, l16);
      SOp opn[CT];
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) opn[rt] = spatial_read<16>(r1, rt, tn, Ls);
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) {
        const f32x4 d = spatial_mm(op[rt], srec);
        spatial_extra<16>(r1, rt, t, op[rt], srec, Ls);
        tile_store(r1, 16 * rt, t * V + Ls.j, true, d, Ls);
      }
      srec = nrec;
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) op[rt] = opn[rt];
    }
  };

  int clip = blockIdx.x;
  float4 gq[4];                                          // K ring: one group in flight, a quarter per register
  {
    xload(xs, clip_res(in, clip, Ci));
    const BufRes du0 = clip_res(dU, clip, Co);
#pragma unroll
    for (int q = 0; q < 4; ++q) gq[q] = qload(du0, 0, q);
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), dures = clip_res(dU, clip, Co);
    const BufRes ores = clip_res(dIn, clip, Ci);
    // group g of a K pass: dU rows first (OT groups), then the pass's second source
    auto kload = [&](int g, int q, const BufRes& res2) { return g < OT ? qload(dures, 16 * g, q) : qload(res2, 16 * (g - OT), q); };
    // ---- stage X = PReLU(U_prev) into the image (fetched during the previous clip) ---------------------------------------------
    __syncthreads();                                     // the previous clip's last readers of the image / window are done
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e4 = tid + 256 * i;
      float4 v = xs[i];
      v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in);
      const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
      // (masked threads write the padding columns of row 0: BOTH halves there -- PADCOL + 2 is row 1's first two positions)
      *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LD + col : PADCOL)) = float2{v.x, v.y};
      *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LD + col + 2 : PADCOL)) = float2{v.z, v.w};
    }
    // first K pass: group 0 (dU rows 0..15, fetched during the previous clip) -> window; group 1 takes the registers
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      qstore(q, gq[q], false, 0.f, LDW);
      gq[q] = kload(1, q, zres);
    }
    __syncthreads();                                     // the image holds X, the window group 0
    // ---- Y = temporal mix of X, in place ------------------------------------------------------------------------------------------
    if (!(FBB_SKIP & 1)) temporal_rr(0);
    // ---- K pass:  acc[tile] += coefficient rows [c0 ..) x dU (OT groups) + rows [c1 ..) x the second source (CT groups) --------
    L = geo();
    const int lq = (L.q * CiP + 16 * ct + L.j) * 4;
    const int jc = L.j < T ? L.j : T - 1;
    auto pos_of = [&](int k) { return k < MAXF ? (f0 + k) * V + L.j : jc * V + 16; };   // tile k of this wave (MAXF: joint 16)
    auto kpass = [&](f32x4 (&acc)[MAXF + 1], f32x4 (&acc2)[MAXF + 1], const BufRes& res2, bool act2, int c0, int c1, int c2,
                     bool with16, bool with16b) {
      // on entry group 0 is staged in the window and group 1 is in the registers.  The dU groups feed BOTH sets of sums: `acc`
      // with coefficient rows [c0 ..) (then the second source's groups with rows [c1 ..)), `acc2` with rows [c2 ..) -- one read
      // of the dU rows for what were two passes in the round-2 kernel
      constexpr bool DUAL = true;
      float wc[2][4], wc2[2][4];
      auto cload = [&](int buf, int g) {
        const int krow = g < OT ? c0 + 16 * g : c1 + 16 * (g - OT);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wc[buf][s] = buf_load1(cres, lq, ((krow + 4 * s) * CiP) * 4);
          if (DUAL && g < OT) wc2[buf][s] = buf_load1(cres, lq, ((c2 + 16 * g + 4 * s) * CiP) * 4);
        }
      };
      cload(0, 0);
      float b[1 + FBB_BDBL][MAXF + 1];
#pragma unroll
      for (int k = 0; k <= MAXF; ++k) b[0][k] = r2[L.q * LDW + pos_of(k)];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) cload((g + 1) & 1, g + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // every wave has read rows 4s .. 4s+3 (a k-step ago); in the last group: its fourth quarter (stored a k-step ago) is visible
          if (g + 1 < NG || s == 0) __syncthreads();
          const int sn = (s + 1) & 3;
          if (FBB_BDBL && (s + 1 < 4 || g + 1 < NG)) {   // the next k-step's operands take off in front of this step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k <= MAXF; ++k) b[(s + 1) & 1][k] = r2[(4 * sn + L.q) * LDW + pos_of(k)];
            __builtin_amdgcn_sched_barrier(0);
          }
          if (g + 1 < NG) {
            qstore(s, gq[s], g + 1 >= OT && act2, a_in, LDW);
            if (g + 2 < NG) gq[s] = kload(g + 2, s, res2);
          }
#pragma unroll
          for (int k = 0; k <= MAXF; ++k)
            if (k < MAXF || with16) acc[k] = mfma(wc[g & 1][s], b[FBB_BDBL ? (s & 1) : 0][k], acc[k]);
          if (DUAL && g < OT) {
#pragma unroll
            for (int k = 0; k <= MAXF; ++k)
              if (k < MAXF || with16b) acc2[k] = mfma(wc2[g & 1][s], b[FBB_BDBL ? (s & 1) : 0][k], acc2[k]);
          }
          if (!FBB_BDBL && (s + 1 < 4 || g + 1 < NG)) {  // single set: the next k-step's operands behind this step's MFMAs
#pragma unroll
            for (int k = 0; k <= MAXF; ++k) b[0][k] = r2[(4 * sn + L.q) * LDW + pos_of(k)];
          }
        }
      }
    };
    f32x4 az[MAXF + 1];
    {
      const float4 a = buf_load4(cres, L.q * 16, (KT0 + 16 * ct) * 4);
#pragma unroll
      for (int k = 0; k <= MAXF; ++k) az[k] = f32x4{a.x, a.y, a.z, a.w};
    }
    f32x4 xr[MAXF + 1];                                  // Br.dU + Kr.X + kr
    {
      const float4 kq = buf_load4(cres, L.q * 16, (KR0 + 16 * ct) * 4);
#pragma unroll
      for (int k = 0; k <= MAXF; ++k) xr[k] = f32x4{kq.x, kq.y, kq.z, kq.w};
    }
    __syncthreads();                                     // the image holds Y (the K pass itself does not touch the image)
    // Bt rows [0, Co), Kt rows [Co, Co + Ci); Br rows [DX0 / CiP ..)
    if (!(FBB_SKIP & 2)) kpass(az, xr, zres, false, 0, Co, DX0 / CiP, true, owns16);
    // dT's first X half takes off behind the dA products (the group registers are free)
#pragma unroll
    for (int q = 0; q < 4; ++q) gq[q] = qload(xres, 0, q);
    // ---- dA += Y^T dZ for this wave's frames and channel tile, dZ over Y --------------------------------------------------------
    if (owns16) {                                        // dA[t = j][16][16]
      const f32x4 y16 = tile_load(r1, 16 * ct, jc * V + 16, L);
#pragma unroll
      for (int r = 0; r < 4; ++r) corner = fmaf(y16[r], az[MAXF][r], corner);
    }
    *reinterpret_cast<f32x4*>(zs + lane * 4) = az[MAXF];  // the joint-16 tile through this wave's 1 KB of LDS: its columns by frame
#pragma unroll
    for (int k = 0; k < ((FBB_SKIP & 4) ? 0 : MAXF); ++k) {
      const int t = f0 + k;
      const f32x4 y = tile_load(r1, 16 * ct, t * V + L.j, L);            // A operand: Y[16 ct + 4q + r][t, v = j]
      const f32x4 y16 = tile_load(r1, 16 * ct, t * V + 16, L);          // Y[..][t, 16] (same address in every column)
      const f32x4 dz16 = *reinterpret_cast<const f32x4*>(zs + (16 * L.q + t) * 4);   // dZ[..][t, 16]: column t of the joint-16 tile
      // the 17th row / column: this lane group's four channels on the VALU, the four groups' shares meet in ONE product with a
      // one-hot row selector (A[i][k] = [i == t]): D[t][j] += sum_q p(j, q)
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dAacc[k] = mfma(y[r], az[k][r], dAacc[k]);
        pa = fmaf(y16[r], az[k][r], pa);                 // dA[t][16][w = j]
        pb = fmaf(y[r], dz16[r], pb);                    // dA[t][v = j][16]
      }
      const float sel = L.j == t ? 1.f : 0.f;
      exA = mfma(sel, pa, exA);
      exB = mfma(sel, pb, exB);
    }
    __syncthreads();                                     // every wave has read Y (frames AND the joint-16 column)
#pragma unroll
    for (int k = 0; k < MAXF; ++k) tile_store(r1, 16 * ct, (f0 + k) * V + L.j, true, az[k], L);   // dZ over Y
    if (owns16) tile_store(r1, 16 * ct, jc * V + 16, L.j < T, az[MAXF], L);
    __syncthreads();                                     // the image holds dZ
    // ---- dY = spatial adjoint of dZ, in place ---------------------------------------------------------------------------------------
    if (!(FBB_SKIP & 8)) spatial_rr();
    __syncthreads();                                     // the image holds dY
    // ---- dT[v] += X_v^T dY_v for this wave's joints: X re-staged 16 rows at a time in the window --------------------------------
    L = geo();
    {
      const int ic = L.j < T ? L.j : T - 1;
#pragma unroll
      for (int h = 0; h < ((FBB_SKIP & 16) ? 0 : CT); ++h) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // X rows 16h .. 16h+15 -> window; the next half / dU group 0 -> registers
          qstore(q, gq[q], true, a_in, LDW);
          if (h + 1 < CT) gq[q] = qload(xres, 16 * (h + 1), q);
        }
        float wk[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) wk[s] = buf_load1(cres, lq, ((DX0 / CiP + Co + 16 * h + 4 * s) * CiP) * 4);   // Kr rows
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (!(FBB_SKIP & 2)) {                         // + Kr.X: the staged half IS the second pass's last group
            float bx[MAXF + 1];
#pragma unroll
            for (int k = 0; k <= MAXF; ++k) bx[k] = r2[(4 * s + L.q) * LDW + pos_of(k)];
#pragma unroll
            for (int k = 0; k <= MAXF; ++k)
              if (k < MAXF || owns16) xr[k] = mfma(wk[s], bx[k], xr[k]);
          }
#pragma unroll
          for (int k = 0; k < MAXJ; ++k) {
            const int v = wave + 4 * k;
            if (v < V) {
              const float a = r2[(4 * s + L.q) * LDW + ic * V + v];
              const float bb = r1[(16 * h + 4 * s + L.q) * LD + ic * V + v];
              dTacc[k] = mfma(L.j < T ? a : 0.f, L.j < T ? bb : 0.f, dTacc[k]);
            }
          }
        }
        __syncthreads();                                 // the window is rewritten next
      }
    }
    float4 u[XL];
    {
      // ---- gcn^T: temporal adjoint in place, + Br.dU + Kr.X + kr (in the registers since the K pass): dX -------------------------
      if (!(FBB_SKIP & 1)) temporal_rr(TEMP_F4 + SPAT_F4);
      __syncthreads();                                   // the image holds gcn^T(dZ)
      xload(u, xres);                                    // the pre-activations come back (from L2) for the row pass
      L = geo();
#pragma unroll
      for (int k = 0; k < MAXF; ++k) {
        const int pos = (f0 + k) * V + L.j;
        tile_store(r1, 16 * ct, pos, true, xr[k] + tile_load(r1, 16 * ct, pos, L), L);
      }
      if (owns16) {
        const int pos = (L.j < T ? L.j : T - 1) * V + 16;
        tile_store(r1, 16 * ct, pos, L.j < T, xr[MAXF] + tile_load(r1, 16 * ct, pos, L), L);
      }
    }
    __syncthreads();                                     // the image holds dX
    // ---- dU_prev = image * PReLU'(U_prev), slope gradient: row-wise, full lines both ways; the next clip's rows take off ---------
    if (!(FBB_SKIP & 32)) {
      const BufRes xn = clip_res(in, clip + gridDim.x, Ci), dun = clip_res(dU, clip + gridDim.x, Co);
      float4 pg[4];
      if constexpr (NS == 1) {                           // the layer below: Z rows 0, 1 and X rows 0, 1 = 2 x 102 float4
        const int fv = tid < 2 * (TV / 4) ? tid * 16 : 0x7ffffff0;
        pg[0] = buf_load4(clip_res(below_z, clip, 2), fv, 0);
        pg[1] = buf_load4(clip_res(below_x, clip, 2), fv, 0);
      }
      if constexpr (NS == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = qload(clip_res(below_z, clip, 16 * CB), 0, q);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = qload(dun, 0, q);
      }
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e4 = tid + 256 * i;
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        float* p = r1 + (e4 < N4 ? row * LD + col : PADCOL);
        float* p2 = r1 + (e4 < N4 ? row * LD + col + 2 : PADCOL);
        const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(p2);
        float g[4] = {g0.x, g0.y, g1.x, g1.y};
        if (e4 < N4) {
          const float uu[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (uu[c] < 0.f) da = fmaf(g[c], uu[c], da);
            g[c] = uu[c] > 0.f ? g[c] : a_in * g[c];
          }
        }
        buf_store4(ores, e4 < N4 ? e4 * 16 : 0x7ffffff0, 0, float4{g[0], g[1], g[2], g[3]});
        if constexpr (NS != 0) {                         // the image keeps dU_prev for the reductions below
          *reinterpret_cast<float2*>(p) = float2{g[0], g[1]};
          *reinterpret_cast<float2*>(p2) = float2{g[2], g[3]};
        }
      }
      xload(xs, xn);
      if constexpr (NS == 1) {
        // ---- the layer below (two input channels): [P | Q] += dU_prev (image rows) x (Z0 Z1 X0 X1)^T (window rows 0..3) ----------
        {
          float4 vz = pg[0], vx = pg[1];
          if (bpre) { vx.x = prelu(vx.x, a_b); vx.y = prelu(vx.y, a_b); vx.z = prelu(vx.z, a_b); vx.w = prelu(vx.w, a_b); }
          const bool fst = tid < 2 * (TV / 4);
          const int frow = tid / (TV / 4), fcol = 4 * (tid - frow * (TV / 4));
          *reinterpret_cast<float2*>(r2 + (fst ? frow * LD + fcol : 3 * LD + PADCOL)) = float2{vz.x, vz.y};
          *reinterpret_cast<float2*>(r2 + (fst ? frow * LD + fcol + 2 : 3 * LD + PADCOL)) = float2{vz.z, vz.w};
          *reinterpret_cast<float2*>(r2 + (fst ? (2 + frow) * LD + fcol : 3 * LD + PADCOL)) = float2{vx.x, vx.y};
          *reinterpret_cast<float2*>(r2 + (fst ? (2 + frow) * LD + fcol + 2 : 3 * LD + PADCOL)) = float2{vx.z, vx.w};
        }
        __syncthreads();                                 // the image holds dU_prev, the window the four rows
        L = geo();
        const float* pb = r2 + (L.j & 3) * LD + 2 * L.q;
        const float* pa = r1 + L.j * LD + 2 * L.q;
        constexpr int NM = (TV + 7) / 8;
        for (int m = wave; m < NM; m += 4) {
          float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
          const bool aok = 8 * m + 2 * L.q < TV, ok = aok && L.j < 4;
          b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            float2 a = *reinterpret_cast<const float2*>(pa + 16 * c * LD + 8 * m);
            a.x = aok ? a.x : 0.f; a.y = aok ? a.y : 0.f;
            nsb[0][c] = mfma(a.x, b.x, nsb[0][c]);
            nsb[0][c] = mfma(a.y, b.y, nsb[0][c]);
            nss[c] += a.x + a.y;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) gq[q] = qload(dun, 0, q);   // (the group registers were not used: the next clip's first dU group)
      }
      if constexpr (NS == 2) {
        // ---- the layer below: [P | Q] += dU_prev (image rows) x group^T (16 window rows at stride LD), k-steps dealt to the waves ----
        auto bres = [&](int g) { return g < CB ? clip_res(below_z, clip, 16 * CB) : clip_res(below_x, clip, 16 * CB); };
        auto brow = [&](int g) { return 16 * (g < CB ? g : g - CB); };
        constexpr int NM = (TV + 7) / 8;
#pragma unroll
        for (int g = 0; g < NGB; ++g) {
          __syncthreads();                               // the row pass's image writes / the previous group's window reads are done
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            qstore(q, gq[q], g >= CB && bpre, a_b, LD);
            gq[q] = g + 1 < NGB ? qload(bres(g + 1), brow(g + 1), q) : qload(dun, 0, q);
          }
          __syncthreads();
          L = geo();
          const float* pb = r2 + L.j * LD + 2 * L.q;
          const float* pa = r1 + L.j * LD + 2 * L.q;
          for (int m = wave; m < NM; m += 4) {
            float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
            const bool ok = 8 * m + 2 * L.q < TV;
            b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              float2 a = *reinterpret_cast<const float2*>(pa + 16 * c * LD + 8 * m);
              a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f;
              nsb[g][c] = mfma(a.x, b.x, nsb[g][c]);
              nsb[g][c] = mfma(a.y, b.y, nsb[g][c]);
              if (g == 0) nss[c] += a.x + a.y;
            }
          }
        }
      }
    }
  }

  // ---- the workgroup's sums: the waves add theirs into ONE lane-major row in LDS one after another (fixed order), then it leaves ----
  __syncthreads();
  float4* row4 = reinterpret_cast<float4*>(lds);         // PR_N records x 64 lanes (32 KB) over the image and the window
  for (int e = tid; e < PR_N * 64; e += 256) row4[e] = float4{0.f, 0.f, 0.f, 0.f};
  float* extra = lds + PR_N * 256;                       // [0] the slope partial, [64 ..) the layer below's row
  constexpr int Cb = NS == 1 ? 2 : 16 * CB;
  constexpr int NSE = NS ? 2 * Ci * Cb + Ci : 0;
  for (int e = tid; e < 64 + NSE; e += 256) extra[e] = 0.f;
  __syncthreads();
  L = geo();
  auto add4 = [&](int rec, const f32x4& v) {
    float4 o = row4[rec * 64 + lane];
    o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
    row4[rec * 64 + lane] = o;
  };
  da = wave_sum(da);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int k = 0; k < MAXF; ++k) add4(PR_A + f0 + k, dAacc[k]);
      add4(PR_XA, exA);
      add4(PR_XB, exB);
      add4(PR_C, f32x4{quad_sum(corner), 0.f, 0.f, 0.f});
#pragma unroll
      for (int k = 0; k < MAXJ; ++k)
        if (wave + 4 * k < V) add4(PR_T + wave + 4 * k, dTacc[k]);
      if (lane == 0) extra[0] += da;
      if constexpr (NS == 1) {                           // [o][Z0 Z1 X0 X1] -> [P Ci x 2][Q Ci x 2][sdU Ci]
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * c + 4 * L.q + r;
            if (L.j < 4) extra[64 + (L.j < 2 ? o * 2 + L.j : 2 * Ci + o * 2 + (L.j - 2))] += nsb[0][c][r];
          }
          const float t = quad_sum(nss[c]);
          if (L.q == 0) extra[64 + 4 * Ci + 16 * c + L.j] += t;
        }
      }
      if constexpr (NS == 2) {                           // [P Ci x Cb][Q Ci x Cb][sdU Ci]
#pragma unroll
        for (int g = 0; g < NGB; ++g)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int o = 16 * c + 4 * L.q + r;
              extra[64 + (g < CB ? 0 : Ci * Cb) + o * Cb + 16 * (g < CB ? g : g - CB) + L.j] += nsb[g][c][r];
            }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float t = quad_sum(nss[c]);
          if (L.q == 0) extra[64 + 2 * Ci * Cb + 16 * c + L.j] += t;
        }
      }
    }
    __syncthreads();
  }
  float4* prow = reinterpret_cast<float4*>(partials + (size_t)blockIdx.x * EROW);
  for (int e = tid; e < PR_N * 64; e += 256) prow[e] = row4[e];
  if (tid == 0 && dap) dap[blockIdx.x] = extra[0];
  if constexpr (NS != 0) {
    for (int e = tid; e < NSE; e += 256) below_stats[(size_t)blockIdx.x * NSE + e] = extra[64 + e];
  }
}

}  // namespace fb

// every workgroup resident: 256 CUs x the waves per SIMD the registers allow
int bwd_bpc_rows(int B, int Ci, bool chain) {
  const int grid = 256 * fb::bpc_occ(Ci / 16, chain ? 1 : 0);
  return B < grid ? B : grid;
}

int launch_layer_bwd_bpc(const float* in, const float* Zg, const float* dU, const float* coef, const float* in_slope, float* dIn,
                         float* btab, float* partials, float* dap, int B, int Ci, int Co, hipStream_t st, int* rows_out,
                         const float* below_z, const float* below_x, const float* below_slope, int below_Ci, float* below_stats) {
  const size_t lds = (size_t)(ff::WAVE_LDS_W + 4 * 256) * sizeof(float);
  const int grid = bwd_bpc_rows(B, Ci, below_stats != nullptr);
  *rows_out = grid;
#define LAUNCH_FBB(CT, OT, NS, CB)                                                                                   \
  do {                                                                                                               \
    auto k = fb::k_layer_bwd_bpc<CT, OT, NS, CB>;                                                                   \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, coef, btab, in_slope, dIn, partials, dap, B,   \
                       below_z, below_x, below_slope, below_stats);                                                  \
  } while (0)
  {
    ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
    if (below_stats) {
      if (Ci == 32 && Co == 16 && below_Ci == 2) LAUNCH_FBB(2, 1, 1, 0);
      else if (Ci == 16 && Co == 32 && below_Ci == 32) LAUNCH_FBB(1, 2, 2, 2);
      else if (Ci == 32 && Co == 64 && below_Ci == 16) LAUNCH_FBB(2, 4, 2, 1);
      else return fail(COSKAD_ERR_SHAPE, "bwd_bpc: no kernel forms the reductions of a %d-channel layer below (%d -> %d)", below_Ci, Ci, Co);
    } else if (Ci == 16 && Co == 16) LAUNCH_FBB(1, 1, 0, 0);
    else if (Ci == 16 && Co == 32) LAUNCH_FBB(1, 2, 0, 0);
    else if (Ci == 16 && Co == 64) LAUNCH_FBB(1, 4, 0, 0);
    else if (Ci == 32 && Co == 16) LAUNCH_FBB(2, 1, 0, 0);
    else if (Ci == 32 && Co == 32) LAUNCH_FBB(2, 2, 0, 0);
    else if (Ci == 32 && Co == 64) LAUNCH_FBB(2, 4, 0, 0);
    else return fail(COSKAD_ERR_SHAPE, "bwd_bpc: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FBB
  return check_launch("bwd_bpc");
}

int launch_reduce_fused(const float* partials, int rows, float* dA, float* dT, const float* dap, float* dslope, int accumulate,
                        hipStream_t st, const float* brows, int bE, double* bout) {
  const int extra = brows ? 1 + ceil_div(bE, fb::RE) : (dap ? 1 : 0);
  hipLaunchKernelGGL(fb::k_reduce_fused, dim3(fb::EROW / fb::RE + extra), dim3(1024), 0, st, partials, rows, dA, dT, dap, rows,
                     dslope, accumulate, brows, rows, bE, bout);
  return check_launch("bwd_reduce_fused");
}

// rows of [2 Ci below_Ci + Ci] floats the data kernel of a (Ci -> Co) layer writes for the layer below it (0: it cannot)
int layer_bwd_below_rows(int T_, int V_, int B, int Ci, int Co, int below_Ci) {
  const bool built = (Ci == 32 && Co == 16 && below_Ci == 2) || (Ci == 16 && Co == 32 && below_Ci == 32) ||
                     (Ci == 32 && Co == 64 && below_Ci == 16);
  if (!(T_ == ff::T && V_ == ff::V && built) || B <= 0) return 0;
  return bwd_bpc_rows(B, Ci, true);
}

bool layer_bwd_fused_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

}  // namespace coskad
