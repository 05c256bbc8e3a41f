// The data path of one ST_GCNN layer's backward on the 25-joint layout, stored-Z path (autograd of
// models/graph_layers/stsgcn.py:94-116 in training mode, behind the batch reductions and the fp64 fold; what k_bwd_data_f,
// stsgcn_bwd.hip, computes) -- ONE CLIP PER WORKGROUP OF FOUR WAVES:
//
//   dZ      = Bt.dU + Kt.Z + kt                          (coefficient matrices from k_bwd_fold; stored for the dA / dT kernel)
//   dX      = gcn^T(dZ) + Br.dU + Kr.X + kr ;  dU_prev = dX * PReLU'(U_prev) ;  dslope_prev = sum dX * U_prev [U_prev < 0]
//
// The two products are fused_apply_flat.hip's K-ring GEMM over FLAT 16-position tiles with two accumulator sets on the rows of
// dU (one read of dU for both, as in fused_bwd.hip): a wave owns one 16-channel tile x half (a quarter at 16 channels) of the 19
// position tiles.  dZ leaves through a 32-row image (full lines) and is mixed there in place -- spatial adjoint by frame,
// temporal adjoint by joint, a wave's mixing operands the same for every clip and held in 63 registers (gcn_params_bpc.hip) --
// then every wave adds its Br.dU + Kr.X tiles and the row pass applies the PReLU mask.  58 KB of LDS per workgroup: two per CU.
// k_bwd_data_f keeps the clip image, both mixing tables (44 KB at 25 joints) and the coefficient matrices in LDS: one 16-wave
// block per CU, every phase behind a block-wide barrier (385 / 273 / 171 us per call at B = 4096).
#include "fused_ops.h"

#ifndef BDB_SKIP   // timing-only builds (wrong results; tools/bd_phases.sh): 1 K-pass products, 2 dA (+ X halves, temporal mix), 4 spatial adjoint, 8 dT,
#define BDB_SKIP 0 // 16 temporal adjoint, 32 row pass.  32 -> 64 at B = 4096: 145 / 66 / 53 / 66 / 26 / 7 us on a 124 us floor of streaming + barriers
#endif
namespace coskad {
namespace bd {

using ff::BufRes;
using ff::buf_load1;
using ff::buf_load4;
using ff::buf_store4;
using ff::f32x4;
using ff::Lane;
using ff::make_res;
using ff::mfma;
using ff::prelu;


constexpr int window_stride(int tv) {
  int l = (tv + 3) / 4 * 4;
  while (l % 64 != 16 && l % 64 != 48) l += 4;
  return l;
}

// CT: 16-row groups of the input; OT: 16-channel tiles of the output (rows of dU); FP: the mixing-parameter gradients are formed
// here as well (dZ never leaves the CU; no k_gcn_params_bpc launch):
//     dA[t] += Yt_t^T dZ_t   (Yt = temporal mix of X, 16 rows at a time through the K window once the GEMM is done with it)
//     dT[v] += X_v^T dYs_v   (dYs = spatial adjoint of dZ; X halves through the window again)
// a wave owns its frames of dA and its joints of dT from the first clip to the last (76 accumulator registers) and writes its part of the
// workgroup's partial row [dA | dT] (`gpart`, summed by k_reduce_gcn).
template <int V, int CT, int OT, bool FP>
__global__ __launch_bounds__(256, 2) void k_bwd_data_bpc(const float* __restrict__ in, const float* __restrict__ Zg,
                                                        const float* __restrict__ dU, const float* __restrict__ Aw,
                                                        const float* __restrict__ Tw, const float* __restrict__ coef,
                                                        const float* __restrict__ in_slope, float* __restrict__ dIn,
                                                        float* __restrict__ dZout, float* __restrict__ dap, int B,
                                                        float* __restrict__ gpart) {
  constexpr int T = 12, TVg = T * V, Ci = 16 * CT, Co = 16 * OT, CiP = Ci;
  // K-pass groups: dU rows, Z rows, the layer input's rows -- FP: the input's rows are staged for dT anyway, Kr.X rides on that staging
  // (fused_bwd.hip does the same) and the K pass is 2 CT groups = 8 CT barriers shorter
  constexpr int NG = FP ? OT + CT : OT + 2 * CT;
  static_assert(TVg % 4 == 0, "rows are staged as float4");
  constexpr int KT0 = (Co + Ci) * CiP, DX0 = KT0 + CiP, KR0 = DX0 + (Co + Ci) * CiP;
  constexpr int R4 = TVg / 4, LDg = TVg + 2, LDWg = window_stride(TVg);
  constexpr int NT = (TVg + 15) / 16;
  constexpr int MAXT = CT == 2 ? (NT + 1) / 2 : (NT + 3) / 4;
  constexpr int Q4 = 4 * R4, NQ = (Q4 + 255) / 256;
  constexpr int N4 = Ci * R4, XL = (N4 + 255) / 256;
  constexpr int NTV = (V + 15) / 16, KV = (V + 3) / 4, MAXF = T / 4, MAXJ = (V + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r2 = lds;                   // 16-row K window (stride LDWg)
  float* r1 = lds + 16 * LDWg;       // 32-row image (stride LDg)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (lane geometry behind an optimisation barrier per phase: its address arithmetic is recomputed there, not held across the K loop)
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  Lane L = geo();
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const BufRes cres = make_res(coef, (KR0 + CiP) * 4u);
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TVg, in_range ? rows * TVg * 4u : 0u);
  };
  const int ct = CT == 2 ? (wave & 1) : 0;
  const int t0 = CT == 2 ? (wave >> 1) * MAXT : wave * MAXT;
  const int nt = NT - t0 < MAXT ? NT - t0 : MAXT;
  float4 gq[4][NQ];
  constexpr bool CARRY = !(FP && CT == 2);
  auto qload = [&](const BufRes& res, int row0, int q, float4 (&dst)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      dst[i] = buf_load4(res, e < Q4 ? e * 16 : 0x7ffffff0, (row0 + 4 * q) * R4 * 16);
    }
  };
  auto qstore = [&](int q, const float4 (&src)[NQ], bool act) {
    int tl = tid;                                        // (behind an optimisation barrier: the staging addresses are recomputed
    asm volatile("" : "+v"(tl));                         //  where they are used instead of being carried through the K loop)
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tl + 256 * i;
      float4 v = src[i];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / R4, col = 4 * (e - row * R4);
      if (e < Q4) *reinterpret_cast<float4*>(r2 + (4 * q + row) * LDWg + col) = v;
    }
  };
  float da = 0.f;
  // (FP) sums over all the workgroup's clips, and a 16-row half of the layer input as the threads own it
  constexpr int H4 = 16 * R4, HL = (H4 + 255) / 256;
  // (32 input channels: the dT sums live in LDS behind the image -- 14.4 KB, two workgroups still fit a CU -- the 28 registers they
  //  took were the ones that spilled)
  constexpr bool TL = FP && CT == 2;
  float* dts = r1 + 32 * LDg;
  if constexpr (TL) {
    for (int e = tid; e < V * T * T; e += 256) dts[e] = 0.f;
  }
  f32x4 accA[FP ? MAXF : 1][NTV][NTV], accT[FP && !TL ? MAXJ : 1];
#pragma unroll
  for (int a = 0; a < (FP ? MAXF : 1); ++a)
#pragma unroll
    for (int b = 0; b < NTV; ++b)
#pragma unroll
      for (int c = 0; c < NTV; ++c) accA[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (FP && !TL ? MAXJ : 1); ++k) accT[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto hload = [&](int clip_, int h, float4 (&dst)[HL]) {
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const float4* g4 = reinterpret_cast<const float4*>(in + ((size_t)clip_ * Ci + 16 * h) * TVg);
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int e = tl + 256 * i;
      dst[i] = e < H4 ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto hstore = [&](const float4 (&src)[HL]) {           // X = PReLU(U_prev) half -> window rows at the IMAGE's stride (row-per-lane reads)
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int e = tl + 256 * i;
      if (e < H4) {
        const int row = e / R4, col = 4 * (e - row * R4);
        float4 v = src[i];
        if (pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        *reinterpret_cast<float2*>(r2 + row * LDg + col) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(r2 + row * LDg + col + 2) = float2{v.z, v.w};
      }
    }
  };
  int clip = blockIdx.x;
  if constexpr (CARRY) {
    const BufRes du0 = clip_res(dU, clip, Co);
#pragma unroll
    for (int q = 0; q < 4; ++q) qload(du0, 0, q, gq[q]);
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), dures = clip_res(dU, clip, Co);
    const BufRes dunext = clip_res(dU, clip + gridDim.x, Co);
    // group g: dU rows (OT groups), then Z's (CT), then the layer input's (CT, PReLU on staging); beyond: the next clip's first
    auto gload = [&](int g, int q, float4 (&dst)[NQ]) {
      if (g < OT) qload(dures, 16 * g, q, dst);
      else if (g < OT + CT) qload(zres, 16 * (g - OT), q, dst);
      else if (g < NG) qload(xres, 16 * (g - OT - CT), q, dst);
      else qload(dunext, 0, q, dst);
    };
    // coefficient rows of group g for the two sums: dZ takes Bt (dU) and Kt (Z), dXres takes Br (dU) and Kr (X)
    const int lq = (L.q * CiP + 16 * ct + L.j) * 4;
    auto cload = [&](int g, float (&w1)[4], float (&w2)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        w1[s] = g < OT + CT ? buf_load1(cres, lq, ((16 * g + 4 * s) * CiP) * 4) : 0.f;                               // Bt rows [0, Co), Kt rows [Co, Co + Ci)
        w2[s] = g < OT ? buf_load1(cres, lq, (DX0 + (16 * g + 4 * s) * CiP) * 4)
                       : (g >= OT + CT ? buf_load1(cres, lq, (DX0 + (Co + 16 * (g - OT - CT) + 4 * s) * CiP) * 4) : 0.f);   // Br / Kr
      }
    };
    auto pos_of = [&](int t) {
      int lj = L.j;
      asm volatile("" : "+v"(lj));
      const int p = 16 * (t0 + (t < nt ? t : 0)) + lj;
      return p < TVg ? p : TVg - 1;
    };
    f32x4 acc1[MAXT], acc2[MAXT];
    {
      const float4 k1 = buf_load4(cres, L.q * 16, (KT0 + 16 * ct) * 4), k2 = buf_load4(cres, L.q * 16, (KR0 + 16 * ct) * 4);
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        acc1[t] = f32x4{k1.x, k1.y, k1.z, k1.w};
        acc2[t] = f32x4{k2.x, k2.y, k2.z, k2.w};
      }
    }
    // entry: the registers hold group 0 (fetched during the previous clip, whose row pass ended with a barrier; !CARRY: fetched here --
    // 32 channels with the dA / dT sums have no registers to carry it through the mixing phases)
    if constexpr (!CARRY) {
#pragma unroll
      for (int q = 0; q < 4; ++q) qload(dures, 0, q, gq[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      qstore(q, gq[q], false);
      gload(1, q, gq[q]);
    }
    float wa[2][4], wb[2][4];
    cload(0, wa[0], wb[0]);
    __syncthreads();                                     // the window holds group 0
    float b[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) b[t] = r2[L.q * LDWg + pos_of(t)];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) cload(g + 1, wa[(g + 1) & 1], wb[(g + 1) & 1]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (g + 1 < NG || s == 0) __syncthreads();
        if (g + 1 < NG) {
          qstore(s, gq[s], g + 1 >= OT + CT && pre);
          if (CARRY || g + 2 < NG) gload(g + 2, s, gq[s]);
        }
        if (g < OT + CT) {
#pragma unroll
          for (int t = 0; t < ((BDB_SKIP & 1) ? 0 : MAXT); ++t) acc1[t] = mfma(wa[g & 1][s], b[t], acc1[t]);
        }
        if (g < OT || g >= OT + CT) {
#pragma unroll
          for (int t = 0; t < ((BDB_SKIP & 1) ? 0 : MAXT); ++t) acc2[t] = mfma(wb[g & 1][s], b[t], acc2[t]);
        }
        if (s + 1 < 4 || g + 1 < NG) {
          const int sn = (s + 1) & 3;
#pragma unroll
          for (int t = 0; t < MAXT; ++t) b[t] = r2[(4 * sn + L.q) * LDWg + pos_of(t)];
        }
      }
    }
      // the B operands of both adjoint mixes (a wave's frames and joints are the same for every clip; fetched from L2 behind the K
      // loop: held for the whole launch they cost the loop 63 registers it does not have at 32 channels)
    //   spatial adjoint   dY[t,v] = sum_w dZ[t,w] A[t][v][w]:   B[k = w][j = v]
    //   temporal adjoint  dX[t,v] = sum_q dY[q,v] T[v][t][q]:   B[k = q][j = t]
    // (the table pointers go through an optimisation barrier per clip: hoisted out of the clip loop the loads would hold their 63
    //  registers through the K loop)
    const float* Awc = Aw;
    const float* Twc = Tw;
    asm volatile("" : "+s"(Awc), "+s"(Twc));
    L = geo();
    float sbv[MAXF][NTV][KV];
    auto load_sbv = [&]() {
#pragma unroll
      for (int tt = 0; tt < MAXF; ++tt) {
        const int t = wave + 4 * tt;
#pragma unroll
        for (int c = 0; c < NTV; ++c)
#pragma unroll
          for (int s = 0; s < KV; ++s)
            sbv[tt][c][s] = (16 * c + L.j < V && 4 * s + L.q < V) ? Awc[(t * V + 16 * c + L.j) * V + 4 * s + L.q] : 0.f;
      }
    };
    float tbv[MAXJ][3];
    auto load_tbv = [&]() {
#pragma unroll
      for (int k = 0; k < MAXJ; ++k) {
        const int v = wave + 4 * k;
#pragma unroll
        for (int s = 0; s < 3; ++s) tbv[k][s] = (v < V && L.j < T) ? Twc[(v * T + L.j) * T + 4 * s + L.q] : 0.f;
      }
    };
    if constexpr (!FP) { load_sbv(); load_tbv(); }       // (FP: fetched where the dA / dT sums leave room for them)
    // ---- dZ -> image (the previous clip's row pass ended with a barrier) -> HBM for the dA / dT kernel ------------------------------
    float4 xh[FP ? HL : 1];
    if constexpr (FP) hload(clip, 0, xh);                // the first X half takes off
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int p = 16 * (t0 + t) + L.j;
      if (t < nt) {
        float* dst = r1 + (16 * ct + 4 * L.q) * LDg + (p < TVg ? p : TVg);
        dst[0] = acc1[t][0]; dst[LDg] = acc1[t][1]; dst[2 * LDg] = acc1[t][2]; dst[3 * LDg] = acc1[t][3];
      }
    }
    __syncthreads();                                     // the image holds dZ (and every wave is done with the K window)
    if constexpr (FP) {
      // ---- dA[t] += Yt_t^T dZ_t: X halves -> window, temporal mix there by joint, products by frame ------------------------------
      //   temporal   Yt[q,v] = sum_t X[t,v] T[v][t][q]:   B[k = t][j = q]
      float tf[MAXJ][3];
#pragma unroll
      for (int k = 0; k < MAXJ; ++k) {
        const int v = wave + 4 * k;
#pragma unroll
        for (int s = 0; s < 3; ++s) tf[k][s] = (v < V && L.j < T) ? Twc[(v * T + 4 * s + L.q) * T + L.j] : 0.f;
      }
#pragma unroll
      for (int h = 0; h < ((BDB_SKIP & 2) ? 0 : CT); ++h) {
        hstore(xh);
        if (h + 1 < CT) hload(clip, h + 1, xh);
        __syncthreads();                                 // the window holds X rows 16 h ..
#pragma unroll
        for (int k = 0; k < MAXJ; ++k) {
          const int v = wave + 4 * k;
          if (v < V) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) d = mfma(r2[L.j * LDg + (4 * s + L.q) * V + v], tf[k][s], d);
            if (L.j < T) {
#pragma unroll
              for (int r = 0; r < 4; ++r) r2[(4 * L.q + r) * LDg + L.j * V + v] = d[r];
            }
          }
        }
        __syncthreads();                                 // the window holds Yt
#pragma unroll
        for (int tt = 0; tt < MAXF; ++tt) {
          const int t = wave + 4 * tt;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int row = 4 * s + L.q;
            float a[NTV], b[NTV];
#pragma unroll
            for (int c = 0; c < NTV; ++c) {
              const bool ok = 16 * c + L.j < V;
              a[c] = ok ? r2[row * LDg + t * V + 16 * c + L.j] : 0.f;
              b[c] = ok ? r1[(16 * h + row) * LDg + t * V + 16 * c + L.j] : 0.f;
            }
#pragma unroll
            for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
              for (int tb2 = 0; tb2 < NTV; ++tb2) accA[tt][ta][tb2] = mfma(a[ta], b[tb2], accA[tt][ta][tb2]);
          }
        }
        if (h + 1 < CT) __syncthreads();                 // Yt's readers are done: the next half over it
      }
      hload(clip, 0, xh);                                // dT's first X half takes off behind the spatial adjoint
      load_sbv();
    }
    if constexpr (!FP) {
      float4* g4 = reinterpret_cast<float4*>(dZout + (size_t)clip * Ci * TVg);
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e = tid + 256 * i;
        if (e < N4) {
          const int row = e / R4, col = 4 * (e - row * R4);
          const float2 g0 = *reinterpret_cast<const float2*>(r1 + row * LDg + col);
          const float2 g1 = *reinterpret_cast<const float2*>(r1 + row * LDg + col + 2);
          g4[e] = float4{g0.x, g0.y, g1.x, g1.y};
        }
      }
    }
    if constexpr (!FP) __syncthreads();                  // the rows have left: the image may be mixed in place
    // ---- dY = spatial adjoint of dZ, in place: frames t = wave, wave + 4, wave + 8 (a frame is touched by its owner only) -------
#pragma unroll
    for (int tt = 0; tt < ((BDB_SKIP & 4) ? 0 : MAXF); ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) {
        float a[KV];
#pragma unroll
        for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? r1[(16 * rt + L.j) * LDg + t * V + 4 * s + L.q] : 0.f;
        f32x4 d[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], sbv[tt][c][s], d[c]);
        }
#pragma unroll
        for (int c = 0; c < NTV; ++c)
          if (16 * c + L.j < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[(16 * rt + 4 * L.q + r) * LDg + t * V + 16 * c + L.j] = d[c][r];
          }
      }
    }
    __syncthreads();                                     // the image holds dY (FP: and every wave has read Yt in the window)
    if constexpr (FP) {
      // ---- dT[v] += X_v^T dYs_v for this wave's joints: X halves through the window ---------------------------------------------
#pragma unroll
      for (int h = 0; h < ((BDB_SKIP & 8) ? 0 : CT); ++h) {
        hstore(xh);
        if (h + 1 < CT) hload(clip, h + 1, xh);
        float wk[4];                                     // Kr rows of this half for the wave's channel tile
#pragma unroll
        for (int s = 0; s < 4; ++s) wk[s] = buf_load1(cres, lq, (DX0 + (Co + 16 * h + 4 * s) * CiP) * 4);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < ((BDB_SKIP & 1) ? 0 : 4); ++s) {  // + Kr.X: the staged half IS the K pass's last group
          float bx[MAXT];
#pragma unroll
          for (int t = 0; t < MAXT; ++t) bx[t] = r2[(4 * s + L.q) * LDg + pos_of(t)];
#pragma unroll
          for (int t = 0; t < MAXT; ++t) acc2[t] = mfma(wk[s], bx[t], acc2[t]);
        }
#pragma unroll
        for (int k = 0; k < MAXJ; ++k) {
          const int v = wave + 4 * k;
          if (v < V) {
            f32x4 acc = TL ? f32x4{0.f, 0.f, 0.f, 0.f} : accT[TL ? 0 : k];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int row = 4 * s + L.q;
              const float a = L.j < T ? r2[row * LDg + L.j * V + v] : 0.f;
              const float b = L.j < T ? r1[(16 * h + row) * LDg + L.j * V + v] : 0.f;
              acc = mfma(a, b, acc);
            }
            if constexpr (TL) {                          // (this wave's joint: nobody else touches its 144 sums)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (4 * L.q + r < T && L.j < T) dts[(v * T + 4 * L.q + r) * T + L.j] += acc[r];
            } else {
              accT[k] = acc;
            }
          }
        }
        if (h + 1 < CT) __syncthreads();                 // the half's readers are done
      }
      load_tbv();
    }
    // ---- temporal adjoint, in place: joints v = wave, wave + 4, .. ---------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < ((BDB_SKIP & 16) ? 0 : MAXJ); ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
#pragma unroll
        for (int rt = 0; rt < CT; ++rt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) d = mfma(r1[(16 * rt + L.j) * LDg + (4 * s + L.q) * V + v], tbv[k][s], d);
          if (L.j < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[(16 * rt + 4 * L.q + r) * LDg + L.j * V + v] = d[r];
          }
        }
      }
    }
    __syncthreads();                                     // the image holds gcn^T(dZ)
    // ---- + Br.dU + Kr.X + kr: every wave adds its tiles in place ---------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int p = 16 * (t0 + t) + L.j;
      if (t < nt && p < TVg) {
        float* dst = r1 + (16 * ct + 4 * L.q) * LDg + p;
        dst[0] += acc2[t][0]; dst[LDg] += acc2[t][1]; dst[2 * LDg] += acc2[t][2]; dst[3 * LDg] += acc2[t][3];
      }
    }
    __syncthreads();                                     // the image holds dX
    float4 u[XL];
    {
      const float4* g4 = reinterpret_cast<const float4*>(in + (size_t)clip * Ci * TVg);
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e = tid + 256 * i;
        u[i] = e < N4 ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};   // the pre-activations come back (from L2) for the row pass
      }
    }
    // ---- dU_prev = dX * PReLU'(U_prev), slope gradient: row-wise, full lines ------------------------------------------------------
    {
      float4* g4 = reinterpret_cast<float4*>(dIn + (size_t)clip * Ci * TVg);
#pragma unroll
      for (int i = 0; i < ((BDB_SKIP & 32) ? 0 : XL); ++i) {
        const int e = tid + 256 * i;
        if (e < N4) {
          const int row = e / R4, col = 4 * (e - row * R4);
          const float2 g0 = *reinterpret_cast<const float2*>(r1 + row * LDg + col);
          const float2 g1 = *reinterpret_cast<const float2*>(r1 + row * LDg + col + 2);
          float g[4] = {g0.x, g0.y, g1.x, g1.y};
          if (pre) {
            const float uu[4] = {u[i].x, u[i].y, u[i].z, u[i].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (uu[c] < 0.f) da = fmaf(g[c], uu[c], da);
              g[c] = uu[c] > 0.f ? g[c] : a_in * g[c];
            }
          }
          g4[e] = float4{g[0], g[1], g[2], g[3]};
        }
      }
    }
    __syncthreads();                                     // (the image and the window are rewritten next)
  }
  if constexpr (FP) {
    // every wave owns its frames of dA and its joints of dT: its part of the workgroup's partial row [dA | dT]
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
    if constexpr (TL) {
      __syncthreads();
      for (int e = tid; e < V * T * T; e += 256) dstT[e] = dts[e];
    } else {
#pragma unroll
      for (int k = 0; k < MAXJ; ++k) {
        const int v = wave + 4 * k;
        if (v < V) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t1 = 4 * L.q + r, t2 = L.j;
            if (t1 < T && t2 < T) dstT[(v * T + t1) * T + t2] = accT[TL ? 0 : k][r];
          }
        }
      }
    }
  }
  if (dap) {
    __shared__ float sred[4];
    da = wave_sum(da);
    if (lane == 0) sred[wave] = da;
    __syncthreads();
    if (tid == 0) dap[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
  }
}

}  // namespace bd

bool bwd_data_bpc_ok(int T_, int V_, int Ci, int Co) {
  return T_ == 12 && V_ == 25 && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

// dZout, dIn: [B, Ci, T, V]; dap: >= *rows_out (<= 512) floats (NULL: no slope gradient)
// gpart != NULL: dA / dT are formed here too, >= *rows_out rows of T V V + V T T floats (dZout is not written)
int launch_bwd_data_bpc(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw, const float* coef,
                        const float* in_slope, float* dIn, float* dZout, float* dap, int B, int Ci, int Co, int T_, int V_,
                        hipStream_t st, int* rows_out, float* gpart) {
  if (!bwd_data_bpc_ok(T_, V_, Ci, Co) || !Zg || !dIn || !(dZout || gpart))
    return fail(COSKAD_ERR_SHAPE, "bwd_data_bpc: built for 12 x 25, 16 / 32 -> 16 / 32 / 64 channels, stored Z");
  constexpr int V = 25;
  const size_t lds = (size_t)(16 * bd::window_stride(12 * V) + 32 * (12 * V + 2) + (gpart && Ci == 32 ? V * 12 * 12 : 0)) * sizeof(float);
  const int grid = B < 512 ? B : 512;
  *rows_out = grid;
#define LAUNCH_BD(CT, OT)                                                                                        \
  do {                                                                                                           \
    if (gpart && lds > 64 * 1024)                                                                                \
      (void)hipFuncSetAttribute((const void*)bd::k_bwd_data_bpc<V, CT, OT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (gpart)                                                                                                   \
      hipLaunchKernelGGL((bd::k_bwd_data_bpc<V, CT, OT, true>), dim3(grid), dim3(256), lds, st, in, Zg, dU, Aw, Tw, coef, in_slope, \
                         dIn, dZout, dap, B, gpart);                                                             \
    else                                                                                                         \
      hipLaunchKernelGGL((bd::k_bwd_data_bpc<V, CT, OT, false>), dim3(grid), dim3(256), lds, st, in, Zg, dU, Aw, Tw, coef, in_slope, \
                         dIn, dZout, dap, B, gpart);                                                             \
  } while (0)
  if (Ci == 16 && Co == 16) LAUNCH_BD(1, 1);
  else if (Ci == 16 && Co == 32) LAUNCH_BD(1, 2);
  else if (Ci == 16 && Co == 64) LAUNCH_BD(1, 4);
  else if (Ci == 32 && Co == 16) LAUNCH_BD(2, 1);
  else if (Ci == 32 && Co == 32) LAUNCH_BD(2, 2);
  else LAUNCH_BD(2, 4);
#undef LAUNCH_BD
  return check_launch("bwd_data_bpc");
}

}  // namespace coskad
