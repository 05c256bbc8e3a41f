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
//           [12, 15)  dA[t = 4c + r][16][j]       (q == 0 lanes)
//           [15, 20)  dA[t = j][v = 4c + r][16]   (q == 0, j < 12 lanes; v < 17)
//           [20, 37)  dT[v][4q + r][j]            (v = record - 20; 4q + r < 12, j < 12)
constexpr int PR_A = 0, PR_XA = 12, PR_XB = 15, PR_T = 20, PR_N = 37, EROW = PR_N * 256;
constexpr int SPAT_F4 = T * 3 * 64;                  // float4 records of one spatial section
constexpr int BTAB_F4 = 2 * TEMP_F4 + SPAT_F4;       // [forward temporal][adjoint spatial][adjoint temporal]

// operand streams of one layer from its A [T,V,V] and T [V,T,T] (lane l: j = l & 15, q = l >> 4):
//   forward temporal  rec[v][l][s]      = T[v][4s+q][j]          (j < 12)
//   adjoint spatial   rec[t][l][0..4]   = A[t][j][4s+q]          (4s+q < 17),  [5..9] = A[t][16][4s+q]
//   adjoint temporal  rec[v][l][s]      = T[v][j][4s+q]          (j < 12)
__global__ void k_build_btab(const float* __restrict__ Aw, const float* __restrict__ Tw, float* __restrict__ tab) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BTAB_F4 * 4) return;
  float val = 0.f;
  if (e < TEMP_F4 * 4 || e >= (TEMP_F4 + SPAT_F4) * 4) {
    const bool adj = e >= TEMP_F4 * 4;
    const int r = adj ? e - (TEMP_F4 + SPAT_F4) * 4 : e;
    const int v = r / 256, l = (r >> 2) & 63, s = r & 3, j = l & 15, q = l >> 4;
    if (s < 3 && j < T) val = adj ? Tw[v * T * T + j * T + 4 * s + q] : Tw[v * T * T + (4 * s + q) * T + j];
  } else {
    const int r = e - TEMP_F4 * 4;
    const int t = r / (3 * 256), c = (r / 256) % 3, l = (r >> 2) & 63, k = 4 * c + (r & 3), j = l & 15, q = l >> 4;
    if (k < 10) {
      const int s = k < 5 ? k : k - 5, w = 4 * s + q;
      if (w < V) val = Aw[t * V * V + (k < 5 ? j : 16) * V + w];
    }
  }
  tab[e] = val;
}

// dA, dT (+)= sum over the P lane-major partial rows (fp64, fixed order); one extra block sums the slope partials
__global__ __launch_bounds__(1024) void k_reduce_fused(const float* __restrict__ partials, int P, float* __restrict__ dA,
                                                       float* __restrict__ dT, const float* __restrict__ dap, int ndap,
                                                       float* __restrict__ dslope, int accumulate) {
  __shared__ double sh[1024];
  constexpr int NB = EROW / 64;
  if ((int)blockIdx.x == NB) {
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
    else if (rec < PR_XB) { if (q == 0) out = dA + (4 * (rec - PR_XA) + r) * V * V + 16 * V + j; }
    else if (rec < PR_T) { const int v = 4 * (rec - PR_XB) + r; if (q == 0 && j < T && v < V) out = dA + j * V * V + v * V + 16; }
    else if (4 * q + r < T && j < T) out = dT + (rec - PR_T) * T * T + (4 * q + r) * T + j;
    if (out) *out = accumulate ? *out + (float)t : (float)t;
  }
}

#ifndef FB_ABLATE
#define FB_ABLATE 0   // timing-only builds (tools/ab_fused.sh): 1 staging + forward temporal, 2 position tiles, 4 dA extras + flush,
#endif                //   8 spatial adjoint, 16 dT, 32 temporal adjoint, 64 epilogue, 128 K-group GEMM only, 256 dA column 16, 512 dA main

template <int CT, int OT>
__global__ __launch_bounds__(256, 1) void k_layer_bwd_fused(const float* __restrict__ in, const float* __restrict__ Zg,
                                                           const float* __restrict__ dU, const float* __restrict__ coef,
                                                           const float* __restrict__ btab, const float* __restrict__ in_slope,
                                                           float* __restrict__ dIn, float* __restrict__ partials,
                                                           float* __restrict__ dap, float* __restrict__ xscr, int B) {
  constexpr int Ci = 16 * CT, Co = 16 * OT, CiP = Ci;
  constexpr int KT0 = (Co + Ci) * CiP, DX0 = KT0 + CiP, KR0 = DX0 + (Co + Ci) * CiP;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = lds_all + wave * WAVE_LDS;
  float* r1 = lds + R1;
  float* r2 = lds + R2;
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  Lane L = geo();
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int l16 = lane * 16;
  const BufRes tabres = make_res(btab, BTAB_F4 * 16u);
  const BufRes cres = make_res(coef, (KR0 + CiP) * 4u);
  const BufRes pres = make_res(partials + (size_t)(blockIdx.x * 4 + wave) * EROW, EROW * 4u);
  // dXres = Br.dU + Kr.X + kr waits for the adjoint mixing outside the register file: a tile-major slab of this wave in the
  // workspace (26 KB, L2-resident; 1 KB per store / load), which frees 104 registers across three phases
  const BufRes sres = make_res(xscr + (size_t)(blockIdx.x * 4 + wave) * (NTILE * 2 * 256), NTILE * 2 * 256 * 4u);
  const int nwaves = gridDim.x * 4;
  float da = 0.f;
  int clip = blockIdx.x * 4 + wave;
#pragma unroll
  for (int k = 0; k < PR_N; ++k) buf_store4(pres, l16, k * 1024, float4{0.f, 0.f, 0.f, 0.f});   // this wave's partial row starts at 0

  for (; clip < B; clip += nwaves) {
    const BufRes xres = make_res(in + (size_t)clip * Ci * TV, Ci * TV * 4u);
    const BufRes zres = make_res(Zg + (size_t)clip * Ci * TV, Ci * TV * 4u);
    const BufRes dures = make_res(dU + (size_t)clip * Co * TV, Co * TV * 4u);
    const BufRes ores = make_res(dIn + (size_t)clip * Ci * TV, Ci * TV * 4u);

    // ---- stage X = PReLU(U_prev) rows [row0, row0 + nrows) into an image (float4 loads: a row is 51 float4) ----------
    auto stage = [&](float* img, int row0, int nrows) {
      const int n4 = nrows * (TV / 4);
#pragma unroll
      for (int i = 0; i < (16 * CT * (TV / 4) + 63) / 64; ++i) {
        if (i * 64 < n4) {
          const int e4 = lane + 64 * i;
          float4 v = buf_load4(xres, l16, (row0 * (TV / 4) + 64 * i) * 16);
          if (pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
          const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
          const bool ok = e4 < n4;
          float* d0 = img + (ok ? row * LD + col : PADCOL);
          float* d1 = img + (ok ? row * LD + col + 2 : PADCOL);
          *reinterpret_cast<float2*>(d0) = float2{v.x, v.y};
          *reinterpret_cast<float2*>(d1) = float2{v.z, v.w};
        }
      }
    };
    L = geo();
    TTab tt;
    load_ttab(tt, tabres, 0, l16);
    if (!(FB_ABLATE & 1)) stage(r1, 0, Ci);

    // ---- Y = temporal mix of X, in place -----------------------------------------------------------------------------
    L = geo();
    if (!(FB_ABLATE & 1)) temporal_phase<16, CT>(r1, tt, L);

    // ---- coefficient matrices as A operands (lane: output channel 16 ct + j, k slot q) and bias quads ------------------
    L = geo();
    f32x4 ktq[CT], krq[CT];
    const int lq = (L.q * CiP + L.j) * 4;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float4 a = buf_load4(cres, L.q * 16, (KT0 + 16 * ct) * 4), b = buf_load4(cres, L.q * 16, (KR0 + 16 * ct) * 4);
      ktq[ct] = f32x4{a.x, a.y, a.z, a.w};
      krq[ct] = f32x4{b.x, b.y, b.z, b.w};
    }
    // coefficient A operands of one 16-row group (lane: output channel 16 ct + j, k slot q), fetched one group ahead:
    // wz multiplies into dZ, wx into dXres (dU groups feed both, Z groups only dZ, X groups only dXres)
    float wz[2][4][CT], wx[2][4][CT];
    auto cload = [&](int buf, int g) {
      constexpr int OTc = OT, CTc = CT;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          if (g < OTc) {
            wz[buf][s][ct] = buf_load1(cres, lq, ((16 * g + 4 * s) * CiP + 16 * ct) * 4);
            wx[buf][s][ct] = buf_load1(cres, lq, (DX0 + (16 * g + 4 * s) * CiP + 16 * ct) * 4);
          } else if (g < OTc + CTc) {
            wz[buf][s][ct] = buf_load1(cres, lq, ((Co + 16 * (g - OTc) + 4 * s) * CiP + 16 * ct) * 4);
          } else {
            wx[buf][s][ct] = buf_load1(cres, lq, (DX0 + (Co + 16 * (g - OTc - CTc) + 4 * s) * CiP + 16 * ct) * 4);
          }
        }
    };

    // ---- dZ = Bt.dU + Kt.Z + kt,  dXres = Br.dU + Kr.X + kr for ALL position tiles at once --------------------------------
    // The K axis (rows of dU, Z, X) is walked in groups of 16 rows: a group is staged into R2 by full-line float4 loads
    // (every byte of dU / Z / X is read once, aligned), then feeds 4 k-steps x 13 tiles x CT (x 2) independent MFMA
    // chains from LDS.  The next group's loads are in flight while the current one is multiplied.
    const int jc = L.j < T ? L.j : T - 1;
    f32x4 az[NTILE][CT], xr[NTILE][CT];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { az[t][ct] = ktq[ct]; xr[t][ct] = krq[ct]; }
    constexpr int G4 = 16 * (TV / 4);                    // float4 of a 16-row group (816)
    constexpr int GL = (G4 + 63) / 64;                   // per lane (13)
    float4 gbuf[GL];
    auto gload = [&](BufRes res, int row0) {
#pragma unroll
      for (int i = 0; i < GL; ++i) gbuf[i] = buf_load4(res, l16, (row0 * (TV / 4) + 64 * i) * 16);
    };
    auto gstore = [&](bool act) {
#pragma unroll
      for (int i = 0; i < GL; ++i) {
        const int e4 = lane + 64 * i;
        float4 v = gbuf[i];
        if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        const bool ok = e4 < G4;
        *reinterpret_cast<float2*>(r2 + (ok ? row * LD + col : PADCOL)) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(r2 + (ok ? row * LD + col + 2 : PADCOL)) = float2{v.z, v.w};
      }
    };
    if (!(FB_ABLATE & (2 | 128))) {
      constexpr int NG = OT + 2 * CT;                    // groups: dU (OT), Z (CT), X (CT)
      gload(dures, 0);
      cload(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        gstore(g >= OT + CT && pre);
        if (g + 1 < NG) {
          if (g + 1 < OT) gload(dures, 16 * (g + 1));
          else if (g + 1 < OT + CT) gload(zres, 16 * (g + 1 - OT));
          else gload(xres, 16 * (g + 1 - OT - CT));
          cload((g + 1) & 1, g + 1);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float b[NTILE];
#pragma unroll
          for (int t = 0; t < NTILE; ++t) b[t] = r2[(4 * s + L.q) * LD + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              if (g < OT) {
                az[t][ct] = mfma(wz[g & 1][s][ct], b[t], az[t][ct]);
                xr[t][ct] = mfma(wx[g & 1][s][ct], b[t], xr[t][ct]);
              } else if (g < OT + CT) {
                az[t][ct] = mfma(wz[g & 1][s][ct], b[t], az[t][ct]);
              } else {
                xr[t][ct] = mfma(wx[g & 1][s][ct], b[t], xr[t][ct]);
              }
            }
        }
      }
    }
    if (!(FB_ABLATE & 2)) {
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          buf_store4(sres, l16, (t * 2 + ct) * 1024, float4{xr[t][ct][0], xr[t][ct][1], xr[t][ct][2], xr[t][ct][3]});
    }
    float exB[V] = {};
    f32x4 dAacc[T];
    float exA[T];
    if (FB_ABLATE) {
#pragma unroll
      for (int t = 0; t < T; ++t) { dAacc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; exA[t] = 0.f; }
    }
    if (!(FB_ABLATE & 2)) {
    {
      // the 17th-joint tile first: dA[t = j][v][16] = sum_c Y[c][t, v] dZ[c][t, 16] needs Y intact
#pragma unroll
      for (int v = 0; v < ((FB_ABLATE & 256) ? 0 : V); ++v) {
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const f32x4 y = tile_load(r1, 16 * ct, jc * V + v, L);
#pragma unroll
          for (int r = 0; r < 4; ++r) s = fmaf(y[r], az[T][ct][r], s);
        }
        exB[v] = (FB_ABLATE & 4) ? s : quad_sum(s);
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      float s16 = 0.f;
#pragma unroll
      for (int ct = 0; ct < ((FB_ABLATE & 512) ? 0 : CT); ++ct) {
        const f32x4 y = tile_load(r1, 16 * ct, t * V + L.j, L);          // A operand: Y[16 ct + 4q + r][t, v = j]
        const f32x4 y16 = tile_load(r1, 16 * ct, t * V + 16, L);        // Y[..][t, 16] (same address in every column)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc = mfma(y[r], az[t][ct][r], acc);
          s16 = fmaf(y16[r], az[t][ct][r], s16);
        }
      }
      dAacc[t] = acc;
      exA[t] = quad_sum(s16);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) tile_store(r1, 16 * ct, t * V + L.j, true, az[t][ct], L);   // dZ over Y's frame t
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) tile_store(r1, 16 * ct, jc * V + 16, L.j < T, az[T][ct], L);
    // dA partial sums of this wave: read-modify-write of its lane-major row (20 x 1 KB each way, loads batched)
    if (!(FB_ABLATE & 4)) {
      float4 pv[PR_T];
#pragma unroll
      for (int k = 0; k < PR_T; ++k) pv[k] = buf_load4(pres, l16, k * 1024);
#pragma unroll
      for (int t = 0; t < T; ++t)
        buf_store4(pres, l16, (PR_A + t) * 1024,
                   float4{pv[t].x + dAacc[t][0], pv[t].y + dAacc[t][1], pv[t].z + dAacc[t][2], pv[t].w + dAacc[t][3]});
#pragma unroll
      for (int c = 0; c < 3; ++c)
        buf_store4(pres, l16, (PR_XA + c) * 1024, float4{pv[PR_XA + c].x + exA[4 * c], pv[PR_XA + c].y + exA[4 * c + 1],
                                                         pv[PR_XA + c].z + exA[4 * c + 2], pv[PR_XA + c].w + exA[4 * c + 3]});
#pragma unroll
      for (int c = 0; c < 5; ++c)
        buf_store4(pres, l16, (PR_XB + c) * 1024,
                   float4{pv[PR_XB + c].x + exB[4 * c], pv[PR_XB + c].y + (4 * c + 1 < V ? exB[4 * c + 1 < V ? 4 * c + 1 : 0] : 0.f),
                          pv[PR_XB + c].z + (4 * c + 2 < V ? exB[4 * c + 2 < V ? 4 * c + 2 : 0] : 0.f),
                          pv[PR_XB + c].w + (4 * c + 3 < V ? exB[4 * c + 3 < V ? 4 * c + 3 : 0] : 0.f)});
    }
    }   // FB_ABLATE & 2

    // ---- dY = spatial adjoint of dZ, in place (operand reads of frame t+1 before the stores of frame t) ----------------
    L = geo();
    if (!(FB_ABLATE & 8)) {
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

    // ---- dT[v] = X_v^T dY_v: X re-staged 16 rows at a time beside the image --------------------------------------------
    L = geo();
    f32x4 dTacc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) dTacc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(FB_ABLATE & 16)) {
      const int ic = L.j < T ? L.j : T - 1;
      gload(xres, 0);
#pragma unroll
      for (int h = 0; h < CT; ++h) {
        gstore(pre);                                   // X rows 16h .. 16h+15 (fetched before the spatial adjoint / the previous half)
        if (h + 1 < CT) gload(xres, 16 * (h + 1));
#pragma unroll
        for (int s = 0; s < 4; ++s) {                  // 17 independent chains per k-step
          float a[V], b[V];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            a[v] = r2[(4 * s + L.q) * LD + ic * V + v];
            b[v] = r1[(16 * h + 4 * s + L.q) * LD + ic * V + v];
          }
#pragma unroll
          for (int v = 0; v < V; ++v) dTacc[v] = mfma(L.j < T ? a[v] : 0.f, L.j < T ? b[v] : 0.f, dTacc[v]);
        }
      }
    }
    if (!(FB_ABLATE & 16)) {                           // dT partial sums: 17 x 1 KB each way
      float4 pv[V];
#pragma unroll
      for (int v = 0; v < V; ++v) pv[v] = buf_load4(pres, l16, (PR_T + v) * 1024);
#pragma unroll
      for (int v = 0; v < V; ++v)
        buf_store4(pres, l16, (PR_T + v) * 1024,
                   float4{pv[v].x + dTacc[v][0], pv[v].y + dTacc[v][1], pv[v].z + dTacc[v][2], pv[v].w + dTacc[v][3]});
    }

    // ---- gcn^T: temporal adjoint in place --------------------------------------------------------------------------------
    L = geo();
    load_ttab(tt, tabres, TEMP_F4 + SPAT_F4, l16);
    if (!(FB_ABLATE & 32)) temporal_phase<16, CT>(r1, tt, L);

    // ---- dU_prev = (gcn^T(dZ) + dXres) * PReLU'(U_prev), slope gradient ---------------------------------------------------
    // dXres joins the image tile by tile (LDS only); the image then leaves row-wise: float4 loads of the pre-activations
    // (the PReLU mask) and float4 stores of dU_prev, full lines both ways.
    L = geo();
    if (!(FB_ABLATE & 64)) {
      f32x4 xq[NTILE][CT];
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float4 v = buf_load4(sres, l16, (t * 2 + ct) * 1024);
          xq[t][ct] = f32x4{v.x, v.y, v.z, v.w};
        }
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        const bool fr = tile < T;
        const int pos = fr ? tile * V + L.j : jc * V + 16;
        const bool ok = fr || L.j < T;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const f32x4 g = tile_load(r1, 16 * ct, pos, L) + xq[tile][ct];
          tile_store(r1, 16 * ct, pos, ok, g, L);
        }
      }
      constexpr int N4 = Ci * (TV / 4), NL = (N4 + 63) / 64;
#pragma unroll
      for (int i0 = 0; i0 < NL; i0 += 7) {
        float4 u[7];
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (i0 + k < NL) u[k] = pre ? buf_load4(xres, l16, 64 * (i0 + k) * 16) : float4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int k = 0; k < 7; ++k)
          if (i0 + k < NL) {
            const int e4 = lane + 64 * (i0 + k);
            const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
            const float* p = r1 + (e4 < N4 ? row * LD + col : PADCOL);
            const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(p + 2);
            float g[4] = {g0.x, g0.y, g1.x, g1.y};
            const float uu[4] = {u[k].x, u[k].y, u[k].z, u[k].w};
            if (pre && e4 < N4) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (uu[c] < 0.f) da = fmaf(g[c], uu[c], da);
                g[c] = uu[c] > 0.f ? g[c] : a_in * g[c];
              }
            }
            buf_store4(ores, l16, 64 * (i0 + k) * 16, float4{g[0], g[1], g[2], g[3]});    // beyond the clip: dropped (bounds check)
          }
      }
    }
  }
  da = wave_sum(da);
  if (lane == 0 && dap) dap[blockIdx.x * 4 + wave] = da;
}

}  // namespace fb

// stage 3 + 4 of launch_layer_bwd for the shapes this kernel is built for; partials: >= 4 * grid rows of T*V*V + V*T*T floats
int launch_layer_bwd_fused(const float* in, const float* Zg, const float* dU, const float* Aw, const float* Tw,
                           const float* coef, const float* in_slope, float* dIn, float* btab, float* partials, float* dap,
                           float* xscr, int B, int Ci, int Co, hipStream_t st, int* rows_out) {
  hipLaunchKernelGGL(fb::k_build_btab, dim3(ceil_div(fb::BTAB_F4 * 4, 256)), dim3(256), 0, st, Aw, Tw, btab);
  int rc;
  if ((rc = check_launch("bwd_build_btab"))) return rc;
  const size_t lds = (size_t)4 * ff::WAVE_LDS * sizeof(float);
  const int nblk = (B + 3) / 4;
  const int grid = nblk < 256 ? nblk : 256;
  *rows_out = grid * 4;
#define LAUNCH_FB(CT, OT)                                                                                              \
  do {                                                                                                                 \
    auto k = fb::k_layer_bwd_fused<CT, OT>;                                                                            \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, coef, btab, in_slope, dIn, partials, dap, xscr, B); \
  } while (0)
  {
    ProbeScope probe(KID_BWD_DATA, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FB(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FB(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_FB(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_FB(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FB(2, 2);
    else if (Ci == 32 && Co == 64) LAUNCH_FB(2, 4);
    else return fail(COSKAD_ERR_SHAPE, "bwd_fused: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FB
  return check_launch("bwd_fused");
}

int launch_reduce_fused(const float* partials, int rows, float* dA, float* dT, const float* dap, float* dslope, int accumulate,
                        hipStream_t st) {
  hipLaunchKernelGGL(fb::k_reduce_fused, dim3(fb::EROW / 64 + (dap ? 1 : 0)), dim3(1024), 0, st, partials, rows, dA, dT, dap, rows,
                     dslope, accumulate);
  return check_launch("bwd_reduce_fused");
}

bool layer_bwd_fused_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

}  // namespace coskad
