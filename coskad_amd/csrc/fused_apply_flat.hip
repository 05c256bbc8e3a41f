// Training-mode layer apply on the stored-Z path for layouts other than 12 x 17 (reference: models/graph_layers/stsgcn.py:94-116
// with both BatchNorms folded from this batch's statistics):
//     U[o][pos] = sum_c Wz[c][o] Z[c][pos] + sum_c Wx[c][o] PReLU(U_prev)[c][pos] + b[o]
// The product does not see frames or joints: fused_apply_bpc.hip's layout -- ONE CLIP PER WORKGROUP, the four waves sharing a
// 16-row K window and a 32-row flush image, the K rows staged by all 256 threads a quarter per k-step, a workgroup barrier per
// k-step -- over FLAT 16-position tiles of a clip's T V positions (the last tile masked), templated on T V.  Built for the
// 25-joint layout (T V = 300: 19 tiles, 58 KB of LDS per workgroup, two workgroups per CU), 16 / 32 input and 16 / 32 / 64
// output channels; replaces the streaming strip GEMM (k_layer_apply_z: nothing staged, every operand fetched per strip) there.
#include "fused_ops.h"

namespace coskad {
namespace fpf {

using ff::BufRes;
using ff::buf_load1;
using ff::buf_load4;
using ff::buf_store4;
using ff::f32x4;
using ff::Lane;
using ff::make_res;
using ff::mfma;
using ff::prelu;

// window row stride: >= TVg, 16-byte rows, rows 4s + q (q = 0..3) on disjoint bank quarters (stride = 16 or 48 mod 64 floats)
constexpr int window_stride(int tv) {
  int l = (tv + 3) / 4 * 4;
  while (l % 64 != 16 && l % 64 != 48) l += 4;
  return l;
}

// CT: 16-row groups of the input; OT: 16-channel output tiles; XO: the commuted convolutions of csrc/commute_layer.hip -- the product
// runs over the layer input alone, [Y; R] = [Wt; Wr] . PReLU(in) with the two [16 x Ci] weights as the parameters store them (`wfold` =
// Wt, `Zg` = Wr, no bias), and the clip's rows, which the flush holds on chip anyway, are mixed before the workgroup moves on:
// Zy = gcn(Y) in place (by joint, then by frame; a wave's operands of both mixes in 63 registers for the launch) -> `Zy`, and the
// per-channel sums of Zy, Zy^2, R, R^2 over the workgroup's clips -> `mixpart` [grid][64] (both BatchNorms' batch statistics).
// NX: the NEXT layer's statistics pass rides on this kernel (C_out <= 32: the flush image holds the clip's whole U): PReLU(out_slope) in
// place, sum x x^T and sum x, Z_next = gcn_next(X) in place -> `Zy`, sum z z^T and sum z -- what k_fwd_moments_bpc does from a re-read of
// U; one partial row [MX][sumX][MZ][sumZ] per workgroup -> `mixpart` (k_reduce_partials + k_train_fold finish).  `Aw`, `Tw`: the next
// layer's mixing parameters.
template <int TVg, int CT, int OT, bool XO = false, bool NX = false>
__global__ __launch_bounds__(256, 2) void k_layer_apply_flat(const float* __restrict__ in, const float* __restrict__ Zg,
                                                            const float* __restrict__ wfold, const float* __restrict__ bias,
                                                            const float* __restrict__ in_slope, float* __restrict__ out, int B,
                                                            const float* __restrict__ Aw, const float* __restrict__ Tw,
                                                            float* __restrict__ Zy, float* __restrict__ mixpart,
                                                            const float* __restrict__ out_slope) {
  static_assert(TVg % 4 == 0, "rows are staged as float4");
  constexpr int Ci = 16 * CT, Co = 16 * OT, CoP = Co, NG = XO ? CT : 2 * CT;
  constexpr int R4 = TVg / 4;                            // float4 per row
  constexpr int LDg = TVg + 2;                           // flush image stride (two padding columns: masked lanes store there)
  constexpr int LDWg = window_stride(TVg);
  constexpr int NT = (TVg + 15) / 16;                    // position tiles
  // a wave's share: 64 output channels: its own output tile x all position tiles; 32: output tile wave & 1 x half of them;
  // 16: a quarter of them
  constexpr int MAXT = OT == 4 ? NT : (OT == 2 ? (NT + 1) / 2 : (NT + 3) / 4);
  constexpr int Q4 = 4 * R4, NQ = (Q4 + 255) / 256;      // float4 of a quarter (4 rows), pieces per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r2 = lds;                   // 16-row K window (stride LDWg)
  float* r1 = lds + 16 * LDWg;       // 32-row flush image (stride LDg)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int ot = OT == 4 ? wave : (OT == 2 ? (wave & 1) : 0);
  // (XO: this wave's output tile is rows of Wt (ot = 0) or of Wr (ot = 1), read transposed)
  const BufRes wres = XO ? make_res(ot ? Zg : wfold, 16 * Ci * 4u) : make_res(wfold, NG * 16 * CoP * 4u);
  const BufRes bres = make_res(XO ? wfold : bias, CoP * 4u);
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TVg, in_range ? rows * TVg * 4u : 0u);
  };
  const int t0 = OT == 4 ? 0 : (OT == 2 ? (wave >> 1) * MAXT : wave * MAXT);
  const int nt = NT - t0 < MAXT ? NT - t0 : MAXT;        // (> 0 for every wave at the shapes built)
  // K ring: one group in flight, a quarter = NQ float4 per thread
  float4 gq[4][NQ];
  auto qload = [&](const BufRes& res, int row0, int q, float4 (&dst)[NQ]) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      dst[i] = buf_load4(res, e < Q4 ? e * 16 : 0x7ffffff0, (row0 + 4 * q) * R4 * 16);
    }
  };
  auto qstore = [&](int q, const float4 (&src)[NQ], bool act) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      float4 v = src[i];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / R4, col = 4 * (e - row * R4);
      if (e < Q4) *reinterpret_cast<float4*>(r2 + (4 * q + row) * LDWg + col) = v;
    }
  };

  // (XO) a wave's joints and frames are the same for every clip: its B operands of both mixes stay in registers
  //   temporal  Y[q,v] = sum_t X[t,v] T[v][t][q]:   B[k = t][j = q];   spatial  Z[t,w] = sum_v Y[t,v] A[t][v][w]:   B[k = v][j = w]
  constexpr int MV = TVg / 12;
  static_assert(!NX || OT <= 2, "the next layer's statistics need the clip's whole output in the 32-row image");
  constexpr bool MIX = XO || NX;
  float mixt[MIX ? (MV + 3) / 4 : 1][3], mixa[MIX ? 3 : 1][(MV + 15) / 16][(MV + 3) / 4], msum[4][4];
  // (NX) Gram accumulators of the next layer's input and of its mixed form: .x / .y halves of the one block (16 channels) or blocks
  // 00, 01, 11 (32); row sums per 16-row group
  constexpr int NACC = OT == 1 ? 2 : 3;
  f32x4 gx[NX ? NACC : 1], gz[NX ? NACC : 1];
  float sx[OT], sz[OT];
  if constexpr (NX) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) { gx[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gz[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < OT; ++i) { sx[i] = 0.f; sz[i] = 0.f; }
  }
  if constexpr (MIX) {
    const Lane L0 = geo();
#pragma unroll
    for (int k = 0; k < (MV + 3) / 4; ++k) {
      const int v = wave + 4 * k;
#pragma unroll
      for (int s = 0; s < 3; ++s) mixt[k][s] = (v < MV && L0.j < 12) ? Tw[(v * 12 + 4 * s + L0.q) * 12 + L0.j] : 0.f;
    }
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int c = 0; c < (MV + 15) / 16; ++c)
#pragma unroll
        for (int s = 0; s < (MV + 3) / 4; ++s)
          mixa[tt][c][s] = (16 * c + L0.j < MV && 4 * s + L0.q < MV) ? Aw[(t * MV + 4 * s + L0.q) * MV + 16 * c + L0.j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { msum[k][0] = 0.f; msum[k][1] = 0.f; msum[k][2] = 0.f; msum[k][3] = 0.f; }
  }
  int clip = blockIdx.x;
  {
    const BufRes z0 = clip_res(XO ? in : Zg, clip, Ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) qload(z0, 0, q, gq[q]);
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(XO ? in : Zg, clip, Ci), ores = clip_res(out, clip, Co);
    const BufRes znext = clip_res(XO ? in : Zg, clip + gridDim.x, Ci);
    // group g of this clip: Z rows first (CT groups), then the layer input (XO: the input alone); beyond: the next clip's first group
    auto gload = [&](int g, int q, float4 (&dst)[NQ]) {
      if (XO) {
        if (g < NG) qload(xres, 16 * g, q, dst);
        else qload(znext, 0, q, dst);
      } else if (g < CT) qload(zres, 16 * g, q, dst);
      else if (g < NG) qload(xres, 16 * (g - CT), q, dst);
      else qload(znext, 0, q, dst);
    };
    Lane L = geo();
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // position of tile t of this wave in the lane's column (beyond the clip: clamped -- those columns are never stored)
    auto pos_of = [&](int t) {
      const int p = 16 * (t0 + (t < nt ? t : 0)) + L.j;
      return p < TVg ? p : TVg - 1;
    };
    // entry: the registers hold group 0 (fetched during the previous clip; its flush ended with a barrier)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      qstore(q, gq[q], XO && pre);
      gload(1, q, gq[q]);
    }
    const int lq = XO ? (L.j * Ci + L.q) * 4 : (L.q * CoP + 16 * ot + L.j) * 4;   // this wave's output tile of the folded weights
    constexpr int WK = XO ? 1 : CoP;                       // floats between consecutive k of the weight operand
    float wc[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wc[0][s] = buf_load1(wres, lq, (4 * s) * WK * 4);
    __syncthreads();                                       // the window holds group 0
    float b[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) b[t] = r2[L.q * LDWg + pos_of(t)];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wc[(g + 1) & 1][s] = buf_load1(wres, lq, ((16 * (g + 1) + 4 * s) * WK) * 4);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // every wave has read rows 4s .. 4s+3 (a k-step ago); in the last group: its fourth quarter (stored a k-step ago) is visible
        if (g + 1 < NG || s == 0) __syncthreads();
        if (g + 1 < NG) {
          qstore(s, gq[s], (XO || g + 1 >= CT) && pre);
          gload(g + 2, s, gq[s]);                          // (beyond this clip: the next clip's first group)
        }
        // (a wave with one tile fewer multiplies its first tile twice instead of branching: that sum is never stored)
#pragma unroll
        for (int t = 0; t < MAXT; ++t) acc[t] = mfma(wc[g & 1][s], b[t], acc[t]);
        if (s + 1 < 4 || g + 1 < NG) {                     // the next k-step's operands behind this step's MFMAs
          const int sn = (s + 1) & 3;
#pragma unroll
          for (int t = 0; t < MAXT; ++t) b[t] = r2[(4 * sn + L.q) * LDWg + pos_of(t)];
        }
      }
    }
    L = geo();
    const float4 b4 = XO ? float4{0.f, 0.f, 0.f, 0.f} : buf_load4(bres, L.q * 16, (16 * ot) * 4);
    const f32x4 bq = {b4.x, b4.y, b4.z, b4.w};
    // ---- flush: 32 channels at a time through the image, full lines to HBM ---------------------------------------------------
    constexpr int NR = (Co + 31) / 32;
#pragma unroll
    for (int rnd = 0; rnd < NR; ++rnd) {
      if (OT < 4 || (wave >> 1) == rnd) {
        const int row0 = OT == 4 ? 16 * (wave & 1) : 16 * ot;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int p = 16 * (t0 + t) + L.j;
          if (t < nt) {
            const f32x4 v = acc[t] + bq;
            float* dst = r1 + (row0 + 4 * L.q) * LDg + (p < TVg ? p : TVg);
            dst[0] = v[0]; dst[LDg] = v[1]; dst[2 * LDg] = v[2]; dst[3 * LDg] = v[3];
          }
        }
      }
      __syncthreads();
      constexpr int rows = Co < 32 ? Co : 32;
      constexpr int n4 = rows * R4;
#pragma unroll
      for (int i = 0; i < (n4 + 255) / 256; ++i) {
        const int e4 = tid + 256 * i;
        const bool ok = e4 < n4;
        const int row = e4 / R4, col = 4 * (e4 - row * R4);
        const float* p = r1 + (ok ? row * LDg + col : TVg);
        const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(ok ? p + 2 : p);
        buf_store4(ores, ok ? e4 * 16 : 0x7ffffff0, (32 * rnd) * R4 * 16, float4{g0.x, g0.y, g1.x, g1.y});
      }
      __syncthreads();                                     // (the image / the window are rewritten next)
    }
    if constexpr (XO) {
      // ---- Zy = gcn(Y) on rows 0 .. 15 of the image, in place; row sums -----------------------------------------------------------
      constexpr int V = TVg / 12, T = 12, MAXJ = (V + 3) / 4, MAXF = T / 4, NTV = (V + 15) / 16, KV = (V + 3) / 4;
      L = geo();
#pragma unroll
      for (int k = 0; k < MAXJ; ++k) {
        const int v = wave + 4 * k;
        if (v < V) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) d = mfma(r1[L.j * LDg + (4 * s + L.q) * V + v], mixt[k][s], d);
          if (L.j < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[(4 * L.q + r) * LDg + L.j * V + v] = d[r];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < MAXF; ++tt) {
        const int t = wave + 4 * tt;
        float a[KV];
#pragma unroll
        for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? r1[L.j * LDg + t * V + 4 * s + L.q] : 0.f;
        f32x4 d[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], mixa[tt][c][s], d[c]);
        }
#pragma unroll
        for (int c = 0; c < NTV; ++c)
          if (16 * c + L.j < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) r1[(4 * L.q + r) * LDg + t * V + 16 * c + L.j] = d[c][r];
          }
      }
      __syncthreads();                                     // rows 0 .. 15 hold Zy
      {
        const BufRes zres = clip_res(Zy, clip, 16);
        constexpr int n4 = 16 * R4;
#pragma unroll
        for (int i = 0; i < (n4 + 255) / 256; ++i) {
          const int e4 = tid + 256 * i;
          const bool ok = e4 < n4;
          const int row = e4 / R4, col = 4 * (e4 - row * R4);
          const float* p = r1 + (ok ? row * LDg + col : TVg);
          const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(ok ? p + 2 : p);
          buf_store4(zres, ok ? e4 * 16 : 0x7ffffff0, 0, float4{g0.x, g0.y, g1.x, g1.y});
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* pz = r1 + (wave + 4 * k) * LDg;
        const float* pr = pz + 16 * LDg;
        for (int p = lane; p < TVg; p += 64) {
          const float z = pz[p], r = pr[p];
          msum[k][0] += z; msum[k][1] = fmaf(z, z, msum[k][1]);
          msum[k][2] += r; msum[k][3] = fmaf(r, r, msum[k][3]);
        }
      }
      __syncthreads();                                     // (the image is rewritten next)
    }
    if constexpr (NX) {
      constexpr int V = TVg / 12, T = 12, MAXJ = (V + 3) / 4, MAXF = T / 4, NTV = (V + 15) / 16, KV = (V + 3) / 4;
      constexpr int NM = (TVg + 7) / 8;                    // double k-steps over the positions
      const float a_out = out_slope[0];
      // this wave's double k-steps m = wave, wave + 4, .. of the image's Gram sum ((row, position) products, ds_read_b64)
      auto gram = [&](f32x4 (&g)[NACC], float (&sm)[OT]) {
        const float* p0 = r1 + L.j * LDg + 2 * L.q;
        const float* p1 = r1 + (16 + L.j) * LDg + 2 * L.q;
        for (int m = wave; m < NM; m += 4) {
          const bool ok = 8 * m + 2 * L.q < TVg;           // (beyond the row: the next row / the padding -- masked)
          float2 a0 = *reinterpret_cast<const float2*>(p0 + 8 * m);
          a0.x = ok ? a0.x : 0.f; a0.y = ok ? a0.y : 0.f;
          if constexpr (OT == 1) {
            g[0] = mfma(a0.x, a0.x, g[0]);
            g[1] = mfma(a0.y, a0.y, g[1]);
            sm[0] += a0.x + a0.y;
          } else {
            float2 a1 = *reinterpret_cast<const float2*>(p1 + 8 * m);
            a1.x = ok ? a1.x : 0.f; a1.y = ok ? a1.y : 0.f;
            g[0] = mfma(a0.x, a0.x, g[0]);
            g[1] = mfma(a0.x, a1.x, g[1]);
            g[2] = mfma(a1.x, a1.x, g[2]);
            g[0] = mfma(a0.y, a0.y, g[0]);
            g[1] = mfma(a0.y, a1.y, g[1]);
            g[2] = mfma(a1.y, a1.y, g[2]);
            sm[0] += a0.x + a0.y;
            sm[OT - 1] += a1.x + a1.y;
          }
        }
      };
      // ---- X_next = PReLU(U) in place (the flush's last barrier is behind us: the rows have left) ----------------------------------
      {
        constexpr int n4 = Co * R4;
#pragma unroll
        for (int i = 0; i < (n4 + 255) / 256; ++i) {
          const int e4 = tid + 256 * i;
          if (e4 < n4) {
            const int row = e4 / R4, col = 4 * (e4 - row * R4);
            float2* p = reinterpret_cast<float2*>(r1 + row * LDg + col);
            float2 g0 = p[0], g1 = p[1];
            g0.x = prelu(g0.x, a_out); g0.y = prelu(g0.y, a_out); g1.x = prelu(g1.x, a_out); g1.y = prelu(g1.y, a_out);
            p[0] = g0; p[1] = g1;
          }
        }
      }
      __syncthreads();                                     // the image holds X_next
      L = geo();
      gram(gx, sx);
      __syncthreads();                                     // every wave has read X_next
#pragma unroll
      for (int k = 0; k < MAXJ; ++k) {
        const int v = wave + 4 * k;
        if (v < V) {
#pragma unroll
          for (int rt = 0; rt < OT; ++rt) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 3; ++s) d = mfma(r1[(16 * rt + L.j) * LDg + (4 * s + L.q) * V + v], mixt[k][s], d);
            if (L.j < T) {
#pragma unroll
              for (int r = 0; r < 4; ++r) r1[(16 * rt + 4 * L.q + r) * LDg + L.j * V + v] = d[r];
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < MAXF; ++tt) {
        const int t = wave + 4 * tt;
#pragma unroll
        for (int rt = 0; rt < OT; ++rt) {
          float a[KV];
#pragma unroll
          for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? r1[(16 * rt + L.j) * LDg + t * V + 4 * s + L.q] : 0.f;
          f32x4 d[NTV];
#pragma unroll
          for (int c = 0; c < NTV; ++c) {
            d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], mixa[tt][c][s], d[c]);
          }
#pragma unroll
          for (int c = 0; c < NTV; ++c)
            if (16 * c + L.j < V) {
#pragma unroll
              for (int r = 0; r < 4; ++r) r1[(16 * rt + 4 * L.q + r) * LDg + t * V + 16 * c + L.j] = d[c][r];
            }
        }
      }
      __syncthreads();                                     // the image holds Z_next
      {
        const BufRes zres = clip_res(Zy, clip, Co);
        constexpr int n4 = Co * R4;
#pragma unroll
        for (int i = 0; i < (n4 + 255) / 256; ++i) {
          const int e4 = tid + 256 * i;
          const bool ok = e4 < n4;
          const int row = e4 / R4, col = 4 * (e4 - row * R4);
          const float* p = r1 + (ok ? row * LDg + col : TVg);
          const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(ok ? p + 2 : p);
          buf_store4(zres, ok ? e4 * 16 : 0x7ffffff0, 0, float4{g0.x, g0.y, g1.x, g1.y});
        }
      }
      gram(gz, sz);
      __syncthreads();                                     // (the image is rewritten by the next clip's flush)
    }
  }
  if constexpr (XO) {
    float* dst = mixpart + (size_t)blockIdx.x * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = wave_sum(msum[k][0]), b = wave_sum(msum[k][1]), c = wave_sum(msum[k][2]), d = wave_sum(msum[k][3]);
      if (lane == 0) {
        const int row = wave + 4 * k;
        dst[row] = a; dst[16 + row] = b; dst[32 + row] = c; dst[48 + row] = d;
      }
    }
  }
  if constexpr (NX) {
    // the waves add their tiles into one row in LDS one after another (fixed order), then the row leaves
    constexpr int E = 2 * (Co * Co + Co);
    float* row = r1;
    __syncthreads();
    const Lane Lp = geo();
    auto put = [&](int w, float* base, const f32x4 (&g)[NACC], const float (&sm)[OT]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {                        // D layout: register r <-> row 4 q + r, column j
        const int i = 4 * Lp.q + r, j = Lp.j;
        if constexpr (OT == 1) {
          float* p = base + i * Co + j;
          p[0] = (w ? p[0] : 0.f) + (g[0][r] + g[1][r]);
        } else {
          float* p00 = base + i * Co + j;
          float* p01 = base + i * Co + 16 + j;
          float* p10 = base + (16 + j) * Co + i;
          float* p11 = base + (16 + i) * Co + 16 + j;
          p00[0] = (w ? p00[0] : 0.f) + g[0][r];
          p01[0] = (w ? p01[0] : 0.f) + g[1][r];
          p10[0] = (w ? p10[0] : 0.f) + g[1][r];
          p11[0] = (w ? p11[0] : 0.f) + g[NACC - 1][r];
        }
      }
#pragma unroll
      for (int rt = 0; rt < OT; ++rt) {
        const float t = ff::quad_sum(sm[rt]);
        if (Lp.q == 0) {
          float* p = base + Co * Co + 16 * rt + Lp.j;
          p[0] = (w ? p[0] : 0.f) + t;
        }
      }
    };
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
        put(w, row, gx, sx);
        put(w, row + Co * Co + Co, gz, sz);
      }
      __syncthreads();
    }
    float* dst = mixpart + (size_t)blockIdx.x * E;
    for (int e = tid; e < E; e += 256) dst[e] = row[e];
  }
}

}  // namespace fpf

bool layer_apply_flat_ok(int TV_, int Ci, int Co) {
  return TV_ == 300 && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

int launch_layer_apply_flat(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                            const float* in_slope, int B, int Ci, int Co, int TV_, hipStream_t st) {
  if (!layer_apply_flat_ok(TV_, Ci, Co)) return fail(COSKAD_ERR_SHAPE, "apply_flat: unsupported shape (%d positions, %d -> %d)", TV_, Ci, Co);
  constexpr int TVg = 300;
  const size_t lds = (size_t)(16 * fpf::window_stride(TVg) + 32 * (TVg + 2)) * sizeof(float);
  const int grid = B < 512 ? B : 512;                      // two workgroups per CU (four at 16 output channels: same speed)
#define LAUNCH_FPF(CT, OT)                                                                                       \
  do {                                                                                                           \
    auto k = fpf::k_layer_apply_flat<TVg, CT, OT>;                                                               \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out, B, (const float*)nullptr, \
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr);          \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FPF(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FPF(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_FPF(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_FPF(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FPF(2, 2);
    else LAUNCH_FPF(2, 4);
  }
#undef LAUNCH_FPF
  return check_launch("layer_apply_flat");
}

bool layer_apply_flat_next_ok(int TV_, int Ci, int Co) { return TV_ == 300 && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32); }

// launch_layer_apply_flat + the next layer's statistics pass (`An`, `Tn`: its mixing parameters; `out_slope`: the PReLU between the
// layers): Z_next [B, Co, TV], partial rows [*rows_out][2 (Co^2 + Co)]
int launch_layer_apply_flat_next(const float* Z, const float* in, float* out, const float* wfold, const float* bias, const float* in_slope,
                                 const float* out_slope, const float* An, const float* Tn, float* z_next, float* partials, int B, int Ci,
                                 int Co, int TV_, hipStream_t st, int* rows_out) {
  if (!layer_apply_flat_next_ok(TV_, Ci, Co)) return fail(COSKAD_ERR_SHAPE, "apply_flat_next: unsupported shape (%d positions, %d -> %d)", TV_, Ci, Co);
  constexpr int TVg = 300;
  const size_t lds = (size_t)(16 * fpf::window_stride(TVg) + 32 * (TVg + 2)) * sizeof(float);
  const int grid = B < 512 ? B : 512;
  *rows_out = grid;
#define LAUNCH_FPN(CT, OT)                                                                                       \
  do {                                                                                                           \
    auto k = fpf::k_layer_apply_flat<TVg, CT, OT, false, true>;                                                  \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out, B, An, Tn, z_next, partials, out_slope); \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FPN(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FPN(1, 2);
    else if (Ci == 32 && Co == 16) LAUNCH_FPN(2, 1);
    else LAUNCH_FPN(2, 2);
  }
#undef LAUNCH_FPN
  return check_launch("layer_apply_flat_next");
}

// [Y; R] = [Wt; Wr] . PReLU(in [B, 32, TV]) -> out [B, 32, TV]; Zy = gcn(Y) -> zy [B, 16, TV]; row sums -> mixpart [*rows_out][64]: the
// forward of a commuted 32 -> 16 layer up to its BatchNorm statistics (csrc/commute_layer.hip) on the K-ring GEMM above
template <int TVg>
static int launch_commute_apply_mix_tv(const float* in, float* out, const float* wt, const float* wr, const float* in_slope, const float* Aw,
                                       const float* Tw, float* zy, float* mixpart, int B, hipStream_t st, int* rows_out) {
  const size_t lds = (size_t)(16 * fpf::window_stride(TVg) + 32 * (TVg + 2)) * sizeof(float);
  const int grid = B < 512 ? B : 512;
  *rows_out = grid;
  auto k = fpf::k_layer_apply_flat<TVg, 2, 2, true>;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, wr, wt, (const float*)nullptr, in_slope, out, B, Aw, Tw, zy, mixpart,
                     (const float*)nullptr);
  return check_launch("commute_apply_mix");
}

int launch_commute_apply_mix(const float* in, float* out, const float* wt, const float* wr, const float* in_slope, const float* Aw,
                             const float* Tw, float* zy, float* mixpart, int B, int Ci, int Jo, int TV_, hipStream_t st, int* rows_out) {
  if ((TV_ != 300 && TV_ != 204) || Ci != 32 || Jo != 32)
    return fail(COSKAD_ERR_SHAPE, "commute_apply_mix: built for 204 / 300 positions, 32 -> 16 + 16 (%d, %d -> %d)", TV_, Ci, Jo);
  if (TV_ == 204) return launch_commute_apply_mix_tv<204>(in, out, wt, wr, in_slope, Aw, Tw, zy, mixpart, B, st, rows_out);
  return launch_commute_apply_mix_tv<300>(in, out, wt, wr, in_slope, Aw, Tw, zy, mixpart, B, st, rows_out);
}

extern "C" {

/* 1: coskad_layer_apply_next_flat_f32 takes a (Ci -> Co) layer on this layout (12 x 25; 16 / 32 -> 16 / 32 channels) */
int coskad_layer_apply_next_flat_ok(int Ci, int Co, int T, int V) { return (T == 12 && layer_apply_flat_next_ok(T * V, Ci, Co)) ? 1 : 0; }

/* partial rows the call writes for a batch of B clips (each 2 (Co^2 + Co) floats) */
int coskad_layer_apply_next_flat_rows(int B) { return B < 512 ? B : 512; }

/* coskad_layer_apply_z_f32 (training-mode apply from the stored Z: U = Wz.Z + Wx.PReLU(in) + b) AND the next layer's statistics pass
 * in one kernel on the 25-joint layout: Z_next = gcn_next(PReLU_out(U)) [B, Co, T, V] and its moment partials (rows of
 * [sum x x^T][sum x][sum z z^T][sum z], finished by coskad_layer_train_fold_f32) -- what coskad_layer_apply_next_f32 does at 17 joints
 * (models/graph_layers/stsgcn.py:94-116 of layer i, 56-80 + the BatchNorm batch statistics of layer i + 1). */
int coskad_layer_apply_next_flat_f32(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                     const float* in_slope, const float* out_slope, const float* A_next, const float* T_next,
                                     float* Z_next, float* partials, size_t partials_bytes, int B, int Ci, int Co, int T, int V,
                                     hipStream_t stream) {
  if (!Z || !in || !out || !wfold || !bias || !out_slope || !A_next || !T_next || !Z_next || !partials)
    return fail(COSKAD_ERR_ARG, "layer_apply_next_flat: null pointer");
  if (B <= 0 || !coskad_layer_apply_next_flat_ok(Ci, Co, T, V))
    return fail(COSKAD_ERR_SHAPE, "layer_apply_next_flat: built for 12 x 25, 16 / 32 -> 16 / 32 channels");
  if (partials_bytes < (size_t)coskad_layer_apply_next_flat_rows(B) * 2 * ((size_t)Co * Co + Co) * sizeof(float))
    return fail(COSKAD_ERR_WORKSPACE, "layer_apply_next_flat: partial table too small");
  int rows = 0;
  return launch_layer_apply_flat_next(Z, in, out, wfold, bias, in_slope, out_slope, A_next, T_next, Z_next, partials, B, Ci, Co, T * V,
                                      stream, &rows);
}

}  // extern "C"

}  // namespace coskad
