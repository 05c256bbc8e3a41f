// Training forward: layer i's apply AND layer i+1's statistics pass (the work of k_layer_apply_next, fused_apply_next.hip;
// reference models/graph_layers/stsgcn.py:94-116, :154-155, :65,76), ONE CLIP PER WORKGROUP.
//
// The wave-per-clip kernel gives every wavefront its own 39.7 KB of LDS: one wave per SIMD, and nothing runs while it waits for
// an MFMA chain, an LDS round trip or a row store.  Here the four waves of a workgroup SHARE a clip's image and K window
// (fused_apply_bpc.hip measured the layout on the plain apply: 114 -> 99 us), so that four workgroups = sixteen waves fit a CU:
//   K-ring GEMM   a wave owns one output tile x half the position tiles (or a quarter of them at 16 output channels); the K rows
//                 are staged by all 256 threads, a workgroup barrier per k-step
//   Gram sums     the 26 double k-steps and the row pieces that leave with them are dealt round-robin to the waves; every wave
//                 keeps its own partial Gram (summed with the others' at the very end, as the four waves of a block always were)
//   temporal mix  joints round-robin (an item = one joint column of the whole image: items never touch each other's columns)
//   spatial mix   frames round-robin (likewise)
// with a workgroup barrier between the phases (six per clip + one per k-step).
#include "fused_ops.h"

namespace coskad {
namespace fnb {

using namespace ff;

#ifndef FNB_OCC
#define FNB_OCC 4
#endif

// CT: 16-row groups of the input (0: TWO input channels: Z rows | X rows are ONE k-step); OTP: 16-channel output tiles
template <int CT, int OTP>
__global__ __launch_bounds__(256, FNB_OCC) void k_layer_apply_next_bpc(const float* __restrict__ in, const float* __restrict__ Zg,
                                                                      const float* __restrict__ wfold, const float* __restrict__ bias,
                                                                      const float* __restrict__ in_slope,
                                                                      const float* __restrict__ out_slope,
                                                                      const float* __restrict__ ftab, float* __restrict__ out,
                                                                      float* __restrict__ Znext, float* __restrict__ partials, int B) {
  constexpr int Ci = CT ? 16 * CT : 2, Co = 16 * OTP, CoP = Co, NG = 2 * CT;
  constexpr int NACC = OTP == 1 ? 2 : 3;                 // Gram accumulators: even / odd k-steps of the one block, or blocks 00, 01, 11
  constexpr int E = 2 * (Co * Co + Co);                  // partial row: [MX Co*Co][sumX Co][MZ Co*Co][sumZ Co]
  // this wave's share of the 13 position tiles of the GEMM: OTP == 2: output tile wave & 1, tiles [0, 7) or [7, 13);
  // OTP == 1: tiles [0, 4), [4, 7), [7, 10), [10, 13)
  constexpr int MAXT = OTP == 2 ? 7 : 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r1 = lds;                 // 32-row image (stride LD)
  float* r2 = lds + 32 * LD;       // 16-row K window (stride LDW)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  auto olane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const float a_out = out_slope[0];
  const int l16 = lane * 16;
  const BufRes wres = make_res(wfold, 2 * Ci * CoP * 4u);
  const BufRes bres = make_res(bias, CoP * 4u);
  const BufRes tabres = make_res(ftab, LAYER_F4 * 16u);
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
  };
  const int ot = OTP == 2 ? (wave & 1) : 0;
  const int t0 = OTP == 2 ? ((wave >> 1) ? 7 : 0) : (wave == 0 ? 0 : 1 + 3 * wave);
  const int nt = OTP == 2 ? ((wave >> 1) ? 6 : 7) : (wave == 0 ? 4 : 3);
  // staging: thread t < 204 owns float4 `t` of a quarter (4 rows x 51 float4)
  constexpr int Q4 = 4 * (TV / 4);
  const bool stg = tid < Q4;
  const int srow = tid / (TV / 4), scol = 4 * (tid - srow * (TV / 4));
  const int svoff = stg ? tid * 16 : 0x7ffffff0;
  // K ring: DEPTH groups in flight [set][quarter] (CT == 0: gq[0][0..1]).  Two input groups (CT == 1) are served by one set: the
  // second would only ever hold the next clip's rows, and its 16 registers are what keeps that kernel from four waves per SIMD
  constexpr int DEPTH = CT == 1 ? 1 : 2;
  float4 gq[DEPTH][4];
  auto qload = [&](const BufRes& res, int row0, int q) { return buf_load4(res, svoff, (row0 + 4 * q) * (TV / 4) * 16); };
  auto qstore = [&](int q, float4 v, bool act) {
    if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
    if (stg) *reinterpret_cast<float4*>(r2 + (4 * q + srow) * LDW + scol) = v;
  };
  // two input channels: Z rows 0, 1 and input rows 0, 1 = 2 x 102 float4: thread t < 102 owns one of each
  const bool fst = tid < 2 * (TV / 4);
  const int frow = tid / (TV / 4), fcol = 4 * (tid - frow * (TV / 4));
  const int fvoff = fst ? tid * 16 : 0x7ffffff0;

  // sums of all this wave's share of the Gram k-steps, over all the workgroup's clips
  f32x4 gx[NACC], gz[NACC];
  float sx[OTP], sz[OTP];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { gx[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gz[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < OTP; ++i) { sx[i] = 0.f; sz[i] = 0.f; }
  // this wave's double k-steps m = wave, wave + 4, .. of the Gram sums of the image, with the image's row pieces of those steps
  // leaving for HBM in the same loop (k_layer_apply_next: gram_rows)
  auto gram_rows = [&](f32x4 (&g)[NACC], float (&s)[OTP], const BufRes& ores, bool act) {
    const Lane L = geo();
    const float* p0 = r1 + L.j * LD + 2 * L.q;
    const float* p1 = r1 + (16 + L.j) * LD + 2 * L.q;
    constexpr int NM = (TV + 7) / 8;                     // 26
    constexpr int n4 = Co * (TV / 4), NI = (n4 + 63) / 64;   // row pieces: 13 (16 rows) / 26 (32 rows)
    const int ln = olane();
    for (int m = wave; m < NM; m += 4) {
      float2 a0 = *reinterpret_cast<const float2*>(p0 + 8 * m);
      float2 a1 = OTP == 2 ? *reinterpret_cast<const float2*>(p1 + 8 * m) : float2{0.f, 0.f};
      if (act) {
        a0.x = prelu(a0.x, a_out); a0.y = prelu(a0.y, a_out);
        a1.x = prelu(a1.x, a_out); a1.y = prelu(a1.y, a_out);
      }
      const bool ok = 8 * m + 2 * L.q < TV;              // the last step's tail lies in the rows' padding
      a0.x = ok ? a0.x : 0.f; a0.y = ok ? a0.y : 0.f;
      a1.x = ok ? a1.x : 0.f; a1.y = ok ? a1.y : 0.f;
      if constexpr (OTP == 1) {
        g[0] = mfma(a0.x, a0.x, g[0]);
        g[1] = mfma(a0.y, a0.y, g[1]);
        s[0] += a0.x + a0.y;
      } else {
        g[0] = mfma(a0.x, a0.x, g[0]);
        g[1] = mfma(a0.x, a1.x, g[1]);
        g[2] = mfma(a1.x, a1.x, g[2]);
        g[0] = mfma(a0.y, a0.y, g[0]);
        g[1] = mfma(a0.y, a1.y, g[1]);
        g[2] = mfma(a1.y, a1.y, g[2]);
        s[0] += a0.x + a0.y;
        s[OTP - 1] += a1.x + a1.y;
      }
      // the rows' piece(s) of this step: 26 pieces (32 rows: one per step) or 13 (16 rows: the even steps carry one)
      const int i = OTP == 2 ? m : (m & 1 ? -1 : m >> 1);
      if (i >= 0 && i < NI) {
        const int e4 = ln + 64 * i;
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        const bool okp = e4 < n4;
        const float* ptr = r1 + (okp ? row * LD + col : PADCOL);
        const float2 g0 = *reinterpret_cast<const float2*>(ptr), g1 = *reinterpret_cast<const float2*>(ptr + 2);
        buf_store4(ores, okp ? e4 * 16 : 0x7ffffff0, 0, float4{g0.x, g0.y, g1.x, g1.y});
      }
    }
  };

  int clip = blockIdx.x;
  {
    const BufRes z0 = clip_res(Zg, clip, Ci), x0 = clip_res(in, clip, Ci);
    if (CT == 0) {
      gq[0][0] = buf_load4(z0, fvoff, 0);
      gq[0][1] = buf_load4(x0, fvoff, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gq[0][q] = qload(z0, 0, q);
        if (DEPTH == 2) gq[DEPTH - 1][q] = qload(z0, 16, q);
      }
    }
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), ores = clip_res(out, clip, Co);
    const BufRes znext = clip_res(Zg, clip + gridDim.x, Ci), xnext = clip_res(in, clip + gridDim.x, Ci);
    const BufRes zores = clip_res(Znext, clip, Co);
    Lane L = geo();
    const int jc = L.j < T ? L.j : T - 1;
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pos_of = [&](int t) { return t < T ? t * V + L.j : jc * V + 16; };   // t: absolute tile index (T: the joint-16 tile)
    const int lq = (L.q * CoP + 16 * ot + L.j) * 4;
    if constexpr (CT == 0) {
      // ---- ONE k-step: rows Z0 Z1 X0 X1 of the window against the four rows of the folded weight ----------------------------
      {
        float4 vz = gq[0][0], vx = gq[0][1];
        if (pre) { vx.x = prelu(vx.x, a_in); vx.y = prelu(vx.y, a_in); vx.z = prelu(vx.z, a_in); vx.w = prelu(vx.w, a_in); }
        if (fst) {
          *reinterpret_cast<float4*>(r2 + frow * LDW + fcol) = vz;
          *reinterpret_cast<float4*>(r2 + (2 + frow) * LDW + fcol) = vx;
        }
      }
      const float wc = buf_load1(wres, lq, 0);
      gq[0][0] = buf_load4(znext, fvoff, 0);             // the next clip's rows take off
      gq[0][1] = buf_load4(xnext, fvoff, 0);
      __syncthreads();                                   // the window holds this clip's four rows
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
        if (t < nt) acc[t] = mfma(wc, r2[L.q * LDW + pos_of(t0 + t)], acc[t]);
    } else {
      auto vload = [&](int vg, int q) {                  // virtual group: this clip's groups 0 .. NG-1 (Z first), then the next clip's
        const bool nxt = vg >= NG;
        const int g = nxt ? vg - NG : vg;
        return g < CT ? qload(nxt ? znext : zres, 16 * g, q) : qload(nxt ? xnext : xres, 16 * (g - CT), q);
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) {                      // entry: set 0 = group 0 (, set 1 = group 1), fetched during the previous clip
        qstore(q, gq[0][q], false);
        gq[0][q] = vload(DEPTH, q);
      }
      float wc[2][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) wc[0][s] = buf_load1(wres, lq, (4 * s) * CoP * 4);
      __syncthreads();                                   // the window holds group 0
      float b[MAXT];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) b[t] = r2[L.q * LDW + pos_of(t0 + (t < nt ? t : 0))];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
          for (int s = 0; s < 4; ++s) wc[(g + 1) & 1][s] = buf_load1(wres, lq, ((16 * (g + 1) + 4 * s) * CoP) * 4);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // every wave has read rows 4s .. 4s+3 (a k-step ago); in the last group: its fourth quarter (stored a k-step ago) is visible
          if (g + 1 < NG || s == 0) __syncthreads();
          if (g + 1 < NG) {
            qstore(s, gq[(g + 1) % DEPTH][s], g + 1 >= CT && pre);
            gq[(g + 1) % DEPTH][s] = vload(g + 1 + DEPTH, s);   // (beyond this clip: the next clip's first group(s))
          }
#pragma unroll
          for (int t = 0; t < MAXT; ++t)
            if (t < nt) acc[t] = mfma(wc[g & 1][s], b[t], acc[t]);
          if (s + 1 < 4 || g + 1 < NG) {                 // the next k-step's operands behind this step's MFMAs
            const int sn = (s + 1) & 3;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) b[t] = r2[(4 * sn + L.q) * LDW + pos_of(t0 + (t < nt ? t : 0))];
          }
        }
      }
    }
    // ---- U = sums + bias -> image (the bias joins the finished sums) ------------------------------------------------------------
    {
      const float4 b4 = buf_load4(bres, L.q * 16, (16 * ot) * 4);
      const f32x4 bq = {b4.x, b4.y, b4.z, b4.w};
      __syncthreads();                                   // the image is free: every wave has finished the previous clip's second Gram
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int ta = t0 + t;
        if (t < nt) tile_store(r1, 16 * ot, pos_of(ta), ta < T || L.j < T, acc[t] + bq, L);
      }
    }
    __syncthreads();                                     // the image holds U
    // ---- U -> HBM while sum x x^T multiplies (X_next = PReLU(U) on the fly); the image keeps U ----------------------------------
    gram_rows(gx, sx, ores, true);
    float4 rec = buf_load4(tabres, l16, (wave * 64) * 16);   // (the first joint's table record travels across the barrier)
    __syncthreads();                                     // every wave has read U
    // ---- Z_next = gcn_next(X_next) in place.  Temporal: joints v = wave, wave + 4, .. (PReLU on the operand reads) ---------------
    L = geo();
    {
      TOp cur[OTP], nxt[OTP];
      f32x4 dprev[OTP];
      int vprev = -1;
#pragma unroll
      for (int rt = 0; rt < OTP; ++rt) cur[rt] = temporal_read<16, true>(r1, rt, wave, L, a_out);
      for (int v = wave; v < V; v += 4) {
        const int vn = v + 4 < V ? v + 4 : v;
        const float4 recn = buf_load4(tabres, l16, (vn * 64) * 16);
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) nxt[rt] = temporal_read<16, true>(r1, rt, vn, L, a_out);
        f32x4 d[OTP];
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) d[rt] = temporal_mm(cur[rt], rec);
        if (vprev >= 0) {
#pragma unroll
          for (int rt = 0; rt < OTP; ++rt) temporal_store<16>(r1, rt, vprev, dprev[rt], L);
        }
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) { dprev[rt] = d[rt]; cur[rt] = nxt[rt]; }
        vprev = v;
        rec = recn;
      }
#pragma unroll
      for (int rt = 0; rt < OTP; ++rt) temporal_store<16>(r1, rt, vprev, dprev[rt], L);
    }
    SpatRec srec = load_spat(tabres, 0, wave, l16);      // (the first frame's table records travel across the barrier)
    __syncthreads();                                     // the image holds the temporal mix
    // ---- spatial: frames t = wave, wave + 4, wave + 8: the next frame's operands and records are fetched before this frame's
    // results are stored (different frames never alias, but only program order tells the compiler) -----------------------------
    L = geo();
    {
      SOp op[OTP];
#pragma unroll
      for (int rt = 0; rt < OTP; ++rt) op[rt] = spatial_read<16>(r1, rt, wave, L);
#pragma unroll
      for (int k = 0; k < T / 4; ++k) {
        const int t = wave + 4 * k, tn = k + 1 < T / 4 ? t + 4 : t;
        const SpatRec nrec = load_spat(tabres, 0, tn, l16);
        SOp opn[OTP];
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) opn[rt] = spatial_read<16>(r1, rt, tn, L);
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) {
          const f32x4 d = spatial_mm(op[rt], srec);
          spatial_extra<16>(r1, rt, t, op[rt], srec, L);
          tile_store(r1, 16 * rt, t * V + L.j, true, d, L);
        }
        srec = nrec;
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) op[rt] = opn[rt];
      }
    }
    __syncthreads();                                     // the image holds Z_next
    // ---- Z_next -> HBM while sum z z^T multiplies -------------------------------------------------------------------------------
    gram_rows(gz, sz, zores, false);
  }

  // ---- workgroup sum: the waves add their tiles into one LDS row one after another (fixed order), then the row leaves ------------
  float* row = lds;                                      // E floats (<= 8.4 KB) over the image: all clip loops are done
  __syncthreads();
  const Lane L = geo();
  auto put = [&](int w, float* base, const f32x4 (&g)[NACC], const float (&s)[OTP]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // D layout: register r <-> row 4 q + r, column j
      const int i = 4 * L.q + r, j = L.j;
      if constexpr (OTP == 1) {
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
        p11[0] = (w ? p11[0] : 0.f) + g[2][r];
      }
    }
#pragma unroll
    for (int rt = 0; rt < OTP; ++rt) {
      const float t = quad_sum(s[rt]);
      if (L.q == 0) {
        float* p = base + Co * Co + 16 * rt + L.j;
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
  float* dst = partials + (size_t)blockIdx.x * E;
  for (int e = threadIdx.x; e < E; e += 256) dst[e] = row[e];
}

}  // namespace fnb

// four workgroups per CU (<= 128 registers per lane in every instantiation): sweep at B = 4096, three waves per SIMD x 768
// workgroups 80 / 86 / 117 us (5.3 clips per workgroup: uneven), x 512 90 / 101 / 123, four x 1024 78 / 76 / 108
int apply_next_bpc_rows(int B) { return B < 1024 ? B : 1024; }

int launch_layer_apply_next_bpc(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                const float* in_slope, const float* out_slope, const float* ftab, float* Znext, float* partials,
                                int B, int Ci, int Co, hipStream_t st) {
  const size_t lds = (size_t)ff::WAVE_LDS_W * sizeof(float);
  const int grid = apply_next_bpc_rows(B);
#define LAUNCH_FNB(CT, OTP)                                                                                           \
  do {                                                                                                                \
    auto k = fnb::k_layer_apply_next_bpc<CT, OTP>;                                                                    \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out_slope, ftab, out, Znext,  \
                       partials, B);                                                                                  \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 2 && Co == 32) LAUNCH_FNB(0, 2);
    else if (Ci == 32 && Co == 16) LAUNCH_FNB(2, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FNB(1, 2);
    else return fail(COSKAD_ERR_SHAPE, "apply_next_bpc: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FNB
  return check_launch("layer_apply_next_bpc");
}

}  // namespace coskad
