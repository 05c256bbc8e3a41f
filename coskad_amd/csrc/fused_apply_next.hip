// Training forward: layer i's apply AND layer i+1's statistics pass in ONE wave-per-clip kernel
// (reference: models/graph_layers/stsgcn.py:94-116 with both BatchNorms folded from this batch's statistics, then
// stsgcn.py:154-155 of the NEXT layer and the moments its two BatchNorms need):
//     U_i        = Wz.Z_i + Wx.PReLU(U_{i-1}) + b                                   (K-ring GEMM of fused_apply.hip)
//     X_{i+1}    = PReLU_i(U_i)                   sum x x^T, sum x   over (clip, position)
//     Z_{i+1}    = gcn_{i+1}(X_{i+1})             sum z z^T, sum z
// The apply kernel holds a clip's whole U_i in accumulators and hands it to HBM through a 32-row LDS image anyway; here
// that image stays: PReLU is written back while the rows leave, the Gram sums read it as (row, position) operands (one
// ds_read_b64 per two k-steps, conflict-free on the 206 stride), the temporal and spatial mixing of the next layer run in
// place (fused_ops.h: the phases of the eval-mode encoder), Z_{i+1} leaves in full lines, the second Gram follows.
// Sums of all of a wave's clips stay in accumulator registers; the four waves of a block add theirs into ONE partial row
// at the very end (fixed order; k_reduce_partials sums the <= 256 rows in fp64: deterministic, no atomics).
// Replaces k_fwd_moments of layers 2..4 (a second read of every activation, block-per-tile staging and barriers) and
// k_first_apply / k_layer_apply_ring of layers 1..3 at T = 12, V = 17, <= 32 output channels.
#include "fused_ops.h"

#ifndef FN_ABLATE
#define FN_ABLATE 0   // timing-only builds (tools/bench_apply_next.py): bit 0 U rows, 1 Gram x, 2 temporal, 3 spatial, 4 Z rows, 5 Gram z, 6 GEMM
#endif

namespace coskad {
namespace fn {

using namespace ff;

// forward operand tables of up to 4 layers, the tab-stream layout of the eval-mode encoder (coskad_amd/fused_plan.py):
//   temporal  rec[v][l][s]     = T[v][4s+q][j]   (s < 3, j < 12)
//   spatial   rec[t][l][0..4]  = A[t][4s+q][j]   (4s+q < 17),  [5..9] = A[t][4s+q][16]
struct FtabArgs {
  const float* A[4];
  const float* Tm[4];
  float* tab[4];
};
__global__ void k_build_ftab(FtabArgs args) {
  const int layer = blockIdx.y;
  const float* __restrict__ Aw = args.A[layer];
  const float* __restrict__ Tw = args.Tm[layer];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= LAYER_F4 * 4) return;
  float val = 0.f;
  if (e < TEMP_F4 * 4) {
    const int v = e / 256, l = (e >> 2) & 63, s = e & 3, j = l & 15, q = l >> 4;
    if (s < 3 && j < T) val = Tw[v * T * T + (4 * s + q) * T + j];
  } else {
    const int r = e - TEMP_F4 * 4;
    const int t = r / (3 * 256), c = (r / 256) % 3, l = (r >> 2) & 63, k = 4 * c + (r & 3), j = l & 15, q = l >> 4;
    if (k < 10) {
      const int s = k < 5 ? k : k - 5, w = 4 * s + q;
      if (w < V) val = Aw[t * V * V + w * V + (k < 5 ? j : 16)];
    }
  }
  args.tab[layer][e] = val;
}

// CT: 16-row groups of the input (0: TWO input channels, the first layer: Z rows | X rows are ONE k-step);
// OTP: 16-channel output tiles (C_out = 16 OTP <= 32: the image holds the whole output)
template <int CT, int OTP>
__global__ __launch_bounds__(256, 1) void k_layer_apply_next(const float* __restrict__ in, const float* __restrict__ Zg,
                                                            const float* __restrict__ wfold, const float* __restrict__ bias,
                                                            const float* __restrict__ in_slope, const float* __restrict__ out_slope,
                                                            const float* __restrict__ ftab, float* __restrict__ out,
                                                            float* __restrict__ Znext, float* __restrict__ partials, int B) {
  constexpr int Ci = CT ? 16 * CT : 2, Co = 16 * OTP, CoP = Co, NG = 2 * CT;
  constexpr int NACC = OTP == 1 ? 2 : 3;                 // Gram accumulators: even / odd k-steps of the one block, or blocks 00, 01, 11
  constexpr int E = 2 * (Co * Co + Co);                  // partial row: [MX Co*Co][sumX Co][MZ Co*Co][sumZ Co] (k_train_fold's layout)
  static_assert(OTP == 1 || OTP == 2, "the 32-row image holds the whole output");
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
  const int nwaves = gridDim.x * 4;
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
#ifdef COSKAD_HOT   // timing-only: every stream from 64 L2-resident clips
    return make_res(base + (size_t)(in_range ? (c & 63) : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#else
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#endif
  };
  constexpr int QTAIL = 4 * (TV / 4) - 192;              // lanes of a quarter's 4th piece (12)
  const int l16t = lane < QTAIL ? l16 : 0x7ffffff0;
  float4 gb[16];                                       // (two input channels: 4 of them)
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
  // two input channels: the clip's Z rows 0, 1 and input rows 0, 1 are four pieces (2 x 204 floats = 64 + 38 float4 per
  // source; lanes beyond a source read 0 through the bounds check) -> window rows 0..3
  auto fload = [&](const BufRes& zr, const BufRes& xr) {
    gb[0] = buf_load4(zr, l16, 0); gb[1] = buf_load4(zr, l16, 1024);
    gb[2] = buf_load4(xr, l16, 0); gb[3] = buf_load4(xr, l16, 1024);
  };
  auto fstore = [&]() {
    const int ln = olane();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ln + 64 * (c & 1);
      float4 v = gb[c];
      if (c >= 2 && pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
      if (e < 2 * (TV / 4)) *reinterpret_cast<float4*>(r2 + (2 * (c >> 1) + row) * LDW + col) = v;
    }
  };

  // sums of all this wave's clips
  f32x4 gx[NACC], gz[NACC];
  float sx[OTP], sz[OTP];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { gx[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gz[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < OTP; ++i) { sx[i] = 0.f; sz[i] = 0.f; }
  // Gram sums of the image with the image's rows leaving for HBM in the same loop: lane (j, q) reads positions 8 m + 2 q, + 1 of
  // rows j and 16 + j as both MFMA operands (A[i][k] and B[k][j] of a symmetric product); step m also reads piece m of the rows
  // in storage order and stores it (full 1 KB lines).  Both only READ the image, so the stores spread over the Gram's MFMAs
  // instead of standing in front of them as a burst (the chip's 1024 waves run their clips in lockstep).  ACT: the image holds
  // pre-activations U -- the rows leave as they are, the Gram operands get PReLU on the fly (no write-back pass).
  auto gram_rows = [&](f32x4 (&g)[NACC], float (&s)[OTP], const BufRes& ores, bool act) {
    const Lane L = geo();
    const float* p0 = r1 + L.j * LD + 2 * L.q;
    const float* p1 = r1 + (16 + L.j) * LD + 2 * L.q;
    constexpr int NM = (TV + 7) / 8;                     // 26
    constexpr int n4 = Co * (TV / 4), NI = (n4 + 63) / 64;   // row pieces: 13 (16 rows) / 26 (32 rows)
    const int ln = olane();
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      float2 a0 = *reinterpret_cast<const float2*>(p0 + 8 * m);
      float2 a1 = OTP == 2 ? *reinterpret_cast<const float2*>(p1 + 8 * m) : float2{0.f, 0.f};
      if (act) {
        a0.x = prelu(a0.x, a_out); a0.y = prelu(a0.y, a_out);
        a1.x = prelu(a1.x, a_out); a1.y = prelu(a1.y, a_out);
      }
      if (8 * (m + 1) > TV) {                            // the last step's tail lies in the rows' padding
        const bool ok = 8 * m + 2 * L.q < TV;
        a0.x = ok ? a0.x : 0.f; a0.y = ok ? a0.y : 0.f;
        a1.x = ok ? a1.x : 0.f; a1.y = ok ? a1.y : 0.f;
      }
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
      // the rows' piece of this step
      const int i = OTP == 2 ? m : (m % 2 == 0 ? m / 2 : -1);
      if (i >= 0 && i < NI && !(FN_ABLATE & 1)) {
        const int e4 = ln + 64 * i;
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        const bool full = 64 * (i + 1) <= n4;
        const bool ok = full || e4 < n4;
        const float* ptr = r1 + (ok ? row * LD + col : PADCOL);
        const float2 g0 = *reinterpret_cast<const float2*>(ptr), g1 = *reinterpret_cast<const float2*>(ptr + 2);
        buf_store4(ores, ok ? l16 : 0x7ffffff0, 64 * i * 16, float4{g0.x, g0.y, g1.x, g1.y});
      }
    }
  };

  int clip = blockIdx.x * 4 + wave;
  {
    const BufRes z0 = clip_res(Zg, clip, Ci);
    if (CT == 0) {
      fload(z0, clip_res(in, clip, Ci));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) qload(z0, 0, q);       // group 0
    }
  }
  for (; clip < B; clip += nwaves) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), ores = clip_res(out, clip, Co);
    const BufRes znext = clip_res(Zg, clip + nwaves, Ci), xnext = clip_res(in, clip + nwaves, Ci);
    const BufRes zores = clip_res(Znext, clip, Co);
    Lane L = geo();
    const int jc = L.j < T ? L.j : T - 1;
    f32x4 acc[NTILE][OTP], bq[OTP];
#pragma unroll
    for (int ot = 0; ot < OTP; ++ot) {
      const float4 b4 = buf_load4(bres, L.q * 16, (16 * ot) * 4);
      bq[ot] = f32x4{b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int t = 0; t < NTILE; ++t) acc[t][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int lq = (L.q * CoP + L.j) * 4;
    if constexpr (CT == 0) {
      // ---- ONE k-step: rows Z0 Z1 X0 X1 of the window against the four rows of the folded weight ------------------------
      fstore();
      float wc[OTP];
#pragma unroll
      for (int ot = 0; ot < OTP; ++ot) wc[ot] = buf_load1(wres, lq, (16 * ot) * 4);
      __builtin_amdgcn_sched_barrier(0);
      fload(znext, xnext);                               // the next clip's rows take off
      __builtin_amdgcn_sched_barrier(0);
      float b[NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) b[t] = r2[L.q * LDW + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
      for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int ot = 0; ot < OTP; ++ot) acc[t][ot] = mfma(wc[ot], b[t], acc[t][ot]);
    } else {
      // ---- K ring (fused_apply.hip): quarter q of group g: Z rows first (CT groups), then the layer input ----------------
      auto kq = [&](int g, int q) {
        if (g < CT) qload(zres, 16 * g, q);
        else qload(xres, 16 * (g - CT), q);
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) { qstore(q, false); kq(1, q); }
      float wc[2][4][OTP];
      auto cload = [&](int buf, int g) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ot = 0; ot < OTP; ++ot) wc[buf][s][ot] = buf_load1(wres, lq, ((16 * g + 4 * s) * CoP + 16 * ot) * 4);
      };
      cload(0, 0);
      float b[2][NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) b[0][t] = r2[L.q * LDW + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) cload((g + 1) & 1, g + 1);
        else {                                           // the ring registers are free: the next clip's group 0 takes off
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) qload(znext, 0, q);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s + 1 < 4 || g + 1 < NG) {
            const int sn = (s + 1) & 3;
#pragma unroll
            for (int t = 0; t < NTILE; ++t) b[(s + 1) & 1][t] = r2[(4 * sn + L.q) * LDW + (t < T ? t * V + L.j : jc * V + 16)];
          }
          if (g + 1 < NG) {
            qstore(s, g + 1 >= CT && pre);
            if (g + 2 < NG) kq(g + 2, s);
          }
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int ot = 0; ot < OTP; ++ot) acc[t][ot] = mfma(wc[g & 1][s][ot], b[s & 1][t], acc[t][ot]);
        }
      }
    }
    // ---- U = sums + bias -> image (the bias joins the finished sums: fused_apply.hip) -------------------------------------
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int ot = 0; ot < OTP; ++ot)
        tile_store(r1, 16 * ot, t < T ? t * V + L.j : jc * V + 16, t < T || L.j < T, acc[t][ot] + bq[ot], L);
    // the next layer's temporal table travels while the rows leave and the first Gram multiplies
    TTab tt;
    load_ttab(tt, tabres, 0, l16);
    // ---- U -> HBM while sum x x^T multiplies (X_next = PReLU(U) on the fly); the image keeps U --------------------------------
    if (!(FN_ABLATE & 2)) gram_rows(gx, sx, ores, true);
    // ---- Z_next = gcn_next(X_next) in place: temporal per joint (PReLU on the operand reads), spatial per frame -----------------
    L = geo();
    if (!(FN_ABLATE & 4)) temporal_phase<16, OTP, true>(r1, tt, L, a_out);
    L = geo();
#ifndef FN_OLD_SPATIAL
    if (!(FN_ABLATE & 8)) spatial_phase<OTP>(r1, tabres, 0, l16, L);
#else
    if (!(FN_ABLATE & 8)) {
      SpatRec rec = load_spat(tabres, 0, 0, l16);
      SOp op[OTP];
#pragma unroll
      for (int rt = 0; rt < OTP; ++rt) op[rt] = spatial_read<16>(r1, rt, 0, L);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : T - 1;
        const SpatRec nxt = load_spat(tabres, 0, tn, l16);
        SOp opn[OTP];
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) opn[rt] = spatial_read<16>(r1, rt, tn, L);
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) {
          const f32x4 d = spatial_mm(op[rt], rec);
          spatial_extra<16>(r1, rt, t, op[rt], rec, L);
          tile_store(r1, 16 * rt, t * V + L.j, true, d, L);
        }
        rec = nxt;
#pragma unroll
        for (int rt = 0; rt < OTP; ++rt) op[rt] = opn[rt];
      }
    }
#endif
    // ---- Z_next -> HBM while sum z z^T multiplies ------------------------------------------------------------------------------
    if (!(FN_ABLATE & 32)) gram_rows(gz, sz, zores, false);
  }

  // ---- block sum: the waves add their tiles into one LDS row one after another (fixed order), then the row leaves ------------
  float* row = lds_all;                                  // E floats (<= 8.4 KB) over wave 0's image: all clip loops are done
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
      const float t = quad_sum(s[rt]);                   // lane (j, q): the positions 8 m + 2 q, + 1 of row 16 rt + j
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

}  // namespace fn

bool layer_apply_next_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && (Ci == 2 || Ci == 16 || Ci == 32) && (Co == 16 || Co == 32);
}

int ftab_floats() { return ff::LAYER_F4 * 4; }

int launch_build_ftab(const float* const* A, const float* const* Tm, float* const* tab, int n, hipStream_t st) {
  if (n < 1 || n > 4) return fail(COSKAD_ERR_ARG, "build_ftab: 1..4 layers per launch, got %d", n);
  fn::FtabArgs args{};
  for (int i = 0; i < n; ++i) {
    if (!A[i] || !Tm[i] || !tab[i]) return fail(COSKAD_ERR_ARG, "build_ftab: null pointer (layer %d)", i);
    args.A[i] = A[i]; args.Tm[i] = Tm[i]; args.tab[i] = tab[i];
  }
  hipLaunchKernelGGL(fn::k_build_ftab, dim3(ceil_div(ff::LAYER_F4 * 4, 256), n), dim3(256), 0, st, args);
  return check_launch("build_ftab");
}

// fused_apply_next_bpc.hip: one clip per workgroup for the three layer shapes of the default stack
int apply_next_bpc_rows(int B);
int launch_layer_apply_next_bpc(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                const float* in_slope, const float* out_slope, const float* ftab, float* Znext, float* partials,
                                int B, int Ci, int Co, hipStream_t st);
// (the three layer shapes of the default stack: 78 / 76 / 108 us against 87 / 86 / 116 us of the wave-per-clip kernel below,
// B = 4096 on one box; the train step 1.572 -> 1.539 ms)
static bool apply_next_bpc_on(int Ci, int Co) { return (Ci == 2 && Co == 32) || (Ci == 32 && Co == 16) || (Ci == 16 && Co == 32); }
int layer_apply_next_rows(int B, int Ci, int Co) {
  if (apply_next_bpc_on(Ci, Co)) return apply_next_bpc_rows(B);
  const int nblk = (B + 3) / 4;
  return nblk < 256 ? nblk : 256;
}

// partial rows written: *rows_out, each 2 (Co^2 + Co) floats
int launch_layer_apply_next(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                            const float* in_slope, const float* out_slope, const float* ftab, float* Znext, float* partials,
                            int B, int Ci, int Co, hipStream_t st, int* rows_out) {
  const size_t lds = (size_t)4 * ff::WAVE_LDS_W * sizeof(float);
  const int grid = layer_apply_next_rows(B, Ci, Co);
  *rows_out = grid;
  if (apply_next_bpc_on(Ci, Co))
    return launch_layer_apply_next_bpc(Z, in, out, wfold, bias, in_slope, out_slope, ftab, Znext, partials, B, Ci, Co, st);
#define LAUNCH_FN(CT, OTP)                                                                                            \
  do {                                                                                                                \
    auto k = fn::k_layer_apply_next<CT, OTP>;                                                                         \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out_slope, ftab, out, Znext,  \
                       partials, B);                                                                                  \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 2 && Co == 16) LAUNCH_FN(0, 1);
    else if (Ci == 2 && Co == 32) LAUNCH_FN(0, 2);
    else if (Ci == 16 && Co == 16) LAUNCH_FN(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FN(1, 2);
    else if (Ci == 32 && Co == 16) LAUNCH_FN(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FN(2, 2);
    else return fail(COSKAD_ERR_SHAPE, "apply_next: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FN
  return check_launch("layer_apply_next");
}

}  // namespace coskad

using namespace coskad;
extern "C" {

int coskad_layer_apply_next_ok(int Ci, int Co, int T, int V) { return layer_apply_next_ok(T, V, Ci, Co) ? 1 : 0; }
int coskad_ftab_floats(void) { return ftab_floats(); }
int coskad_layer_apply_next_rows(int B, int Ci, int Co) { return layer_apply_next_rows(B, Ci, Co); }

int coskad_build_ftab_f32(const float* const* A, const float* const* Tm, float* const* tab, int n, int T, int V,
                          hipStream_t stream) {
  if (!A || !Tm || !tab) return fail(COSKAD_ERR_ARG, "build_ftab: null pointer");
  if (T != ff::T || V != ff::V) return fail(COSKAD_ERR_SHAPE, "build_ftab: built for T=12, V=17 (got %d, %d)", T, V);
  return launch_build_ftab(A, Tm, tab, n, stream);
}

int coskad_layer_apply_next_f32(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                                const float* in_slope, const float* out_slope, const float* ftab_next, float* Z_next,
                                float* partials, size_t partials_bytes, int B, int Ci, int Co, int T, int V,
                                hipStream_t stream) {
  if (!Z || !in || !out || !wfold || !bias || !out_slope || !ftab_next || !Z_next || !partials)
    return fail(COSKAD_ERR_ARG, "layer_apply_next: null pointer");
  if (B <= 0) return fail(COSKAD_ERR_ARG, "layer_apply_next: B=%d", B);
  if (!layer_apply_next_ok(T, V, Ci, Co))
    return fail(COSKAD_ERR_SHAPE, "layer_apply_next: built for T=12, V=17, C_in in {2,16,32}, C_out in {16,32} (got %d, %d, %d, %d)", T, V, Ci, Co);
  const size_t need = (size_t)coskad_layer_apply_next_rows(B, Ci, Co) * 2 * ((size_t)Co * Co + Co) * sizeof(float);
  if (partials_bytes < need) return fail(COSKAD_ERR_WORKSPACE, "layer_apply_next: partials %zu < %zu bytes", partials_bytes, need);
  int rows = 0;
  return launch_layer_apply_next(Z, in, out, wfold, bias, in_slope, out_slope, ftab_next, Z_next, partials, B, Ci, Co, stream, &rows);
}

}  // extern "C"
