// Training-mode layer apply on the stored-Z path, ONE CLIP PER WORKGROUP (reference: models/graph_layers/stsgcn.py:94-116 with
// both BatchNorms folded from this batch's statistics):
//     U[o][pos] = sum_c Wz[c][o] Z[c][pos] + sum_c Wx[c][o] PReLU(U_prev)[c][pos] + b[o]
// The wave-per-clip kernel (fused_apply.hip) gives every wavefront its own 39.7 KB of LDS and up to 208 accumulators: one wave
// per SIMD, nothing to run while it waits.  Here the four waves of a workgroup SHARE one clip's K window and flush image
// (39.7 KB per workgroup: four workgroups = sixteen waves per CU, four per SIMD), each wave owns one 16-channel output tile of all
// 13 position tiles (52 accumulators), the K rows are staged by all 256 threads (one float4 each per quarter), and a workgroup
// barrier per k-step separates "every wave has read rows 4s.." from "rows 4s.. take the next group's quarter".
// Built for 32 -> 64 channels at T = 12, V = 17 (layer 4 of the default stack).
#include "fused_ops.h"

namespace coskad {
namespace fpc {

using namespace ff;

// Measured (B = 4096, 32 -> 64; the wave-per-clip ring kernel: 114-116 us on the same box): two waves per SIMD 101-104 us, three
// (148 registers with ONE set of B operands, read behind the MFMAs that used the previous ones) 98-101 us, four waves per SIMD only
// with scratch spills in the K loop (212 B: 214 us).  Two groups of K rows in flight (DEPTH 2) beat one by 2 us.
#ifndef BPC_GRID
#define BPC_GRID (256 * BPC_OCC)   // every workgroup resident: 92-94 us against 96 with 1024 (a second, thinner round of workgroups)
#endif
#ifndef BPC_OCC
#define BPC_OCC 3
#endif
template <int CT>
__global__ __launch_bounds__(256, BPC_OCC) void k_layer_apply_bpc(const float* __restrict__ in, const float* __restrict__ Zg,
                                                           const float* __restrict__ wfold, const float* __restrict__ bias,
                                                           const float* __restrict__ in_slope, float* __restrict__ out, int B) {
  constexpr int Ci = 16 * CT, Co = 64, CoP = Co, NG = 2 * CT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r1 = lds;                 // 32-row flush image (stride LD)
  float* r2 = lds + 32 * LD;       // 16-row K window (stride LDW)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // lane geometry behind an optimisation barrier, refreshed where it is used: address arithmetic is then recomputed instead of being
  // hoisted out of the clip loop and parked in registers (fused_bwd.hip)
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  Lane L = geo();
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const BufRes wres = make_res(wfold, 2 * Ci * CoP * 4u);
  const BufRes bres = make_res(bias, CoP * 4u);
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
#ifdef COSKAD_HOT   // timing-only: every stream from 64 L2-resident clips
    return make_res(base + (size_t)(in_range ? (c & 63) : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#else
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#endif
  };
  // staging: thread t < 204 owns float4 `t` of a quarter (4 rows x 51 float4)
  constexpr int Q4 = 4 * (TV / 4);
  const bool stg = tid < Q4;
  const int srow = tid / (TV / 4), scol = 4 * (tid - srow * (TV / 4));
  const int svoff = stg ? tid * 16 : 0x7ffffff0;           // (beyond the quarter: out of range -> 0, no traffic)
#ifndef BPC_DEPTH
#define BPC_DEPTH 2
#endif
  constexpr int DEPTH = BPC_DEPTH;                         // groups in flight (register sets)
  float4 gq[DEPTH][4];                                     // [set][quarter]
  auto qload = [&](const BufRes& res, int row0, int q) { return buf_load4(res, svoff, (row0 + 4 * q) * (TV / 4) * 16); };
  auto qstore = [&](int q, float4 v, bool act) {
    if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
    if (stg) *reinterpret_cast<float4*>(r2 + (4 * q + srow) * LDW + scol) = v;
  };

  int clip = blockIdx.x;
  {
    const BufRes z0 = clip_res(Zg, clip, Ci), x0 = clip_res(in, clip, Ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gq[0][q] = qload(z0, 0, q);
      if (DEPTH == 2) gq[DEPTH - 1][q] = CT > 1 ? qload(z0, 16, q) : qload(x0, 0, q);
    }
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), ores = clip_res(out, clip, Co);
    const BufRes znext = clip_res(Zg, clip + gridDim.x, Ci), xnext = clip_res(in, clip + gridDim.x, Ci);
    // virtual group vg: this clip's groups 0 .. NG-1 (Z rows, then the layer input), then the next clip's
    auto vload = [&](int vg, int q) {
      const bool nxt = vg >= NG;
      const int g = nxt ? vg - NG : vg;
      return g < CT ? qload(nxt ? znext : zres, 16 * g, q) : qload(nxt ? xnext : xres, 16 * (g - CT), q);
    };
    L = geo();
    const int jc = L.j < T ? L.j : T - 1;
    f32x4 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // entry: set 0 = group 0, set 1 = group 1 (fetched during the previous clip).  The previous clip's flush ended with a barrier.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      qstore(q, gq[0][q], false);
      gq[0][q] = vload(DEPTH, q);
    }
    const int lq = (L.q * CoP + 16 * wave + L.j) * 4;      // this wave's output tile of the folded weights
    float wc[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) wc[0][s] = buf_load1(wres, lq, (4 * s) * CoP * 4);
    __syncthreads();                                       // the window holds group 0
#ifndef BPC_BDBL
#define BPC_BDBL 0   // operands of k-step s+1 read while step s multiplies (two register sets) / read just in time (one)
#endif
    float b[1 + BPC_BDBL][NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) b[0][t] = r2[L.q * LDW + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
#pragma unroll
        for (int s = 0; s < 4; ++s) wc[(g + 1) & 1][s] = buf_load1(wres, lq, ((16 * (g + 1) + 4 * s) * CoP) * 4);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (BPC_BDBL && (s + 1 < 4 || g + 1 < NG)) {       // operands of the next k-step (the next group's first: stored at s = 0)
          const int sn = (s + 1) & 3;
#pragma unroll
          for (int t = 0; t < NTILE; ++t) b[(s + 1) & 1][t] = r2[(4 * sn + L.q) * LDW + (t < T ? t * V + L.j : jc * V + 16)];
        }
        // every wave has read rows 4s .. 4s+3 (a k-step ago); in the last group: its fourth quarter (stored a k-step ago) is visible
        if (g + 1 < NG || s == 0) __syncthreads();
        if (g + 1 < NG) {
          qstore(s, gq[(g + 1) % DEPTH][s], g + 1 >= CT && pre);
          if (DEPTH == 2 || g + 2 < NG) gq[(g + 1) % DEPTH][s] = vload(g + 1 + DEPTH, s);   // (beyond this clip: the next clip's first groups)
        }
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc[t] = mfma(wc[g & 1][s], b[BPC_BDBL ? (s & 1) : 0][t], acc[t]);
        if (!BPC_BDBL && (s + 1 < 4 || g + 1 < NG)) {      // single set: the next k-step's operands behind this step's MFMAs
          const int sn = (s + 1) & 3;
#pragma unroll
          for (int t = 0; t < NTILE; ++t) b[0][t] = r2[(4 * sn + L.q) * LDW + (t < T ? t * V + L.j : jc * V + 16)];
        }
      }
    }
    if (DEPTH == 1) {                                      // the next clip's group 0 takes off behind the last group
#pragma unroll
      for (int q = 0; q < 4; ++q) gq[0][q] = vload(NG, q);
    }
    L = geo();
    const float4 b4 = buf_load4(bres, L.q * 16, (16 * wave) * 4);
    const f32x4 bq = {b4.x, b4.y, b4.z, b4.w};
    const int jf = L.j < T ? L.j : T - 1;
    // ---- flush: 32 channels at a time through the image (waves 0, 1, then 2, 3), full lines to HBM -------------------------
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      if ((wave >> 1) == rnd) {
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
          tile_store(r1, 16 * (wave & 1), t < T ? t * V + L.j : jf * V + 16, t < T || L.j < T, acc[t] + bq, L);
      }
      __syncthreads();
      constexpr int n4 = 32 * (TV / 4);                    // 1632 float4 of 32 rows
#pragma unroll
      for (int i = 0; i < (n4 + 255) / 256; ++i) {
        const int e4 = tid + 256 * i;
        const bool ok = e4 < n4;
        const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
        const float* p = r1 + (ok ? row * LD + col : PADCOL);
        const float2 g0 = *reinterpret_cast<const float2*>(p), g1 = *reinterpret_cast<const float2*>(p + 2);
        buf_store4(ores, ok ? e4 * 16 : 0x7ffffff0, (32 * rnd) * (TV / 4) * 16, float4{g0.x, g0.y, g1.x, g1.y});
      }
      __syncthreads();                                     // (the image / the window are rewritten next)
    }
  }
}

}  // namespace fpc

int launch_layer_apply_bpc(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                           const float* in_slope, int B, int Ci, int Co, hipStream_t st) {
  if (!(Ci == 32 && Co == 64)) return fail(COSKAD_ERR_SHAPE, "apply_bpc: built for 32 -> 64 channels (%d, %d)", Ci, Co);
  const size_t lds = (size_t)ff::WAVE_LDS_W * sizeof(float);
  const int grid = B < BPC_GRID ? B : BPC_GRID;
  auto k = fpc::k_layer_apply_bpc<2>;
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out, B);
  }
  return check_launch("layer_apply_bpc");
}

}  // namespace coskad
