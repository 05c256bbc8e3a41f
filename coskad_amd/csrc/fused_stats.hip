// Stage 1 of the layer backward on the stored-Z path, wave-per-clip: the batch reductions behind both BatchNorm
// backward folds (autograd of models/graph_layers/stsgcn.py:94-116 in training mode)
//     P[o][c] = sum_{b,pos} dU[b][o][pos] Z[b][c][pos]      Q[o][c] = sum dU[b][o][pos] PReLU(U_prev)[b][c][pos]
//     s[o]    = sum_{b,pos} dU[b][o][pos]
// for T = 12, V = 17, 16 or 32 input channels.  The contraction runs over positions: both MFMA operands are (row,
// position) reads of LDS row images of stride 206 (conflict-free; one ds_read_b64 serves two k-steps -- any assignment
// of positions to k slots is as good as any other as long as both operands use the same one).  One clip per wavefront,
// no workgroup barrier in the clip loop: the 32-row image holds the B side (Z and X together for 16 input channels; Z,
// then X in a second phase for 32), the 16-row window the current group of dU rows; the next group / the next source /
// the next clip's rows travel in registers while the current group multiplies.  Sums of all of a wave's clips stay in
// accumulator registers; the four waves of a block add theirs into one [P][Q][s] row at the very end (fixed order).
// Replaces k_bwd_reduce_z (block-per-tile: LDS images + a block barrier per 68-position slab) for these shapes.
#include "fused_ops.h"

namespace coskad {
namespace fs {

using namespace ff;

template <int CT, int OT>
__global__ __launch_bounds__(256, 1) void k_bwd_stats_ring(const float* __restrict__ in, const float* __restrict__ Zg,
                                                          const float* __restrict__ dU, const float* __restrict__ in_slope,
                                                          float* __restrict__ partials, int B) {
  constexpr int Ci = 16 * CT, Co = 16 * OT, NPH = CT;    // phases: one with Z | X side by side (16 channels), else Z then X
  constexpr int E = 2 * Co * Ci + Co;
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
  auto olane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int l16 = lane * 16;
  const int nwaves = gridDim.x * 4;
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
  };
  // ---- a 16-row group of dU through 16 registers per lane (quarters of 4 rows: 3 full 64-lane pieces + one of 12) ----------
  constexpr int QTAIL = 4 * (TV / 4) - 192;
  const int l16t = lane < QTAIL ? l16 : 0x7ffffff0;
  float4 gb[16];
  auto gload = [&](const BufRes& res, int row0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) gb[4 * q + c] = buf_load4(res, c < 3 ? l16 : l16t, ((row0 + 4 * q) * (TV / 4) + 64 * c) * 16);
  };
  auto gstore = [&]() {
    const int ln = olane();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int e = ln + 64 * c;
        const float4 v = gb[4 * q + c];
        const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
        float* d = r2 + (4 * q + row) * LD + col;
        if (c < 3 || lane < QTAIL) {
          *reinterpret_cast<float2*>(d) = float2{v.x, v.y};
          *reinterpret_cast<float2*>(d + 2) = float2{v.z, v.w};
        }
      }
  };
  // ---- 16 rows of a B-side source through 13 registers per lane into image rows [row0, row0 + 16) ----------------------------
  constexpr int SL = (16 * (TV / 4) + 63) / 64;          // 13
  auto sload = [&](float4 (&dst)[SL], const BufRes& res, int src_row0) {
#pragma unroll
    for (int i = 0; i < SL; ++i) dst[i] = buf_load4(res, l16, (src_row0 * (TV / 4) + 64 * i) * 16);
  };
  auto sstore = [&](const float4 (&src)[SL], int row0, bool act) {
    const int ln = olane();
    constexpr int n4 = 16 * (TV / 4);
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int e4 = ln + 64 * i;
      float4 v = src[i];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
      const bool ok = 64 * (i + 1) <= n4 || e4 < n4;
      *reinterpret_cast<float2*>(r1 + (ok ? (row0 + row) * LD + col : PADCOL)) = float2{v.x, v.y};
      *reinterpret_cast<float2*>(r1 + (ok ? (row0 + row) * LD + col + 2 : PADCOL)) = float2{v.z, v.w};
    }
  };
  // sums of all this wave's clips
  f32x4 pacc[OT][CT], qacc[OT][CT];
  float srow[OT];
#pragma unroll
  for (int g = 0; g < OT; ++g) {
    srow[g] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { pacc[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; qacc[g][ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  // source halves in flight: sa / sb = the two 16-row halves the next phase puts into the image
  float4 sa[SL], sb[SL];
  int clip = blockIdx.x * 4 + wave;
  {
    const BufRes z0 = clip_res(Zg, clip, Ci), x0 = clip_res(in, clip, Ci), d0 = clip_res(dU, clip, Co);
    sload(sa, z0, 0);
    if (CT == 1) sload(sb, x0, 0); else sload(sb, z0, 16);
    gload(d0, 0);
  }
  for (; clip < B; clip += nwaves) {
    const BufRes xres = clip_res(in, clip, Ci), dures = clip_res(dU, clip, Co);
    const BufRes zn = clip_res(Zg, clip + nwaves, Ci), xn = clip_res(in, clip + nwaves, Ci), dn = clip_res(dU, clip + nwaves, Co);
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      Lane L = geo();
      // the image takes this phase's two halves; the next phase's (or the next clip's first) take off
      const bool xphase = ph == 1;                       // (32 channels) second phase: both halves are X
      sstore(sa, 0, xphase && pre);
      sstore(sb, 16, (CT == 1 || xphase) && pre);
      if (CT == 2 && ph == 0) { sload(sa, xres, 0); sload(sb, xres, 16); }
      else if (CT == 2) { sload(sa, zn, 0); sload(sb, zn, 16); }
      else { sload(sa, zn, 0); sload(sb, xn, 0); }
#pragma unroll
      for (int g = 0; g < OT; ++g) {
        gstore();                                        // group g of dU -> window (all reads of the previous group are issued)
        if (g + 1 < OT) gload(dures, 16 * (g + 1));
        else if (ph + 1 < NPH) gload(dures, 0);
        else gload(dn, 0);
        // 26 double steps: lane (j, q) reads positions 8 m + 2 q, + 1 of row j of each operand
        const float* ap = r2 + L.j * LD + 2 * L.q;
        const float* bp0 = r1 + L.j * LD + 2 * L.q;
        const float* bp1 = r1 + (16 + L.j) * LD + 2 * L.q;
        f32x4 c0 = (CT == 2 && xphase) ? qacc[g][0] : pacc[g][0];
        f32x4 c1 = (CT == 1 || xphase) ? qacc[g][CT - 1] : pacc[g][CT - 1];
        float ssum = 0.f;
        constexpr int NM = (TV + 7) / 8;                 // 26
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          float2 a = *reinterpret_cast<const float2*>(ap + 8 * m);
          float2 b0 = *reinterpret_cast<const float2*>(bp0 + 8 * m);
          float2 b1 = *reinterpret_cast<const float2*>(bp1 + 8 * m);
          if (8 * (m + 1) > TV) {                        // the last step's tail lies in the rows' padding: zero the A side
            const bool ok = 8 * m + 2 * L.q < TV;
            a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f;
            b0.x = ok ? b0.x : 0.f; b0.y = ok ? b0.y : 0.f;
            b1.x = ok ? b1.x : 0.f; b1.y = ok ? b1.y : 0.f;
          }
          c0 = mfma(a.x, b0.x, c0);
          c1 = mfma(a.x, b1.x, c1);
          c0 = mfma(a.y, b0.y, c0);
          c1 = mfma(a.y, b1.y, c1);
          ssum += a.x + a.y;
        }
        if (CT == 2 && xphase) qacc[g][0] = c0; else pacc[g][0] = c0;
        if (CT == 1 || xphase) qacc[g][CT - 1] = c1; else pacc[g][CT - 1] = c1;
        if (ph == 0) srow[g] += ssum;
      }
    }
  }

  // ---- block sum: the waves add their tiles into one LDS row one after another (fixed order), then the row leaves ------------
  float* row = lds_all;                                  // E floats (<= 16.6 KB) over wave 0's image: all clip loops are done
  __syncthreads();
  const Lane L = geo();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int g = 0; g < OT; ++g) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * g + 4 * L.q + r, c = 16 * ct + L.j;   // D layout: register r <-> row 4 q + r, column j
            float* p = row + o * Ci + c;
            p[0] = (w ? p[0] : 0.f) + pacc[g][ct][r];
            p[Co * Ci] = (w ? p[Co * Ci] : 0.f) + qacc[g][ct][r];
          }
        const float s = quad_sum(srow[g]);               // lane (j, q): the positions 8 m + 2 q, + 1 of row 16 g + j
        if (L.q == 0) {
          float* p = row + 2 * Co * Ci + 16 * g + L.j;
          p[0] = (w ? p[0] : 0.f) + s;
        }
      }
    }
    __syncthreads();
  }
  float* dst = partials + (size_t)blockIdx.x * E;
  for (int e = threadIdx.x; e < E; e += 256) dst[e] = row[e];
}


// The same sums, ONE CLIP PER WORKGROUP, for 32 input channels and 32 / 64 output channels (the top layer of the default stack,
// whose dU comes from the bottleneck's split-K kernel -- no producer holds a clip's rows together for the backward chain):
// the image holds the clip's dU rows (Co x 206 floats), the B side -- Z's two 16-row groups, then PReLU(U_prev)'s -- alternates
// between two 16-row windows, every row staged by all 256 threads (one float4 each per quarter), ONE workgroup barrier per
// group; the 26 double k-steps of a group are dealt to the four waves round-robin (each keeps its own partial sums: they meet
// at the end of the launch).  The next group travels in 16 registers, the next clip's dU rows in 52, while the current group
// multiplies.  65.9 + 13.2 KB of LDS per workgroup: two workgroups per CU.
template <int CT, int OT>
__global__ __launch_bounds__(256, 2) void k_bwd_stats_bpc(const float* __restrict__ in, const float* __restrict__ Zg,
                                                         const float* __restrict__ dU, const float* __restrict__ in_slope,
                                                         float* __restrict__ partials, int B) {
  constexpr int Ci = 16 * CT, Co = 16 * OT, NG = 2 * CT, E = 2 * Co * Ci + Co;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r1 = lds;                       // dU image: Co rows (stride LD)
  float* w0 = lds + Co * LD;             // two 16-row windows (stride LD: (row, position) operand reads on both sides)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
  };
  constexpr int Q4 = 4 * (TV / 4);
  const bool stg = tid < Q4;
  const int srow = tid / (TV / 4), scol = 4 * (tid - srow * (TV / 4));
  const int svoff = stg ? tid * 16 : 0x7ffffff0;
  auto qload = [&](const BufRes& res, int row0, int q) { return buf_load4(res, svoff, (row0 + 4 * q) * (TV / 4) * 16); };
  auto qstore = [&](float* win, int q, float4 v, bool act) {
    if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
    // (threads beyond the quarter: both halves to the padding columns of the window's last row)
    *reinterpret_cast<float2*>(win + (stg ? (4 * q + srow) * LD + scol : 15 * LD + PADCOL)) = float2{v.x, v.y};
    *reinterpret_cast<float2*>(win + (stg ? (4 * q + srow) * LD + scol + 2 : 15 * LD + PADCOL)) = float2{v.z, v.w};
  };
  constexpr int N4 = Co * (TV / 4), XL = (N4 + 255) / 256;
  float4 xs[XL];
  auto xload = [&](const BufRes& r) {
#pragma unroll
    for (int i = 0; i < XL; ++i) xs[i] = buf_load4(r, (tid + 256 * i) < N4 ? (tid + 256 * i) * 16 : 0x7ffffff0, 0);
  };
  f32x4 acc[NG][OT];
  float rs[OT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int c = 0; c < OT; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < OT; ++c) rs[c] = 0.f;

  int clip = blockIdx.x;
  float4 gq[4];
  xload(clip_res(dU, clip, Co));
  {
    const BufRes z0 = clip_res(Zg, clip, Ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) gq[q] = qload(z0, 0, q);
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes zres = clip_res(Zg, clip, Ci), xres = clip_res(in, clip, Ci);
    const BufRes znext = clip_res(Zg, clip + gridDim.x, Ci);
    // group g: Z rows 16g .. (g < CT), then the layer input's; beyond this clip: the next clip's first group
    auto gload = [&](int g, int q) {
      return g < CT ? qload(zres, 16 * g, q) : (g < NG ? qload(xres, 16 * (g - CT), q) : qload(znext, 0, q));
    };
    __syncthreads();                                     // the previous clip's readers of the image and window 0 are done
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e4 = tid + 256 * i;
      const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
      *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LD + col : PADCOL)) = float2{xs[i].x, xs[i].y};
      *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LD + col + 2 : PADCOL)) = float2{xs[i].z, xs[i].w};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      qstore(w0, q, gq[q], false);
      gq[q] = gload(1, q);
    }
    xload(clip_res(dU, clip + gridDim.x, Co));           // the next clip's rows: a whole clip of MFMAs to arrive
    __syncthreads();                                     // the image holds dU, window 0 group 0
    constexpr int NM = (TV + 7) / 8;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // (window (g + 1) & 1 was last read in group g - 1: the barrier at its end has passed)
      if (g + 1 < NG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          qstore(w0 + ((g + 1) & 1) * 16 * LD, q, gq[q], g + 1 >= CT && pre);
          gq[q] = gload(g + 2, q);
        }
      }
      const Lane L = geo();
      const float* pb = w0 + (g & 1) * 16 * LD + L.j * LD + 2 * L.q;
      const float* pa = r1 + L.j * LD + 2 * L.q;
      // (operands read just in time, the loop rolled: reading step i + 1's in front of step i's products measured 20 us
      // slower per launch, rolled or unrolled)
      for (int m = wave; m < NM; m += 4) {
        const bool ok = 8 * m + 2 * L.q < TV;            // (beyond the row: the window's / image's next row -- masked)
        float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
        b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
        for (int c = 0; c < OT; ++c) {
          float2 a = *reinterpret_cast<const float2*>(pa + 16 * c * LD + 8 * m);
          a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f;
          acc[g][c] = mfma(a.x, b.x, acc[g][c]);
          acc[g][c] = mfma(a.y, b.y, acc[g][c]);
          if (g == 0) rs[c] += a.x + a.y;
        }
      }
      if (g + 1 < NG) __syncthreads();                   // group g + 1 is staged; every wave has left window g & 1
    }
  }
  // ---- the workgroup's sums: the waves add theirs into ONE [P][Q][s] row in LDS one after another (fixed order) ----------------
  __syncthreads();
  for (int e = tid; e < E; e += 256) lds[e] = 0.f;
  __syncthreads();
  const Lane L = geo();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < OT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o = 16 * c + 4 * L.q + r;
            lds[(g < CT ? 0 : Co * Ci) + o * Ci + 16 * (g < CT ? g : g - CT) + L.j] += acc[g][c][r];
          }
#pragma unroll
      for (int c = 0; c < OT; ++c) {
        const float t = quad_sum(rs[c]);
        if (L.q == 0) lds[2 * Co * Ci + 16 * c + L.j] += t;
      }
    }
    __syncthreads();
  }
  float* dst = partials + (size_t)blockIdx.x * E;
  for (int e = tid; e < E; e += 256) dst[e] = lds[e];
}


// The same sums on layouts other than 12 x 17: nothing in them sees frames or joints, so k_bwd_stats_bpc's scheme runs over a
// clip's FLAT T V positions, templated on T V (built for the 25-joint layout, T V = 300).  The image holds up to 32 rows of dU
// at a time (64 output channels: two passes, the B side -- Z's 16-row groups, then PReLU(U_prev)'s -- streamed through the two
// windows once per pass, the second time from L2): (32 + 2 x 16) x 302 floats = 77 KB per workgroup, two workgroups per CU.
template <int TVg, int CT, int OT>
__global__ __launch_bounds__(256, 2) void k_bwd_stats_flat(const float* __restrict__ in, const float* __restrict__ Zg,
                                                          const float* __restrict__ dU, const float* __restrict__ in_slope,
                                                          float* __restrict__ partials, int B) {
  static_assert(TVg % 4 == 0, "rows are staged as float4");
  constexpr int Ci = 16 * CT, Co = 16 * OT, NG = 2 * CT, E = 2 * Co * Ci + Co;
  constexpr int OTP = OT < 2 ? OT : 2, NP = OT / OTP;    // dU tiles per pass, passes
  constexpr int R4 = TVg / 4, LDg = TVg + 2;             // float4 per row; row stride of the image and the windows
  constexpr int Q4 = 4 * R4, NQ = (Q4 + 255) / 256;      // float4 of a quarter (4 rows), pieces per thread
  constexpr int N4 = 16 * OTP * R4, XL = (N4 + 255) / 256;   // float4 of a pass's dU rows, pieces per thread
  constexpr int NM = (TVg + 7) / 8;                      // double k-steps
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* r1 = lds;                       // dU image: 32 rows
  float* w0 = lds + 32 * LDg;            // two 16-row windows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TVg, in_range ? rows * TVg * 4u : 0u);
  };
  float4 gq[4][NQ];
  auto qload = [&](const BufRes& res, int row0, int q) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      gq[q][i] = buf_load4(res, e < Q4 ? e * 16 : 0x7ffffff0, (row0 + 4 * q) * R4 * 16);
    }
  };
  auto qstore = [&](float* win, int q, bool act) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + 256 * i;
      float4 v = gq[q][i];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / R4, col = 4 * (e - row * R4);
      // (pieces beyond the quarter: both halves to the padding columns of the window's last row)
      *reinterpret_cast<float2*>(win + (e < Q4 ? (4 * q + row) * LDg + col : 15 * LDg + TVg)) = float2{v.x, v.y};
      *reinterpret_cast<float2*>(win + (e < Q4 ? (4 * q + row) * LDg + col + 2 : 15 * LDg + TVg)) = float2{v.z, v.w};
    }
  };
  float4 xs[XL];
  auto xload = [&](const BufRes& r, int row0) {
#pragma unroll
    for (int i = 0; i < XL; ++i) xs[i] = buf_load4(r, (tid + 256 * i) < N4 ? (tid + 256 * i) * 16 : 0x7ffffff0, row0 * R4 * 16);
  };
  f32x4 acc[NP][NG][OTP];
  float rs[NP][OTP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int c = 0; c < OTP; ++c) acc[p][g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < OTP; ++c) rs[p][c] = 0.f;
  }

  int clip = blockIdx.x;
  xload(clip_res(dU, clip, Co), 0);
  {
    const BufRes z0 = clip_res(Zg, clip, Ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) qload(z0, 0, q);
  }
  for (; clip < B; clip += gridDim.x) {
    const BufRes zres = clip_res(Zg, clip, Ci), xres = clip_res(in, clip, Ci), dures = clip_res(dU, clip, Co);
    const BufRes znext = clip_res(Zg, clip + gridDim.x, Ci), dunext = clip_res(dU, clip + gridDim.x, Co);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // group g of this pass: Z rows 16g .. (g < CT), then the layer input's; beyond: the first group of the next pass / clip
      auto gload = [&](int g, int q) {
        if (g < CT) qload(zres, 16 * g, q);
        else if (g < NG) qload(xres, 16 * (g - CT), q);
        else qload(p + 1 < NP ? zres : znext, 0, q);
      };
      __syncthreads();                                   // the previous pass's readers of the image and window 0 are done
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e4 = tid + 256 * i;
        const int row = e4 / R4, col = 4 * (e4 - row * R4);
        *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LDg + col : TVg)) = float2{xs[i].x, xs[i].y};
        *reinterpret_cast<float2*>(r1 + (e4 < N4 ? row * LDg + col + 2 : TVg)) = float2{xs[i].z, xs[i].w};
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        qstore(w0, q, false);
        gload(1, q);
      }
      if (p + 1 < NP) xload(dures, 16 * OTP * (p + 1));  // the next pass's rows / the next clip's: a whole pass of MFMAs to arrive
      else xload(dunext, 0);
      __syncthreads();                                   // the image holds the pass's dU rows, window 0 group 0
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // (window (g + 1) & 1 was last read in group g - 1: the barrier at its end has passed)
        if (g + 1 < NG) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            qstore(w0 + ((g + 1) & 1) * 16 * LDg, q, g + 1 >= CT && pre);
            gload(g + 2, q);
          }
        }
        const Lane L = geo();
        const float* pb = w0 + (g & 1) * 16 * LDg + L.j * LDg + 2 * L.q;
        const float* pa = r1 + L.j * LDg + 2 * L.q;
        for (int m = wave; m < NM; m += 4) {
          const bool ok = 8 * m + 2 * L.q < TVg;         // (beyond the row: the next row / the padding -- masked)
          float2 b = *reinterpret_cast<const float2*>(pb + 8 * m);
          b.x = ok ? b.x : 0.f; b.y = ok ? b.y : 0.f;
#pragma unroll
          for (int c = 0; c < OTP; ++c) {
            float2 a = *reinterpret_cast<const float2*>(pa + 16 * c * LDg + 8 * m);
            a.x = ok ? a.x : 0.f; a.y = ok ? a.y : 0.f;
            acc[p][g][c] = mfma(a.x, b.x, acc[p][g][c]);
            acc[p][g][c] = mfma(a.y, b.y, acc[p][g][c]);
            if (g == 0) rs[p][c] += a.x + a.y;
          }
        }
        if (g + 1 < NG) __syncthreads();                 // group g + 1 is staged; every wave has left window g & 1
      }
    }
  }
  // ---- the workgroup's sums: the waves add theirs into ONE [P][Q][s] row in LDS one after another (fixed order) ----------------
  __syncthreads();
  for (int e = tid; e < E; e += 256) lds[e] = 0.f;
  __syncthreads();
  const Lane L = geo();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int c = 0; c < OTP; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int o = 16 * (OTP * p + c) + 4 * L.q + r;
              lds[(g < CT ? 0 : Co * Ci) + o * Ci + 16 * (g < CT ? g : g - CT) + L.j] += acc[p][g][c][r];
            }
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c = 0; c < OTP; ++c) {
          const float t = quad_sum(rs[p][c]);
          if (L.q == 0) lds[2 * Co * Ci + 16 * (OTP * p + c) + L.j] += t;
        }
    }
    __syncthreads();
  }
  float* dst = partials + (size_t)blockIdx.x * E;
  for (int e = tid; e < E; e += 256) dst[e] = lds[e];
}

}  // namespace fs

// Used where it wins (B = 4096, in-step timings): 32 -> 16 channels 54 vs 73 us, 16 -> 32 channels 44-55 vs 55 us; at
// 32 -> 64 the rows of dU pass the window twice (two phases x four groups) and the block-per-tile kernel is faster (125 vs
// 148 us): wide outputs on 32 input channels take k_bwd_stats_bpc below (118 us).
bool bwd_stats_ring_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && ((Ci == 16 && (Co == 16 || Co == 32 || Co == 64)) || (Ci == 32 && Co == 16));
}

// partial rows written: *rows_out (<= 256), each 2 Co Ci + Co floats
int launch_bwd_stats_ring(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                          int Ci, int Co, hipStream_t st, int* rows_out) {
  const size_t lds = (size_t)4 * ff::WAVE_LDS * sizeof(float);
  const int nblk = (B + 3) / 4;
  const int grid = nblk < 256 ? nblk : 256;
  *rows_out = grid;
#define LAUNCH_FS(CT, OT)                                                                                        \
  do {                                                                                                           \
    auto k = fs::k_bwd_stats_ring<CT, OT>;                                                                       \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, in_slope, partials, B);                    \
  } while (0)
  {
    ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FS(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FS(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_FS(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_FS(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FS(2, 2);
    else if (Ci == 32 && Co == 64) LAUNCH_FS(2, 4);
    else return fail(COSKAD_ERR_SHAPE, "bwd_stats_ring: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FS
  return check_launch("bwd_stats_ring");
}


// One clip per workgroup: 32 input channels, 32 / 64 output channels.  Partial rows written: *rows_out (<= 512).
bool bwd_stats_bpc_ok(int T_, int V_, int Ci, int Co) { return T_ == ff::T && V_ == ff::V && Ci == 32 && (Co == 32 || Co == 64); }

int launch_bwd_stats_bpc(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                         int Ci, int Co, hipStream_t st, int* rows_out) {
  const size_t lds = (size_t)(Co + 32) * ff::LD * sizeof(float);
  const int per_cu = lds <= (size_t)52 * 1024 ? 3 : 2;
  const int grid = B < 256 * per_cu ? B : 256 * per_cu;
  *rows_out = grid;
#define LAUNCH_FSB(CT, OT)                                                                                       \
  do {                                                                                                           \
    auto k = fs::k_bwd_stats_bpc<CT, OT>;                                                                        \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, in_slope, partials, B);                    \
  } while (0)
  {
    ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
    if (Ci == 32 && Co == 32) LAUNCH_FSB(2, 2);
    else if (Ci == 32 && Co == 64) LAUNCH_FSB(2, 4);
    else return fail(COSKAD_ERR_SHAPE, "bwd_stats_bpc: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FSB
  return check_launch("bwd_stats_bpc");
}


// Flat-position form for other layouts (the 25-joint one): 16 / 32 input, 16 / 32 / 64 output channels.  *rows_out <= 512.
bool bwd_stats_flat_ok(int TV_, int Ci, int Co) { return TV_ == 300 && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64); }

int launch_bwd_stats_flat(const float* in, const float* Zg, const float* dU, const float* in_slope, float* partials, int B,
                          int Ci, int Co, int TV_, hipStream_t st, int* rows_out) {
  if (!bwd_stats_flat_ok(TV_, Ci, Co)) return fail(COSKAD_ERR_SHAPE, "bwd_stats_flat: unsupported shape (%d positions, %d -> %d)", TV_, Ci, Co);
  constexpr int TVg = 300;
  const size_t lds = (size_t)64 * (TVg + 2) * sizeof(float);
  const int grid = B < 512 ? B : 512;
  *rows_out = grid;
#define LAUNCH_FSF(CT, OT)                                                                                       \
  do {                                                                                                           \
    auto k = fs::k_bwd_stats_flat<TVg, CT, OT>;                                                                  \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Zg, dU, in_slope, partials, B);                    \
  } while (0)
  {
    ProbeScope probe(KID_BWD_REDUCE, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FSF(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FSF(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_FSF(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_FSF(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FSF(2, 2);
    else LAUNCH_FSF(2, 4);
  }
#undef LAUNCH_FSF
  return check_launch("bwd_stats_flat");
}

}  // namespace coskad
