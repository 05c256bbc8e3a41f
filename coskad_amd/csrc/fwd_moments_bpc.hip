// The statistics pass of a training-mode layer on the 25-joint layout (reference models/graph_layers/stsgcn.py:56-80, 94-110;
// what k_fwd_moments, stsgcn_train.hip, computes): per clip X = PReLU(U_prev), sum x x^T and sum x, Z = gcn(X) stored for the
// rest of the step, sum z z^T and sum z -- ONE CLIP PER WORKGROUP OF FOUR WAVES.
//
// k_fwd_moments keeps the clip image AND both mixing tables (44 KB at 25 joints) in LDS: one 16-wave block per CU, every phase
// behind a block-wide barrier (150 us per call at 32 channels, B = 4096).  Here, as in gcn_params_bpc.hip, the work is dealt by
// joint (temporal mix) and by frame (spatial mix), so a wave's mixing operands are the same for every clip and live in 63
// registers; the image takes 38.6 KB (three workgroups per CU at 168 registers); the Gram sums are the (row, position)
// `ds_read_b64` products of fused_apply_next_bpc.hip over flat positions, their 38 double k-steps dealt to the waves; the next
// clip's rows travel in registers.  One partial row [MX][sumX][MZ][sumZ] per workgroup (k_reduce_partials + k_train_fold finish).
#include "fused_ops.h"

namespace coskad {
namespace fm {

using ff::f32x4;
using ff::Lane;
using ff::mfma;
using ff::prelu;
using ff::quad_sum;

// CT: 16-row groups of the input; COMB (16 channels): the input is formed on the way in from the two branches of a commuted layer
// (commute_layer.hip): U = a_t Zy + a_r R + shift per channel (`in` = Zy [B, 16, TV], `Rr` = the R rows of [B, 32, TV], `stat` = a_t | a_r |
// shift), stored to `Uout` for the layer's other readers -- the commuted layer's element-wise combine pass and this pass's read of it gone
template <int V, int CT, bool COMB = false>
__global__ __launch_bounds__(256, (CT == 1 ? 3 : 2)) void k_fwd_moments_bpc(const float* __restrict__ in, const float* __restrict__ Aw,
                                                           const float* __restrict__ Tw, const float* __restrict__ in_slope,
                                                           float* __restrict__ partials, int B, int need_x,
                                                           float* __restrict__ Zout, const float* __restrict__ Rr,
                                                           const float* __restrict__ stat, float* __restrict__ Uout) {
  static_assert(!COMB || CT == 1, "the commuted layers have 16 output channels");
  __shared__ float cst[COMB ? 48 : 1];
  if constexpr (COMB) {
    if (threadIdx.x < 48) cst[threadIdx.x] = stat[threadIdx.x];
  }
  constexpr int T = 12, TV = T * V, LD = TV + 2, R4 = TV / 4, Ci = 16 * CT;
  static_assert(TV % 4 == 0, "rows are staged as float4");
  constexpr int N4 = Ci * R4, XL = (N4 + 255) / 256;
  constexpr int NTV = (V + 15) / 16, KV = (V + 3) / 4;
  constexpr int MAXF = T / 4, MAXJ = (V + 3) / 4;
  constexpr int NACC = CT == 1 ? 2 : 3;                  // Gram accumulators: .x / .y halves of the one block, or blocks 00, 01, 11
  constexpr int E = 2 * (Ci * Ci + Ci);
  constexpr int NM = (TV + 7) / 8;                       // double k-steps over the positions
  extern __shared__ __attribute__((aligned(16))) float img[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lane L{lane & 15, lane >> 4};
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  // a wave's joints and frames are the same for every clip: its B operands of both mixes stay in registers
  //   temporal  Y[q,v] = sum_t X[t,v] T[v][t][q]:   B[k = t][j = q]
  //   spatial   Z[t,w] = sum_v Y[t,v] A[t][v][w]:   B[k = v][j = w]
  float tbv[MAXJ][3], bbv[MAXF][NTV][KV];
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) {
    const int v = wave + 4 * k;
#pragma unroll
    for (int s = 0; s < 3; ++s) tbv[k][s] = (v < V && L.j < T) ? Tw[(v * T + 4 * s + L.q) * T + L.j] : 0.f;
  }
#pragma unroll
  for (int tt = 0; tt < MAXF; ++tt) {
    const int t = wave + 4 * tt;
#pragma unroll
    for (int c = 0; c < NTV; ++c)
#pragma unroll
      for (int s = 0; s < KV; ++s)
        bbv[tt][c][s] = (16 * c + L.j < V && 4 * s + L.q < V) ? Aw[(t * V + 4 * s + L.q) * V + 16 * c + L.j] : 0.f;
  }
  f32x4 gx[NACC], gz[NACC];
  float sx[CT], sz[CT];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { gx[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gz[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int i = 0; i < CT; ++i) { sx[i] = 0.f; sz[i] = 0.f; }
  // this wave's double k-steps m = wave, wave + 4, .. of the image's Gram sum
  auto gram = [&](f32x4 (&g)[NACC], float (&s)[CT]) {
    const float* p0 = img + L.j * LD + 2 * L.q;
    const float* p1 = img + (16 + L.j) * LD + 2 * L.q;
    for (int m = wave; m < NM; m += 4) {
      const bool ok = 8 * m + 2 * L.q < TV;              // (beyond the row: the next row / the padding -- masked)
      float2 a0 = *reinterpret_cast<const float2*>(p0 + 8 * m);
      a0.x = ok ? a0.x : 0.f; a0.y = ok ? a0.y : 0.f;
      if constexpr (CT == 1) {
        g[0] = mfma(a0.x, a0.x, g[0]);
        g[1] = mfma(a0.y, a0.y, g[1]);
        s[0] += a0.x + a0.y;
      } else {
        float2 a1 = *reinterpret_cast<const float2*>(p1 + 8 * m);
        a1.x = ok ? a1.x : 0.f; a1.y = ok ? a1.y : 0.f;
        g[0] = mfma(a0.x, a0.x, g[0]);
        g[1] = mfma(a0.x, a1.x, g[1]);
        g[2] = mfma(a1.x, a1.x, g[2]);
        g[0] = mfma(a0.y, a0.y, g[0]);
        g[1] = mfma(a0.y, a1.y, g[1]);
        g[2] = mfma(a1.y, a1.y, g[2]);
        s[0] += a0.x + a0.y;
        s[CT - 1] += a1.x + a1.y;
      }
    }
  };
  float4 px[XL], pr[COMB ? XL : 1];
  auto xload = [&](int clip) {
    const float4* g4 = reinterpret_cast<const float4*>(in + (size_t)(clip < B ? clip : 0) * Ci * TV);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + 256 * i;
      px[i] = (e < N4 && clip < B) ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (COMB) {
      const float4* r4 = reinterpret_cast<const float4*>(Rr + (size_t)(clip < B ? clip : 0) * 2 * Ci * TV);
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e = tid + 256 * i;
        pr[i] = (e < N4 && clip < B) ? r4[e] : float4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  int clip = blockIdx.x;
  xload(clip);
  for (; clip < B; clip += gridDim.x) {
    __syncthreads();                                     // the previous clip's readers of the image are done
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + 256 * i;
      if (e < N4) {
        const int row = e / R4, col = 4 * (e - row * R4);
        float4 v = px[i];
        if constexpr (COMB) {
          const float at = cst[row], ar = cst[16 + row], sh = cst[32 + row];
          const float4 r = pr[i];
          v = float4{fmaf(at, v.x, fmaf(ar, r.x, sh)), fmaf(at, v.y, fmaf(ar, r.y, sh)), fmaf(at, v.z, fmaf(ar, r.z, sh)),
                     fmaf(at, v.w, fmaf(ar, r.w, sh))};
          reinterpret_cast<float4*>(Uout + (size_t)clip * Ci * TV)[e] = v;
        }
        if (pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        *reinterpret_cast<float2*>(img + row * LD + col) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(img + row * LD + col + 2) = float2{v.z, v.w};
      }
    }
    xload(clip + gridDim.x);                             // the next clip's rows: a whole clip of products to arrive
    __syncthreads();                                     // the image holds X
    if (need_x) {
      gram(gx, sx);
      __syncthreads();                                   // every wave has read X
    }
    // ---- Y = temporal mix of X, in place: joints v = wave, wave + 4, .. --------------------------------------------------------
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
#pragma unroll
        for (int rt = 0; rt < CT; ++rt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) d = mfma(img[(16 * rt + L.j) * LD + (4 * s + L.q) * V + v], tbv[k][s], d);
          if (L.j < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * rt + 4 * L.q + r) * LD + L.j * V + v] = d[r];
          }
        }
      }
    }
    __syncthreads();                                     // the image holds Y
    // ---- Z = spatial mix of Y, in place: frames t = wave, wave + 4, wave + 8 ---------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) {
        float a[KV];
#pragma unroll
        for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? img[(16 * rt + L.j) * LD + t * V + 4 * s + L.q] : 0.f;
        f32x4 d[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], bbv[tt][c][s], d[c]);
        }
#pragma unroll
        for (int c = 0; c < NTV; ++c)
          if (16 * c + L.j < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * rt + 4 * L.q + r) * LD + t * V + 16 * c + L.j] = d[c][r];
          }
      }
    }
    __syncthreads();                                     // the image holds Z
    // ---- Z -> HBM (full lines, all threads), sum z z^T ----------------------------------------------------------------------------
    {
      float4* g4 = reinterpret_cast<float4*>(Zout + (size_t)clip * Ci * TV);
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e = tid + 256 * i;
        if (e < N4) {
          const int row = e / R4, col = 4 * (e - row * R4);
          const float2 g0 = *reinterpret_cast<const float2*>(img + row * LD + col);
          const float2 g1 = *reinterpret_cast<const float2*>(img + row * LD + col + 2);
          g4[e] = float4{g0.x, g0.y, g1.x, g1.y};
        }
      }
    }
    gram(gz, sz);
  }
  // ---- workgroup sum: the waves add their tiles into one row in LDS one after another (fixed order), then the row leaves --------
  float* row = img;                                      // E floats over the image: all clip loops are done
  __syncthreads();
  auto put = [&](int w, float* base, const f32x4 (&g)[NACC], const float (&s)[CT]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // D layout: register r <-> row 4 q + r, column j
      const int i = 4 * L.q + r, j = L.j;
      if constexpr (CT == 1) {
        float* p = base + i * Ci + j;
        p[0] = (w ? p[0] : 0.f) + (g[0][r] + g[1][r]);
      } else {
        float* p00 = base + i * Ci + j;
        float* p01 = base + i * Ci + 16 + j;
        float* p10 = base + (16 + j) * Ci + i;
        float* p11 = base + (16 + i) * Ci + 16 + j;
        p00[0] = (w ? p00[0] : 0.f) + g[0][r];
        p01[0] = (w ? p01[0] : 0.f) + g[1][r];
        p10[0] = (w ? p10[0] : 0.f) + g[1][r];
        p11[0] = (w ? p11[0] : 0.f) + g[2][r];
      }
    }
#pragma unroll
    for (int rt = 0; rt < CT; ++rt) {
      const float t = quad_sum(s[rt]);
      if (L.q == 0) {
        float* p = base + Ci * Ci + 16 * rt + L.j;
        p[0] = (w ? p[0] : 0.f) + t;
      }
    }
  };
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      put(w, row, gx, sx);
      put(w, row + Ci * Ci + Ci, gz, sz);
    }
    __syncthreads();
  }
  float* dst = partials + (size_t)blockIdx.x * E;
  for (int e = tid; e < E; e += 256) dst[e] = row[e];
}

}  // namespace fm

bool fwd_moments_bpc_ok(int T_, int V_, int Ci) { return T_ == 12 && V_ == 25 && (Ci == 16 || Ci == 32); }

// partials: >= *rows_out (<= 768) rows of 2 (Ci Ci + Ci) floats; Zout: [B, Ci, T, V]
int launch_fwd_moments_bpc(const float* in, const float* Aw, const float* Tw, const float* in_slope, float* partials, int B, int Ci,
                           int T_, int V_, int need_x, float* Zout, hipStream_t st, int* rows_out) {
  if (!fwd_moments_bpc_ok(T_, V_, Ci) || !Zout) return fail(COSKAD_ERR_SHAPE, "fwd_moments_bpc: built for 12 x 25, 16 / 32 channels, stored Z");
  constexpr int V = 25;
  const size_t lds = (size_t)32 * (12 * V + 2) * sizeof(float);
  const int per_cu = Ci == 16 ? 3 : 2;                   // (32 channels: 80 B of scratch at three waves per SIMD)
  const int grid = B < 256 * per_cu ? B : 256 * per_cu;
  *rows_out = grid;
  if (Ci == 16)
    hipLaunchKernelGGL((fm::k_fwd_moments_bpc<V, 1>), dim3(grid), dim3(256), lds, st, in, Aw, Tw, in_slope, partials, B, need_x, Zout,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL((fm::k_fwd_moments_bpc<V, 2>), dim3(grid), dim3(256), lds, st, in, Aw, Tw, in_slope, partials, B, need_x, Zout,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  return check_launch("fwd_moments_bpc");
}

// the statistics pass of the layer BEHIND a commuted (32 -> 16) layer, with that layer's combine U = a_t Zy + a_r R + shift formed on the
// way in: YR [B, 32, TV] (rows 16 .. 31 = R), Zy [B, 16, TV], stat (a_t | a_r | shift) -> U [B, 16, TV], Zout = gcn(PReLU(U)), partials
int launch_combine_moments_bpc(const float* Zy, const float* YR, const float* stat, float* U, const float* Aw, const float* Tw,
                               const float* slope, float* partials, int B, int T_, int V_, float* Zout, hipStream_t st, int* rows_out) {
  if (!fwd_moments_bpc_ok(T_, V_, 16) || !Zout || !slope) return fail(COSKAD_ERR_SHAPE, "combine_moments_bpc: built for 12 x 25, 16 channels");
  constexpr int V = 25;
  const size_t lds = (size_t)32 * (12 * V + 2) * sizeof(float);
  const int grid = B < 768 ? B : 768;
  *rows_out = grid;
  hipLaunchKernelGGL((fm::k_fwd_moments_bpc<V, 1, true>), dim3(grid), dim3(256), lds, st, Zy, Aw, Tw, slope, partials, B, 1, Zout,
                     YR + 16 * 12 * V, stat, U);
  return check_launch("combine_moments_bpc");
}

}  // namespace coskad
