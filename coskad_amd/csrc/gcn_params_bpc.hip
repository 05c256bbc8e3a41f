// Stage 4 of the layer backward on the 25-joint layout: the gradients of the mixing parameters (autograd of
// models/graph_layers/stsgcn.py:154-155)
//     dA[t][v][w]   = sum_rows Y[row][t,v] dZ[row][t,w]      Y = temporal mix of X = PReLU(U_prev)
//     dT[v][t1][t2] = sum_rows X[row][t1,v] dY[row][t2,v]    dY = spatial adjoint of dZ
// over rows = (clip, channel), 32 consecutive rows of the [B C_in, T V] matrices per tile (both products sum over rows and the
// mixing is per row, so a tile is any run of rows), ONE TILE PER WORKGROUP OF FOUR WAVES.  k_bwd_gcn_params (stsgcn_bwd.hip)
// keeps the two 32-row images AND both mixing tables (44 KB at 25 joints) in LDS: one 16-wave block per CU, every phase behind
// a block-wide barrier, nothing to run while a tile's rows arrive (1.6 TB/s, 180 us per call at 32 channels).  Here the work is
// dealt by joint (temporal mix, dT) and by frame (dA, spatial adjoint), so a wave's B operands of both mixes are the same for
// every tile and live in 63 registers instead of LDS; the images take 77 KB (two workgroups per CU: one stages while the other
// multiplies), and every wave owns its sums from the first tile to the last: no combine, each wave writes its own part of the
// workgroup's partial row ([dA | dT], summed by k_reduce_gcn in a fixed order).  168 us per call.
#include "fused_ops.h"

namespace coskad {
namespace gp {

using ff::f32x4;
using ff::Lane;
using ff::mfma;
using ff::prelu;

template <int V>
__global__ __launch_bounds__(256, 2) void k_gcn_params_bpc(const float* __restrict__ in, const float* __restrict__ dZ,
                                                          const float* __restrict__ Aw, const float* __restrict__ Tw,
                                                          const float* __restrict__ in_slope, float* __restrict__ partials,
                                                          int rows_total) {
  constexpr int T = 12, TV = T * V, LD = TV + 2, R4 = TV / 4;
  static_assert(TV % 4 == 0, "rows are staged as float4");
  constexpr int NR = 32;                                 // rows per tile
  constexpr int N4 = NR * R4, XL = (N4 + 255) / 256;
  constexpr int NTV = (V + 15) / 16;                     // joint column tiles
  constexpr int KV = (V + 3) / 4;                        // k-steps over the joints
  constexpr int MAXF = T / 4, MAXJ = (V + 3) / 4;        // frames / joints per wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img1 = lds;                 // X -> Y -> X again
  float* img2 = lds + NR * LD;       // dZ -> dY
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Lane L{lane & 15, lane >> 4};
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  f32x4 accA[MAXF][NTV][NTV], accT[MAXJ];
#pragma unroll
  for (int a = 0; a < MAXF; ++a)
#pragma unroll
    for (int b = 0; b < NTV; ++b)
#pragma unroll
      for (int c = 0; c < NTV; ++c) accA[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) accT[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (rows_total + NR - 1) / NR;
  // a tile's rows as the threads own them (rows beyond the matrix: zeros -- they add nothing to either sum)
  auto tload = [&](const float* base, int tile, float4 (&r)[XL]) {
    const int rows = rows_total - tile * NR < NR ? rows_total - tile * NR : NR;
    const float4* g4 = reinterpret_cast<const float4*>(base + (size_t)tile * NR * TV);
    const int n4 = rows * R4;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + 256 * i;
      r[i] = e < n4 ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto tstore = [&](const float4 (&r)[XL], float* img, bool act) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + 256 * i;
      if (e < N4) {
        const int row = e / R4, col = 4 * (e - row * R4);
        float4 v = r[i];
        if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
        *reinterpret_cast<float2*>(img + row * LD + col) = float2{v.x, v.y};
        *reinterpret_cast<float2*>(img + row * LD + col + 2) = float2{v.z, v.w};
      }
    }
  };
  // a wave's joints and frames are the same for every tile: its B operands of both mixes (T: 3 floats per lane and joint, A: 14
  // per lane and frame) stay in registers for the launch.  (Measured against this: the next tile's rows prefetched into 80 more
  // registers with the operands fetched per item -- 308 B of scratch, +85 us per step; this form: -37 us per step.)
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
        bbv[tt][c][s] = (16 * c + L.j < V && 4 * s + L.q < V) ? Aw[(t * V + 16 * c + L.j) * V + 4 * s + L.q] : 0.f;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                     // the previous tile's readers are done
    {
      float4 px[XL];
      tload(in, tile, px);
      tstore(px, img1, pre);
      tload(dZ, tile, px);
      tstore(px, img2, false);
    }
    __syncthreads();
    // ---- Y = temporal mix of X, in place: joints v = wave, wave + 4, ..  (Y[q,v] = sum_t X[t,v] T[v][t][q]) -------------------
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
#pragma unroll
        for (int rt = 0; rt < NR / 16; ++rt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) d = mfma(img1[(16 * rt + L.j) * LD + (4 * s + L.q) * V + v], tbv[k][s], d);
          if (L.j < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) img1[(16 * rt + 4 * L.q + r) * LD + L.j * V + v] = d[r];
          }
        }
      }
    }
    __syncthreads();                                     // the first image holds Y
    // ---- dA[t] += Y_t^T dZ_t: frames t = wave, wave + 4, wave + 8; K = the tile's rows -------------------------------------------
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int s = 0; s < NR / 4; ++s) {
        const int row = 4 * s + L.q;
        float a[NTV], b[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          const bool ok = 16 * c + L.j < V;
          a[c] = ok ? img1[row * LD + t * V + 16 * c + L.j] : 0.f;
          b[c] = ok ? img2[row * LD + t * V + 16 * c + L.j] : 0.f;
        }
#pragma unroll
        for (int ta = 0; ta < NTV; ++ta)
#pragma unroll
          for (int tb2 = 0; tb2 < NTV; ++tb2) accA[tt][ta][tb2] = mfma(a[ta], b[tb2], accA[tt][ta][tb2]);
      }
    }
    __syncthreads();                                     // every wave has read Y and dZ
    // ---- dY = spatial adjoint of dZ, in place: dY[t,v] = sum_w dZ[t,w] A[t][v][w]; frames as above -------------------------------
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int rt = 0; rt < NR / 16; ++rt) {
        float a[KV];
#pragma unroll
        for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? img2[(16 * rt + L.j) * LD + t * V + 4 * s + L.q] : 0.f;
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
            for (int r = 0; r < 4; ++r) img2[(16 * rt + 4 * L.q + r) * LD + t * V + 16 * c + L.j] = d[c][r];
          }
      }
    }
    {                                                    // X again, from L2 (Y's readers left before the barrier above)
      float4 px[XL];
      tload(in, tile, px);
      tstore(px, img1, pre);
    }
    __syncthreads();                                     // the images hold X and dY
    // ---- dT[v] += X_v^T dY_v: joints v = wave, wave + 4, .. ------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
#pragma unroll
        for (int s = 0; s < NR / 4; ++s) {
          const int row = 4 * s + L.q;
          const float a = L.j < T ? img1[row * LD + L.j * V + v] : 0.f;
          const float b = L.j < T ? img2[row * LD + L.j * V + v] : 0.f;
          accT[k] = mfma(a, b, accT[k]);
        }
      }
    }
  }
  // ---- every wave owns its frames of dA and its joints of dT: its part of the workgroup's partial row [dA | dT] ------------------
  float* dstA = partials + (size_t)blockIdx.x * (T * V * V + V * T * T);
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
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) {
    const int v = wave + 4 * k;
    if (v < V) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t1 = 4 * L.q + r, t2 = L.j;
        if (t1 < T && t2 < T) dstT[(v * T + t1) * T + t2] = accT[k][r];
      }
    }
  }
}

}  // namespace gp

bool gcn_params_bpc_ok(int T_, int V_) { return T_ == 12 && V_ == 25; }

// partials: >= *rows_out (<= 512) rows of T V V + V T T floats
int launch_gcn_params_bpc(const float* in, const float* in_slope, const float* dz, const float* Aw, const float* Tw, float* partials,
                          int rows_total, int T_, int V_, hipStream_t st, int* rows_out) {
  if (!gcn_params_bpc_ok(T_, V_)) return fail(COSKAD_ERR_SHAPE, "gcn_params_bpc: built for 12 x 25 (%d x %d)", T_, V_);
  constexpr int V = 25;
  const size_t lds = (size_t)2 * 32 * (12 * V + 2) * sizeof(float);
  const int ntiles = (rows_total + 31) / 32;
  const int grid = ntiles < 512 ? ntiles : 512;
  *rows_out = grid;
  auto k = gp::k_gcn_params_bpc<V>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, dz, Aw, Tw, in_slope, partials, rows_total);
  return check_launch("gcn_params_bpc");
}

}  // namespace coskad
