// MFMA (v_mfma_f32_16x16x4_f32) versions of the tile phases.
//
// The f32 MFMA is an exact fp32 FMA chain at the VALU FLOP rate, but it needs ~1 instruction
// per 1024 FMAs, no scalar-load / waitcnt chatter, and runs on the matrix pipe.  Everything in
// the ST-GCN block that is a small GEMM goes there:
//
//   temporal mix  (stsgcn.py:154)  per joint v : Y_v[rows x T]  = X_v[rows x T] . T_v[T x T]
//   spatial  mix  (stsgcn.py:155)  per frame t : Z_t[rows x V]  = Y_t[rows x V] . A_t[V x V]
//   1x1 convs                       per position tile : Out[Co x 16] = W[Co x K] . In[K x 16]
//
// Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l&15][k = l>>4]   B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, reg r  <->  D[row = 4*(l>>4) + r][col = l&15]
//
// The mixing matrices (A: T*V*V, T: V*T*T floats = 23.6 KB for 12x17) and the conv weights live
// in LDS tables, loaded once per persistent block.
#pragma once
#include "tile_ops.h"

namespace coskad {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += kBlock) dst[i] = src[i];
}

// ---- separable mixing on the LDS row image, in place -----------------------------------
// rows: valid rows of the image (row tiles of 16; rows beyond `rows` read as 0, never written)
// TwL / AwL: LDS copies of T[V][T][T] and A[T][V][V].
template <int T, int V, bool ADJ>
__device__ __forceinline__ void temporal_mfma(float* img, int rows, const float* TwL) {
  constexpr int LD = Geo<T, V>::LD;
  constexpr int KS = (T + 3) / 4;
  static_assert(T <= 16, "temporal_mfma: one 16-wide column tile");
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  const int RT = (rows + 15) >> 4;
  const int jc = i < T ? i : T - 1;  // column clamp (columns >= T are never stored)
  for (int it = wave; it < RT * V; it += kBlock / 64) {
    const int rt = it / V, v = it - rt * V;
    const int rowA = 16 * rt + i;
    const bool rokA = rowA < rows;
    const float* ra = img + (rokA ? rowA : 0) * LD + v;
    const float* tb = TwL + v * T * T;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 4 * s + k;
      const bool kok = kk < T;
      const int kc = kok ? kk : T - 1;
      const float a = (rokA && kok) ? ra[kc * V] : 0.f;
      // forward: B[k = t][j = q] = T[v][t][q];  adjoint: B[k = q][j = t] = T[v][t][q]
      const float b = ADJ ? tb[jc * T + kc] : tb[kc * T + jc];
      acc = mfma4(a, b, acc);
    }
    if (i < T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + 4 * k + r;
        if (row < rows) img[row * LD + i * V + v] = acc[r];
      }
    }
  }
}

template <int T, int V, bool ADJ>
__device__ __forceinline__ void spatial_mfma(float* img, int rows, const float* AwL) {
  constexpr int LD = Geo<T, V>::LD;
  constexpr int KS = (V + 3) / 4;
  // column tiles on MFMA; up to 2 leftover columns (V = 17, 18) are cheaper on the VALU
  constexpr int VX = (V > 16 && V - 16 <= 2) ? V - 16 : ((V > 32 && V - 32 <= 2) ? V - 32 : 0);
  constexpr int NT = (V - VX + 15) / 16;
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  const int RT = (rows + 15) >> 4;
  for (int it = wave; it < RT * T; it += kBlock / 64) {
    const int rt = it / T, t = it - rt * T;
    const int rowA = 16 * rt + i;
    const bool rokA = rowA < rows;
    float* ra = img + (rokA ? rowA : 0) * LD + t * V;
    const float* ab = AwL + t * V * V;
    float a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 4 * s + k;
      a[s] = (rokA && kk < V) ? ra[kk] : 0.f;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int col = 16 * nt + i;
      const int jc = col < V ? col : V - 1;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int kk = 4 * s + k;
        const int kc = kk < V ? kk : V - 1;
        // forward: B[k = v][j = w] = A[t][v][w];  adjoint: B[k = w][j = v] = A[t][v][w]
        const float b = ADJ ? ab[jc * V + kc] : ab[kc * V + jc];
        acc[nt] = mfma4(a[s], b, acc[nt]);
      }
    }
    float ex[VX > 0 ? VX : 1];
    if constexpr (VX > 0) {
#pragma unroll
      for (int x = 0; x < VX; ++x) {
        const int col = V - VX + x;
        float p = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int kk = 4 * s + k;
          const int kc = kk < V ? kk : V - 1;
          const float b = ADJ ? ab[col * V + kc] : ab[kc * V + col];
          p = fmaf(a[s], b, p);   // a[s] is 0 for kk >= V
        }
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        ex[x] = p;                // every lane: value for row 16*rt + i
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = 16 * nt + i;
      if (col < V - VX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rt + 4 * k + r;
          if (row < rows) img[row * LD + t * V + col] = acc[nt][r];
        }
      }
    }
    if constexpr (VX > 0) {
      if (k == 0 && rokA) {
#pragma unroll
        for (int x = 0; x < VX; ++x) ra[V - VX + x] = ex[x];
      }
    }
  }
}

template <int T, int V, bool ADJ>
__device__ __forceinline__ void gcn_mfma(float* img, int rows, const float* AwL, const float* TwL) {
  if constexpr (!ADJ) {
    temporal_mfma<T, V, false>(img, rows, TwL);
    __syncthreads();
    spatial_mfma<T, V, false>(img, rows, AwL);
  } else {
    spatial_mfma<T, V, true>(img, rows, AwL);
    __syncthreads();
    temporal_mfma<T, V, true>(img, rows, TwL);
  }
}

}  // namespace coskad
