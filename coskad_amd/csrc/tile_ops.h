// Device building blocks shared by the STS-GCN tile kernels (gfx950).
//
// A *tile* is NB consecutive clips x C channels of one activation tensor
// [N, C, T, V] (contiguous, fp32).  In HBM a tile is one contiguous run of
// NB*C*T*V floats, so staging is a pure streaming copy.  In LDS a tile is
// `rows = NB*C` rows of LD = T*V (+1 if even) floats: row = (clip, channel),
// column = position p = t*V + v.
//
// Two thread mappings alternate on the same LDS image:
//   * row phase   (the learned space/time mixing, reference stsgcn.py:154-155):
//     lane <-> row, wave <-> a wave-uniform slice of joints (temporal mix) or
//     frames (spatial mix).  The mixing weights T[v,:,:] / A[t,:,:] are then
//     wave-uniform: they stream through SGPRs (s_load) and every FMA is
//     v_fma(vgpr, sgpr, vgpr) -- no LDS or VGPR traffic for weights.
//   * position phase (1x1 convolutions / epilogues): lane <-> position,
//     channel loop inside the thread, conv weights wave-uniform in SGPRs.
#pragma once
#include "common.h"

namespace coskad {

// Stream `nfloats` contiguous floats (whole rows of TV) from HBM into the LDS row image,
// optionally applying PReLU on the way (the producer layer stores pre-activations).
// LDX: LDS row stride override (0 = the odd Geo stride; the batch-reduction kernels use TV + 2, see RedGeo)
template <int T, int V, int LDX = 0>
__device__ __forceinline__ void stage_rows(const float* __restrict__ g, float* lds, int nfloats,
                                           bool do_prelu, float slope, int tid = -1) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  if (tid < 0) tid = threadIdx.x;
  constexpr int TV = Geo<T, V>::TV, LD = LDX ? LDX : Geo<T, V>::LD;
  if constexpr (TV % 4 == 0) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int n4 = nfloats >> 2;
    constexpr int UB = 4;   // HBM loads in flight per thread before the first LDS write
    for (int i0 = tid; i0 < n4; i0 += UB * kBlock) {
      float4 v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = i0 + u * kBlock;
        v[u] = i < n4 ? g4[i] : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = i0 + u * kBlock;
        if (i < n4) {
          const int e = i << 2;
          const int row = e / TV;
          const int col = e - row * TV;
          if (do_prelu) {
            v[u].x = prelu_f(v[u].x, slope); v[u].y = prelu_f(v[u].y, slope);
            v[u].z = prelu_f(v[u].z, slope); v[u].w = prelu_f(v[u].w, slope);
          }
          float* d = lds + row * LD + col;
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    }
  } else {
    for (int e = threadIdx.x; e < nfloats; e += kBlock) {
      float v = g[e];
      const int row = e / TV;
      const int col = e - row * TV;
      if (do_prelu) v = prelu_f(v, slope);
      lds[row * LD + col] = v;
    }
  }
}

// Write the LDS row image back to a contiguous HBM tile.
template <int T, int V, int LDX = 0>
__device__ __forceinline__ void unstage_rows(float* __restrict__ g, const float* lds, int nfloats, int tid = -1) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  if (tid < 0) tid = threadIdx.x;
  constexpr int TV = Geo<T, V>::TV, LD = LDX ? LDX : Geo<T, V>::LD;
  if constexpr (TV % 4 == 0) {
    float4* g4 = reinterpret_cast<float4*>(g);
    const int n4 = nfloats >> 2;
#pragma unroll 2
    for (int i = tid; i < n4; i += kBlock) {
      const int e = i << 2;
      const int row = e / TV;
      const int col = e - row * TV;
      const float* s = lds + row * LD + col;
      g4[i] = float4{s[0], s[1], s[2], s[3]};
    }
  } else {
    for (int e = threadIdx.x; e < nfloats; e += kBlock) {
      const int row = e / TV;
      const int col = e - row * TV;
      g[e] = lds[row * LD + col];
    }
  }
}

// ---- row phase ------------------------------------------------------------------------
// Temporal mix of one joint column v of one row, in place.
//   forward : y[q] = sum_t x[t] * Tw[v][t][q]        (einsum 'nctv,vtq->ncqv', stsgcn.py:154)
//   adjoint : x'[t] = sum_q y'[q] * Tw[v][t][q]
template <int T, int V, bool ADJ>
__device__ __forceinline__ void temporal_col(float* r, int v, const float* __restrict__ Tw) {
  float x[T];
#pragma unroll
  for (int t = 0; t < T; ++t) x[t] = r[t * V + v];
  const float* w = Tw + v * T * T;  // wave-uniform -> SGPR stream
  float y[T];
  if constexpr (!ADJ) {
#pragma unroll
    for (int q = 0; q < T; ++q) y[q] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int q = 0; q < T; ++q) y[q] = fmaf(x[t], w[t * T + q], y[q]);
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < T; ++q) s = fmaf(x[q], w[t * T + q], s);
      y[t] = s;
    }
  }
#pragma unroll
  for (int q = 0; q < T; ++q) r[q * V + v] = y[q];
}

// Spatial mix of one frame t of one row, in place.
//   forward : z[w] = sum_v y[v] * Aw[t][v][w]        (einsum 'nctv,tvw->nctw', stsgcn.py:155)
//   adjoint : y'[v] = sum_w z'[w] * Aw[t][v][w]
template <int T, int V, bool ADJ>
__device__ __forceinline__ void spatial_row(float* r, int t, const float* __restrict__ Aw) {
  float y[V];
#pragma unroll
  for (int v = 0; v < V; ++v) y[v] = r[t * V + v];
  const float* w = Aw + t * V * V;  // wave-uniform -> SGPR stream
  float z[V];
  if constexpr (!ADJ) {
#pragma unroll
    for (int j = 0; j < V; ++j) z[j] = 0.f;
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int j = 0; j < V; ++j) z[j] = fmaf(y[v], w[v * V + j], z[j]);
  } else {
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) s = fmaf(y[j], w[v * V + j], s);
      z[v] = s;
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) r[t * V + j] = z[j];
}

// One half of the separable mixing over all rows of the LDS image, in place.
//   TEMPORAL=true : every row's joint columns;  TEMPORAL=false : every row's frames.
// Work split: lane <-> row (64 rows per batch), wave <-> slice of v (or t).
template <int T, int V, bool TEMPORAL, bool ADJ>
__device__ __forceinline__ void mix_rows(float* lds, int rows, const float* __restrict__ W) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = Geo<T, V>::LD;
  constexpr int NPART = kBlock / 64;
  constexpr int EXT = TEMPORAL ? V : T;
  const int lane = threadIdx.x & 63;
  const int part = uniform(threadIdx.x >> 6);
  const int beg = (EXT * part) / NPART, end = (EXT * (part + 1)) / NPART;
  for (int rb = 0; rb < rows; rb += 64) {
    const int row = rb + lane;
    if (row < rows) {
      float* r = lds + row * LD;
      for (int i = beg; i < end; ++i) {
        if constexpr (TEMPORAL) temporal_col<T, V, ADJ>(r, i, W);
        else spatial_row<T, V, ADJ>(r, i, W);
      }
    }
  }
}

// Full ConvTemporalGraphical on the LDS image, in place (forward: temporal then spatial;
// adjoint: spatial^T then temporal^T).  Contains the barrier between the two halves;
// caller provides the barriers before and after.
template <int T, int V, bool ADJ>
__device__ __forceinline__ void gcn_rows(float* lds, int rows, const float* __restrict__ Aw,
                                         const float* __restrict__ Tw) {
  if constexpr (!ADJ) {
    mix_rows<T, V, true, false>(lds, rows, Tw);
    lds_barrier();
    mix_rows<T, V, false, false>(lds, rows, Aw);
  } else {
    mix_rows<T, V, false, true>(lds, rows, Aw);
    lds_barrier();
    mix_rows<T, V, true, true>(lds, rows, Tw);
  }
}

}  // namespace coskad

// ---- MFMA phase -----------------------------------------------------------------------
// Batch reductions that are GEMMs with K = positions (BatchNorm second moments, conv weight
// gradients): Out[a][b] += sum_p Arow[a][p] * Brow[b][p].  v_mfma_f32_16x16x4_f32 is an exact
// fp32 FMA chain (same numerics as VALU) at the VALU FLOP rate, on the otherwise idle matrix
// pipe, and does the cross-lane reduction over K for free.
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: B[k = l>>4][j = l&15]
//   C/D      : lane l, reg r  <->  D[row = 4*(l>>4) + r][col = l&15]
namespace coskad {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// One clip's worth of positions.  The 4 waves of the block split the K (position) steps;
// each wave keeps its own accumulators for ALL (ta, tb) tile pairs.
//   ldsA/ldsB : first row of the clip in the A / B row image (LD stride)
//   va / vb   : number of valid rows (rows >= valid read as 0)
//   SUMS      : also accumulate row sums of A (B operand == 1) into sacc
template <int T, int V, int NTA, int NTB, bool SUMS>
__device__ __forceinline__ void outer_accum(const float* ldsA, int va, const float* ldsB, int vb,
                                            f32x4 (&acc)[NTA][NTB], f32x4 (&sacc)[NTA]) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  static_assert(TV % 4 == 0, "positions must be a multiple of the MFMA K step");
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  for (int p0 = 4 * wave; p0 < TV; p0 += 4 * (kBlock / 64)) {
    float a[NTA], b[NTB];
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
      const int row = 16 * ta + i;
      a[ta] = row < va ? ldsA[row * LD + p0 + k] : 0.f;
    }
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      const int row = 16 * tb + i;
      b[tb] = row < vb ? ldsB[row * LD + p0 + k] : 0.f;
    }
#pragma unroll
    for (int ta = 0; ta < NTA; ++ta) {
#pragma unroll
      for (int tb = 0; tb < NTB; ++tb)
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
      if constexpr (SUMS)
        sacc[ta] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], 1.0f, sacc[ta], 0, 0, 0);
    }
  }
}

// Second moments of ONE row image: M[a][b] += sum_p X[a][p] X[b][p], rs[a] += sum_p X[a][p].
// Symmetric: only tiles tb >= ta are computed, and the A fragment of tile t IS the B fragment of tile t
// (lane (i,k) holds X[16t+i][p0+k] either way), so one LDS read per tile and k-step feeds everything.
// Row sums ride along on the VALU (one add per tile and k-step) instead of an extra MFMA against ones.
template <int T, int V, int NT, int LDX = 0>
__device__ __forceinline__ void moment_accum(const float* img, int valid, f32x4 (&acc)[NT][NT], float (&rs)[NT]) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = Geo<T, V>::TV, LD = LDX ? LDX : Geo<T, V>::LD;
  static_assert(TV % 4 == 0, "positions must be a multiple of the MFMA K step");
  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int i = lane & 15, k = lane >> 4;
  for (int p0 = 4 * wave; p0 < TV; p0 += 4 * (kBlock / 64)) {
    float a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = 16 * t + i;
      a[t] = row < valid ? img[row * LD + p0 + k] : 0.f;
      rs[t] += a[t];
    }
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
      for (int tb = ta; tb < NT; ++tb)
        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], a[tb], acc[ta][tb], 0, 0, 0);
  }
}

// Store of moment_accum results: tiles tb >= ta are mirrored into the full [valid x valid] matrix at dst
// (row stride ldd); row sums (per lane: row 16t + (lane&15), partial over its k group) go to sums[].
template <int NT>
__device__ __forceinline__ void store_moments(const f32x4 (&acc)[NT][NT], const float (&rs)[NT], float* scratch,
                                              float* dst, int ldd, float* sums, int valid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NW = blockDim.x >> 6;
#pragma unroll
  for (int ta = 0; ta < NT; ++ta)
#pragma unroll
    for (int tb = ta; tb < NT; ++tb) {
      lds_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[ta][tb][r];
      lds_barrier();
      const int e = threadIdx.x;
      if (e < 256) {
        const int row = e >> 4, col = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += scratch[w * 256 + e];
        const int ra = 16 * ta + row, cb = 16 * tb + col;
        if (ra < valid && cb < valid) {
          dst[ra * ldd + cb] = s;
          if (ta != tb) dst[cb * ldd + ra] = s;
        }
      }
    }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v = rs[t];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    lds_barrier();
    if (lane < 16) scratch[wave * 16 + lane] = v;
    lds_barrier();
    const int e = threadIdx.x;
    if (e < 16 && 16 * t + e < valid) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += scratch[w * 16 + e];
      sums[16 * t + e] = s;
    }
  }
}

// Cross-wave reduction of MFMA accumulators and store of this block's partial.
//   dst[(16*ta + row) * ldd + 16*tb + col]  (only row < va, col < vb are written)
// scratch: LDS, >= kScratchFloats.  Contains barriers; all threads must call.
template <int NTA, int NTB>
__device__ __forceinline__ void store_outer(const f32x4 (&acc)[NTA][NTB], float* scratch, float* dst,
                                            int ldd, int va, int vb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ta = 0; ta < NTA; ++ta)
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
      lds_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[wave * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[ta][tb][r];
      lds_barrier();
      const int e = threadIdx.x;  // first 256 threads <-> 16x16 tile elements
      if (e < 256) {
        const int row = e >> 4, col = e & 15;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w * 256 + e];
        if (16 * ta + row < va && 16 * tb + col < vb) dst[(16 * ta + row) * ldd + 16 * tb + col] = s;
      }
    }
}

// Same for the row-sum accumulators (all 16 columns of a sum tile are identical: take col 0).
template <int NTA>
__device__ __forceinline__ void store_sums(const f32x4 (&sacc)[NTA], float* scratch, float* dst, int va) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ta = 0; ta < NTA; ++ta) {
    lds_barrier();
    if ((lane & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[wave * 16 + 4 * (lane >> 4) + r] = sacc[ta][r];
    }
    lds_barrier();
    const int e = threadIdx.x;
    if (e < 16 && 16 * ta + e < va) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w * 16 + e];
      dst[16 * ta + e] = s;
    }
  }
}

// Row sums accumulated per lane on the VALU (lane (i,k): row 16t+i, partial over its k group and k-steps):
// reduce over the k groups, then over the waves, store sums[16t + i] for rows < valid.
template <int NT>
__device__ __forceinline__ void store_rowsums(const float (&rs)[NT], float* scratch, float* sums, int valid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v = rs[t];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    lds_barrier();
    if (lane < 16) scratch[wave * 16 + lane] = v;
    lds_barrier();
    const int e = threadIdx.x;
    if (e < 16 && 16 * t + e < valid) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += scratch[w * 16 + e];
      sums[16 * t + e] = s;
    }
  }
}

template <int N, int M>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[N][M]) {
#pragma unroll
  for (int a = 0; a < N; ++a)
#pragma unroll
    for (int b = 0; b < M; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
}
template <int N>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[N]) {
#pragma unroll
  for (int a = 0; a < N; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace coskad
