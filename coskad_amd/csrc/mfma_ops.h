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

#ifndef COSKAD_XB
#define COSKAD_XB 4
#endif
#ifndef COSKAD_RP
#define COSKAD_RP 1
#endif

namespace coskad {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// ---- separable mixing on the LDS row image, in place -----------------------------------
// rows: valid rows of the image (row tiles of 16; rows beyond `rows` read as 0, never written)
// TwL / AwL: LDS copies of T[V][T][T] and A[T][V][V].
// LDX: row stride override of the LDS image (0 = Geo's odd stride, which the strip-conv kernels need; kernels without
// a strip phase use TV + 2: conflict-free (row, k) operand reads, see RedGeo in stsgcn_bwd.hip)
template <int T, int V, bool ADJ, int LDX = 0>
__device__ __forceinline__ void temporal_mfma(float* img, int rows, const float* TwL, int tid = -1) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = LDX ? LDX : Geo<T, V>::LD;
  constexpr int KS = (T + 3) / 4;
  constexpr int RP = COSKAD_RP;   // row tiles per item: they share the B operand and give independent MFMA chains
  static_assert(T <= 16, "temporal_mfma: one 16-wide column tile");
  if (tid < 0) tid = threadIdx.x;   // callers short of registers pass tid_here()
  const int lane = tid & 63, wave = uniform(tid >> 6);
  const int i = lane & 15, k = lane >> 4;
  const int RT = (rows + 15) >> 4;
  const int RG = (RT + RP - 1) / RP;
  const int jc = i < T ? i : T - 1;  // column clamp (columns >= T are never stored)
  for (int it = wave; it < RG * V; it += kBlock / 64) {
    const int rg = it / V, v = it - rg * V;
    const float* tb = TwL + v * T * T;
    float b[KS], a[RP][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 4 * s + k;
      const bool kok = kk < T;
      const int kc = kok ? kk : T - 1;
      // forward: B[k = t][j = q] = T[v][t][q];  adjoint: B[k = q][j = t] = T[v][t][q]
      b[s] = ADJ ? tb[jc * T + kc] : tb[kc * T + jc];
#pragma unroll
      for (int q = 0; q < RP; ++q) {
        const int rowA = 16 * (rg * RP + q) + i;
        const bool ok = rowA < rows && kok;
        a[q][s] = ok ? img[rowA * LD + kc * V + v] : 0.f;
      }
    }
    f32x4 acc[RP];
#pragma unroll
    for (int q = 0; q < RP; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int q = 0; q < RP; ++q) acc[q] = mfma4(a[q][s], b[s], acc[q]);
    if (i < T) {
#pragma unroll
      for (int q = 0; q < RP; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * (rg * RP + q) + 4 * k + r;
          if (row < rows) img[row * LD + i * V + v] = acc[q][r];
        }
    }
  }
}

template <int T, int V, bool ADJ, int LDX = 0>
__device__ __forceinline__ void spatial_mfma(float* img, int rows, const float* AwL, int tid = -1) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int LD = LDX ? LDX : Geo<T, V>::LD;
  constexpr int KS = (V + 3) / 4;
  constexpr int RP = COSKAD_RP;   // row tiles per item (share B, independent chains)
  // column tiles on MFMA; up to 2 leftover columns (V = 17, 18) are cheaper on the VALU
  constexpr int VX = (V > 16 && V - 16 <= 2) ? V - 16 : ((V > 32 && V - 32 <= 2) ? V - 32 : 0);
  constexpr int NT = (V - VX + 15) / 16;
  if (tid < 0) tid = threadIdx.x;   // callers short of registers pass tid_here()
  const int lane = tid & 63, wave = uniform(tid >> 6);
  const int i = lane & 15, k = lane >> 4;
  const int RT = (rows + 15) >> 4;
  const int RG = (RT + RP - 1) / RP;
  for (int it = wave; it < RG * T; it += kBlock / 64) {
    const int rg = it / T, t = it - rg * T;
    const float* ab = AwL + t * V * V;
    float a[RP][KS];
#pragma unroll
    for (int q = 0; q < RP; ++q) {
      const int rowA = 16 * (rg * RP + q) + i;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int kk = 4 * s + k;
        a[q][s] = (rowA < rows && kk < V) ? img[rowA * LD + t * V + kk] : 0.f;
      }
    }
    f32x4 acc[RP][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = 16 * nt + i;
      const int jc = col < V ? col : V - 1;
      float b[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int kk = 4 * s + k;
        const int kc = kk < V ? kk : V - 1;
        // forward: B[k = v][j = w] = A[t][v][w];  adjoint: B[k = w][j = v] = A[t][v][w]
        b[s] = ADJ ? ab[jc * V + kc] : ab[kc * V + jc];
      }
#pragma unroll
      for (int q = 0; q < RP; ++q) acc[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < RP; ++q) acc[q][nt] = mfma4(a[q][s], b[s], acc[q][nt]);
    }
    float ex[RP][VX > 0 ? VX : 1];
    if constexpr (VX > 0) {
#pragma unroll
      for (int x = 0; x < VX; ++x) {
        const int col = V - VX + x;
        float bw[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int kk = 4 * s + k;
          const int kc = kk < V ? kk : V - 1;
          bw[s] = ADJ ? ab[col * V + kc] : ab[kc * V + col];
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
          float p = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s) p = fmaf(a[q][s], bw[s], p);   // a is 0 for kk >= V
          p += __shfl_xor(p, 16, 64);
          p += __shfl_xor(p, 32, 64);
          ex[q][x] = p;             // every lane: value for row 16*tile + i
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RP; ++q) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + i;
        if (col < V - VX) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * (rg * RP + q) + 4 * k + r;
            if (row < rows) img[row * LD + t * V + col] = acc[q][nt][r];
          }
        }
      }
      if constexpr (VX > 0) {
        const int rowA = 16 * (rg * RP + q) + i;
        if (k == 0 && rowA < rows) {
#pragma unroll
          for (int x = 0; x < VX; ++x) img[rowA * LD + t * V + V - VX + x] = ex[q][x];
        }
      }
    }
  }
}

template <int T, int V, bool ADJ, int LDX = 0>
__device__ __forceinline__ void gcn_mfma(float* img, int rows, const float* AwL, const float* TwL, int tid = -1) {
  if constexpr (!ADJ) {
    temporal_mfma<T, V, false, LDX>(img, rows, TwL, tid);
    lds_barrier();
    spatial_mfma<T, V, false, LDX>(img, rows, AwL, tid);
  } else {
    spatial_mfma<T, V, true, LDX>(img, rows, AwL, tid);
    lds_barrier();
    temporal_mfma<T, V, true, LDX>(img, rows, TwL, tid);
  }
}

}  // namespace coskad

// ---- 1x1 convolutions of one clip as an MFMA GEMM ---------------------------------------
//   Out[o][p] = bias[o] + sum_k Wl[k][o] * In_k[p]
// K-image of the clip: k in [0, KZ) = rows of the LDS image (KZ = Ci rounded up to 4, rows >= Ci have zero
// weights), then rows of up to two global tensors (stride TV, optional PReLU on load): see conv_mfma_s below.
namespace coskad {

// LDS weight table of the folded layer: Wl[k][o], k over the padded K-image.
// src: [(2*Ci)][CoP] (rows 0..Ci-1: Z part, Ci..2Ci-1: X part)
__device__ __forceinline__ void load_wfold_padded(float* Wl, const float* __restrict__ src, int Ci, int KZ,
                                                  int CoP, int nsrc) {
  for (int e = threadIdx.x; e < nsrc * KZ * CoP; e += blockDim.x) {
    const int k = e / CoP, o = e - k * CoP;
    const int part = k / KZ, c = k - part * KZ;
    Wl[e] = c < Ci ? src[(part * Ci + c) * CoP + o] : 0.f;
  }
}

}  // namespace coskad

// ---- strip version of the conv GEMM: full-line global traffic ----------------------------
// A work item is a STRIP of 32 consecutive positions, handled as two MFMA column tiles with the
// position map  tile m, column j  <->  p = p0 + 2*j + m.  Then
//   * a global B operand is ONE 8-byte load per lane (16 lanes = one contiguous 128-B line per row)
//     feeding both tiles,
//   * outputs leave as 8-byte stores (128-B lines),
//   * LDS operands are two stride-2 ds_read_b32 (conflict-free on the odd-LD image).
// K-image: [LDS rows KZ][global source 1 rows K1 (no activation)][global source 2 rows K2 (optional
// PReLU)]; KZ, K1, K2 are multiples of 4 (weights of padded rows are zero); Wl[k][CoP] in LDS.
namespace coskad {

template <int T, int V, int OTI, int XB = COSKAD_XB, class Epilogue>
__device__ __forceinline__ void conv_mfma_s(const float* zimg, int KZ, int nz, const float* __restrict__ g1,
                                            int K1, int n1, const float* __restrict__ g2, int K2, int n2,
                                            bool act2, float a2, const float* Wl, int CoP, int og, int s0,
                                            int sstep, Epilogue&& epi) {
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  static_assert(TV % 2 == 0, "strip conv needs an even number of positions");
  constexpr int PS = (TV + 31) / 32;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kk = lane >> 4;
  const int KZS = KZ / 4, KGS = (K1 + K2) / 4, K1S = K1 / 4;
  const float* wcol = Wl + 16 * og * OTI + j;

  for (int sp = s0; sp < PS; sp += sstep) {
    const int p = 32 * sp + 2 * j;
    const bool pok = p < TV;
    const int pc = pok ? p : TV - 2;
    auto gload = [&](int g) -> float2 {          // B operand of global k-step g for this lane
      if (g >= KGS) return float2{0.f, 0.f};
      const bool first = g < K1S;
      const int c = first ? 4 * g + kk : 4 * (g - K1S) + kk;
      const int nr = first ? n1 : n2;
      const float* base = first ? g1 : g2;
      const int cc = c < nr ? c : nr - 1;
      float2 v = *reinterpret_cast<const float2*>(base + (size_t)cc * TV + pc);
      if (!first && act2) { v.x = prelu_f(v.x, a2); v.y = prelu_f(v.y, a2); }
      return v;
    };
    f32x4 acc[OTI][2];
#pragma unroll
    for (int t = 0; t < OTI; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float2 cur[XB], nxt[XB];
#pragma unroll
    for (int u = 0; u < XB; ++u) cur[u] = gload(u);
    // LDS source while the first global batch is in flight
    for (int s = 0; s < KZS; s += 2) {
      float b[2][2], a[2][OTI];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = 4 * (s + u) + kk;
        const int cc = c < nz ? c : nz - 1;
        const bool ok = s + u < KZS;
        b[u][0] = ok ? zimg[cc * LD + pc] : 0.f;
        b[u][1] = ok ? zimg[cc * LD + pc + 1] : 0.f;
        const float* w = wcol + (ok ? c : 0) * CoP;
#pragma unroll
        for (int t = 0; t < OTI; ++t) a[u][t] = w[16 * t];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < OTI; ++t) {
          acc[t][0] = mfma4(a[u][t], b[u][0], acc[t][0]);
          acc[t][1] = mfma4(a[u][t], b[u][1], acc[t][1]);
        }
    }
    for (int g0 = 0; g0 < KGS; g0 += XB) {
#pragma unroll
      for (int u = 0; u < XB; ++u) nxt[u] = gload(g0 + XB + u);
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        if (g0 + u < KGS) {
          const float* w = wcol + (KZ + 4 * (g0 + u) + kk) * CoP;
#pragma unroll
          for (int t = 0; t < OTI; ++t) {
            const float a = w[16 * t];
            acc[t][0] = mfma4(a, cur[u].x, acc[t][0]);
            acc[t][1] = mfma4(a, cur[u].y, acc[t][1]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < XB; ++u) cur[u] = nxt[u];
    }
    // D[row = 4*kk + r][col = j] of tile m  <->  (o, p + m)
#pragma unroll
    for (int t = 0; t < OTI; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) epi(16 * (og * OTI + t) + 4 * kk + r, p, pok, acc[t][0][r], acc[t][1][r]);
  }
}

}  // namespace coskad
