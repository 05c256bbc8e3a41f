// Device building blocks shared by the wave-per-clip kernels (fused_fwd.hip: eval-mode encoder; fused_bwd.hip: the
// layer backward): lane geometry, buffer-addressed global memory, the hand-pipelined temporal / spatial mixing phases on
// an LDS row image of stride 206, accumulator-layout tiles, cross-lane sums.
#pragma once
#include "common.h"

namespace coskad {
namespace ff {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int T = 12, V = 17, TV = T * V;
#ifndef FF_LD
#define FF_LD 206
#endif
constexpr int LD = FF_LD;               // row stride: = 2 (mod 4) -> (row, k) operand reads of the mixing phases are conflict-free
constexpr int R1 = 0, R2 = 32 * LD;     // float offsets inside the wave's LDS image
constexpr int PADCOL = TV;              // columns 204, 205 of every row are padding: masked lanes store there (no divergent branches)
constexpr int WAVE_LDS = 48 * LD;       // floats per wave
// The K-ring kernels (fused_bwd.hip, fused_apply.hip) give their 16-row window its own stride: 208 = 16 (mod 64) makes the
// window's only read pattern -- lane (j, q) reads row 4 s + q, position base + j -- conflict-free, and rows 16-byte aligned,
// so a staged float4 is ONE ds_write_b128 instead of two half-bank ds_write_b64 (the LDS pipe, shared by the CU's four
// waves, was ~85 % busy in the K passes with the 206 stride)
constexpr int LDW = 208;
constexpr int WAVE_LDS_W = 32 * LD + 16 * LDW;   // floats per wave with that window (9 920: 4 x 39 680 B <= 160 KB)
constexpr int NTILE = T + 1;
#ifndef FF_AB_UNROLL
#define FF_AB_UNROLL _Pragma("unroll")     // tile loops of layers 1-2: fully unrolled like layers 3-4 (rolled: 213 vs 199 us at B = 4096 --
                                           // the register rotation of the hand-written pipeline costs ~20 v_mov per iteration)
#endif
#ifndef FF_ABLATE
#define FF_ABLATE 0   // timing-only builds (tools/ab_fused.sh): bit 0 skips the temporal phases, bits 1..4 the chains of layers 1..4
#endif
#ifndef FF_CABL
#define FF_CABL 0     // timing-only: pieces of the layer-3 chain (1 table loads, 2 accumulator loads, 4 stores, 8 joint 16, 16 PReLU)
#endif
#ifndef FF_TILE_FENCE
#define FF_TILE_FENCE   // (A/B hook: -DFF_TILE_FENCE="__builtin_amdgcn_sched_barrier(0)" keeps the scheduler inside one tile)
#endif
constexpr int KP = NTILE * 4 * 64 * 4;  // tile-major output floats per clip (13 312)
constexpr int TEMP_F4 = V * 64;         // float4 records of the temporal part of one layer
constexpr int LAYER_F4 = TEMP_F4 + T * 3 * 64;
enum { W1A = 0, W1B = 2, WP = 4, WR = 12, WX3 = 20, WZ3 = 28, WZ4 = 36, WX4 = 68, B1 = 100, B2 = 108, B3 = 112, B4 = 120, NWREG = 136 };

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float prelu(float x, float a) { return x > 0.f ? x : a * x; }
__device__ __forceinline__ f32x4 prelu4(f32x4 v, float a) {
  return f32x4{prelu(v[0], a), prelu(v[1], a), prelu(v[2], a), prelu(v[3], a)};
}

struct Lane {
  int j, q;
};

// Buffer-addressed global memory (a 128-bit descriptor in SGPRs + ONE 32-bit lane offset + a wave-uniform SGPR/immediate
// offset): with flat addressing hipcc precomputes a 64-bit VGPR address pair per 4 KB window of every stream and holds
// ~70 of them across the clip loop.  Out-of-range lanes read 0 / do not store (hardware bounds check).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using BufRes = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRes make_res(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(BufRes r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return float4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ float buf_load1(BufRes r, int voff, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4(BufRes r, int voff, int soff, const float4& v) {
  const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, soff, 0);
}

// ---- temporal mixing (stsgcn.py:154), in place -----------------------------------------------------------------------
// One item = one joint column v of one row tile: D[row][q] = sum_t X[row][t,v] T[v][t][q].  The phase is software-
// pipelined by hand: the operands of group g+1 are read BEFORE the results of group g are written (the compiler cannot
// prove that those LDS accesses never alias, so program order is what it executes), which keeps several independent
// MFMA chains and LDS round trips in flight from one wave.
struct TOp {
  float a0, a1, a2;
};
template <int ROWS, bool ACT = false>
__device__ __forceinline__ TOp temporal_read(const float* img, int rt, int v, const Lane& L, float slope = 0.f) {
  const float* p = img + (16 * rt + L.j) * LD + L.q * V + v;
  TOp o{p[0], p[4 * V], p[8 * V]};
  if (ACT) { o.a0 = prelu(o.a0, slope); o.a1 = prelu(o.a1, slope); o.a2 = prelu(o.a2, slope); }
  if (ROWS < 16) {
    const bool ok = L.j < ROWS;
    o.a0 = ok ? o.a0 : 0.f; o.a1 = ok ? o.a1 : 0.f; o.a2 = ok ? o.a2 : 0.f;
  }
  return o;
}
__device__ __forceinline__ f32x4 temporal_mm(const TOp& o, const float4& rec) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma(o.a0, rec.x, acc);
  acc = mfma(o.a1, rec.y, acc);
  acc = mfma(o.a2, rec.z, acc);
  return acc;
}
template <int ROWS>
__device__ __forceinline__ void temporal_store(float* img, int rt, int v, const f32x4& acc, const Lane& L) {
  float* rowp = img + (16 * rt + 4 * L.q) * LD;
  float* p;
  if (ROWS >= 16) {
    p = rowp + (L.j < T ? L.j * V + v : PADCOL);      // masked lanes (columns 12..15 of the tile): the rows' padding column
    p[0] = acc[0]; p[LD] = acc[1]; p[2 * LD] = acc[2]; p[3 * LD] = acc[3];
  } else {
    p = rowp + ((L.j < T && L.q == 0) ? L.j * V + v : PADCOL);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < ROWS) p[r * LD] = acc[r];
  }
}

// the table of one layer's temporal mixing as B operands: 17 records per lane
struct TTab {
  float4 r[V];
};
// tabres: the whole tab stream; `base4`: float4 index of the layer's first record; l16 = lane * 16 bytes
__device__ __forceinline__ void load_ttab(TTab& t, BufRes tabres, int base4, int l16) {
#pragma unroll
  for (int v = 0; v < V; ++v) t.r[v] = buf_load4(tabres, l16, (base4 + v * 64) * 16);
}

template <int ROWS, int NRT, bool ACT = false>   // ACT: the image holds pre-activations, PReLU(slope) is applied to the operands as they are read
__device__ __forceinline__ void temporal_phase(float* img, const TTab& tt, const Lane& L, float slope = 0.f) {
  constexpr int GV = NRT == 2 ? 2 : 4;          // joints per group: 4 independent chains in flight either way
  constexpr int NG = (V + GV - 1) / GV;
  // three stages in flight: operand reads of group g+1, MFMAs of group g, result writes of group g-1 -- a result is
  // written one group after its MFMA chain was issued, so the wave never idles on the MFMA -> LDS-store hazard
  TOp cur[GV][NRT], nxt[GV][NRT];
  f32x4 d[GV][NRT], dp[GV][NRT];
#pragma unroll
  for (int u = 0; u < GV; ++u)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) cur[u][rt] = temporal_read<ROWS, ACT>(img, rt, u, L, slope);
#pragma unroll
  for (int g = 0; g <= NG; ++g) {
    const int v0 = g * GV;
    if (g < NG) {
#pragma unroll
      for (int u = 0; u < GV; ++u)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (v0 + GV + u < V) nxt[u][rt] = temporal_read<ROWS, ACT>(img, rt, v0 + GV + u, L, slope);
#pragma unroll
      for (int u = 0; u < GV; ++u)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (v0 + u < V) d[u][rt] = temporal_mm(cur[u][rt], tt.r[v0 + u < V ? v0 + u : V - 1]);
    }
    if (g > 0) {
#pragma unroll
      for (int u = 0; u < GV; ++u)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (v0 - GV + u < V) temporal_store<ROWS>(img, rt, v0 - GV + u, dp[u][rt], L);
    }
#pragma unroll
    for (int u = 0; u < GV; ++u)
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        cur[u][rt] = nxt[u][rt];
        dp[u][rt] = d[u][rt];
      }
  }
}

// ---- spatial mixing of one frame (stsgcn.py:155) -> accumulator tile; joint 16 goes back to the image in place -------
struct SpatRec {
  float4 c0, c1, c2;   // b[0..3] | b[4], bw[0..2] | bw[3], bw[4], -, -
};
__device__ __forceinline__ SpatRec load_spat(BufRes tabres, int base4, int t, int l16) {
  const int o = (base4 + TEMP_F4 + t * 3 * 64) * 16;
  return SpatRec{buf_load4(tabres, l16, o), buf_load4(tabres, l16, o + 1024), buf_load4(tabres, l16, o + 2048)};
}
struct SOp {
  float a0, a1, a2, a3, a4;
};
template <int ROWS>
__device__ __forceinline__ SOp spatial_read(const float* img, int rt, int t, const Lane& L) {
  const float* row = img + (16 * rt + L.j) * LD + t * V;
  const float* p = row + L.q;
  SOp o{p[0], p[4], p[8], p[12], row[16]};
  o.a4 = L.q == 0 ? o.a4 : 0.f;                        // joint 16: k slot 0 of the fifth step only
  if (ROWS < 16) {
    const bool ok = L.j < ROWS;
    o.a0 = ok ? o.a0 : 0.f; o.a1 = ok ? o.a1 : 0.f; o.a2 = ok ? o.a2 : 0.f; o.a3 = ok ? o.a3 : 0.f; o.a4 = ok ? o.a4 : 0.f;
  }
  return o;
}
__device__ __forceinline__ f32x4 spatial_mm(const SOp& o, const SpatRec& R) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma(o.a0, R.c0.x, acc);
  acc = mfma(o.a1, R.c0.y, acc);
  acc = mfma(o.a2, R.c0.z, acc);
  acc = mfma(o.a3, R.c0.w, acc);
  acc = mfma(o.a4, R.c1.x, acc);
  return acc;
}
// sum over the four k slots (lanes l, l^16, l^32, l^48) without leaving the VALU: v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ float quad_sum(float x) {
  unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  unsigned u2 = __float_as_uint(s);
  auto b = __builtin_amdgcn_permlane32_swap(u2, u2, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// joint 16 of the frame: Z[row][t,16] = sum_v Y[row][t,v] A[t][v][16], written over Y[row][t,16] (dead by now)
template <int ROWS>
__device__ __forceinline__ void spatial_extra(float* img, int rt, int t, const SOp& o, const SpatRec& R, const Lane& L) {
  float ex = o.a0 * R.c1.y;
  ex = fmaf(o.a1, R.c1.z, ex);
  ex = fmaf(o.a2, R.c1.w, ex);
  ex = fmaf(o.a3, R.c2.x, ex);
  ex = fmaf(o.a4, R.c2.y, ex);
  ex = quad_sum(ex);
  float* e = img + (16 * rt + L.j) * LD + ((L.q == 0 && (ROWS >= 16 || L.j < ROWS)) ? t * V + 16 : PADCOL);
  *e = ex;
}

// ---- spatial mixing of a whole image, in place, software-pipelined by hand ------------------------------------------------
// One wave issues in order: an MFMA chain that is followed in program order by the stores of its own result stalls the wave
// for the chain's whole latency, and the VALU / LDS work of the frame (joint 16, operand reads, tile stores) then runs with
// the matrix pipe idle.  Here the frame loop is a three-stage pipeline -- MFMA chains of frame t (the NRT row tiles
// interleaved step by step), joint 16 of frame t on the VALU, operand reads of frame t+1, stores of frame t-1 -- and
// sched_group_barriers lay the stages out between the MFMAs (cdna_hip_programming.md T19).  Used with the forward tables
// (fused_apply_next.hip) and the adjoint ones (fused_bwd.hip): the same code, `base4` selects the table.
template <int NRT>
__device__ __forceinline__ void spatial_phase(float* img, BufRes tabres, int base4, int l16, const Lane& L) {
  constexpr int FP = NRT == 1 ? 2 : 1;      // frames per pipeline step: at least two independent MFMA chains in flight
  constexpr int NC = NRT * FP;              // chains per step; chain c = frame f * NRT + row tile rt
  constexpr int NS = T / FP;
  static_assert(T % FP == 0, "frames per step divide the window");
  SpatRec rec[FP];
  SOp op[NC];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    rec[f] = load_spat(tabres, base4, f, l16);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) op[f * NRT + rt] = spatial_read<16>(img, rt, f, L);
  }
  f32x4 dprev[NC];
  float exprev[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { dprev[c] = f32x4{0.f, 0.f, 0.f, 0.f}; exprev[c] = 0.f; }
#pragma unroll
  for (int st = 0; st <= NS; ++st) {
    f32x4 d[NC];
    float ex[NC];
    SOp opn[NC];
    SpatRec nxt[FP];
    const bool live = st < NS, more = st + 1 < NS;
    if (live) {
#pragma unroll
      for (int f = 0; f < FP; ++f) nxt[f] = more ? load_spat(tabres, base4, (st + 1) * FP + f, l16) : rec[f];
      // stage 1: the MFMA chains of this step's frames, interleaved step by step (a chain's next MFMA is NC MFMAs away)
#pragma unroll
      for (int c = 0; c < NC; ++c) d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const SpatRec& R = rec[c / NRT];
          const float bk = s == 0 ? R.c0.x : s == 1 ? R.c0.y : s == 2 ? R.c0.z : s == 3 ? R.c0.w : R.c1.x;
          const float a = s == 0 ? op[c].a0 : s == 1 ? op[c].a1 : s == 2 ? op[c].a2 : s == 3 ? op[c].a3 : op[c].a4;
          d[c] = mfma(a, bk, d[c]);
        }
      // stage 2: joint 16 of these frames (VALU)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const SpatRec& R = rec[c / NRT];
        float e = op[c].a0 * R.c1.y;
        e = fmaf(op[c].a1, R.c1.z, e);
        e = fmaf(op[c].a2, R.c1.w, e);
        e = fmaf(op[c].a3, R.c2.x, e);
        e = fmaf(op[c].a4, R.c2.y, e);
        ex[c] = quad_sum(e);
      }
      // stage 3: operands of the next step's frames
      if (more) {
#pragma unroll
        for (int c = 0; c < NC; ++c) opn[c] = spatial_read<16>(img, c % NRT, (st + 1) * FP + c / NRT, L);
      }
    }
    // stage 4: results of the previous step (their chains were issued a step ago)
    if (st > 0) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int t = (st - 1) * FP + c / NRT, rt = c % NRT;
        float* e = img + (16 * rt + L.j) * LD + (L.q == 0 ? t * V + 16 : PADCOL);
        *e = exprev[c];
        tile_store(img, 16 * rt, t * V + L.j, true, dprev[c], L);
      }
    }
    if (live) {
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, NC, 0);          // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2 + 2 * NC, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x100, NC, 0);          // DS read
        __builtin_amdgcn_sched_group_barrier(0x200, NC, 0);          // DS write
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        dprev[c] = d[c];
        exprev[c] = ex[c];
        if (more) op[c] = opn[c];
      }
#pragma unroll
      for (int f = 0; f < FP; ++f) rec[f] = nxt[f];
    }
  }
}

// ---- accumulator-layout tiles in LDS -------------------------------------------------------------------------------
// lane (j, q), register r  <->  row row0 + 4q + r, position `pos` (lane-dependent; `ok` masks the padding columns)
__device__ __forceinline__ void tile_store(float* img, int row0, int pos, bool ok, const f32x4& a, const Lane& L) {
  float* p = img + (row0 + 4 * L.q) * LD + (ok ? pos : PADCOL);
  p[0] = a[0]; p[LD] = a[1]; p[2 * LD] = a[2]; p[3 * LD] = a[3];
}
__device__ __forceinline__ f32x4 tile_load(const float* img, int row0, int pos, const Lane& L) {
  const float* p = img + (row0 + 4 * L.q) * LD + pos;
  return f32x4{p[0], p[LD], p[2 * LD], p[3 * LD]};
}

// ---- operand tables of the fused layer backward (fused_bwd.hip) from a layer's A [T,V,V] and T [V,T,T] --------------------
// (lane l: j = l & 15, q = l >> 4):
//   forward temporal  rec[v][l][s]      = T[v][4s+q][j]          (j < 12)
//   adjoint spatial   rec[t][l][0..4]   = A[t][j][4s+q]          (4s+q < 17),  [5..9] = A[t][16][4s+q]
//   adjoint temporal  rec[v][l][s]      = T[v][j][4s+q]          (j < 12)
constexpr int SPAT_F4 = T * 3 * 64;                  // float4 records of one spatial section
constexpr int BTAB_F4 = 2 * TEMP_F4 + SPAT_F4;       // [forward temporal][adjoint spatial][adjoint temporal]
__device__ __forceinline__ float btab_value(const float* __restrict__ Aw, const float* __restrict__ Tw, int e) {
  float val = 0.f;
  if (e < TEMP_F4 * 4 || e >= (TEMP_F4 + SPAT_F4) * 4) {
    const bool adj = e >= TEMP_F4 * 4;
    const int r = adj ? e - (TEMP_F4 + SPAT_F4) * 4 : e;
    const int v = r / 256, l = (r >> 2) & 63, s = r & 3, j = l & 15, q = l >> 4;
    if (s < 3 && j < T) val = adj ? Tw[v * T * T + j * T + 4 * s + q] : Tw[v * T * T + (4 * s + q) * T + j];
  } else {
    const int r = e - TEMP_F4 * 4;
    const int t = r / (3 * 256), c = (r / 256) % 3, l = (r >> 2) & 63, k = 4 * c + (r & 3), j = l & 15, q = l >> 4;
    if (k < 10) {
      const int s = k < 5 ? k : k - 5, w = 4 * s + q;
      if (w < V) val = Aw[t * V * V + (k < 5 ? j : 16) * V + w];
    }
  }
  return val;
}

}  // namespace ff
}  // namespace coskad
