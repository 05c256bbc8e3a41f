// Fused eval-mode STS-GCN encoder: the reference's whole `Encoder.forward` (models/common/components.py:94-105 = four
// ST_GCNN_layer.forward, models/graph_layers/stsgcn.py:94-116, BatchNorm folded from its running statistics) in ONE
// kernel, one clip per wavefront, every activation resident in LDS or registers.  HBM traffic per clip: the 1.6 KB clip
// in, the activated last layer out (52 KB, tile-major, consumed by the bottleneck kernel with a permuted weight).
//
// Design ("fused forward"):
//   * one wavefront owns one clip end to end: no workgroup barrier anywhere, only the wave's own LDS ordering;
//     four waves per CU (one per SIMD, up to 512 VGPRs each), 39.6 KB of LDS per wave: R1 = 32 rows x 206, R2 = 16 rows;
//   * everything is v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains): temporal mixing per joint, spatial mixing per
//     frame, 1x1 convs per 16-position tile;
//   * an accumulator tile D[channel][position] IS the B operand of the next 1x1 conv (register r = channel 4q + r on
//     k slot q): spatial mixing -> conv -> (PReLU) -> next conv chain without touching LDS; the conv weights live in
//     registers for the whole launch in that k order (coskad_amd/fused_plan.py builds the operand streams);
//   * layer 2 (32 -> 16) is commuted: gcn(Wz X) = Wz gcn(X), so the mixing runs on 16 channels;
//   * the 17th joint: spatial column 16 is a 5-FMA dot + two cross-lane adds, written back in place; its 12 positions
//     form a 13th position tile;
//   * layer 4 needs its input twice (mixing, residual conv) but LDS holds it once (mixed in place): the second copy
//     stays in 104 registers per lane (x4[13][2]), which is why the tile loops of layers 3/4 are fully unrolled;
//   * the mixing matrices stream from L2 as ready-made B operands (16-byte records per lane), prefetched one item ahead.
//
// tests/test_fused_plan.py replays exactly this schedule in numpy from the same operand streams.
#include "common.h"

namespace coskad {
namespace ff {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int T = 12, V = 17, TV = T * V;
constexpr int LD = 206;                 // row stride: = 2 (mod 4) -> (row, k) operand reads of the mixing phases are conflict-free
constexpr int R1 = 0, R2 = 32 * LD;     // float offsets inside the wave's LDS image
constexpr int PADCOL = TV;              // columns 204, 205 of every row are padding: masked lanes store there (no divergent branches)
constexpr int WAVE_LDS = 48 * LD;       // floats per wave
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
template <int ROWS>
__device__ __forceinline__ TOp temporal_read(const float* img, int rt, int v, const Lane& L) {
  const float* p = img + (16 * rt + L.j) * LD + L.q * V + v;
  TOp o{p[0], p[4 * V], p[8 * V]};
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

template <int ROWS, int NRT>
__device__ __forceinline__ void temporal_phase(float* img, const TTab& tt, const Lane& L) {
  constexpr int GV = NRT == 2 ? 2 : 4;          // joints per group: 4 independent chains in flight either way
  constexpr int NG = (V + GV - 1) / GV;
  // three stages in flight: operand reads of group g+1, MFMAs of group g, result writes of group g-1 -- a result is
  // written one group after its MFMA chain was issued, so the wave never idles on the MFMA -> LDS-store hazard
  TOp cur[GV][NRT], nxt[GV][NRT];
  f32x4 d[GV][NRT], dp[GV][NRT];
#pragma unroll
  for (int u = 0; u < GV; ++u)
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) cur[u][rt] = temporal_read<ROWS>(img, rt, u, L);
#pragma unroll
  for (int g = 0; g <= NG; ++g) {
    const int v0 = g * GV;
    if (g < NG) {
#pragma unroll
      for (int u = 0; u < GV; ++u)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (v0 + GV + u < V) nxt[u][rt] = temporal_read<ROWS>(img, rt, v0 + GV + u, L);
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

__global__ __launch_bounds__(256, 1) void k_fused_encoder(const float* __restrict__ x, float* __restrict__ out,
                                                         const float4* __restrict__ tab, const float* __restrict__ wreg,
                                                         const float* __restrict__ slopes, int B) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = lds_all + wave * WAVE_LDS;
  // Lane geometry behind an optimisation barrier, refreshed at the start of every phase: everything address-like is
  // then recomputed where it is used (a few VALU ops) instead of being hoisted out of the persistent clip loop and
  // held in ~150 registers across all phases.
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  Lane L = geo();
  float* r1 = lds + R1;
  float* r2 = lds + R2;

  // conv weights (A operands) and bias quads: registers for the whole launch
  // (rows WZ3.. and the layer-4 biases; the operands of layers 1-2 and of the residual conv of layer 3 are re-read from
  // L2 at the start of every clip, so that their 48 registers are free while layers 3-4 hold X4 in registers)
  const int l4 = lane * 4, l16 = lane * 16;
  const BufRes tabres = make_res(tab, 4u * LAYER_F4 * 16u);
  const BufRes wres = make_res(wreg, NWREG * 64u * 4u);
  float w[NWREG];
#pragma unroll
  for (int i = WZ3; i < B1; ++i) w[i] = buf_load1(wres, l4, i * 256);
#pragma unroll
  for (int i = B4; i < NWREG; ++i) w[i] = buf_load1(wres, l4, i * 256);
#define BQ(row) f32x4{w[(row)], w[(row) + 1], w[(row) + 2], w[(row) + 3]}
  const float s1 = slopes[0], s2 = slopes[1], s3 = slopes[2], s4 = slopes[3];

  const int nwaves = gridDim.x * 4;
  int clip = blockIdx.x * 4 + wave;
  // the clip's 408 floats: element e = lane + 64 i
  float xin[7];
  auto load_x = [&](int c) {      // (bounds-checked: elements >= 408 and clips >= B read 0)
    const BufRes xr = make_res(x + (size_t)(c < B ? c : 0) * (2 * TV), c < B ? 2 * TV * 4 : 0);
#pragma unroll
    for (int i = 0; i < 7; ++i) xin[i] = buf_load1(xr, l4, i * 256);
  };
  load_x(clip);
  TTab tt;                          // temporal table of the phase about to run (loaded while the previous chain computes)
  load_ttab(tt, tabres, 0, l16);

  // position of this lane's column in tile `tile` (frame tiles: joints 0..15 of frame `tile`; tile 12: joint 16 of frame j)
#define TILE_GEO(tile)                                                                   \
  const bool fr = (tile) < T;                                                           \
  const int pos = fr ? (tile) * V + L.j : (L.j < T ? L.j : T - 1) * V + 16;              \
  const bool ok = fr || L.j < T

  for (; clip < B; clip += nwaves) {
    // ---- stage: R2 rows 0,1 = mixing copy, rows 2,3 = the copy the residual conv of layer 1 reads
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int e = lane + 64 * i;
      const int c = e >= TV ? 1 : 0;
      const int p = e - c * TV;
      float* d = r2 + (e < 2 * TV ? c * LD + p : PADCOL);
      d[0] = xin[i];
      d[2 * LD] = xin[i];
    }
    load_x(clip + nwaves);      // next clip's input travels while this one is computed
#pragma unroll
    for (int i = 0; i < WZ3; ++i) w[i] = buf_load1(wres, l4, i * 256);
#pragma unroll
    for (int i = B1; i < B4; ++i) w[i] = buf_load1(wres, l4, i * 256);

    // Every chain below is a two-stage software pipeline written out by hand: iteration `tile` issues the operand reads
    // of frame tile+2, runs the spatial mixing of frame tile+1 and the convolutions of tile `tile` -- two independent
    // dependency chains per iteration, so that the in-order wave always has MFMAs to issue while the other chain waits
    // for its accumulators, LDS round trips or the VALU epilogue.
    // ================= layer 1 (2 -> 32) and the convs of layer 2 (32 -> 16, commuted) =================
    L = geo();
    if (!(FF_ABLATE & 1)) temporal_phase<2, 1>(r2, tt, L);
    load_ttab(tt, tabres, LAYER_F4, l16);              // layer 2's temporal table: needed after this chain
    L = geo();
    if (!(FF_ABLATE & 2)) {
      SpatRec rec1 = load_spat(tabres, 0, 0, l16);
      SOp op1 = spatial_read<2>(r2, 0, 0, L);
      f32x4 zc = spatial_mm(op1, rec1);
      spatial_extra<2>(r2, 0, 0, op1, rec1, L);
      rec1 = load_spat(tabres, 0, 1, l16);
      op1 = spatial_read<2>(r2, 0, 1, L);
      f32x4 Pp = {0.f, 0.f, 0.f, 0.f}, Rp = Pp;      // results of the previous tile: stored one tile later (see temporal_phase)
      int posp = PADCOL;
      bool okp = false;
      FF_AB_UNROLL
      for (int tile = 0; tile < NTILE; ++tile) {
        TILE_GEO(tile);
        const int t2 = tile + 2 < T ? tile + 2 : T - 1;
        const SpatRec rec2 = load_spat(tabres, 0, t2, l16);
        const SOp op2 = spatial_read<2>(r2, 0, t2, L);            // frame tile+2's operands before this tile's stores
        f32x4 zn;
        if (tile + 1 < T) {
          zn = spatial_mm(op1, rec1);
          spatial_extra<2>(r2, 0, tile + 1, op1, rec1, L);
        } else {                                                    // next = the 17th-joint tile: joint 16 of frame j
          const int pe = (L.j < T ? L.j : T - 1) * V + 16;
          zn = f32x4{r2[pe], r2[LD + pe], 0.f, 0.f};
        }
        const float xc = r2[(L.q == 1 ? 2 : 3) * LD + pos];
        const float bA = L.q == 0 ? zc[0] : ((L.q == 1 || L.q == 2) ? xc : 0.f);
        const float bB = L.q == 0 ? zc[1] : 0.f;
        f32x4 u0 = mfma(w[W1A], bA, BQ(B1));
        f32x4 u1 = mfma(w[W1A + 1], bA, BQ(B1 + 4));
        u0 = mfma(w[W1B], bB, u0);
        u1 = mfma(w[W1B + 1], bB, u1);
        const f32x4 x20 = prelu4(u0, s1), x21 = prelu4(u1, s1);
        f32x4 P = {0.f, 0.f, 0.f, 0.f};
        f32x4 Rr = BQ(B2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P = mfma(w[WP + r], x20[r], P);
          Rr = mfma(w[WR + r], x20[r], Rr);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          P = mfma(w[WP + 4 + r], x21[r], P);
          Rr = mfma(w[WR + 4 + r], x21[r], Rr);
        }
        if (tile > 0) {
          tile_store(r1, 0, posp, okp, Pp, L);
          tile_store(r1, 16, posp, okp, Rp, L);
        }
        Pp = P; Rp = Rr; posp = pos; okp = ok;
        zc = zn; rec1 = rec2; op1 = op2;
      }
      tile_store(r1, 0, posp, okp, Pp, L);
      tile_store(r1, 16, posp, okp, Rp, L);
    }

    // ================= layer 2 mixing on P; U2 = gcn(P) + R; X3 -> R2; residual conv of layer 3 -> R1 (in place) ======
    L = geo();
    if (!(FF_ABLATE & 1)) temporal_phase<16, 1>(r1, tt, L);
    load_ttab(tt, tabres, 2 * LAYER_F4, l16);
    L = geo();
    if (!(FF_ABLATE & 4)) {
      constexpr int tb = LAYER_F4;
      SpatRec rec1 = load_spat(tabres, tb, 0, l16);
      SOp op1 = spatial_read<16>(r1, 0, 0, L);
      f32x4 zc = spatial_mm(op1, rec1);
      spatial_extra<16>(r1, 0, 0, op1, rec1, L);
      rec1 = load_spat(tabres, tb, 1, l16);
      op1 = spatial_read<16>(r1, 0, 1, L);
      f32x4 a0p = {0.f, 0.f, 0.f, 0.f}, a1p = a0p;
      int posp = PADCOL;
      bool okp = false;
      FF_AB_UNROLL
      for (int tile = 0; tile < NTILE; ++tile) {
        TILE_GEO(tile);
        const int t2 = tile + 2 < T ? tile + 2 : T - 1;
        const SpatRec rec2 = load_spat(tabres, tb, t2, l16);
        const SOp op2 = spatial_read<16>(r1, 0, t2, L);
        const f32x4 rr = tile_load(r1, 16, pos, L);
        f32x4 zn;
        if (tile + 1 < T) {
          zn = spatial_mm(op1, rec1);
          spatial_extra<16>(r1, 0, tile + 1, op1, rec1, L);
        } else {
          zn = tile_load(r1, 0, (L.j < T ? L.j : T - 1) * V + 16, L);
        }
        const f32x4 x3 = prelu4(zc + rr, s2);
        tile_store(r2, 0, pos, ok, x3, L);
        f32x4 a0 = BQ(B3), a1 = BQ(B3 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0 = mfma(w[WX3 + r], x3[r], a0);
          a1 = mfma(w[WX3 + 4 + r], x3[r], a1);
        }
        if (tile > 0) {
          tile_store(r1, 0, posp, okp, a0p, L);
          tile_store(r1, 16, posp, okp, a1p, L);
        }
        a0p = a0; a1p = a1; posp = pos; okp = ok;
        zc = zn; rec1 = rec2; op1 = op2;
      }
      tile_store(r1, 0, posp, okp, a0p, L);
      tile_store(r1, 16, posp, okp, a1p, L);
    }

    // ================= layer 3 mixing on X3; conv3 on top of the stored residual part; X4 -> R1 and registers =========
    L = geo();
    if (!(FF_ABLATE & 1)) temporal_phase<16, 1>(r2, tt, L);
    f32x4 x4[NTILE][2];
    L = geo();
    if (FF_ABLATE & 8) {
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) x4[tile][0] = x4[tile][1] = f32x4{xin[0], xin[1], xin[2], xin[3]};
    } else {
      constexpr int tb = 2 * LAYER_F4;
      SpatRec rec1 = load_spat(tabres, tb, 0, l16);
      SOp op1 = spatial_read<16>(r2, 0, 0, L);
      f32x4 zc = spatial_mm(op1, rec1);
      spatial_extra<16>(r2, 0, 0, op1, rec1, L);
      rec1 = load_spat(tabres, tb, 1, l16);
      op1 = spatial_read<16>(r2, 0, 1, L);
      f32x4 a0p = {0.f, 0.f, 0.f, 0.f}, a1p = a0p;
      int posp = PADCOL;
      bool okp = false;
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        TILE_GEO(tile);
        const int t2 = tile + 2 < T ? tile + 2 : T - 1;
        const SpatRec rec2 = (FF_CABL & 1) ? rec1 : load_spat(tabres, tb, t2, l16);
        const SOp op2 = spatial_read<16>(r2, 0, t2, L);
        f32x4 a0 = (FF_CABL & 2) ? BQ(B3) : tile_load(r1, 0, pos, L), a1 = (FF_CABL & 2) ? BQ(B3 + 4) : tile_load(r1, 16, pos, L);
        f32x4 zn;
        if (tile + 1 < T) {
          zn = spatial_mm(op1, rec1);
          if (!(FF_CABL & 8)) spatial_extra<16>(r2, 0, tile + 1, op1, rec1, L);
        } else {
          zn = tile_load(r2, 0, (L.j < T ? L.j : T - 1) * V + 16, L);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0 = mfma(w[WZ3 + r], zc[r], a0);
          a1 = mfma(w[WZ3 + 4 + r], zc[r], a1);
        }
        if (tile > 0) {                                   // epilogue of the previous tile, behind this tile's MFMAs
          const f32x4 e0 = prelu4(a0p, s3), e1 = prelu4(a1p, s3);
          tile_store(r1, 0, posp, okp, e0, L);
          tile_store(r1, 16, posp, okp, e1, L);
          x4[tile > 0 ? tile - 1 : 0][0] = e0;
          x4[tile > 0 ? tile - 1 : 0][1] = e1;
        }
        a0p = a0; a1p = a1; posp = pos; okp = ok;
        zc = zn; rec1 = rec2; op1 = op2;
        FF_TILE_FENCE;
      }
      {
        const f32x4 e0 = prelu4(a0p, s3), e1 = prelu4(a1p, s3);
        tile_store(r1, 0, posp, okp, e0, L);
        tile_store(r1, 16, posp, okp, e1, L);
        x4[NTILE - 1][0] = e0;
        x4[NTILE - 1][1] = e1;
      }
    }

    // ================= layer 4 (32 -> 64): mixing in place, conv from the mixing accumulators + the X4 registers ======
    load_ttab(tt, tabres, 3 * LAYER_F4, l16);
    L = geo();
    if (!(FF_ABLATE & 1)) temporal_phase<16, 2>(r1, tt, L);
    L = geo();
    if (!(FF_ABLATE & 16)) {
      constexpr int tb = 3 * LAYER_F4;
      const BufRes ores = make_res(out + (size_t)clip * KP, KP * 4);
      SpatRec rec1 = load_spat(tabres, tb, 0, l16);
      SOp op10 = spatial_read<16>(r1, 0, 0, L), op11 = spatial_read<16>(r1, 1, 0, L);
      f32x4 zc0 = spatial_mm(op10, rec1), zc1 = spatial_mm(op11, rec1);
      spatial_extra<16>(r1, 0, 0, op10, rec1, L);
      spatial_extra<16>(r1, 1, 0, op11, rec1, L);
      rec1 = load_spat(tabres, tb, 1, l16);
      op10 = spatial_read<16>(r1, 0, 1, L);
      op11 = spatial_read<16>(r1, 1, 1, L);
      f32x4 ap[4];
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) ap[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tile = 0; tile < NTILE; ++tile) {
        TILE_GEO(tile);
        const int t2 = tile + 2 < T ? tile + 2 : T - 1;
        const SpatRec rec2 = load_spat(tabres, tb, t2, l16);
        const SOp op20 = spatial_read<16>(r1, 0, t2, L), op21 = spatial_read<16>(r1, 1, t2, L);
        f32x4 zn0, zn1;
        if (tile + 1 < T) {
          zn0 = spatial_mm(op10, rec1);
          zn1 = spatial_mm(op11, rec1);
          spatial_extra<16>(r1, 0, tile + 1, op10, rec1, L);
          spatial_extra<16>(r1, 1, tile + 1, op11, rec1, L);
        } else {
          const int pe = (L.j < T ? L.j : T - 1) * V + 16;
          zn0 = tile_load(r1, 0, pe, L);
          zn1 = tile_load(r1, 16, pe, L);
        }
        f32x4 a[4];
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) a[ot] = BQ(B4 + 4 * ot);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) a[ot] = mfma(w[WX4 + 8 * ot + r], x4[tile][0][r], a[ot]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) a[ot] = mfma(w[WX4 + 8 * ot + 4 + r], x4[tile][1][r], a[ot]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) a[ot] = mfma(w[WZ4 + 8 * ot + r], zc0[r], a[ot]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) a[ot] = mfma(w[WZ4 + 8 * ot + 4 + r], zc1[r], a[ot]);
        if (tile > 0) {                                   // epilogue of the previous tile, behind this tile's MFMAs
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) {
            const f32x4 v = prelu4(ap[ot], s4);
            buf_store4(ores, l16, ((tile > 0 ? tile - 1 : 0) * 4 + ot) * 1024, float4{v[0], v[1], v[2], v[3]});   // frame tiles: every column valid
          }
        }
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) ap[ot] = a[ot];
        zc0 = zn0; zc1 = zn1; rec1 = rec2; op10 = op20; op11 = op21;
        FF_TILE_FENCE;
      }
      {
        const bool ok = L.j < T;                          // the 17th-joint tile: columns 12..15 are padding
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
          const f32x4 v = prelu4(ap[ot], s4);
          buf_store4(ores, l16, ((NTILE - 1) * 4 + ot) * 1024, ok ? float4{v[0], v[1], v[2], v[3]} : float4{0.f, 0.f, 0.f, 0.f});
        }
      }
    }
    load_ttab(tt, tabres, 0, l16);   // layer 1's table for the next clip
  }
#undef BQ
#undef TILE_GEO
}

// out[i] = idx[i] >= 0 ? src[idx[i]] : 0   (builds the operand streams from the concatenated parameters)
__global__ void k_gather(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int s = idx[i];
    out[i] = s >= 0 ? src[s] : 0.f;
  }
}

}  // namespace ff
}  // namespace coskad

using namespace coskad;

extern "C" {

int coskad_gather_f32(const float* src, const int* idx, float* out, size_t n, hipStream_t stream) {
  if (!src || !idx || !out) return fail(COSKAD_ERR_ARG, "gather: null pointer");
  if (n == 0) return COSKAD_OK;
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(ff::k_gather, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, src, idx, out, n);
  return check_launch("gather");
}

int coskad_fused_encoder_out_floats(void) { return ff::KP; }

int coskad_fused_encoder_f32(const float* x, float* out, const float* tab, const float* wreg, const float* slopes, int B,
                             int T, int V, hipStream_t stream) {
  if (!x || !out || !tab || !wreg || !slopes) return fail(COSKAD_ERR_ARG, "fused_encoder: null pointer");
  if (B <= 0) return fail(COSKAD_ERR_ARG, "fused_encoder: B=%d", B);
  if (T != ff::T || V != ff::V)
    return fail(COSKAD_ERR_SHAPE, "fused_encoder: built for n_frames=12, n_joints=17 and channels 2-32-16-32-64 (got T=%d V=%d)", T, V);
  const size_t lds = (size_t)4 * ff::WAVE_LDS * sizeof(float);
  auto k = ff::k_fused_encoder;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nblk = (B + 3) / 4;
  const int grid = nblk < 256 ? nblk : 256;          // one 4-wave block per CU, persistent over clips
  {
    ProbeScope probe(KID_FUSED_FWD, 2, 64, stream);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, stream, x, out, reinterpret_cast<const float4*>(tab), wreg, slopes, B);
  }
  return check_launch("fused_encoder");
}

}  // extern "C"
