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
#include "fused_ops.h"

namespace coskad {
namespace ff {

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
