// Bottleneck backward + the batch reductions of the encoder's TOP layer in one pass (round 4).
//
// The bottleneck's backward (reference models/sts/ae.py:97-101,157 under autograd) produces dU = (dz W) * PReLU'(U) for the
// last ST_GCNN layer; that layer's BatchNorm backward (models/graph_layers/stsgcn.py:94-116 in training mode) first needs
//     P[o][c] = sum_{n,p} dU[n][o][p] Z[n][c][p]     Q[o][c] = sum_{n,p} dU[n][o][p] PReLU(U_prev)[n][c][p]     s[o] = sum dU[n][o][p]
// over the whole batch.  Rounds 1-3 ran them as a pass of their own (k_bwd_stats_bpc: dU read back, 120 us / 436 MB at
// B = 4096) because k_btlnk_bwd tiles K = (channel, position) channel-major and never holds a clip's 64 channels of one
// position together.  This kernel tiles K POSITION-major: a workgroup owns 16 positions x all 64 channels and walks its chunk
// of clips 16 at a time, so the dU tile it forms meets Z and PReLU(U_prev) of the same (clips, positions) once, on chip.
//
//   * 512 threads = 8 waves, two phases per 16-clip tile with an LDS-only barrier behind each.
//   * Phase 1, wave (wc, h) owns channels 16 wc + 8 h .. +7 at all 16 positions: dx[clip][position] of ONE channel is 4 x
//     v_mfma_f32_16x16x4_f32 (K = latent; the W fragments of the wave's 8 channels live in 32 registers for the launch); lanes
//     (q, c) hold clip 4 q + r at position c, so U arrives and dU leaves as 64-byte runs per clip and channel (a first version
//     with channels across the lanes moved 16 bytes per 128-byte line and lane: 190 of its 347 us).  The masked, PReLU'd
//     accumulator registers are the B operand of dW[j][col] += dz^T PReLU(U) as they stand (the MFMA's K axis is the CLIP axis),
//     and go to an LDS image G[clip][channel][position].
//   * Phase 2, wave (wc, h) owns channels 16 wc .. +15 at positions 8 h .. +7: the contraction of P / Q runs over (clip,
//     position), so K is the clip axis again: A = G[clip 4 q + r][channel c][p], B = Z[clip 4 q + r][c][p] resp. the layer
//     input's, both (row, position) ds_read_b32 of images whose position runs are 17 floats apart and whose clip stride is
//     = 4 (mod 8) floats: conflict-free.
//   * Z / U_prev tiles (2 C_in rows x 16 positions x 16 clips) are fetched as 64-byte runs one tile ahead, through registers
//     (at the top of phase 2); the next tile's U rows at the top of phase 1, into a second register set.  Fetching everything at
//     one point measured 258 (top of phase 2) / 278 us (top of phase 1) against 242: the chip's workgroups run in lockstep, so
//     what matters is that the bursts are spread over the tile.
//   * Measured (B = 4096, 12 x 17, rocprofv3): 242 us alone against 122 + 133 us for k_btlnk_bwd + k_bwd_stats_bpc alone; in the
//     train step 1.356 vs 1.374 ms.  MfmaUtil 31.5 % (its 76 us of fp32 MFMA issue are the floor beside 642 MB of HBM traffic),
//     30 % of the wave cycles in s_waitcnt / barriers, 54 % waiting to issue; L2 fetch = the algorithmic bytes with the XCD remap.
//   * Everything a workgroup sums (dW columns, P, Q, s, the slope gradient) stays in accumulators for the launch; one partial
//     row / dW slab per workgroup, summed in fp64 in a fixed order by the reduce launches (deterministic, no atomics).
#include "fused_ops.h"

namespace coskad {
namespace bc {

using ff::f32x4; using ff::mfma; using ff::BufRes; using ff::make_res; using ff::buf_load4; using ff::buf_load1;
__device__ __forceinline__ void buf_store1(BufRes r, int voff, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

constexpr int HID = 64;          // channels of the encoder's last layer (four channel waves)
constexpr int PT = 16;           // positions per workgroup
constexpr int NCL = 16;          // clips per tile (MFMA rows)
constexpr int RSTR = 17;         // floats between the position runs of consecutive rows (odd: lanes <-> rows spread over all banks)
constexpr int kThreads = 512;
#ifndef BC_XCD
#define BC_XCD 1    // (A/B hook: 0 = tiles dealt round-robin over the XCDs)
#endif
#ifndef BC_PF
#define BC_PF 1     // U rows of the next tile fetched at the top of phase 1 (second register set) instead of the top of phase 2
#endif
#ifndef BC_SKIP
#define BC_SKIP 0   // timing-only builds (tools/ab_fused.sh, AB_SRC=btlnk_chain): 1 no U loads, 2 no dU stores, 4 no Z / input loads,
                    // 8 no LDS staging of them, 16 no P / Q products, 32 no dW products, 64 no G image
#endif

template <int CT>
__global__ __launch_bounds__(kThreads, 1) void k_btlnk_bwd_stats(
    const float* __restrict__ U, const float* __restrict__ W, const float* __restrict__ dz, const float* __restrict__ slope,
    float* __restrict__ dU, float* __restrict__ dWp, float* __restrict__ dap, const float* __restrict__ xin,
    const float* __restrict__ Zg, const float* __restrict__ in_slope, float* __restrict__ prow, int B, int TV, int L, int chunk) {
  constexpr int Ci = 16 * CT, NR = 2 * Ci;              // B-side rows: Z's channels, then the layer input's
  constexpr int GCS = HID * RSTR + 4, ZCS = NR * RSTR + 4;   // clip strides (both = 4 mod 8)
  constexpr int NL = NR * NCL * (PT / 4) / kThreads;    // float4 per thread and tile (4 / 8)
  constexpr int E = 2 * HID * Ci + HID;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* G = lds;
  float* ZX = lds + NCL * GCS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, h = wave >> 2;
  const int c = lane & 15, q = lane >> 4;
  const int K = HID * TV;
  // Workgroup -> (chunk, position tile).  A tile's 64-byte runs share their 128-byte lines with the neighbouring position tiles
  // (rows are 4 T V bytes, never line-aligned), so the tiles of one chunk must sit behind ONE L2: blocks b and b + 8 share an XCD
  // (observed round-robin placement: speed only), hence the bijective remap of cdna_hip_programming.md T1 -- each XCD takes a
  // contiguous run of the chunk-major (chunk, tile) order.  Without it every line crosses the fabric twice (315 vs 254 us at B = 4096).
  const int nwg = gridDim.x, npt = (TV + PT - 1) / PT;
  int wg = blockIdx.x;
  if (BC_XCD) {
    const int xq = nwg / 8, xr = nwg % 8, xcd = wg % 8;
    wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + wg / 8;
  }
  const int ychunk = wg / npt;
  const int p0 = (wg - ychunk * npt) * PT;              // first position of the workgroup's tile
  const int nbeg = ychunk * chunk, nend = min(B, nbeg + chunk);
  const bool pre = slope != nullptr;
  const float a = pre ? slope[0] : 0.f;
  const bool xact = in_slope != nullptr;
  const float a_in = xact ? in_slope[0] : 0.f;
  const int ch0 = 16 * wc + 8 * h;                      // phase 1: this wave's first channel
  const bool pok = p0 + c < TV;                         // phase 1: this lane's position exists
  constexpr int MASK = 0x7ffffff0;

  // W fragments: B operand of dx, k = latent 4 g + q, column = position p0 + c, channel ch0 + i
  float wf[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = 4 * g + q;
      wf[i][g] = (j < L && pok) ? W[(size_t)j * K + (ch0 + i) * TV + p0 + c] : 0.f;
    }

  f32x4 dw[8], pacc[CT], qacc[CT];
#pragma unroll
  for (int i = 0; i < 8; ++i) dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < CT; ++t) pacc[t] = qacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f, da = 0.f;

  // ---- tile movers (buffer-addressed: one 32-bit lane offset per stream; clips beyond the chunk fall outside the descriptor and
  // read 0 / are not stored; positions beyond T V are masked through the lane offset) ---------------------------------------------
  auto tile_res = [&](const float* base, int n0, int rowfloats) {
    const int left = nend - n0;
    return make_res(base + (size_t)(left > 0 ? n0 : 0) * rowfloats, left > 0 ? (unsigned)left * (unsigned)rowfloats * 4u : 0u);
  };
  // Z / input rows: item (wave, i) <-> clip wave + 8 (i & 1), row block i / 2 (Z's 16-row blocks, then the input's); lanes <-> (row of
  // the block, float4 of the 16 positions): a wave's 64 lanes fetch 16 rows x 64 contiguous bytes of one clip
  const int zx_p4 = lane & 3, zx_row = lane >> 2;
  const int zx_voff = (p0 + 4 * zx_p4 < TV) ? (zx_row * TV + p0 + 4 * zx_p4) * 4 : MASK;
  auto zx_load = [&](int n0, float4 (&v)[NL]) {
    const BufRes rz = tile_res(Zg, n0, Ci * TV), rx = tile_res(xin, n0, Ci * TV);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int cl = wave + 8 * (i & 1), rb = i >> 1;
      v[i] = (BC_SKIP & 4) ? float4{0.f, 0.f, 0.f, 0.f} : buf_load4(rb < CT ? rz : rx, zx_voff, (cl * Ci + 16 * (rb % CT)) * TV * 4);
    }
  };
  auto zx_store = [&](const float4 (&v)[NL]) {
    if (BC_SKIP & 8) return;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int cl = wave + 8 * (i & 1), rb = i >> 1;
      float4 x = v[i];
      if (xact && rb >= CT) { x.x = prelu_f(x.x, a_in); x.y = prelu_f(x.y, a_in); x.z = prelu_f(x.z, a_in); x.w = prelu_f(x.w, a_in); }
      float* d = ZX + cl * ZCS + (16 * rb + zx_row) * RSTR + 4 * zx_p4;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
  };
  // U rows: lane (q, c), register [i][r] <-> clip n0 + 4 q + r, channel ch0 + i, position p0 + c
  const int u_voff = pok ? (4 * q * K + p0 + c) * 4 : MASK;
  auto u_load = [&](int n0, float (&u)[8][4]) {
    const BufRes ru = tile_res(U, n0, K);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) u[i][r] = (BC_SKIP & 1) ? ((i + r) & 1 ? 1.f : -1.f) : buf_load1(ru, u_voff, (r * K + (ch0 + i) * TV) * 4);
  };
  const int dza_voff = (c * L + q) * 4, dzb_voff = (4 * q * L + c) * 4;
  auto dz_load = [&](int n0, float (&da_)[4], float (&db_)[4]) {
    const BufRes rd = tile_res(dz, n0, L);
#pragma unroll
    for (int g = 0; g < 4; ++g) {          // A operand of dx: row = clip n0 + c, k = latent 4 g + q
      const float v = buf_load1(rd, dza_voff, 16 * g);
      da_[g] = 4 * g + q < L ? v : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {          // A operand of dW: row = latent c, k = clip n0 + 4 q + r
      const float v = buf_load1(rd, dzb_voff, r * L * 4);
      db_[r] = c < L ? v : 0.f;
    }
  };

  float4 zx[NL];
  float ucur[8][4], dza[4], dzb[4];
  u_load(nbeg, ucur);
  dz_load(nbeg, dza, dzb);
  zx_load(nbeg, zx);

  float* gw = G + 4 * q * GCS + ch0 * RSTR + c;                      // phase 1: G[clip 4 q + r][channel ch0 + i][position c]
  const float* ga = G + 4 * q * GCS + (16 * wc + c) * RSTR + 8 * h;  // phase 2: G[clip 4 q + r][channel 16 wc + c][position 8 h + pl]
  const float* zb = ZX + 4 * q * ZCS + c * RSTR + 8 * h;             // phase 2: ZX[clip 4 q + r][row 16 t + c][position 8 h + pl]

  for (int n0 = nbeg; n0 < nend; n0 += NCL) {
    // ---- phase 1: dx, dU rows, dW, the G image; the B-side tile into LDS ------------------------------------------------
    float unext[8][4];
    if (BC_PF == 2) { zx_store(zx); zx_load(n0 + NCL, zx); }
    if (BC_PF) u_load(n0 + NCL, unext);    // in front of this tile's stores: a whole tile to arrive
    {
      const BufRes rdu = tile_res(dU, n0, K);
      // (eight independent MFMA chains at a time: k-step outermost for dx, clip register outermost for dW)
      f32x4 dx[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) dx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i) dx[i] = mfma(dza[g], wf[i][g], dx[i]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float u = ucur[i][r];
          float g = dx[i][r], x = u;
          if (pre) {
            da += u < 0.f ? g * u : 0.f;
            g = u > 0.f ? g : a * g;
            x = prelu_f(u, a);
          }
          if (!(BC_SKIP & 32)) dw[i] = mfma(dzb[r], x, dw[i]);
          if (!(BC_SKIP & 64)) gw[r * GCS + i * RSTR] = g;
          if (!(BC_SKIP & 2) || g == 123.456f) buf_store1(rdu, u_voff, (r * K + (ch0 + i) * TV) * 4, g);
        }
    }
    if (BC_PF != 2) zx_store(zx);
    lds_barrier();                         // G and the B-side tile are complete
    // ---- phase 2: the next tile's fetches (issued from phase 1, interleaved with its stores, they measured 10 us slower), then
    // P / Q / s ---------------------------------------------------------------------------------------------------------------
    if (BC_PF) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) ucur[i][r] = unext[i][r];
    } else {
      u_load(n0 + NCL, ucur);              // (dead since phase 1: the next tile travels in the same registers)
    }
    dz_load(n0 + NCL, dza, dzb);
    if (BC_PF != 2) zx_load(n0 + NCL, zx);
#pragma unroll
    for (int pl = 0; pl < 8; ++pl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float av = (BC_SKIP & 64) ? 1.f : ga[r * GCS + pl];
        ssum += av;
        if (BC_SKIP & 16) continue;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
          pacc[t] = mfma(av, zb[r * ZCS + (16 * t) * RSTR + pl], pacc[t]);
          qacc[t] = mfma(av, zb[r * ZCS + (Ci + 16 * t) * RSTR + pl], qacc[t]);
        }
      }
    lds_barrier();                         // every wave has left G and the B-side tile
  }

  // ---- what the workgroup leaves behind -------------------------------------------------------------------------------
  // dW slab of this chunk: tile i, register r <-> latent 4 q + r, column (ch0 + i) T V + p0 + c
  {
    float* dst = dWp + (size_t)ychunk * L * K + p0 + c;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * q + r;
        if (j < L && pok) dst[(size_t)j * K + (ch0 + i) * TV] = dw[i][r];
      }
  }
  // [P][Q][s] row: the two position halves add theirs one after the other (fixed order); channel waves own disjoint rows
  float* row = lds;                        // E floats over G (the loop's last barrier has passed)
  float* red = lds + E;                    // 8 floats: the waves' slope-gradient sums
  const float s4 = ssum + __shfl_xor(ssum, 16, 64);
  const float sch = s4 + __shfl_xor(s4, 32, 64);           // lanes (q, c) of every q: channel 16 wc + c
  da = wave_sum(da);
  for (int hh = 0; hh < 2; ++hh) {
    if (h == hh) {
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * wc + 4 * q + r, cc = 16 * t + c;      // D layout: register r <-> row 4 q + r, column c
          float* p = row + o * Ci + cc;
          p[0] = (hh ? p[0] : 0.f) + pacc[t][r];
          p[HID * Ci] = (hh ? p[HID * Ci] : 0.f) + qacc[t][r];
        }
      if (q == 0) {
        float* p = row + 2 * HID * Ci + 16 * wc + c;
        p[0] = (hh ? p[0] : 0.f) + sch;
      }
      if (lane == 0) red[wave] = da;
    }
    __syncthreads();
  }
  float* dstrow = prow + (size_t)wg * E;
  for (int e = tid; e < E; e += kThreads) dstrow[e] = row[e];
  if (tid == 0) dap[wg] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

struct Plan {
  int npt, S, chunk;
};
static Plan plan(int B, int TV) {
  Plan p;
  p.npt = ceil_div(TV, PT);
  int s = 256 / p.npt;                                     // one 8-wave workgroup per CU, all resident at once
  const int smax = ceil_div(B, NCL);
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  p.chunk = round_up(ceil_div(B, s), NCL);
  p.S = ceil_div(B, p.chunk);
  return p;
}
static inline size_t sums_offset(int rows, int E) { return ((size_t)rows * E + 1) / 2 * 2; }   // = chain_sums_offset (stsgcn_bwd.hip)

}  // namespace bc

// bottleneck.hip
int launch_btlnk_reduce(const float* partials, int P, size_t E, float* out, const float* dz, int B, int L, float* db,
                        const float* dap, int nda, float* dslope, int accumulate, hipStream_t stream, const float* rows,
                        int RP, int RE, double* rsum);
}  // namespace coskad

using namespace coskad;

extern "C" {

/* 1 when coskad_btlnk_bwd_chain_f32 takes the shape: K = 64 TV (a 64-channel last layer), TV % 4 == 0, 16 or 32 channels below */
int coskad_btlnk_bwd_chain_ok(int K, int TV, int below_Ci) {
  return TV > 0 && TV % 4 == 0 && K == bc::HID * TV && (below_Ci == 16 || below_Ci == 32);
}

/* partial rows the call writes for the top layer (each 2 * 64 * below_Ci + 64 floats) */
int coskad_btlnk_bwd_chain_rows(int B, int TV) {
  if (B <= 0 || TV <= 0) return 0;
  const bc::Plan p = bc::plan(B, TV);
  return p.npt * p.S;
}

/* floats of the chain buffer (partial rows, then their fp64 sums) */
size_t coskad_btlnk_bwd_chain_floats(int B, int TV, int below_Ci) {
  const size_t E = 2 * (size_t)bc::HID * below_Ci + bc::HID;
  return bc::sums_offset(coskad_btlnk_bwd_chain_rows(B, TV), (int)E) + 2 * E;
}

size_t coskad_btlnk_bwd_chain_ws_bytes(int B, int K, int L, int TV) {
  if (B <= 0 || TV <= 0) return 0;
  const bc::Plan p = bc::plan(B, TV);
  return ((size_t)p.S * L * K + (size_t)p.S * p.npt + 64) * sizeof(float);
}

int coskad_btlnk_bwd_chain_f32(const float* U, const float* W, const float* dz, const float* slope, float* dU, float* dW,
                               float* db, float* dslope, void* ws, size_t ws_bytes, int accumulate, int B, int K, int L,
                               const float* below_in, const float* below_Z, const float* below_in_slope, int below_Ci, int TV,
                               float* stats_out, size_t stats_out_bytes, int* stats_rows, hipStream_t stream) {
  if (!U || !W || !dz || !dU || !dW || !ws || !below_in || !below_Z || !stats_out || !stats_rows)
    return fail(COSKAD_ERR_ARG, "btlnk_bwd_chain: null pointer");
  if (B <= 0 || K <= 0 || L <= 0) return fail(COSKAD_ERR_ARG, "btlnk_bwd_chain: B=%d K=%d L=%d", B, K, L);
  if (L > 16) return fail(COSKAD_ERR_SHAPE, "btlnk_bwd_chain: latent_dim=%d > 16 not supported", L);
  if (!coskad_btlnk_bwd_chain_ok(K, TV, below_Ci))
    return fail(COSKAD_ERR_SHAPE, "btlnk_bwd_chain: K=%d TV=%d below_Ci=%d (needs K = 64 TV, TV %% 4 == 0, 16 / 32 channels below)", K, TV, below_Ci);
  if ((size_t)stats_out & 7) return fail(COSKAD_ERR_ARG, "btlnk_bwd_chain: stats_out must be 8-byte aligned");
  if (ws_bytes < coskad_btlnk_bwd_chain_ws_bytes(B, K, L, TV)) return fail(COSKAD_ERR_WORKSPACE, "btlnk_bwd_chain: workspace too small");
  if (stats_out_bytes < coskad_btlnk_bwd_chain_floats(B, TV, below_Ci) * sizeof(float))
    return fail(COSKAD_ERR_WORKSPACE, "btlnk_bwd_chain: stats_out %zu bytes too small", stats_out_bytes);
  const bc::Plan p = bc::plan(B, TV);
  const int rows = p.npt * p.S;
  const int E = 2 * bc::HID * below_Ci + bc::HID;
  float* dWp = reinterpret_cast<float*>(ws);
  float* dap = dWp + (size_t)p.S * L * K;
  const size_t lds = (size_t)bc::NCL * ((bc::HID * bc::RSTR + 4) + (2 * below_Ci * bc::RSTR + 4)) * sizeof(float);
  int rc;
  {
  ProbeScope probe(KID_BTLNK_BWD, below_Ci, L, stream);
  if (below_Ci == 32) {
    auto k = bc::k_btlnk_bwd_stats<2>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(p.npt * p.S), dim3(bc::kThreads), lds, stream, U, W, dz, slope, dU, dWp, dap, below_in, below_Z,
                       below_in_slope, stats_out, B, TV, L, p.chunk);
  } else {
    auto k = bc::k_btlnk_bwd_stats<1>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(p.npt * p.S), dim3(bc::kThreads), lds, stream, U, W, dz, slope, dU, dWp, dap, below_in, below_Z,
                       below_in_slope, stats_out, B, TV, L, p.chunk);
  }
  }
  if ((rc = check_launch("btlnk_bwd_stats"))) return rc;
  // one launch sums the dW slabs, the bias / slope gradients and the chain buffer's partial rows
  *stats_rows = rows;
  return launch_btlnk_reduce(dWp, p.S, (size_t)L * K, dW, dz, B, L, db, dap, rows, (dslope && slope) ? dslope : nullptr, accumulate, stream,
                             stats_out, rows, E, reinterpret_cast<double*>(stats_out + bc::sums_offset(rows, E)));
}

}  // extern "C"
