// MFMA version of the fused ST_GCNN layer forward (same contract as k_layer_apply in stsgcn_fwd.hip):
//   U = Wz . gcn(PReLU_in(in)) + Wx . PReLU_in(in) + b
// Persistent blocks; the mixing matrices and the folded conv weights are loaded into LDS once
// per block; gcn and both 1x1 convs run on v_mfma_f32_16x16x4_f32.
#include "mfma_ops.h"
#include <cstdlib>

namespace coskad {

template <int T, int V, int OTI>
__global__ __launch_bounds__((Geo<T, V>::Block)) void k_layer_apply_m(
    const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ wfold, const float* __restrict__ bias,
    const float* __restrict__ in_slope, const float* __restrict__ out_slope, int B, int Ci, int Co,
    int CoP, int NB, int dbg, const float* __restrict__ Zg) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
#ifndef COSKAD_ABLATE
  dbg = 0;   // product build: every `dbg &` test below folds away (phase ablation needs -DCOSKAD_ABLATE)
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int KZ = round_up(Ci, 4);
  float* img = lds;
  float* AwL = img + NB * Ci * LD;
  float* TwL = AwL + T * V * V;
  float* Wl = TwL + V * T * T;
  float* bl = Wl + 2 * KZ * CoP;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const bool post = out_slope != nullptr;
  const float a_out = post ? out_slope[0] : 0.f;
  copy_to_lds(AwL, Aw, T * V * V);
  copy_to_lds(TwL, Tw, V * T * T);
  load_wfold_padded(Wl, wfold, Ci, KZ, CoP, 2);
  copy_to_lds(bl, bias, CoP);
  const int wave = uniform(threadIdx.x >> 6);

  const int ntiles = ceil_div(B, NB);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int clip0 = tile * NB;
    const int nb = min(NB, B - clip0);
    const int rows = nb * Ci;
    const float* gin = in + (size_t)clip0 * Ci * TV;
    lds_barrier();  // tables ready / previous tile's conv reads done
    if (Zg) {              // stored Z = gcn(X) (training: written by the statistics pass): stage it, skip the mixing
      stage_rows<T, V>(Zg + (size_t)clip0 * Ci * TV, img, rows * TV, false, 0.f);
    } else {
      if (!(dbg & 4)) stage_rows<T, V>(gin, img, rows * TV, pre, a_in);
      lds_barrier();
      if (!(dbg & 1)) gcn_mfma<T, V, false>(img, rows, AwL, TwL);
    }
    lds_barrier();
    for (int n = 0; n < nb && !(dbg & 2); ++n) {
      float* og = out + (size_t)(clip0 + n) * Co * TV;
      auto epi = [&](int o, int p, bool pok, float v0, float v1) {
        if (o < Co && pok && !(dbg & 8)) {
          v0 += bl[o];
          v1 += bl[o];
          if (post) { v0 = prelu_f(v0, a_out); v1 = prelu_f(v1, a_out); }
          *reinterpret_cast<float2*>(og + (size_t)o * TV + p) = float2{v0, v1};
        }
      };
      const int NOG = (CoP / 16 + OTI - 1) / OTI;
      for (int g = 0; g < NOG; ++g)
        conv_mfma_s<T, V, OTI>(img + n * Ci * LD, (dbg & 32) ? 0 : KZ, Ci, nullptr, 0, 1, gin + (size_t)n * Ci * TV,
                               (dbg & 16) ? 0 : KZ, Ci, pre, a_in, Wl, CoP, g, (wave + n + g) % (kBlock / 64), kBlock / 64, epi);
    }
  }
}

// Training-mode layer when Z = gcn(X) was stored by the statistics pass: U = Wz.Z + Wx.PReLU(Xpre) + b is then a pure
// streaming GEMM -- both operands come from HBM as 8-byte strip loads, nothing is staged, no barrier after the weight
// load.  A work item is (clip, 32-position strip, output-tile group); waves take items grid-stride, so the MFMA load is
// balanced over the SIMDs regardless of 7 strips per clip.  LDS: folded weights + bias only.
#ifndef COSKAD_XBZ
#define COSKAD_XBZ 4   // k-steps of 8-byte operand loads in flight per wave (everything comes from HBM here)
#endif
template <int T, int V, int OTI>
__global__ __launch_bounds__(256) void k_layer_apply_z(
    const float* __restrict__ Z, const float* __restrict__ in, float* __restrict__ out,
    const float* __restrict__ wfold, const float* __restrict__ bias, const float* __restrict__ in_slope,
    const float* __restrict__ out_slope, int B, int Ci, int Co, int CoP) {
  constexpr int TV = Geo<T, V>::TV;
  constexpr int PS = (TV + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int KZ = round_up(Ci, 4);
  float* Wl = lds;
  float* bl = Wl + 2 * KZ * CoP;
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const bool post = out_slope != nullptr;
  const float a_out = post ? out_slope[0] : 0.f;
  for (int e = threadIdx.x; e < 2 * KZ * CoP; e += 256) {      // load_wfold_padded for a 256-thread block
    const int k = e / CoP, o = e - k * CoP;
    const int src = k / KZ, c = k - src * KZ;
    Wl[e] = c < Ci ? wfold[(size_t)(src * Ci + c) * CoP + o] : 0.f;
  }
  for (int e = threadIdx.x; e < CoP; e += 256) bl[e] = bias[e];
  lds_barrier();
  const int NOG = (CoP / 16 + OTI - 1) / OTI;
  const int wave = uniform(threadIdx.x >> 6);
  const int KS = KZ / 4;                       // k-steps per source
  constexpr int XB = COSKAD_XBZ;
  const long long nitems = (long long)B * PS * NOG;
  for (long long item = (long long)blockIdx.x * 4 + wave; item < nitems; item += (long long)gridDim.x * 4) {
    const int og = (int)(item % NOG);
    const long long r = item / NOG;
    const int sp = (int)(r % PS);
    const int clip = (int)(r / PS);
    const int lane = tid_here() & 63;          // lane geometry recomputed per item: nothing address-like stays live
    const int j = lane & 15, kk = lane >> 4;
    const int p = 32 * sp + 2 * j;
    const bool pok = p < TV;
    const int pc = pok ? p : TV - 2;
    // lane pointers; k-step g of a source is `+ g * 4 * TV` (rows beyond C_in are clamped: their weights are zero)
    const int rowk = kk < Ci ? kk : Ci - 1;
    const float* zp = Z + ((size_t)clip * Ci + rowk) * TV + pc;
    const float* xp = in + ((size_t)clip * Ci + rowk) * TV + pc;
    const bool full = (Ci & 3) == 0;
    auto gload = [&](int g) -> float2 {        // g in [0, 2*KS): Z rows then X rows
      if (g >= 2 * KS) return float2{0.f, 0.f};
      const bool isz = g < KS;
      const int gs = isz ? g : g - KS;
      const float* base = isz ? zp : xp;
      int off = gs * 4 * TV;
      if (!full && 4 * gs + kk >= Ci) off = (Ci - 1 - rowk) * TV;
      float2 v = *reinterpret_cast<const float2*>(base + off);
      if (!isz && pre) { v.x = prelu_f(v.x, a_in); v.y = prelu_f(v.y, a_in); }
      return v;
    };
    f32x4 acc[OTI][2];
#pragma unroll
    for (int t = 0; t < OTI; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float* wcol = Wl + 16 * og * OTI + j;
    float2 cur[XB], nxt[XB];
#pragma unroll
    for (int u = 0; u < XB; ++u) cur[u] = gload(u);
    for (int g0 = 0; g0 < 2 * KS; g0 += XB) {
#pragma unroll
      for (int u = 0; u < XB; ++u) nxt[u] = gload(g0 + XB + u);
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        if (g0 + u < 2 * KS) {
          const float* w = wcol + (4 * (g0 + u) + kk) * CoP;     // weight rows: [Z part KZ][X part KZ]
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
    if (pok) {
      float* ogp = out + (size_t)clip * Co * TV + p;
#pragma unroll
      for (int t = 0; t < OTI; ++t)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int o = 16 * (og * OTI + t) + 4 * kk + r4;
          if (o < Co) {
            float v0 = acc[t][0][r4] + bl[o], v1 = acc[t][1][r4] + bl[o];
            if (post) { v0 = prelu_f(v0, a_out); v1 = prelu_f(v1, a_out); }
            *reinterpret_cast<float2*>(ogp + (size_t)o * TV) = float2{v0, v1};
          }
        }
    }
  }
}

template <int T, int V>
int launch_layer_apply_z(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                         const float* in_slope, const float* out_slope, int B, int Ci, int Co, hipStream_t st) {
  const int CoP = round_up(Co, 16), KZ = round_up(Ci, 4);
  const size_t lds = ((size_t)2 * KZ * CoP + CoP) * sizeof(float);
  if (lds > (size_t)kMaxLdsBytes) return fail(COSKAD_ERR_SHAPE, "layer_apply_z: LDS %zu too large", lds);
  constexpr int PS = (Geo<T, V>::TV + 31) / 32;
  const long long witems = ((long long)B * PS + 3) / 4;
  const int grid = (int)(witems < 256 * 8 ? witems : 256 * 8);   // 8 four-wave blocks per CU (sweep: 4..16 within 5 %)
#define LAUNCH_Z(OTI)                                                                                 \
  do {                                                                                                \
    auto k = k_layer_apply_z<T, V, OTI>;                                                              \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, Z, in, out, wfold, bias, in_slope, out_slope, B, Ci, Co, CoP); \
  } while (0)
  ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
  if (CoP == 16) LAUNCH_Z(1);
  else if (CoP == 32) LAUNCH_Z(2);
  else if (CoP == 48) LAUNCH_Z(3);
  else LAUNCH_Z(4);
#undef LAUNCH_Z
  return check_launch("layer_apply_z");
}

size_t layer_apply_m_lds(int T, int V, int Ci, int CoP, int NB) {
  const int TV = T * V, LD = TV % 2 == 0 ? TV + 1 : TV;
  return ((size_t)NB * Ci * LD + (size_t)T * V * V + (size_t)V * T * T + 2 * (size_t)round_up(Ci, 4) * CoP + CoP) * sizeof(float);
}

template <int T, int V>
int launch_layer_apply_m(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold,
                         const float* bias, const float* in_slope, const float* out_slope, int B, int Ci,
                         int Co, hipStream_t st, const float* Zg) {
  constexpr int kBlock = Geo<T, V>::Block;   // threads per block of this geometry
  const int CoP = round_up(Co, 16);
  int NB = Ci >= 32 ? 1 : 32 / Ci;   // 32 rows per tile: 2 row tiles of 16
  if (NB > B) NB = B;
  const size_t lds = layer_apply_m_lds(T, V, Ci, CoP, NB);
  if (lds > (size_t)kMaxLdsBytes) return 1;  // caller falls back to the VALU kernel
  const int ntiles = ceil_div(B, NB);
  const int per_cu = (int)((size_t)kMaxLdsBytes / lds);
  int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 2048 / kBlock ? 2048 / kBlock : per_cu));   // (a CU holds 32 waves)
  if (grid > ntiles) grid = ntiles;
#define LAUNCH_OTI(OTI)                                                                             \
  do {                                                                                              \
    auto k = k_layer_apply_m<T, V, OTI>;                                                            \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, out, Aw, Tw, wfold, bias, in_slope,   \
                       out_slope, B, Ci, Co, CoP, NB, dbg, Zg);                                     \
  } while (0)
#ifdef COSKAD_ABLATE   // phase-ablation builds only (tools/ablate_bwd.py): the product library has no runtime switches
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("COSKAD_DBG"); dbg = e ? atoi(e) : 0; }
#else
  constexpr int dbg = 0;
#endif
  ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
  if (CoP == 16) LAUNCH_OTI(1);
  else if (CoP == 32) LAUNCH_OTI(2);
  else if (CoP == 48) LAUNCH_OTI(3);
  else LAUNCH_OTI(4);
#undef LAUNCH_OTI
  return check_launch("layer_apply_m");
}

// explicit instantiations used by stsgcn_fwd.hip's dispatcher
template int launch_layer_apply_m<12, 17>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t, const float*);
template int launch_layer_apply_m<12, 25>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t, const float*);
template int launch_layer_apply_m<12, 14>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t, const float*);
template int launch_layer_apply_m<12, 18>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t, const float*);

}  // namespace coskad

namespace coskad {
// fused_apply.hip
bool layer_apply_ring_ok(int T_, int V_, int Ci, int Co);
int launch_layer_apply_ring(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                            const float* in_slope, int B, int Ci, int Co, hipStream_t st);
// fused_apply_bpc.hip
int launch_layer_apply_bpc(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                           const float* in_slope, int B, int Ci, int Co, hipStream_t st);
// first_layer.hip
int launch_first_apply(const float* Z, const float* in, float* out, const float* wfold, const float* bias, const float* in_slope,
                       int B, int Ci, int Co, int TVr, hipStream_t st);
// fused_apply_flat.hip
bool layer_apply_flat_ok(int TV_, int Ci, int Co);
int launch_layer_apply_flat(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                            const float* in_slope, int B, int Ci, int Co, int TV_, hipStream_t st);
}  // namespace coskad

using namespace coskad;
extern "C" int coskad_layer_apply_z_f32(const float* Z, const float* in, float* out, const float* A, const float* Tm,
                                        const float* wfold, const float* bias, const float* in_slope,
                                        const float* out_slope, int B, int Ci, int Co, int T, int V,
                                        hipStream_t stream) {
  if (!Z || !in || !out || !A || !Tm || !wfold || !bias) return fail(COSKAD_ERR_ARG, "layer_apply_z: null pointer");
  if (B <= 0 || Ci <= 0 || Co <= 0 || Co > 64 || Ci > 64) return fail(COSKAD_ERR_ARG, "layer_apply_z: B=%d Ci=%d Co=%d", B, Ci, Co);
  // a handful of input channels (the first layer): plain FMAs on full-line stores (first_layer.hip)
  if (!out_slope && Ci <= 4 && (T * V) % 4 == 0) return launch_first_apply(Z, in, out, wfold, bias, in_slope, B, Ci, Co, T * V, stream);
  // default geometry, 16 / 32 input channels, pre-activation output: the wave-per-clip K-ring GEMM (fused_apply.hip)
  // default geometry, 32 -> 64 channels, pre-activation output: one clip per WORKGROUP, four waves sharing its K window
  // (fused_apply_bpc.hip: three waves per SIMD instead of one)
  if (!out_slope && T == 12 && V == 17 && Ci == 32 && Co == 64)
    return launch_layer_apply_bpc(Z, in, out, wfold, bias, in_slope, B, Ci, Co, stream);
  if (!out_slope && layer_apply_ring_ok(T, V, Ci, Co)) return launch_layer_apply_ring(Z, in, out, wfold, bias, in_slope, B, Ci, Co, stream);
  // the 25-joint layout, 16 / 32 input channels: the same K-ring GEMM, one clip per workgroup, over flat position tiles
  // (fused_apply_flat.hip)
  if (!out_slope && layer_apply_flat_ok(T * V, Ci, Co))
    return launch_layer_apply_flat(Z, in, out, wfold, bias, in_slope, B, Ci, Co, T * V, stream);
  // the streaming GEMM over Z and `in` (nothing staged) for every width up to 64: at 64 output channels on the 25-joint layout it
  // beats the LDS-tiled kernel with Z staged (258 -> ~235 us; encoder step 3.15 -> 3.13 ms)
  (void)A; (void)Tm;
#define CALL(T_, V_) return launch_layer_apply_z<T_, V_>(Z, in, out, wfold, bias, in_slope, out_slope, B, Ci, Co, stream)
  COSKAD_DISPATCH_TV(T, V, CALL);
#undef CALL
}
