// MFMA version of the fused ST_GCNN layer forward (same contract as k_layer_apply in stsgcn_fwd.hip):
//   U = Wz . gcn(PReLU_in(in)) + Wx . PReLU_in(in) + b
// Persistent blocks; the mixing matrices and the folded conv weights are loaded into LDS once
// per block; gcn and both 1x1 convs run on v_mfma_f32_16x16x4_f32.
#include "mfma_ops.h"
#include <cstdlib>

namespace coskad {

template <int T, int V, int OTI>
__global__ __launch_bounds__(kBlock) void k_layer_apply_m(
    const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ wfold, const float* __restrict__ bias,
    const float* __restrict__ in_slope, const float* __restrict__ out_slope, int B, int Ci, int Co,
    int CoP, int NB, int dbg) {
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
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
    __syncthreads();  // tables ready / previous tile's conv reads done
    if (!(dbg & 4)) stage_rows<T, V>(gin, img, rows * TV, pre, a_in);
    __syncthreads();
    if (!(dbg & 1)) gcn_mfma<T, V, false>(img, rows, AwL, TwL);
    __syncthreads();
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

size_t layer_apply_m_lds(int T, int V, int Ci, int CoP, int NB) {
  const int TV = T * V, LD = TV % 2 == 0 ? TV + 1 : TV;
  return ((size_t)NB * Ci * LD + (size_t)T * V * V + (size_t)V * T * T + 2 * (size_t)round_up(Ci, 4) * CoP + CoP) * sizeof(float);
}

template <int T, int V>
int launch_layer_apply_m(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold,
                         const float* bias, const float* in_slope, const float* out_slope, int B, int Ci,
                         int Co, hipStream_t st) {
  const int CoP = round_up(Co, 16);
  int NB = Ci >= 32 ? 1 : 32 / Ci;   // 32 rows per tile: 2 row tiles of 16
  if (NB > B) NB = B;
  const size_t lds = layer_apply_m_lds(T, V, Ci, CoP, NB);
  if (lds > (size_t)kMaxLdsBytes) return 1;  // caller falls back to the VALU kernel
  const int ntiles = ceil_div(B, NB);
  const int per_cu = (int)((size_t)kMaxLdsBytes / lds);
  int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
  if (grid > ntiles) grid = ntiles;
#define LAUNCH_OTI(OTI)                                                                             \
  do {                                                                                              \
    auto k = k_layer_apply_m<T, V, OTI>;                                                            \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), lds, st, in, out, Aw, Tw, wfold, bias, in_slope,   \
                       out_slope, B, Ci, Co, CoP, NB, dbg);                                         \
  } while (0)
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("COSKAD_DBG"); dbg = e ? atoi(e) : 0; }
  ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
  if (CoP == 16) LAUNCH_OTI(1);
  else if (CoP == 32) LAUNCH_OTI(2);
  else if (CoP == 48) LAUNCH_OTI(3);
  else LAUNCH_OTI(4);
#undef LAUNCH_OTI
  return check_launch("layer_apply_m");
}

// explicit instantiations used by stsgcn_fwd.hip's dispatcher
template int launch_layer_apply_m<12, 17>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t);
template int launch_layer_apply_m<12, 25>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t);
template int launch_layer_apply_m<12, 14>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t);
template int launch_layer_apply_m<12, 18>(const float*, float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, hipStream_t);

}  // namespace coskad
