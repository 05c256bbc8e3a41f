// Narrow-output ST_GCNN layers (C_out <= 4 behind >= 16 input channels: the decoder's LAST layer, reference
// models/common/components.py:143-179 -> models/graph_layers/stsgcn.py:94-116) by commutation.
//
// The mixing acts per channel on (frame, joint), a 1x1 convolution mixes channels at one position: they commute,
//     Wt gcn(X) = gcn(Wt X).
// So both convolutions of the layer run FIRST, on the wide input, as ONE streaming pass
//     [Y; R] = [Wt; Wr] PReLU(U_prev)          (J = 2 C_out rows out of C_in)
// and the mixing, both BatchNorms, the add and the PReLU see 2 C_out-channel tensors: the few-channel kernels of
// first_layer.hip / the generic backward on a virtual (2 C_out -> C_out) layer with selector weights (coskad_amd/trainer.py,
// `narrow` segments).  Instead of the statistics / apply / batch-reduction / data / dA-dT passes over 32-channel tensors (738 us
// per step for the 32 -> 2 layer of the 25-joint decoder) the layer costs this file's two passes -- forward: read U_prev;
// backward: read U_prev and d[Y; R], write dU_prev, per-workgroup partial rows of d[Wt; Wr] and of the producer's slope
// gradient (summed in a fixed order by coskad_gemm_sum_f32: deterministic) -- plus launches on 4-channel tensors.
#include "common.h"

namespace coskad {
namespace nl {

constexpr int JMAX = 8, CMAX = 64, kThreads = 256;

__device__ __forceinline__ float prelu1(float v, float a) { return v > 0.f ? v : a * v; }

// sum over the 64 lanes as six DPP adds (quad swaps, half-row / row mirrors, row broadcasts) + a readlane -- the shuffle butterfly of
// wave_sum (six ds_bpermute + adds) costs three times as much, and this kernel does J C_in of them per 256 positions
__device__ __forceinline__ float wave_total(float v) {
  int x = __float_as_int(v);
#define DPP_ADD(ctrl, rmask)                                                                                         \
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true)))
  DPP_ADD(0xB1, 0xf);    // quad_perm [1, 0, 3, 2]
  DPP_ADD(0x4E, 0xf);    // quad_perm [2, 3, 0, 1]
  DPP_ADD(0x141, 0xf);   // row_half_mirror
  DPP_ADD(0x140, 0xf);   // row_mirror: every lane of a 16-lane row holds the row's sum
  DPP_ADD(0x142, 0xa);   // row_bcast15 into rows 1 and 3
  DPP_ADD(0x143, 0xc);   // row_bcast31 into rows 2 and 3: row 3 holds the total
#undef DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// thread <-> (clip, float4 of positions); channels in the loop: a wave's loads are 1 KB contiguous per channel
template <int J>
__global__ __launch_bounds__(kThreads) void k_narrow_fwd(const float* __restrict__ U, const float* __restrict__ in_slope,
                                                          const float* __restrict__ W, float* __restrict__ out, int B, int Ci, int R4) {
  __shared__ float wl[JMAX * CMAX];
  for (int e = threadIdx.x; e < J * Ci; e += kThreads) wl[e] = W[e];
  __syncthreads();
  const bool pre = in_slope != nullptr;
  const float a = pre ? in_slope[0] : 0.f;
  const long long total = (long long)B * R4;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long long)gridDim.x * kThreads) {
    const int n = (int)(idx / R4), p4 = (int)(idx - (long long)n * R4);
    const float4* src = reinterpret_cast<const float4*>(U) + (size_t)n * Ci * R4 + p4;
    float4 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < Ci; ++c) {
      float4 x = src[(size_t)c * R4];
      if (pre) { x.x = prelu1(x.x, a); x.y = prelu1(x.y, a); x.z = prelu1(x.z, a); x.w = prelu1(x.w, a); }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float w = wl[j * Ci + c];
        acc[j].x = fmaf(w, x.x, acc[j].x); acc[j].y = fmaf(w, x.y, acc[j].y);
        acc[j].z = fmaf(w, x.z, acc[j].z); acc[j].w = fmaf(w, x.w, acc[j].w);
      }
    }
    float4* dst = reinterpret_cast<float4*>(out) + (size_t)n * J * R4 + p4;
#pragma unroll
    for (int j = 0; j < J; ++j) dst[(size_t)j * R4] = acc[j];
  }
}

// dU_prev = (W^T dOut) * PReLU'(U_prev);  partial row of the workgroup: [dW (J x Ci)][slope gradient]
template <int J, int Ci>
__global__ __launch_bounds__(kThreads) void k_narrow_bwd(const float* __restrict__ U, const float* __restrict__ in_slope,
                                                          const float* __restrict__ W, const float* __restrict__ dOut,
                                                          float* __restrict__ dU, float* __restrict__ partials, int B, int R4) {
  __shared__ float wl[J * Ci];
  __shared__ float red[kThreads / 64][J * Ci + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int E = J * Ci;
  for (int e = threadIdx.x; e < E; e += kThreads) wl[e] = W[e];
  __syncthreads();
  const bool pre = in_slope != nullptr;
  const float a = pre ? in_slope[0] : 0.f;
  // wave totals of dW live in LDS (one slot per (j, c) and wave, added by lane 0): a register array indexed by the channel would need
  // the channel loop fully unrolled, and hipcc then hoists all C_in loads (512 registers at 32 channels)
  for (int e = threadIdx.x; e < (kThreads / 64) * (E + 1); e += kThreads) (&red[0][0])[e] = 0.f;
  __syncthreads();
  float da = 0.f;
  const long long total = (long long)B * R4;
  const long long span = (long long)gridDim.x * kThreads;
  // every lane of a wave runs every round (the wave sums below are butterflies over all 64 lanes); lanes beyond the data carry zeros
  for (long long base = (long long)blockIdx.x * kThreads; base < total; base += span) {
    const long long idx = base + threadIdx.x;
    const bool live = idx < total;
    const int n = live ? (int)(idx / R4) : 0, p4 = live ? (int)(idx - (long long)n * R4) : 0;
    const float4* src = reinterpret_cast<const float4*>(U) + (size_t)n * Ci * R4 + p4;
    const float4* gsrc = reinterpret_cast<const float4*>(dOut) + (size_t)n * J * R4 + p4;
    float4* dst = reinterpret_cast<float4*>(dU) + (size_t)n * Ci * R4 + p4;
    float4 g[J];
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = live ? gsrc[(size_t)j * R4] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < Ci; ++c) {
      const float4 u = live ? src[(size_t)c * R4] : float4{0.f, 0.f, 0.f, 0.f};
      float4 x = u;
      if (pre) { x.x = prelu1(u.x, a); x.y = prelu1(u.y, a); x.z = prelu1(u.z, a); x.w = prelu1(u.w, a); }
      float4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float w = wl[j * Ci + c];
        d.x = fmaf(w, g[j].x, d.x); d.y = fmaf(w, g[j].y, d.y); d.z = fmaf(w, g[j].z, d.z); d.w = fmaf(w, g[j].w, d.w);
        // d W[j][c] += <dOut_j, X_c> over this wave's 64 position quads
        float s = fmaf(g[j].x, x.x, fmaf(g[j].y, x.y, fmaf(g[j].z, x.z, g[j].w * x.w)));
        s = wave_total(s);
        if (lane == 0) red[wave][j * Ci + c] += s;
      }
      if (pre) {
        da += (u.x < 0.f ? d.x * u.x : 0.f) + (u.y < 0.f ? d.y * u.y : 0.f) + (u.z < 0.f ? d.z * u.z : 0.f) + (u.w < 0.f ? d.w * u.w : 0.f);
        d.x = u.x > 0.f ? d.x : a * d.x; d.y = u.y > 0.f ? d.y : a * d.y; d.z = u.z > 0.f ? d.z : a * d.z; d.w = u.w > 0.f ? d.w : a * d.w;
      }
      if (live) dst[(size_t)c * R4] = d;
    }
  }
  da = wave_sum(da);
  if (lane == 0) red[wave][E] = da;
  __syncthreads();
  float* row = partials + (size_t)blockIdx.x * (E + 1);
  for (int e = threadIdx.x; e <= E; e += kThreads) row[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

static int grid_for(int B, int R4) {
  const long long items = (long long)B * R4;
  long long g = (items + kThreads - 1) / kThreads;
  return (int)(g < 1024 ? g : 1024);
}

}  // namespace nl
}  // namespace coskad

using namespace coskad;

extern "C" {

/* rows of the partial table coskad_narrow_conv_bwd_f32 writes (each J * Ci + 1 floats) */
int coskad_narrow_conv_rows(int B, int TV) { return (B > 0 && TV > 0 && TV % 4 == 0) ? nl::grid_for(B, TV / 4) : 0; }

/* out [B, J, TV] = W [J, Ci] . PReLU(U [B, Ci, TV])  (in_slope NULL: U is already activated); J <= 8, Ci <= 64, TV % 4 == 0 */
int coskad_narrow_conv_fwd_f32(const float* U, const float* in_slope, const float* W, float* out, int B, int Ci, int J, int TV,
                               hipStream_t stream) {
  if (!U || !W || !out) return fail(COSKAD_ERR_ARG, "narrow_conv_fwd: null pointer");
  if (B <= 0 || Ci <= 0 || Ci > nl::CMAX || TV <= 0 || TV % 4 || (J != 2 && J != 4 && J != 6 && J != 8))
    return fail(COSKAD_ERR_SHAPE, "narrow_conv_fwd: B=%d Ci=%d J=%d TV=%d (J in {2, 4, 6, 8}, Ci <= 64, TV %% 4 == 0)", B, Ci, J, TV);
  const int R4 = TV / 4, grid = nl::grid_for(B, R4);
#define LAUNCH_NF(J_) hipLaunchKernelGGL((nl::k_narrow_fwd<J_>), dim3(grid), dim3(nl::kThreads), 0, stream, U, in_slope, W, out, B, Ci, R4)
  if (J == 2) LAUNCH_NF(2); else if (J == 4) LAUNCH_NF(4); else if (J == 6) LAUNCH_NF(6); else LAUNCH_NF(8);
#undef LAUNCH_NF
  return check_launch("narrow_conv_fwd");
}

/* dU [B, Ci, TV] = (W^T dOut) * PReLU'(U);  partials [coskad_narrow_conv_rows(B, TV)][J * Ci + 1]: per-workgroup sums of
 * dW[j][c] = sum dOut_j PReLU(U)_c and (last column) of the producer's slope gradient sum (W^T dOut) U [U < 0] -- add the rows with
 * coskad_gemm_sum_f32 (fixed order) */
int coskad_narrow_conv_bwd_f32(const float* U, const float* in_slope, const float* W, const float* dOut, float* dU, float* partials,
                               size_t partials_floats, int B, int Ci, int J, int TV, hipStream_t stream) {
  if (!U || !W || !dOut || !dU || !partials) return fail(COSKAD_ERR_ARG, "narrow_conv_bwd: null pointer");
  if (B <= 0 || Ci <= 0 || Ci > nl::CMAX || TV <= 0 || TV % 4 || (J != 2 && J != 4 && J != 6 && J != 8))
    return fail(COSKAD_ERR_SHAPE, "narrow_conv_bwd: B=%d Ci=%d J=%d TV=%d (J in {2, 4, 6, 8}, Ci <= 64, TV %% 4 == 0)", B, Ci, J, TV);
  const int R4 = TV / 4, grid = nl::grid_for(B, R4);
  if (partials_floats < (size_t)grid * (J * Ci + 1)) return fail(COSKAD_ERR_WORKSPACE, "narrow_conv_bwd: partial table too small");
  if (Ci != 16 && Ci != 32 && Ci != 64) return fail(COSKAD_ERR_SHAPE, "narrow_conv_bwd: built for 16 / 32 / 64 input channels (%d)", Ci);
#define LAUNCH_NB(J_, C_) hipLaunchKernelGGL((nl::k_narrow_bwd<J_, C_>), dim3(grid), dim3(nl::kThreads), 0, stream, U, in_slope, W, dOut, dU, partials, B, R4)
#define LAUNCH_NBJ(J_) do { if (Ci == 16) LAUNCH_NB(J_, 16); else if (Ci == 32) LAUNCH_NB(J_, 32); else LAUNCH_NB(J_, 64); } while (0)
  if (J == 2) LAUNCH_NBJ(2); else if (J == 4) LAUNCH_NBJ(4); else if (J == 6) LAUNCH_NBJ(6); else LAUNCH_NBJ(8);
#undef LAUNCH_NBJ
#undef LAUNCH_NB
  return check_launch("narrow_conv_bwd");
}

}  // extern "C"
