// Training-mode layer apply on the stored-Z path (reference: models/graph_layers/stsgcn.py:94-116 with both BatchNorms
// folded from THIS batch's statistics by k_train_fold):
//     U[o][pos] = sum_c Wz[c][o] Z[c][pos] + sum_c Wx[c][o] PReLU(U_prev)[c][pos] + b[o]
// as a wave-per-clip K-ring GEMM, the forward twin of the K passes of fused_bwd.hip: one clip per wavefront, no workgroup
// barrier, the rows of Z and X stream through a 16-row LDS window quarter by quarter (full-line buffer loads; a
// quarter is stored behind the k-step that consumed those rows and its registers are refilled with the next group's),
// all 13 position tiles x all output channels accumulate at once (up to 208 accumulator registers: nothing else lives in
// that half of the register file here), operands of k-step s+1 are read while step s multiplies.  The result leaves
// through a 32-row LDS image, 32 channels at a time: full 1 KB lines.
// Replaces k_layer_apply_z (streaming strip GEMM, <= 32 output channels) and k_layer_apply_m (LDS tile kernel that
// re-mixes X although Z is stored) at T = 12, V = 17, 16 / 32 input channels.
#include "fused_ops.h"

namespace coskad {
namespace fa {

using namespace ff;

// CT: 16-row groups of the input; OTP: 16-channel output tiles per pass; NP: passes  (C_in = 16 CT, C_out = 16 OTP NP)
template <int CT, int OTP, int NP>
__global__ __launch_bounds__(256, 1) void k_layer_apply_ring(const float* __restrict__ in, const float* __restrict__ Zg,
                                                            const float* __restrict__ wfold, const float* __restrict__ bias,
                                                            const float* __restrict__ in_slope, float* __restrict__ out, int B) {
  constexpr int Ci = 16 * CT, Co = 16 * OTP * NP, CoP = Co, NG = 2 * CT;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = lds_all + wave * WAVE_LDS_W;
  float* r1 = lds + R1;
  float* r2 = lds + R2;
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  auto olane = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  const bool pre = in_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f;
  const int l16 = lane * 16;
  const BufRes wres = make_res(wfold, 2 * Ci * CoP * 4u);
  const BufRes bres = make_res(bias, CoP * 4u);
  const int nwaves = gridDim.x * 4;
  auto clip_res = [&](const float* base, int c, int rows) {
    const bool in_range = c < B;
#ifdef COSKAD_HOT   // timing-only: every stream from 64 L2-resident clips
    return make_res(base + (size_t)(in_range ? (c & 63) : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#else
    return make_res(base + (size_t)(in_range ? c : 0) * rows * TV, in_range ? rows * TV * 4u : 0u);
#endif
  };
  constexpr int QTAIL = 4 * (TV / 4) - 192;              // lanes of a quarter's 4th piece (12)
  const int l16t = lane < QTAIL ? l16 : 0x7ffffff0;
  float4 gb[16];
  auto qload = [&](const BufRes& res, int row0, int q) {
#pragma unroll
    for (int c = 0; c < 4; ++c) gb[4 * q + c] = buf_load4(res, c < 3 ? l16 : l16t, ((row0 + 4 * q) * (TV / 4) + 64 * c) * 16);
  };
  auto qstore = [&](int q, bool act) {                   // rows 4q .. 4q+3 of R2
    const int ln = olane();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ln + 64 * c;
      float4 v = gb[4 * q + c];
      if (act) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
      const int row = e / (TV / 4), col = 4 * (e - row * (TV / 4));
      if (c < 3 || lane < QTAIL) *reinterpret_cast<float4*>(r2 + (4 * q + row) * LDW + col) = v;
    }
  };
  int clip = blockIdx.x * 4 + wave;
  {
    const BufRes z0 = clip_res(Zg, clip, Ci);
#pragma unroll
    for (int q = 0; q < 4; ++q) qload(z0, 0, q);         // first pass, group 0
  }
  for (; clip < B; clip += nwaves) {
    const BufRes xres = clip_res(in, clip, Ci), zres = clip_res(Zg, clip, Ci), ores = clip_res(out, clip, Co);
    const BufRes znext = clip_res(Zg, clip + nwaves, Ci);
    // quarter q of group g: Z rows first (CT groups), then the layer input
    auto kq = [&](int g, int q) {
      if (g < CT) qload(zres, 16 * g, q);
      else qload(xres, 16 * (g - CT), q);
    };
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      Lane L = geo();
      const int jc = L.j < T ? L.j : T - 1;
      // group 0 (fetched a phase ago) goes to the window, group 1 takes its registers
#pragma unroll
      for (int q = 0; q < 4; ++q) { qstore(q, false); kq(1, q); }
      // the bias joins the finished sums (added first, a folded BatchNorm bias much larger than the result would cost every
      // partial sum an ulp of the BIAS: a per-channel offset, the kind of error tools/dbg_sens.py shows the model amplifies)
      f32x4 acc[NTILE][OTP], bq[OTP];
#pragma unroll
      for (int ot = 0; ot < OTP; ++ot) {
        const float4 b4 = buf_load4(bres, L.q * 16, (16 * (p * OTP + ot)) * 4);
        bq[ot] = f32x4{b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int t = 0; t < NTILE; ++t) acc[t][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int lq = (L.q * CoP + L.j) * 4;
      float wc[2][4][OTP];
      auto cload = [&](int buf, int g) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int ot = 0; ot < OTP; ++ot)
            wc[buf][s][ot] = buf_load1(wres, lq, ((16 * g + 4 * s) * CoP + 16 * (p * OTP + ot)) * 4);
      };
      cload(0, 0);
      float b[2][NTILE];
#pragma unroll
      for (int t = 0; t < NTILE; ++t) b[0][t] = r2[L.q * LDW + (t < T ? t * V + L.j : jc * V + 16)];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) cload((g + 1) & 1, g + 1);
        else {                                           // the ring registers are free: the next pass's / clip's group 0 takes off
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q) qload(p + 1 < NP ? zres : znext, 0, q);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s + 1 < 4 || g + 1 < NG) {
            const int sn = (s + 1) & 3;
#pragma unroll
            for (int t = 0; t < NTILE; ++t) b[(s + 1) & 1][t] = r2[(4 * sn + L.q) * LDW + (t < T ? t * V + L.j : jc * V + 16)];
          }
          if (g + 1 < NG) {
            qstore(s, g + 1 >= CT && pre);
            if (g + 2 < NG) kq(g + 2, s);
          }
#pragma unroll
          for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int ot = 0; ot < OTP; ++ot) acc[t][ot] = mfma(wc[g & 1][s][ot], b[s & 1][t], acc[t][ot]);
        }
      }
      // ---- the pass's 16 OTP output channels leave through the 32-row image, two tiles (32 channels) at a time --------------
#pragma unroll
      for (int h = 0; h < (OTP + 1) / 2; ++h) {
        const int nt = OTP - 2 * h < 2 ? OTP - 2 * h : 2;          // tiles in this flush (compile-time after unrolling)
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
          for (int o2 = 0; o2 < 2; ++o2)
            if (o2 < nt) tile_store(r1, 16 * o2, t < T ? t * V + L.j : jc * V + 16, t < T || L.j < T, acc[t][2 * h + o2] + bq[2 * h + o2], L);
        const int n4 = 16 * nt * (TV / 4);
        const int ch0 = 16 * (p * OTP + 2 * h);                    // first output channel of this flush
        const int ln = olane();
#pragma unroll
        for (int i = 0; i < (32 * (TV / 4) + 63) / 64; ++i) {
          if (64 * i < n4) {
            if (i % 4 == 0) __builtin_amdgcn_sched_barrier(0);
            const int e4 = ln + 64 * i;
            const int row = e4 / (TV / 4), col = 4 * (e4 - row * (TV / 4));
            const bool full = 64 * (i + 1) <= n4;
            const float* ptr = r1 + ((full || e4 < n4) ? row * LD + col : PADCOL);
            const float2 g0 = *reinterpret_cast<const float2*>(ptr), g1 = *reinterpret_cast<const float2*>(ptr + 2);
            // lanes beyond this flush's rows must not reach the next channels: out of range
            const int voff = (full || e4 < n4) ? l16 : 0x7ffffff0;
            buf_store4(ores, voff, (ch0 * (TV / 4) + 64 * i) * 16, float4{g0.x, g0.y, g1.x, g1.y});
          }
        }
      }
    }
  }
}

}  // namespace fa

bool layer_apply_ring_ok(int T_, int V_, int Ci, int Co) {
  return T_ == ff::T && V_ == ff::V && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

int launch_layer_apply_ring(const float* Z, const float* in, float* out, const float* wfold, const float* bias,
                            const float* in_slope, int B, int Ci, int Co, hipStream_t st) {
  const size_t lds = (size_t)4 * ff::WAVE_LDS_W * sizeof(float);
  const int nblk = (B + 3) / 4;
  const int grid = nblk < 256 ? nblk : 256;
#define LAUNCH_FA(CT, OTP, NP)                                                                                   \
  do {                                                                                                           \
    auto k = fa::k_layer_apply_ring<CT, OTP, NP>;                                                                \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, Z, wfold, bias, in_slope, out, B);                 \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_FA(1, 1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_FA(1, 2, 1);
    else if (Ci == 16 && Co == 64) LAUNCH_FA(1, 4, 1);
    else if (Ci == 32 && Co == 16) LAUNCH_FA(2, 1, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_FA(2, 2, 1);
    else if (Ci == 32 && Co == 64) LAUNCH_FA(2, 4, 1);
    else return fail(COSKAD_ERR_SHAPE, "apply_ring: unsupported channels (%d, %d)", Ci, Co);
  }
#undef LAUNCH_FA
  return check_launch("layer_apply_ring");
}

}  // namespace coskad
