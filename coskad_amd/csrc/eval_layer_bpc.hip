// One ST_GCNN layer with BatchNorm folded (eval mode, or any caller that hands over folded weights; reference
// models/graph_layers/stsgcn.py:56-80 the mixing, 94-116 the layer), 17 or 25 joints, ONE CLIP PER WORKGROUP OF FOUR WAVES, nothing
// but the layer's input and output in HBM:
//     U = Wz . gcn(X) + Wx . X + b          X = PReLU(in) (in_slope), optional PReLU on the way out (out_slope)
// k_layer_apply_m (stsgcn_fwd_mfma.hip, round 1) keeps the clip image, both mixing tables (44 KB at 25 joints) and the folded weights
// in LDS: one 16-wave block per CU, every phase behind a block-wide barrier (265 / 157 / 81 us per call at 32 -> 64 / 32 -> 16 /
// 16 -> 32, B = 4096).  Here, as in fwd_moments_bpc.hip, the mixes are dealt by joint and by frame so that a wave's operands of both
// are the same for every clip and live in 63 registers; ONE image: the residual convolution Wx . X runs first, straight from the staged
// rows (B operands read from the image, the folded weights as A operands held in registers for the launch; a wave owns output tiles x
// position tiles), its sums wait in registers while the image is mixed in place into Z, then Wz . Z is added; the sums leave through the
// image in full lines.  The next clip's rows travel in registers.  (Measured against this: layers 3 + 4 in ONE kernel with the
// 32-channel activation between them kept on chip -- 262 us against 72 + 182 for the two launches: these kernels wait on their own phase
// chain at two waves per SIMD, not on HBM, so the saved round trip buys nothing.)  LDS: 32 rows (38.6 KB; 64 output channels: 64 rows).
#include "fused_ops.h"

namespace coskad {
namespace ev {

using ff::f32x4;
using ff::Lane;
using ff::mfma;
using ff::prelu;

// the FIRST layer (2 -> 32 channels, folded) in front of a 32-input layer, as the FIRST form's parameters: its mixing parameters, folded
// weights [4][32] (rows Z0 Z1 X0 X1) and bias [32]; the PReLU between the layers is the main kernel's `in_slope`
struct FirstLayer {
  const float* A;
  const float* T;
  const float* wfold;
  const float* bias;
};

// CT: 16-row groups of the input; OT: 16-channel output tiles; FIRST (CT = 2): `in` is the NETWORK input [B, 2, T, V] and the 32-channel
// input of this layer is the first layer's output, formed per clip on the VALU (two rows: 87 multiply-adds per thread for both mixes, 150
// for the 4 -> 32 convolution) straight into the image -- the first layer's kernel, its 32-row write and this kernel's 32-row read gone
template <int V, int CT, int OT, bool FIRST = false>
__global__ __launch_bounds__(256, (OT == 4 || (CT == 2 && OT == 2) ? 2 : 3)) void k_eval_layer_bpc(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ Aw, const float* __restrict__ Tw,
                                                          const float* __restrict__ wfold, const float* __restrict__ bias,
                                                          const float* __restrict__ in_slope, const float* __restrict__ out_slope,
                                                          int B, FirstLayer fl) {
  static_assert(!FIRST || CT == 2, "the first layer has 32 output channels");
  constexpr int T = 12, TV = T * V, LD = TV + 2, R4 = TV / 4, Ci = 16 * CT, Co = 16 * OT, CoP = Co;
  static_assert(TV % 4 == 0, "rows are staged as float4");
  constexpr int N4 = (FIRST ? 2 : Ci) * R4, XL = (N4 + 255) / 256;
  constexpr int NTV = (V + 15) / 16, KV = (V + 3) / 4, MAXF = T / 4, MAXJ = (V + 3) / 4;
  constexpr int NT = (TV + 15) / 16;                     // position tiles
  // a wave's share of the output: 64 channels: its own tile x all position tiles; 32: tile wave & 1 x half of them; 16: a quarter
  constexpr int MAXT = OT == 4 ? NT : (OT == 2 ? (NT + 1) / 2 : (NT + 3) / 4);
  constexpr int KS = Ci / 4;                             // k-steps of each of the two products
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* imz = lds;                  // X -> Y -> Z (mixed in place); then the flush
  __shared__ float fx[FIRST ? 2 * LD : 1], fy[FIRST ? 2 * LD : 1], fz[FIRST ? 2 * LD : 1], fw[FIRST ? 160 : 1];   // (FIRST) x, its mixes, W1 | b1
  if constexpr (FIRST) {
    if (threadIdx.x < 160) fw[threadIdx.x] = threadIdx.x < 128 ? fl.wfold[threadIdx.x] : fl.bias[threadIdx.x - 128];
  }
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  auto geo = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return Lane{l & 15, l >> 4};
  };
  auto tid_now = [&]() {
    int t = tid0;
    asm volatile("" : "+v"(t));
    return t;
  };
  Lane L = geo();
  const bool pre = in_slope != nullptr, post = out_slope != nullptr;
  const float a_in = pre ? in_slope[0] : 0.f, a_out = post ? out_slope[0] : 0.f;
  // a wave's joints and frames are the same for every clip: its B operands of both mixes stay in registers
  //   temporal  Y[q,v] = sum_t X[t,v] T[v][t][q]:   B[k = t][j = q];   spatial  Z[t,w] = sum_v Y[t,v] A[t][v][w]:   B[k = v][j = w]
  float tbv[MAXJ][3], bbv[MAXF][NTV][KV];
#pragma unroll
  for (int k = 0; k < MAXJ; ++k) {
    const int v = wave + 4 * k;
#pragma unroll
    for (int s = 0; s < 3; ++s) tbv[k][s] = (v < V && L.j < T) ? Tw[(v * T + 4 * s + L.q) * T + L.j] : 0.f;
  }
#pragma unroll
  for (int tt = 0; tt < MAXF; ++tt) {
    const int t = wave + 4 * tt;
#pragma unroll
    for (int c = 0; c < NTV; ++c)
#pragma unroll
      for (int s = 0; s < KV; ++s)
        bbv[tt][c][s] = (16 * c + L.j < V && 4 * s + L.q < V) ? Aw[(t * V + 4 * s + L.q) * V + 16 * c + L.j] : 0.f;
  }
  const int ot = OT == 4 ? wave : (OT == 2 ? (wave & 1) : 0);
  const int t0 = OT == 4 ? 0 : (OT == 2 ? (wave >> 1) * MAXT : wave * MAXT);
  const int nt = NT - t0 < MAXT ? NT - t0 : MAXT;
  // the folded weights of this wave's output tile, for the launch: A[i = o][k = 4 s + q] = wfold[k][16 ot + o]; rows [0, Ci) act on Z,
  // rows [Ci, 2 Ci) on X
  float wz[KS], wx[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    wz[s] = wfold[(4 * s + L.q) * CoP + 16 * ot + L.j];
    wx[s] = wfold[(Ci + 4 * s + L.q) * CoP + 16 * ot + L.j];
  }
  float4 px[XL];
  auto xload = [&](int clip) {
    const int tid = tid_now();
    const float4* g4 = reinterpret_cast<const float4*>(in + (size_t)(clip < B ? clip : 0) * (FIRST ? 2 : Ci) * TV);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + 256 * i;
      px[i] = (e < N4 && clip < B) ? g4[e] : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int clip = blockIdx.x;
  xload(clip);
  for (; clip < B; clip += gridDim.x) {
    __syncthreads();                                     // the previous clip's flush is done with the images
    if constexpr (FIRST) {
      // ---- the first layer on the VALU: x -> Y0 (temporal) -> Z0 (spatial) -> X = PReLU(W1 [Z0; x] + b1) into the image ---------------
      const int tid = tid_now();
      if (tid < N4) {
        const int row = tid / R4, col = 4 * (tid - row * R4);
        const float4 v = px[0];
        fx[row * LD + col] = v.x; fx[row * LD + col + 1] = v.y; fx[row * LD + col + 2] = v.z; fx[row * LD + col + 3] = v.w;
      }
      xload(clip + gridDim.x);
      // (the table pointers through an optimisation barrier per clip: a thread's 37 table values are L1 hits, not 37 held registers)
      const float* A1 = fl.A;
      const float* T1 = fl.T;
      asm volatile("" : "+s"(A1), "+s"(T1));
      __syncthreads();
      for (int idx = tid; idx < TV; idx += 256) {        // idx = v * 12 + q:  Y0[r][q, v] = sum_t x[r][t, v] T1[v][t][q]
        const int v = idx / T, q = idx - v * T;
        float y0 = 0.f, y1 = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const float w = T1[(v * T + t) * T + q];
          y0 = fmaf(fx[t * V + v], w, y0);
          y1 = fmaf(fx[LD + t * V + v], w, y1);
        }
        fy[q * V + v] = y0;
        fy[LD + q * V + v] = y1;
      }
      __syncthreads();
      for (int idx = tid; idx < TV; idx += 256) {        // idx = t * V + w:   Z0[r][t, w] = sum_v Y0[r][t, v] A1[t][v][w]
        const int t = idx / V, w = idx - t * V;
        float z0 = 0.f, z1 = 0.f;
#pragma unroll 5
        for (int v = 0; v < V; ++v) {
          const float a = A1[(t * V + v) * V + w];
          z0 = fmaf(fy[t * V + v], a, z0);
          z1 = fmaf(fy[LD + t * V + v], a, z1);
        }
        fz[idx] = z0;
        fz[LD + idx] = z1;
      }
      __syncthreads();
      // thread <-> (channel, four positions): full 16-byte rows of the image
      for (int e = tid; e < 32 * R4; e += 256) {
        const int o = e / R4, col = 4 * (e - o * R4);
        const float w0 = fw[o], w1 = fw[32 + o], w2 = fw[64 + o], w3 = fw[96 + o], bb = fw[128 + o];
        float u[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // (bias added to the finished sum, as the first layer's own kernels do)
          float s = w0 * fz[col + c];
          s = fmaf(w1, fz[LD + col + c], s);
          s = fmaf(w2, fx[col + c], s);
          s = fmaf(w3, fx[LD + col + c], s);
          s += bb;
          u[c] = pre ? prelu(s, a_in) : s;
        }
        *reinterpret_cast<float2*>(imz + o * LD + col) = float2{u[0], u[1]};
        *reinterpret_cast<float2*>(imz + o * LD + col + 2) = float2{u[2], u[3]};
      }
    } else {
      const int tid = tid_now();
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        const int e = tid + 256 * i;
        if (e < N4) {
          const int row = e / R4, col = 4 * (e - row * R4);
          float4 v = px[i];
          if (pre) { v.x = prelu(v.x, a_in); v.y = prelu(v.y, a_in); v.z = prelu(v.z, a_in); v.w = prelu(v.w, a_in); }
          *reinterpret_cast<float2*>(imz + row * LD + col) = float2{v.x, v.y};
          *reinterpret_cast<float2*>(imz + row * LD + col + 2) = float2{v.z, v.w};
        }
      }
    }
    if constexpr (OT < 4 && !FIRST) xload(clip + gridDim.x);       // the next clip's rows: a whole clip of products to arrive
    __syncthreads();                                     // the image holds X
    L = geo();
    // ---- residual convolution: U tile = sum_k Wx[k][o] X[k][p], in registers while the image is mixed --------------------------------
    f32x4 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto product = [&](const float (&w)[KS]) {
      const float* bp = lds + L.q * LD + L.j;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int p = 16 * (t0 + (t < nt ? t : 0)) + L.j;          // (beyond the clip: clamped -- those columns are never stored)
          acc[t] = mfma(w[s], bp[4 * s * LD + (p < TV ? p : TV - 1) - L.j], acc[t]);
        }
      }
    };
    product(wx);
    __syncthreads();                                     // every wave has read X
    // ---- Y = temporal mix, in place: joints v = wave, wave + 4, .. ---------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < MAXJ; ++k) {
      const int v = wave + 4 * k;
      if (v < V) {
#pragma unroll
        for (int rt = 0; rt < CT; ++rt) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 3; ++s) d = mfma(imz[(16 * rt + L.j) * LD + (4 * s + L.q) * V + v], tbv[k][s], d);
          if (L.j < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) imz[(16 * rt + 4 * L.q + r) * LD + L.j * V + v] = d[r];
          }
        }
      }
    }
    __syncthreads();
    // ---- Z = spatial mix, in place: frames t = wave, wave + 4, wave + 8 ------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < MAXF; ++tt) {
      const int t = wave + 4 * tt;
#pragma unroll
      for (int rt = 0; rt < CT; ++rt) {
        float a[KV];
#pragma unroll
        for (int s = 0; s < KV; ++s) a[s] = 4 * s + L.q < V ? imz[(16 * rt + L.j) * LD + t * V + 4 * s + L.q] : 0.f;
        f32x4 d[NTV];
#pragma unroll
        for (int c = 0; c < NTV; ++c) {
          d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KV; ++s) d[c] = mfma(a[s], bbv[tt][c][s], d[c]);
        }
#pragma unroll
        for (int c = 0; c < NTV; ++c)
          if (16 * c + L.j < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) imz[(16 * rt + 4 * L.q + r) * LD + t * V + 16 * c + L.j] = d[c][r];
          }
      }
    }
    __syncthreads();                                     // the image holds Z
    L = geo();
    if constexpr (OT == 4 && !FIRST) xload(clip + gridDim.x);      // (64 output channels: no registers for them through the mixing phases)
    product(wz);
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * ot + 4 * L.q);
    const f32x4 bq = {b4.x, b4.y, b4.z, b4.w};
    // ---- flush through the image (64 output channels: 64 rows of LDS, every wave at once), full lines to HBM --------------------------
    constexpr int FR = Co < 32 ? Co : (OT == 4 ? 64 : 32);  // rows per round
    constexpr int NR = Co / FR;
#pragma unroll
    for (int rnd = 0; rnd < NR; ++rnd) {
      __syncthreads();                                   // the product's readers / the previous round's rows are done with the image
      {
        const int row0 = 16 * ot;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int p = 16 * (t0 + t) + L.j;
          if (t < nt) {
            f32x4 v = acc[t] + bq;
            if (post) { v[0] = prelu(v[0], a_out); v[1] = prelu(v[1], a_out); v[2] = prelu(v[2], a_out); v[3] = prelu(v[3], a_out); }
            float* dst = lds + (row0 + 4 * L.q) * LD + (p < TV ? p : TV);
            dst[0] = v[0]; dst[LD] = v[1]; dst[2 * LD] = v[2]; dst[3 * LD] = v[3];
          }
        }
      }
      __syncthreads();
      constexpr int n4 = FR * R4;
      const int tid = tid_now();
      float4* g4 = reinterpret_cast<float4*>(out + ((size_t)clip * Co + FR * rnd) * TV);
#pragma unroll
      for (int i = 0; i < (n4 + 255) / 256; ++i) {
        const int e4 = tid + 256 * i;
        if (e4 < n4) {
          const int row = e4 / R4, col = 4 * (e4 - row * R4);
          const float2 g0 = *reinterpret_cast<const float2*>(lds + row * LD + col);
          const float2 g1 = *reinterpret_cast<const float2*>(lds + row * LD + col + 2);
          g4[e4] = float4{g0.x, g0.y, g1.x, g1.y};
        }
      }
    }
  }
}

}  // namespace ev

bool eval_layer_bpc_ok(int T_, int V_, int Ci, int Co) {
  return T_ == 12 && (V_ == 25 || V_ == 17) && (Ci == 16 || Ci == 32) && (Co == 16 || Co == 32 || Co == 64);
}

template <int V>
static int launch_eval_layer_v(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold, const float* bias,
                               const float* in_slope, const float* out_slope, int B, int Ci, int Co, hipStream_t st) {
  const size_t lds = (size_t)(Co == 64 ? 64 : 32) * (12 * V + 2) * sizeof(float);    // (the flush needs min(C_out, 64) rows)
  const int per_cu = (Co == 64 || (Ci == 32 && Co == 32)) ? 2 : 3;
  const int grid = B < 256 * per_cu ? B : 256 * per_cu;
#define LAUNCH_EV(CT, OT)                                                                                                 \
  do {                                                                                                                    \
    auto k = ev::k_eval_layer_bpc<V, CT, OT>;                                                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, in, out, Aw, Tw, wfold, bias, in_slope, out_slope, B, ev::FirstLayer{}); \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, Ci, Co, st);
    if (Ci == 16 && Co == 16) LAUNCH_EV(1, 1);
    else if (Ci == 16 && Co == 32) LAUNCH_EV(1, 2);
    else if (Ci == 16 && Co == 64) LAUNCH_EV(1, 4);
    else if (Ci == 32 && Co == 16) LAUNCH_EV(2, 1);
    else if (Ci == 32 && Co == 32) LAUNCH_EV(2, 2);
    else LAUNCH_EV(2, 4);
  }
#undef LAUNCH_EV
  return check_launch("eval_layer_bpc");
}

bool eval_first_pair_ok(int T_, int V_, int Ci, int Cm, int Co) {
  return T_ == 12 && (V_ == 25 || V_ == 17) && Ci == 2 && Cm == 32 && (Co == 16 || Co == 32 || Co == 64);
}

template <int V>
static int launch_eval_first_pair_v(const float* x, float* out, const ev::FirstLayer& fl, const float* Aw, const float* Tw,
                                    const float* wfold, const float* bias, const float* mid_slope, const float* out_slope, int B, int Co,
                                    hipStream_t st) {
  const size_t lds = (size_t)(Co == 64 ? 64 : 32) * (12 * V + 2) * sizeof(float);
  const int per_cu = (Co == 64 || Co == 32) ? 2 : 3;
  const int grid = B < 256 * per_cu ? B : 256 * per_cu;
#define LAUNCH_EF(OT)                                                                                                     \
  do {                                                                                                                    \
    auto k = ev::k_eval_layer_bpc<V, 2, OT, true>;                                                                        \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, x, out, Aw, Tw, wfold, bias, mid_slope, out_slope, B, fl);      \
  } while (0)
  {
    ProbeScope probe(KID_LAYER_APPLY, 2, Co, st);
    if (Co == 16) LAUNCH_EF(1);
    else if (Co == 32) LAUNCH_EF(2);
    else LAUNCH_EF(4);
  }
#undef LAUNCH_EF
  return check_launch("eval_first_pair");
}

int launch_eval_layer_bpc(const float* in, float* out, const float* Aw, const float* Tw, const float* wfold, const float* bias,
                          const float* in_slope, const float* out_slope, int B, int Ci, int Co, int T_, int V_, hipStream_t st) {
  if (!eval_layer_bpc_ok(T_, V_, Ci, Co))
    return fail(COSKAD_ERR_SHAPE, "eval_layer_bpc: built for 12 x 17 / 25, 16 / 32 -> 16 / 32 / 64 channels");
  if (V_ == 17) return launch_eval_layer_v<17>(in, out, Aw, Tw, wfold, bias, in_slope, out_slope, B, Ci, Co, st);
  return launch_eval_layer_v<25>(in, out, Aw, Tw, wfold, bias, in_slope, out_slope, B, Ci, Co, st);
}

extern "C" {

/* 1: coskad_layer_first_pair_apply_f32 takes the first layer (Ci = 2 -> Cm) with the layer behind it (Cm -> Co) on this layout */
int coskad_layer_first_pair_ok(int T, int V, int Ci, int Cm, int Co) { return eval_first_pair_ok(T, V, Ci, Cm, Co) ? 1 : 0; }

/* The first two ST_GCNN layers of the encoder with folded BatchNorm (models/common/components.py:94-105 -> models/graph_layers/
 * stsgcn.py:94-116 twice, eval mode) in ONE pass: out [B, Co, T, V] = layer2(PReLU_mid(layer1(x))), x [B, 2, T, V] the network input; the
 * 32-channel activation between them never reaches HBM.  wfold1 [4, 32] / bias1 [32], wfold2 [64, Co] / bias2 from coskad_bn_fold_f32;
 * out_slope may be NULL (pre-activation output). */
int coskad_layer_first_pair_apply_f32(const float* x, float* out, const float* A1, const float* T1, const float* wfold1, const float* bias1,
                                      const float* A2, const float* T2, const float* wfold2, const float* bias2, const float* mid_slope,
                                      const float* out_slope, int B, int Cm, int Co, int T, int V, hipStream_t stream) {
  if (!x || !out || !A1 || !T1 || !wfold1 || !bias1 || !A2 || !T2 || !wfold2 || !bias2 || !mid_slope)
    return fail(COSKAD_ERR_ARG, "layer_first_pair_apply: null pointer");
  if (B <= 0 || !eval_first_pair_ok(T, V, 2, Cm, Co))
    return fail(COSKAD_ERR_SHAPE, "layer_first_pair_apply: built for 12 x 17 / 25, 2 -> 32 -> 16 / 32 / 64 channels");
  const ev::FirstLayer fl{A1, T1, wfold1, bias1};
  if (V == 17) return launch_eval_first_pair_v<17>(x, out, fl, A2, T2, wfold2, bias2, mid_slope, out_slope, B, Co, stream);
  return launch_eval_first_pair_v<25>(x, out, fl, A2, T2, wfold2, bias2, mid_slope, out_slope, B, Co, stream);
}

}  // extern "C"

}  // namespace coskad
