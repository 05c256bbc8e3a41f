// Shared host/device helpers for the COSKAD gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#define COSKAD_ABI_VERSION 1

// error codes returned through the C ABI (0 = ok)
#define COSKAD_OK 0
#define COSKAD_ERR_ARG (-1)      // bad argument (null pointer, non-positive size)
#define COSKAD_ERR_SHAPE (-2)    // unsupported (T,V) / channel count
#define COSKAD_ERR_LAUNCH (-3)   // HIP launch failure
#define COSKAD_ERR_WORKSPACE (-4)// workspace too small

namespace coskad {

// thread-local last-error text, read through coskad_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

#ifndef COSKAD_BLOCK
#define COSKAD_BLOCK 512
#endif
constexpr int kFlatBlock = COSKAD_BLOCK;   // threads per block of the element-wise kernels (heads.hip)
#ifndef COSKAD_MINWAVES
#define COSKAD_MINWAVES 4
#endif
constexpr int kMinWaves = COSKAD_MINWAVES;   // waves per SIMD the big tile kernels are compiled for (4 = two 512-thread blocks per CU)
constexpr int kMaxLdsBytes = 160 * 1024;
constexpr float kBnEps = 1e-5f;      // nn.BatchNorm2d default (reference stsgcn.py:65,76)

template <int T, int V>
struct Geo {
  static constexpr int TV = T * V;
  // LDS row stride in floats: odd, so that lanes<->rows (stride LD) and lanes<->positions
  // (stride 1) are both bank-conflict-free for ds_read_b32/ds_write_b32 (32 banks).
  static constexpr int LD = (TV % 2 == 0) ? TV + 1 : TV;
  // Threads per block of the LDS-tile kernels (block per tile of clips).  A clip of more than 256 positions (25 joints) has more
  // than eight 32-position strips and an LDS footprint (image + mixing tables) that leaves room for ONE block per CU: that block
  // is then 16 waves, so that the CU still runs four waves per SIMD and every strip has its own wave (measured, round 3: the
  // 25-joint encoder step 4.36 -> 3.31 ms; at 17 joints two 8-wave blocks per CU are faster: 1.67 vs 1.71 ms).
  static constexpr int Block = TV > 256 ? 2 * COSKAD_BLOCK : COSKAD_BLOCK;
  static constexpr int Scratch = (Block / 64) * 256;   // floats of LDS scratch of the cross-wave reductions (tile_ops.h)
};

// kernel ids of the timing probe (api.hip)
enum { KID_LAYER_APPLY = 1, KID_BWD_DATA = 2, KID_BWD_REDUCE = 3, KID_FWD_MOMENTS = 4, KID_GCN_PARAMS = 5,
       KID_LAYER_BWD = 6 /* every launch of one coskad_layer_bwd*_f32 call together */, KID_FUSED_FWD = 7,
       KID_BTLNK_BWD = 8 /* the bottleneck backward's main kernel (Ci = channels below or 0, Co = latent) */ };
struct ProbeScope {   // brackets ONE kernel launch with events when the probe is armed for it
  ProbeScope(int kernel, int ci, int co, hipStream_t st);
  ~ProbeScope();
  hipStream_t st_;
  bool armed_ = false;
};

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

__device__ __forceinline__ float prelu_f(float x, float a) { return x > 0.f ? x : a * x; }

// threadIdx.x behind an optimisation barrier: address arithmetic derived from it is recomputed where it is used
// instead of being hoisted out of the persistent tile loop and held in VGPRs across every phase.
__device__ __forceinline__ int tid_here() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// Block barrier for phases that exchange data through LDS only.  __syncthreads() also drains the vector-memory
// counter (a workgroup-scope release covers global memory), so every barrier after a store phase (unstage, epilogue)
// or with prefetched global loads in flight would wait for HBM round trips.  Here only the LDS/scalar counter is
// drained: the compiler still waits for a global load where its value is used, and no tile kernel passes data
// between waves through global memory.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wave-uniform value -> SGPR (lets hipcc use s_load for everything indexed by it)
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }

// Column `e` of P partial rows `stride` floats apart, summed in fp64 in a fixed order by a 1024-thread block of COLS columns:
// 1024 / COLS row slices, eight loads of a slice in flight, then the slices one after another.  The sum is returned to the
// threads of slice 0 (threadIdx.x < COLS); `sh`: 1024 doubles.  Partial-row sums are latency kernels: their time is the number
// of dependent load rounds, so many small blocks beat few wide ones (64 columns x 16 slices: 16 us for 768 rows of 32 KB, this
// at 32 columns: 8 us).
template <int COLS>
__device__ __forceinline__ double column_sum_f64(const float* __restrict__ rows, int P, size_t stride, int e, bool ok, double* sh) {
  constexpr int NSL = 1024 / COLS;
  const int slice = threadIdx.x / COLS;
  double s = 0.0;
  if (ok) {
    const float* base = rows + e;
    int p = slice;
    for (; p + 7 * NSL < P; p += 8 * NSL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(p + u * NSL) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; p < P; p += NSL) s += (double)base[(size_t)p * stride];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  double t = 0.0;
  if (slice == 0) {
#pragma unroll 8
    for (int k = 0; k < NSL; ++k) t += sh[threadIdx.x + COLS * k];
  }
  return t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// (T,V) geometries the kernels are instantiated for.  17 = COCO joints (default),
// 25 = NTU layout (BASELINE config 4), 14 = headless, 18 = kp18 (wrappers, staticCenter.py:70-75).
#define COSKAD_DISPATCH_TV(T_, V_, CALL)                                   \
  do {                                                                      \
    if ((T_) == 12 && (V_) == 17) { CALL(12, 17); }                         \
    else if ((T_) == 12 && (V_) == 25) { CALL(12, 25); }                    \
    else if ((T_) == 12 && (V_) == 14) { CALL(12, 14); }                    \
    else if ((T_) == 12 && (V_) == 18) { CALL(12, 18); }                    \
    else return coskad::fail(COSKAD_ERR_SHAPE,                              \
        "unsupported (n_frames=%d, n_joints=%d): built for T=12, V in {14,17,18,25}", (T_), (V_)); \
  } while (0)

}  // namespace coskad
