// Micro-benchmark: what does merging a partial-row reduce launch and the fold launch behind it cost / save on MI355X?
//   A: k_red<<<NR>>> (column sums of a [P][E] float table -> E doubles)  then  k_fold<<<NF>>> (reads all E doubles, a little fp64 work)
//   B: ONE launch of NR + NF blocks: reduce blocks publish their sums (agent-scope release fence + atomicAdd on an arrival counter),
//      fold blocks (highest block ids: dispatched last) spin on the counter with a bounded loop, acquire, then fold
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/fence_cost.hip -o tools/micro/fence_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int P = 1024, E = 2112, COLS = 16, NR = E / COLS, NF = 8;

__device__ double colsum(const float* __restrict__ t, int e, double* sh) {
  const int col = threadIdx.x % COLS, slice = threadIdx.x / COLS;   // 64 slices of 16 rows
  double s = 0.0;
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = t[(size_t)(slice + 64 * u) * E + e];
#pragma unroll
  for (int u = 0; u < 16; ++u) s += (double)v[u];
  sh[threadIdx.x] = s;
  __syncthreads();
  double r = 0.0;
  if (slice == 0)
    for (int k = 0; k < 64; ++k) r += sh[col + COLS * k];
  return r;
}

__device__ void fold_work(const double* __restrict__ red, float* __restrict__ out, double* sh) {
  // stand-in for the fold: every block reads all sums into LDS, then each thread forms a short fp64 dot
  for (int e = threadIdx.x; e < E; e += blockDim.x) sh[e] = red[e];
  __syncthreads();
  double a = 0.0;
  for (int k = 0; k < 64; ++k) a += sh[(threadIdx.x + 33 * k) % E] * sh[(threadIdx.x * 7 + k) % E];
  out[blockIdx.x * 1024 + threadIdx.x] = (float)a;
}

__global__ __launch_bounds__(1024) void k_red(const float* __restrict__ t, double* __restrict__ red) {
  __shared__ double sh[1024];
  const int e = blockIdx.x * COLS + threadIdx.x % COLS;
  const double r = colsum(t, e, sh);
  if (threadIdx.x < COLS) red[e] = r;
}

__global__ __launch_bounds__(1024) void k_fold(const double* __restrict__ red, float* __restrict__ out) {
  __shared__ double sh[E];
  fold_work(red, out, sh);
}

__global__ __launch_bounds__(1024) void k_merged(const float* __restrict__ t, double* __restrict__ red, float* __restrict__ out,
                                                  unsigned* __restrict__ ctr, int* __restrict__ err) {
  __shared__ double sh[E];
  if (blockIdx.x < NR) {
    const int e = blockIdx.x * COLS + threadIdx.x % COLS;
    const double r = colsum(t, e, sh);
    if (threadIdx.x < COLS) red[e] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                   // release: the sums are visible device-wide before the arrival
      atomicAdd(&ctr[0], 1u);
    }
    return;
  }
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NR) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { err[0] = 1; break; }    // bounded: never hang the device
    }
    __threadfence();                                     // acquire
  }
  __syncthreads();
  fold_work(red, out + 0, sh);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(&ctr[1], 1u);
    if (done == NF - 1) { ctr[0] = 0; ctr[1] = 0; }      // the last fold block re-arms the counters for the next launch
  }
}

int main() {
  float* t; double* red; float* out; unsigned* ctr; int* err;
  hipMalloc(&t, (size_t)P * E * 4); hipMalloc(&red, E * 8); hipMalloc(&out, (NR + NF) * 1024 * 4); hipMalloc(&ctr, 8); hipMalloc(&err, 4);
  std::vector<float> h((size_t)P * E);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  hipMemcpy(t, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(ctr, 0, 8); hipMemset(err, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 300;
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_red, dim3(NR), dim3(1024), 0, 0, t, red); hipLaunchKernelGGL(k_fold, dim3(NF), dim3(1024), 0, 0, red, out); }
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) { hipLaunchKernelGGL(k_red, dim3(NR), dim3(1024), 0, 0, t, red); hipLaunchKernelGGL(k_fold, dim3(NF), dim3(1024), 0, 0, red, out); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float msA; hipEventElapsedTime(&msA, e0, e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_merged, dim3(NR + NF), dim3(1024), 0, 0, t, red, out, ctr, err);
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k_merged, dim3(NR + NF), dim3(1024), 0, 0, t, red, out, ctr, err);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float msB; hipEventElapsedTime(&msB, e0, e1);
    int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("two launches %.2f us   merged (fence + arrival counter + spin) %.2f us   spin timeout %d\n", msA * 1e3 / it, msB * 1e3 / it, herr);
  }
  // the merged launch computes the same fold output as the pair
  std::vector<float> a(1024), b(1024);
  hipLaunchKernelGGL(k_red, dim3(NR), dim3(1024), 0, 0, t, red); hipLaunchKernelGGL(k_fold, dim3(1), dim3(1024), 0, 0, red, out);
  hipMemcpy(a.data(), out, 4096, hipMemcpyDeviceToHost);
  hipMemset(red, 0, E * 8);
  hipLaunchKernelGGL(k_merged, dim3(NR + NF), dim3(1024), 0, 0, t, red, out, ctr, err);
  hipMemcpy(b.data(), out, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += a[i] != b[i];
  printf("mismatches %d\n", bad);
  return 0;
}
