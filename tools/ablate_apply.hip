// Ablation harness for k_layer_apply (scratch tool, not part of the product).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I coskad_amd/csrc tools/ablate_apply.hip coskad_amd/csrc/api.hip -o tools/ablate_apply
#include "tile_ops.h"
#include <vector>
#include <cstdlib>
using namespace coskad;

// MODE bits: 1 = skip temporal, 2 = skip spatial, 4 = skip conv FMAs (copy), 8 = skip staging, 16 = x from LDS copy (no global x reload)
template <int T, int V, int CB, int MODE>
__global__ __launch_bounds__(kBlock) void k_apply(
    const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Aw,
    const float* __restrict__ Tw, const float* __restrict__ wfold, const float* __restrict__ bias,
    int B, int Ci, int Co, int CoP, int NB) {
  constexpr int TV = Geo<T, V>::TV, LD = Geo<T, V>::LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int clip0 = blockIdx.x * NB;
  const int nb = min(NB, B - clip0);
  const int rows = nb * Ci;
  const float* gin = in + (size_t)clip0 * Ci * TV;
  if (!(MODE & 8)) stage_rows<T, V>(gin, lds, rows * TV, false, 0.f);
  __syncthreads();
  if (!(MODE & 1)) mix_rows<T, V, true, false>(lds, rows, Tw);
  __syncthreads();
  if (!(MODE & 2)) mix_rows<T, V, false, false>(lds, rows, Aw);
  __syncthreads();
  const int P = nb * TV;
  const int rounds = ceil_div(P, kBlock);
  for (int r = 0; r < rounds; ++r) {
    const int pos = r * kBlock + threadIdx.x;
    const bool act = pos < P;
    const int pc = act ? pos : 0;
    const int n = pc / TV;
    const int p = pc - n * TV;
    const float* zrow = lds + (n * Ci) * LD + p;
    const float* xg = gin + (size_t)n * Ci * TV + p;
    float* og = out + ((size_t)(clip0 + n) * Co) * TV + p;
    for (int o0 = 0; o0 < CoP; o0 += 16) {
      float acc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = bias[o0 + j];
      if (!(MODE & 4)) {
        for (int c0 = 0; c0 < Ci; c0 += CB) {
          float zr[CB], xr[CB];
#pragma unroll
          for (int k = 0; k < CB; ++k) {
            zr[k] = zrow[(c0 + k) * LD];
            xr[k] = (MODE & 16) ? zrow[(c0 + k) * LD] : xg[(c0 + k) * TV];
          }
#pragma unroll
          for (int k = 0; k < CB; ++k) {
            const float* wz = wfold + (c0 + k) * CoP + o0;
            const float* wx = wfold + (Ci + c0 + k) * CoP + o0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              acc[j] = fmaf(wz[j], zr[k], acc[j]);
              acc[j] = fmaf(wx[j], xr[k], acc[j]);
            }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += zrow[((o0 + j) % Ci) * LD];
      }
      if (act) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (o0 + j < Co) og[(o0 + j) * TV] = acc[j];
      }
    }
  }
}

template <int MODE>
float run(const float* in, float* out, const float* A, const float* Tm, const float* wf, const float* b, int B, int Ci, int Co, int NB) {
  constexpr int LD = Geo<12, 17>::LD;
  size_t lds = (size_t)NB * Ci * LD * 4;
  int grid = (B + NB - 1) / NB;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_apply<12, 17, 8, MODE>), dim3(grid), dim3(kBlock), lds, 0, in, out, A, Tm, wf, b, B, Ci, Co, (Co + 15) / 16 * 16, NB);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_apply<12, 17, 8, MODE>), dim3(grid), dim3(kBlock), lds, 0, in, out, A, Tm, wf, b, B, Ci, Co, (Co + 15) / 16 * 16, NB);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10 * 1000;
}

int main(int argc, char** argv) {
  int B = 4096, Ci = argc > 1 ? atoi(argv[1]) : 32, Co = argc > 2 ? atoi(argv[2]) : 64, NB = argc > 3 ? atoi(argv[3]) : 64 / Ci;
  const int TV = 204;
  std::vector<float> h((size_t)B * Ci * TV);
  for (auto& v : h) v = (rand() % 2001 - 1000) / 1000.f;
  float *in, *out, *A, *Tm, *wf, *b;
  hipMalloc(&in, h.size() * 4); hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, (size_t)B * Co * TV * 4);
  std::vector<float> w(12 * 17 * 17 + 17 * 144 + 2 * Ci * 64 + 64);
  for (auto& v : w) v = (rand() % 2001 - 1000) / 5000.f;
  hipMalloc(&A, 12 * 17 * 17 * 4); hipMemcpy(A, w.data(), 12 * 17 * 17 * 4, hipMemcpyHostToDevice);
  hipMalloc(&Tm, 17 * 144 * 4); hipMemcpy(Tm, w.data() + 3468, 17 * 144 * 4, hipMemcpyHostToDevice);
  hipMalloc(&wf, 2 * Ci * 64 * 4); hipMemcpy(wf, w.data() + 3468 + 2448, 2 * Ci * 64 * 4, hipMemcpyHostToDevice);
  hipMalloc(&b, 64 * 4); hipMemcpy(b, w.data(), 64 * 4, hipMemcpyHostToDevice);
  printf("B=%d Ci=%d Co=%d NB=%d\n", B, Ci, Co, NB);
  printf("full              %8.1f us\n", run<0>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("no temporal       %8.1f us\n", run<1>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("no spatial        %8.1f us\n", run<2>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("no gcn            %8.1f us\n", run<3>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("no conv           %8.1f us\n", run<4>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("no gcn no conv    %8.1f us\n", run<7>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("only conv (no stage/gcn) %8.1f us\n", run<11>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("conv x-from-lds   %8.1f us\n", run<16>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  printf("only conv x-lds   %8.1f us\n", run<27>(in, out, A, Tm, wf, b, B, Ci, Co, NB));
  return 0;
}
