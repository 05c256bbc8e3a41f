// Batch formation on the device: the window table [N, 2, T*V] stays resident in HBM; a batch is
// one gather + affine-transform launch.  Replaces the reference's per-item numpy path
// utils/dataset.py:65-77 (`sample = index % N`, `trans = index // N`, `transform_list[trans](data)[:num_coords]`) with
// utils/dataset_utils.py:272-286 (`einsum('ktv,ck->ctv', (x, y, 1), M)`).
#include "common.h"

namespace coskad {

// one thread = 4 consecutive positions of one batch item (TV % 4 == 0) or 1 position (generic)
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_transform(const float* __restrict__ xy, const long long* __restrict__ index,
                                                         const float* __restrict__ mats, float* __restrict__ out, int B,
                                                         int N, int ntrans, int TV) {
  const int per = TV / VEC;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * per) return;
  const int b = (int)(e / per), q = (int)(e - (long long)b * per) * VEC;
  const long long id = index[b];
  const long long t = id / N, s = id - t * N;
  float* o0 = out + ((size_t)b * 2) * TV + q;
  float* o1 = o0 + TV;
  if (id < 0 || t >= ntrans) {   // out-of-range index: a zero clip rather than an out-of-bounds read
#pragma unroll
    for (int u = 0; u < VEC; ++u) { o0[u] = 0.f; o1[u] = 0.f; }
    return;
  }
  const float* m = mats + t * 9;
  const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5];
  const float* px = xy + ((size_t)s * 2) * TV + q;
  const float* py = px + TV;
  if constexpr (VEC == 4) {
    const float4 x = *reinterpret_cast<const float4*>(px), y = *reinterpret_cast<const float4*>(py);
    // (x*m0 + y*m1) + 1*m2, in the einsum's summation order (no FMA contraction: -ffp-contract=off)
    *reinterpret_cast<float4*>(o0) = float4{(x.x * m00 + y.x * m01) + m02, (x.y * m00 + y.y * m01) + m02,
                                            (x.z * m00 + y.z * m01) + m02, (x.w * m00 + y.w * m01) + m02};
    *reinterpret_cast<float4*>(o1) = float4{(x.x * m10 + y.x * m11) + m12, (x.y * m10 + y.y * m11) + m12,
                                            (x.z * m10 + y.z * m11) + m12, (x.w * m10 + y.w * m11) + m12};
  } else {
    const float x = px[0], y = py[0];
    o0[0] = (x * m00 + y * m01) + m02;
    o1[0] = (x * m10 + y * m11) + m12;
  }
}

}  // namespace coskad

using namespace coskad;

extern "C" {

/* out[b, c, p] = M[t][c][0] x[s, p] + M[t][c][1] y[s, p] + M[t][c][2],  s = index[b] % N, t = index[b] / N, c < 2. */
int coskad_gather_transform_f32(const float* xy, const long long* index, const float* mats, float* out, int B, int N,
                                int ntrans, int TV, hipStream_t stream) {
  if (!xy || !index || !mats || !out) return fail(COSKAD_ERR_ARG, "gather_transform: null pointer");
  if (B <= 0 || N <= 0 || ntrans <= 0 || TV <= 0) return fail(COSKAD_ERR_ARG, "gather_transform: B=%d N=%d ntrans=%d TV=%d", B, N, ntrans, TV);
  if (TV % 4 == 0) {
    const long long n = (long long)B * (TV / 4);
    hipLaunchKernelGGL(k_gather_transform<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xy, index, mats, out, B, N, ntrans, TV);
  } else {
    const long long n = (long long)B * TV;
    hipLaunchKernelGGL(k_gather_transform<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, xy, index, mats, out, B, N, ntrans, TV);
  }
  return check_launch("gather_transform");
}

}  // extern "C"
