// chamfer.hip -- fused per-cloud reduction of the K=1 nearest-neighbour distances.
//
// Fuses the tail of _chamfer_distance_single_direction (reference:
// functions/chamfer.py:135-185) for point_reduction in {"sum","mean"}:
//   mask rows i >= lengths[n]  ->  sum over points  ->  * weights[n]
//   ->  / max(lengths[n], 1) ("mean")
// into one launch with one workgroup per cloud (the reference runs ~6 elementwise /
// reduction torch kernels and one host sync here).  Fixed-shape tree reduction, so
// the result is deterministic run to run.
#include "common.h"

namespace pointops {

constexpr int kChBlock = 1024;

__global__ __launch_bounds__(kChBlock) void chamfer_reduce_kernel(
    const float* __restrict__ dists, const int64_t* __restrict__ lengths,
    const float* __restrict__ weights, int64_t P, int mean, float* __restrict__ out) {
  const int n = blockIdx.x;
  int64_t len = lengths[n];
  if (len > P) len = P;
  const float* __restrict__ row = dists + (int64_t)n * P;
  float acc = 0.0f;
  for (int64_t i = threadIdx.x; i < len; i += kChBlock) acc += row[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  __shared__ float s[kChBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) s[wave] = acc;
  __syncthreads();
  if (wave == 0) {
    float v = lane < kChBlock / kWave ? s[lane] : 0.0f;
#pragma unroll
    for (int off = kChBlock / kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) {
      if (weights != nullptr) v *= weights[n];
      if (mean) v /= (float)(len < 1 ? 1 : len);
      out[n] = v;
    }
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_chamfer_reduce(const float* dists, const int64_t* lengths,
                                       const float* weights, int64_t N, int64_t P, int mean,
                                       float* out, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && N < (1LL << 31), "chamfer_reduce: bad sizes");
  if (N == 0) return POINTOPS_OK;
  hipLaunchKernelGGL(chamfer_reduce_kernel, dim3((unsigned)N), dim3(kChBlock), 0, (hipStream_t)stream_,
                     dists, lengths, weights, P, mean, out);
  return check_launch("chamfer_reduce");
}
