// covariance.hip -- per-point covariance of a gathered K-neighbourhood, fused.
//
// Device half of get_point_covariances (reference: functions/utils.py:111-153), which builds the
// result from torch ops and materialises a (N,P,K,D,D) tensor of outer products (576 bytes per point
// at K = 16, D = 3) next to the centred copy of the neighbourhood:
//     m = mean_k x_k ;  cov[a][b] = mean_k (x_k[a] - m[a]) (x_k[b] - m[b]).
// Here one lane owns one point: the K x D neighbourhood row (contiguous, K*D*4 bytes) is read
// twice from L1/L2 (mean pass, covariance pass), the D x D accumulators live in registers, and
// only the (N,P,D,D) result is written.  Backward (closed form, the mean terms cancel because
// sum_k (x_k - m) = 0):  grad_x_k = (G + G^T) (x_k - m) / K.
// fp32 throughout; sums run in k order (the torch reduction order is unspecified: parity within 1e-5).
#include "common.h"

namespace pointops {

constexpr int kCovBlock = 256;
constexpr int kCovMaxD = 8;

template <int DT>  // DT = compile-time D (2, 3) or 0 = runtime D <= kCovMaxD
__global__ __launch_bounds__(kCovBlock) void covariance_kernel(const float* __restrict__ knn, int64_t rows, int K,
                                                              int Drt, float* __restrict__ cov) {
  constexpr int DM = DT > 0 ? DT : kCovMaxD;
  const int D = DT > 0 ? DT : Drt;
  const int64_t r = (int64_t)blockIdx.x * kCovBlock + threadIdx.x;
  if (r >= rows) return;
  const float* __restrict__ x = knn + r * K * D;
  const float inv_k = 1.0f / (float)K;
  float m[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] = 0.0f;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int d = 0; d < DM; ++d)
      if (d < D) m[d] += x[k * D + d];
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] *= inv_k;
  float c[DM][DM];
#pragma unroll
  for (int a = 0; a < DM; ++a)
#pragma unroll
    for (int b = 0; b < DM; ++b) c[a][b] = 0.0f;
  for (int k = 0; k < K; ++k) {
    float v[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) v[d] = d < D ? x[k * D + d] - m[d] : 0.0f;
#pragma unroll
    for (int a = 0; a < DM; ++a)
#pragma unroll
      for (int b = 0; b < DM; ++b)
        if (a < D && b < D) c[a][b] += v[a] * v[b];
  }
  float* __restrict__ o = cov + r * D * D;
#pragma unroll
  for (int a = 0; a < DM; ++a)
#pragma unroll
    for (int b = 0; b < DM; ++b)
      if (a < D && b < D) o[a * D + b] = c[a][b] * inv_k;
}

template <int DT>
__global__ __launch_bounds__(kCovBlock) void covariance_backward_kernel(const float* __restrict__ knn,
                                                                       const float* __restrict__ grad_cov,
                                                                       int64_t rows, int K, int Drt,
                                                                       float* __restrict__ grad_knn) {
  constexpr int DM = DT > 0 ? DT : kCovMaxD;
  const int D = DT > 0 ? DT : Drt;
  const int64_t r = (int64_t)blockIdx.x * kCovBlock + threadIdx.x;
  if (r >= rows) return;
  const float* __restrict__ x = knn + r * K * D;
  const float* __restrict__ g = grad_cov + r * D * D;
  float* __restrict__ o = grad_knn + r * K * D;
  const float inv_k = 1.0f / (float)K;
  float m[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] = 0.0f;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int d = 0; d < DM; ++d)
      if (d < D) m[d] += x[k * D + d];
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] *= inv_k;
  float s[DM][DM];  // (G + G^T) / K
#pragma unroll
  for (int a = 0; a < DM; ++a)
#pragma unroll
    for (int b = 0; b < DM; ++b) s[a][b] = (a < D && b < D) ? (g[a * D + b] + g[b * D + a]) * inv_k : 0.0f;
  for (int k = 0; k < K; ++k) {
    float v[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) v[d] = d < D ? x[k * D + d] - m[d] : 0.0f;
#pragma unroll
    for (int a = 0; a < DM; ++a) {
      if (a < D) {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < DM; ++b) acc += s[a][b] * v[b];
        o[k * D + a] = acc;
      }
    }
  }
}


// Staged form for K * D <= 64: a 128-lane workgroup loads its 128 neighbourhood rows COALESCED into
// LDS (row stride K*D + 1 words: conflict-free for the per-lane walk), then every lane makes both
// passes over its own row from LDS -- the direct kernels above read one 4..256-byte row per lane
// with 64 cache lines per wave instruction (0.26 ms for 8 x 65536 x 16 x 3, no faster than the five
// torch kernels they replace).
constexpr int kCovTile = 128;
constexpr int kCovStageMax = 64;  // floats per row

template <int DT, bool BACKWARD>
__global__ __launch_bounds__(kCovTile) void covariance_staged_kernel(const float* __restrict__ knn,
                                                                    const float* __restrict__ grad_cov,
                                                                    int64_t rows, int K, int Drt,
                                                                    float* __restrict__ out) {
  extern __shared__ float s_x[];  // [kCovTile][K*D + 1]
  constexpr int DM = DT > 0 ? DT : kCovMaxD;
  const int D = DT > 0 ? DT : Drt;
  const int T = K * D, stride = T + 1;
  const int64_t row0 = (int64_t)blockIdx.x * kCovTile;
  const int nrows = (int)min((int64_t)kCovTile, rows - row0);
  const float* __restrict__ src = knn + row0 * T;
  for (int f = threadIdx.x; f < nrows * T; f += kCovTile) {
    const int r = f / T;
    s_x[r * stride + (f - r * T)] = src[f];
  }
  __syncthreads();
  const bool active = (int)threadIdx.x < nrows;
  if (!BACKWARD && !active) return;  // the backward form has one more barrier: every lane must reach it
  if (active) {
  const float* __restrict__ x = s_x + threadIdx.x * stride;
  const int64_t r = row0 + threadIdx.x;
  const float inv_k = 1.0f / (float)K;
  float m[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] = 0.0f;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int d = 0; d < DM; ++d)
      if (d < D) m[d] += x[k * D + d];
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) m[d] *= inv_k;
  if constexpr (!BACKWARD) {
    float c[DM][DM];
#pragma unroll
    for (int a = 0; a < DM; ++a)
#pragma unroll
      for (int b = 0; b < DM; ++b) c[a][b] = 0.0f;
    for (int k = 0; k < K; ++k) {
      float v[DM];
#pragma unroll
      for (int d = 0; d < DM; ++d) v[d] = d < D ? x[k * D + d] - m[d] : 0.0f;
#pragma unroll
      for (int a = 0; a < DM; ++a)
#pragma unroll
        for (int b = 0; b < DM; ++b)
          if (a < D && b < D) c[a][b] += v[a] * v[b];
    }
    float* __restrict__ o = out + r * D * D;
#pragma unroll
    for (int a = 0; a < DM; ++a)
#pragma unroll
      for (int b = 0; b < DM; ++b)
        if (a < D && b < D) o[a * D + b] = c[a][b] * inv_k;
  } else {
    const float* __restrict__ g = grad_cov + r * D * D;
    float sm[DM][DM];  // (G + G^T) / K
#pragma unroll
    for (int a = 0; a < DM; ++a)
#pragma unroll
      for (int b = 0; b < DM; ++b) sm[a][b] = (a < D && b < D) ? (g[a * D + b] + g[b * D + a]) * inv_k : 0.0f;
    // results go back through LDS (in place) so that the global store is coalesced as well
    float* __restrict__ xo = s_x + threadIdx.x * stride;
    for (int k = 0; k < K; ++k) {
      float v[DM], w[DM];
#pragma unroll
      for (int d = 0; d < DM; ++d) v[d] = d < D ? xo[k * D + d] - m[d] : 0.0f;
#pragma unroll
      for (int a = 0; a < DM; ++a) {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < DM; ++b) acc += sm[a][b] * v[b];
        w[a] = acc;
      }
#pragma unroll
      for (int a = 0; a < DM; ++a)
        if (a < D) xo[k * D + a] = w[a];
    }
  }
  }  // active
  if constexpr (BACKWARD) {
    __syncthreads();
    float* __restrict__ dst = out + row0 * T;
    for (int f = threadIdx.x; f < nrows * T; f += kCovTile) {
      const int rr = f / T;
      dst[f] = s_x[rr * stride + (f - rr * T)];
    }
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_point_covariances(const float* knn, int64_t N, int64_t P, int64_t K, int64_t D, float* cov,
                                          void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && K >= 1 && D >= 1 && D <= kCovMaxD && K < (1LL << 31),
                   "point_covariances: need K >= 1 and 1 <= D <= %d", kCovMaxD);
  const int64_t rows = N * P;
  if (rows == 0) return POINTOPS_OK;
  const int64_t blocks = ceil_div(rows, kCovBlock);
  POINTOPS_REQUIRE(blocks < (1LL << 31), "point_covariances: grid too large");
  hipStream_t stream = (hipStream_t)stream_;
  if (K * D <= kCovStageMax) {
    const dim3 gs((unsigned)ceil_div(rows, kCovTile)), bs(kCovTile);
    const size_t lds = sizeof(float) * (size_t)kCovTile * (size_t)(K * D + 1);
    if (D == 3) hipLaunchKernelGGL((covariance_staged_kernel<3, false>), gs, bs, lds, stream, knn, nullptr, rows, (int)K, 3, cov);
    else if (D == 2) hipLaunchKernelGGL((covariance_staged_kernel<2, false>), gs, bs, lds, stream, knn, nullptr, rows, (int)K, 2, cov);
    else hipLaunchKernelGGL((covariance_staged_kernel<0, false>), gs, bs, lds, stream, knn, nullptr, rows, (int)K, (int)D, cov);
    return check_launch("point_covariances");
  }
  const dim3 grid((unsigned)blocks), block(kCovBlock);
  if (D == 3) hipLaunchKernelGGL(covariance_kernel<3>, grid, block, 0, stream, knn, rows, (int)K, 3, cov);
  else if (D == 2) hipLaunchKernelGGL(covariance_kernel<2>, grid, block, 0, stream, knn, rows, (int)K, 2, cov);
  else hipLaunchKernelGGL(covariance_kernel<0>, grid, block, 0, stream, knn, rows, (int)K, (int)D, cov);
  return check_launch("point_covariances");
}

extern "C" int pointops_point_covariances_backward(const float* knn, const float* grad_cov, int64_t N, int64_t P,
                                                   int64_t K, int64_t D, float* grad_knn, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && K >= 1 && D >= 1 && D <= kCovMaxD && K < (1LL << 31),
                   "point_covariances_backward: need K >= 1 and 1 <= D <= %d", kCovMaxD);
  const int64_t rows = N * P;
  if (rows == 0) return POINTOPS_OK;
  const int64_t blocks = ceil_div(rows, kCovBlock);
  POINTOPS_REQUIRE(blocks < (1LL << 31), "point_covariances_backward: grid too large");
  hipStream_t stream = (hipStream_t)stream_;
  if (K * D <= kCovStageMax) {
    const dim3 gs((unsigned)ceil_div(rows, kCovTile)), bs(kCovTile);
    const size_t lds = sizeof(float) * (size_t)kCovTile * (size_t)(K * D + 1);
    if (D == 3) hipLaunchKernelGGL((covariance_staged_kernel<3, true>), gs, bs, lds, stream, knn, grad_cov, rows, (int)K, 3, grad_knn);
    else if (D == 2) hipLaunchKernelGGL((covariance_staged_kernel<2, true>), gs, bs, lds, stream, knn, grad_cov, rows, (int)K, 2, grad_knn);
    else hipLaunchKernelGGL((covariance_staged_kernel<0, true>), gs, bs, lds, stream, knn, grad_cov, rows, (int)K, (int)D, grad_knn);
    return check_launch("point_covariances_backward");
  }
  const dim3 grid((unsigned)blocks), block(kCovBlock);
  if (D == 3)
    hipLaunchKernelGGL(covariance_backward_kernel<3>, grid, block, 0, stream, knn, grad_cov, rows, (int)K, 3, grad_knn);
  else if (D == 2)
    hipLaunchKernelGGL(covariance_backward_kernel<2>, grid, block, 0, stream, knn, grad_cov, rows, (int)K, 2, grad_knn);
  else
    hipLaunchKernelGGL(covariance_backward_kernel<0>, grid, block, 0, stream, knn, grad_cov, rows, (int)K, (int)D,
                       grad_knn);
  return check_launch("point_covariances_backward");
}
