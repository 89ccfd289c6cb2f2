// knn_backward.hip -- gradient of the KNN / ball-query distances w.r.t. both clouds.
//
// Replaces KNearestNeighborBackward (reference: csrc/knn/knn.h:127-149) with the
// CPU path's semantics (csrc/knn/knn_cpu.cpp:75-128).  One lane per query point:
//   grad_p1[n,i,:] = sum_k c(n,i,k,:)   accumulated in registers in k order -- no
//       atomics, deterministic and in the same order as the CPU loop, so grad_p1 is
//       bit-equal to the reference CPU result;
//   grad_p2[n,idx,:] -= c               scatter-add with global_atomic_add_f32
//       (fp32, agent scope): the only cross-lane accumulation; order-dependent in
//       the last bits like the reference CUDA path (csrc/knn/knn.cu:514-515,538).
//   c = 2*g*(p1-p2) for L2, g*sign(p1>p2) for L1.
#include "common.h"

namespace pointops {

constexpr int kBwdBlock = 256;

template <int DT, int NORM>  // DT = compile-time D, 0 = runtime D
__global__ __launch_bounds__(kBwdBlock) void knn_backward_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2,
    const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists, int P1, int P2, int Drt,
    int K, int tiles_per_cloud, float* __restrict__ grad_p1, float* __restrict__ grad_p2) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int i = tile * kBwdBlock + threadIdx.x;
  if (i >= P1) return;
  const int len1 = (int)lengths1[n];
  int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  float* __restrict__ g1 = grad_p1 + row * D;
  if (i >= len1) {
    for (int d = 0; d < D; ++d) g1[d] = 0.0f;
    return;
  }
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float* __restrict__ a = p1 + row * D;
  if constexpr (DT > 0) {
    float av[DT], acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      av[d] = a[d];
      acc[d] = 0.0f;
    }
    for (int k = 0; k < kmax; ++k) {
      const int64_t i2 = irow[k];
      if (i2 < 0 || i2 >= P2) continue;  // -1 padding (ball query); also guards bad input
      const float g = grow[k];
      const float* __restrict__ b = p2 + ((int64_t)n * P2 + i2) * DT;
      float* __restrict__ g2 = grad_p2 + ((int64_t)n * P2 + i2) * DT;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const float bv = b[d];
        float diff;
        if (NORM == 1) diff = g * ((av[d] > bv) ? 1.0f : -1.0f);
        else diff = 2.0f * g * (av[d] - bv);
        acc[d] = acc[d] + diff;
        if (diff != 0.0f) atomicAdd(g2 + d, -1.0f * diff);  // +-0 never changes a sum started at +0
      }
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) g1[d] = acc[d];
  } else {
    for (int d = 0; d < D; ++d) {
      const float av = a[d];
      float acc = 0.0f;
      for (int k = 0; k < kmax; ++k) {
        const int64_t i2 = irow[k];
        if (i2 < 0 || i2 >= P2) continue;
        const float g = grow[k];
        const float bv = p2[((int64_t)n * P2 + i2) * D + d];
        float diff;
        if (NORM == 1) diff = g * ((av > bv) ? 1.0f : -1.0f);
        else diff = 2.0f * g * (av - bv);
        acc = acc + diff;
        if (diff != 0.0f) atomicAdd(grad_p2 + ((int64_t)n * P2 + i2) * D + d, -1.0f * diff);
      }
      g1[d] = acc;
    }
  }
}


// D <= 4: FOUR lanes per query, lane c owning coordinate c.  grad_p1[n,i,c] is still one lane's
// register sum in k order (bit-equal to the CPU loop); the grad_p2 atomics of one neighbour row
// now leave as ONE wave instruction over 16 rows x 12-16 contiguous bytes instead of three
// instructions over 64 different rows each (the memory-side fp32 atomic rate depends on the rows
// touched per instruction: MI355X_MICROARCH.md, global float atomics).
template <int NORM>
__global__ __launch_bounds__(kBwdBlock) void knn_backward4_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2,
    const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists, int P1, int P2, int D, int K,
    int tiles_per_cloud, float* __restrict__ grad_p1, float* __restrict__ grad_p2) {
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int t = tile * kBwdBlock + threadIdx.x;
  const int i = t >> 2, c = t & 3;
  if (i >= P1 || c >= D) return;
  const int len1 = (int)lengths1[n];
  const int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  if (i >= len1) {
    grad_p1[row * D + c] = 0.0f;
    return;
  }
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float av = p1[row * D + c];
  float acc = 0.0f;
  for (int k = 0; k < kmax; ++k) {
    const int64_t i2 = irow[k];
    if (i2 < 0 || i2 >= P2) continue;  // -1 padding (ball query); also guards bad input
    const float g = grow[k];
    const int64_t o = ((int64_t)n * P2 + i2) * D + c;
    const float bv = p2[o];
    float diff;
    if (NORM == 1) diff = g * ((av > bv) ? 1.0f : -1.0f);
    else diff = 2.0f * g * (av - bv);
    acc = acc + diff;
    if (diff != 0.0f) atomicAdd(grad_p2 + o, -1.0f * diff);
  }
  grad_p1[row * D + c] = acc;
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_knn_points_backward(const float* p1, const float* p2,
                                            const int64_t* lengths1, const int64_t* lengths2,
                                            const int64_t* idxs, const float* grad_dists, int64_t N,
                                            int64_t P1, int64_t P2, int64_t D, int64_t K, int norm,
                                            float* grad_p1, float* grad_p2, void* stream_) {
  POINTOPS_REQUIRE(norm == 1 || norm == 2, "knn_points_backward: norm must be 1 or 2");
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && K >= 0, "knn_points_backward: bad sizes");
  POINTOPS_REQUIRE(P1 < (1LL << 31) && P2 < (1LL << 31) && K < (1LL << 31) && D < (1LL << 16),
                   "knn_points_backward: sizes must fit int32");
  hipStream_t stream = (hipStream_t)stream_;
  if (N * P2 * D > 0) {
    if (hipMemsetAsync(grad_p2, 0, sizeof(float) * (size_t)(N * P2 * D), stream) != hipSuccess)
      return check_launch("knn_points_backward(memset)");
  }
  if (N == 0 || P1 == 0) return POINTOPS_OK;
  if (D <= 4) {
    const int tiles4 = (int)ceil_div(P1 * 4, kBwdBlock);
    POINTOPS_REQUIRE(N * tiles4 < (1LL << 31), "knn_points_backward: grid too large");
    const dim3 grid4((unsigned)(N * tiles4)), block4(kBwdBlock);
    if (norm == 1)
      hipLaunchKernelGGL(knn_backward4_kernel<1>, grid4, block4, 0, stream, p1, p2, lengths1, lengths2, idxs,
                         grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles4, grad_p1, grad_p2);
    else
      hipLaunchKernelGGL(knn_backward4_kernel<2>, grid4, block4, 0, stream, p1, p2, lengths1, lengths2, idxs,
                         grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles4, grad_p1, grad_p2);
    return check_launch("knn_points_backward");
  }
  const int tiles = (int)ceil_div(P1, kBwdBlock);
  POINTOPS_REQUIRE(N * tiles < (1LL << 31), "knn_points_backward: grid too large");
  const dim3 grid((unsigned)(N * tiles)), block(kBwdBlock);
#define PO_LAUNCH(DT, NORM)                                                                       \
  hipLaunchKernelGGL((knn_backward_kernel<DT, NORM>), grid, block, 0, stream, p1, p2, lengths1,   \
                     lengths2, idxs, grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles, grad_p1, \
                     grad_p2)
  if (norm == 1) {
    if (D == 3) PO_LAUNCH(3, 1);
    else if (D == 2) PO_LAUNCH(2, 1);
    else PO_LAUNCH(0, 1);
  } else {
    if (D == 3) PO_LAUNCH(3, 2);
    else if (D == 2) PO_LAUNCH(2, 2);
    else PO_LAUNCH(0, 2);
  }
#undef PO_LAUNCH
  return check_launch("knn_points_backward");
}
