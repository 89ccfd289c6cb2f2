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
#include "tiled_scatter.h"

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
template <int NORM, bool SCATTER>
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
    if (SCATTER && diff != 0.0f) atomicAdd(grad_p2 + o, -1.0f * diff);
  }
  grad_p1[row * D + c] = acc;
}


// grad_p1 only (the LDS-tile path below owns grad_p2): ONE lane per query row, the row's idx and
// grad entries fetched as 16-byte vectors eight neighbours at a time, all issued before use, so a
// row's 128-byte line is consumed while it is in flight instead of being re-fetched per k (the
// 4-lanes-per-row kernel above exists for contiguous atomics, which this path does not issue).
// The sum stays one register chain in k order: bit-equal to the CPU loop.
// Measured and dropped: an LDS-staged form with fully coalesced table reads (lane per entry, then
// lane per row over the parked contributions) -- 0.845 vs 0.841 ms for the whole backward: the kernel
// is bound by the 33.5 M random 12-byte gathers of p2 rows (~145 G L2 requests/s), not by its table reads.
struct alignas(8) I64x2 { int64_t v[2]; };
struct alignas(4) F32x4 { float v[4]; };
struct alignas(4) F32x3 { float v[3]; };

template <int DT, int NORM>  // DT = compile-time D (1..4)
__global__ __launch_bounds__(kBwdBlock) void knn_backward_rows_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists,
    int P1, int P2, int K, int tiles_per_cloud, float* __restrict__ grad_p1) {
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int i = tile * kBwdBlock + threadIdx.x;
  if (i >= P1) return;
  const int len1 = (int)lengths1[n];
  const int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  float* __restrict__ g1 = grad_p1 + row * DT;
  float acc[DT], av[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) acc[d] = 0.0f;
  if (i >= len1) {
#pragma unroll
    for (int d = 0; d < DT; ++d) g1[d] = 0.0f;
    return;
  }
#pragma unroll
  for (int d = 0; d < DT; ++d) av[d] = p1[row * DT + d];
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float* __restrict__ p2n = p2 + (int64_t)n * P2 * DT;
  auto one = [&](int64_t i2, float g) {
    // -1 padding (ball query) and bad input contribute an exact +0 (branch-free: the eight
    // gathers of a chunk can all be in flight)
    const bool valid = i2 >= 0 && i2 < P2;
    i2 = valid ? i2 : 0;
    float bv[DT];
    if constexpr (DT == 3) {
      const F32x3 b = *reinterpret_cast<const F32x3*>(p2n + i2 * 3);
      bv[0] = b.v[0], bv[1] = b.v[1], bv[2] = b.v[2];
    } else {
#pragma unroll
      for (int d = 0; d < DT; ++d) bv[d] = p2n[i2 * DT + d];
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      float diff;
      if (NORM == 1) diff = g * ((av[d] > bv[d]) ? 1.0f : -1.0f);
      else diff = 2.0f * g * (av[d] - bv[d]);
      acc[d] = acc[d] + (valid ? diff : 0.0f);
    }
  };
  int k = 0;
  for (; k + 8 <= kmax; k += 8) {
    I64x2 iv[4];
    F32x4 gv[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) iv[q] = *reinterpret_cast<const I64x2*>(irow + k + 2 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) gv[q] = *reinterpret_cast<const F32x4*>(grow + k + 4 * q);
#pragma unroll
    for (int q = 0; q < 8; ++q) one(iv[q >> 1].v[q & 1], gv[q >> 2].v[q & 3]);
  }
  for (; k < kmax; ++k) one(irow[k], grow[k]);
#pragma unroll
  for (int d = 0; d < DT; ++d) g1[d] = acc[d];
}

// grad_p2 through the LDS-tile scatter of tiled_scatter.h: the addend of entry (i, k) -> row j is
// -c(n,i,k,:), fetched as the entry's grad and the two points' rows.
template <int DT, int NORM>
struct KnnGradSrc {
  static constexpr int kChannels = DT;
  struct Regs {
    float g, a[DT], b[DT];
  };
  const float* p1;
  const float* p2;
  const int64_t* lengths1;
  const int64_t* lengths2;
  const float* grad_dists;
  int P1, P2, K;
  __device__ int rows(int n) const { return (int)min((int64_t)P1, lengths1[n]); }
  __device__ int kmax(int n) const { return (int)min((int64_t)K, lengths2[n]); }
  __device__ void issue(int n, int e, int i, int k, int j, Regs& r) const {
    r.g = grad_dists[(int64_t)n * P1 * K + e];
    const float* __restrict__ a = p1 + ((int64_t)n * P1 + i) * DT;
    const float* __restrict__ b = p2 + ((int64_t)n * P2 + j) * DT;
    if constexpr (DT == 3) {
      const F32x3 av = *reinterpret_cast<const F32x3*>(a);
      const F32x3 bv = *reinterpret_cast<const F32x3*>(b);
#pragma unroll
      for (int c = 0; c < 3; ++c) r.a[c] = av.v[c], r.b[c] = bv.v[c];
    } else {
#pragma unroll
      for (int c = 0; c < DT; ++c) r.a[c] = a[c], r.b[c] = b[c];
    }
  }
  __device__ float value(const Regs& r, int c) const {
    float diff;
    if (NORM == 1) diff = r.g * ((r.a[c] > r.b[c]) ? 1.0f : -1.0f);
    else diff = 2.0f * r.g * (r.a[c] - r.b[c]);
    return -1.0f * diff;
  }
};

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
  // grad_p2 mode: LDS tiles when the neighbour table is large and p2 splits into few tiles
  const TiledPlan plan = tiled_plan(N, P1, K, P2, (int)D, "knn_bwd_mode", "knn_bwd_split");
  const bool tiled = plan.tiled;
  const int S = plan.S;
  if (N * P2 * D > 0 && !(tiled && S == 1)) {
    if (hipMemsetAsync(grad_p2, 0, sizeof(float) * (size_t)(N * P2 * D), stream) != hipSuccess)
      return check_launch("knn_points_backward(memset)");
  }
  if (N == 0 || P1 == 0) return POINTOPS_OK;
  if (D <= 4) {
    const int tiles4 = (int)ceil_div(P1 * 4, kBwdBlock);
    POINTOPS_REQUIRE(N * tiles4 < (1LL << 31), "knn_points_backward: grid too large");
    const dim3 grid4((unsigned)(N * tiles4)), block4(kBwdBlock);
#define PO_LAUNCH4(NORM, SCATTER)                                                                               \
  hipLaunchKernelGGL((knn_backward4_kernel<NORM, SCATTER>), grid4, block4, 0, stream, p1, p2, lengths1, lengths2, \
                     idxs, grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles4, grad_p1, grad_p2)
    if (!tiled) {
      if (norm == 1) PO_LAUNCH4(1, true);
      else PO_LAUNCH4(2, true);
      return check_launch("knn_points_backward");
    }
#undef PO_LAUNCH4
    const int tiles = (int)ceil_div(P1, kBwdBlock);
    const dim3 gridr((unsigned)(N * tiles)), blockr(kBwdBlock);
    const DivMagic dm = division_magic((unsigned)K);
#define PO_TILED(DT, NORM)                                                                                     \
  do {                                                                                                         \
    hipLaunchKernelGGL((knn_backward_rows_kernel<DT, NORM>), gridr, blockr, 0, stream, p1, p2, lengths1,       \
                       lengths2, idxs, grad_dists, (int)P1, (int)P2, (int)K, tiles, grad_p1);                  \
    const KnnGradSrc<DT, NORM> src{p1, p2, lengths1, lengths2, grad_dists, (int)P1, (int)P2, (int)K};          \
    hipLaunchKernelGGL((tiled_scatter_kernel<KnnGradSrc<DT, NORM>>), plan.grid, dim3(kTiledBlock), 0, stream,  \
                       src, idxs, (int)N, (int)P1, (int)P2, (int)K, dm, plan.parts, S, grad_p2);               \
  } while (0)
    if (norm == 1) {
      if (D == 1) PO_TILED(1, 1);
      else if (D == 2) PO_TILED(2, 1);
      else if (D == 3) PO_TILED(3, 1);
      else PO_TILED(4, 1);
    } else {
      if (D == 1) PO_TILED(1, 2);
      else if (D == 2) PO_TILED(2, 2);
      else if (D == 3) PO_TILED(3, 2);
      else PO_TILED(4, 2);
    }
#undef PO_TILED
    return check_launch("knn_points_backward(tiled)");
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
