// gather.hip -- neighbour gather and its scatter-add backward for gfx950.
//
// Device half of knn_gather / masked_gather (reference: functions/knn.py:200-250,
// functions/utils.py:20-65): out[n,l,k,:] = x[n, idx[n,l,k], :], zero where
// k >= lengths[n] (knn padding) or idx < 0 (ball-query / FPS -1 padding).  The
// reference builds this from expand + torch.gather, which materialises an
// (N,L,K,U) int64 index (8*U bytes per output element); here the index is read
// once per (n,l,k) row and the output is written coalesced, one lane per element.
// Backward scatters grad_out into grad_x (same masks): LDS tiles without device atomics
// (tiled_scatter.h) when U <= 4 and the table is large, fp32 device atomics otherwise.
#include "common.h"
#include "tiled_scatter.h"

namespace pointops {

constexpr int kGaBlock = 256;

__global__ __launch_bounds__(kGaBlock) void gather_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ lengths, int64_t total, int64_t M, int U, int64_t LK, int K,
    float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kGaBlock + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / U;  // (n,l,k) row
  const int u = (int)(e - r * U);
  const int64_t n = r / LK;
  const int k = (int)(r % K);
  const int64_t j = idx[r];
  bool ok = j >= 0 && j < M;
  if (lengths != nullptr) ok = ok && (int64_t)k < lengths[n];
  out[e] = ok ? x[(n * M + j) * U + u] : 0.0f;
}

// (Measured, round 3: the cfg2-shape gather -- 33.5 M rows of 12 bytes out of 786 KB clouds -- runs at 2.4 TB/s of
// algorithmic bytes whatever the store width or the rows per lane (four rows per lane with 16-byte index loads and
// output stores: 0.2865 against 0.2876 ms).  Every 12-byte row pulls a whole 128-byte line from L2 into L1 -- 4.3 GB of
// L2->L1 traffic per call, ~15 TB/s --, and neighbours are close in space, not in storage order, so lines are not shared
// between lanes: the op is bound by L2 line traffic, not by HBM.)
// U <= 4: one lane per (n,l,k) ROW -- the index is read once, the U values leave as one 4/8/12/16
// byte store per lane (contiguous across the wave), and all index arithmetic is 32-bit within a
// cloud (grid.y = cloud); the per-element kernel above spends most of its time in 64-bit divisions.
template <int U>
struct alignas(4) GaRow { float v[U]; };

template <int U>
__global__ __launch_bounds__(kGaBlock) void gather_rows_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ idx, const int64_t* __restrict__ lengths, int M,
    int LK, int K, float* __restrict__ out) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * kGaBlock + threadIdx.x;
  if (e >= LK) return;
  const int k = e % K;
  const int64_t r = (int64_t)n * LK + e;
  const int64_t j = idx[r];
  bool ok = j >= 0 && j < M;
  if (lengths != nullptr) ok = ok && (int64_t)k < lengths[n];
  GaRow<U> v;
#pragma unroll
  for (int u = 0; u < U; ++u) v.v[u] = 0.0f;
  if (ok) v = *reinterpret_cast<const GaRow<U>*>(x + ((int64_t)n * M + j) * U);
  *reinterpret_cast<GaRow<U>*>(out + r * U) = v;
}

// any U, one lane per element, 32-bit arithmetic within a cloud (LK * U < 2^31)
__global__ __launch_bounds__(kGaBlock) void gather_elems32_kernel(
    const float* __restrict__ x, const int64_t* __restrict__ idx, const int64_t* __restrict__ lengths, int M, int U,
    int LK, int K, float* __restrict__ out) {
  const int n = blockIdx.y;
  const unsigned e = blockIdx.x * (unsigned)kGaBlock + threadIdx.x;
  if (e >= (unsigned)LK * (unsigned)U) return;
  const unsigned r = e / (unsigned)U, u = e - r * (unsigned)U;
  const int k = (int)(r % (unsigned)K);
  const int64_t j = idx[(int64_t)n * LK + r];
  bool ok = j >= 0 && j < M;
  if (lengths != nullptr) ok = ok && (int64_t)k < lengths[n];
  out[((int64_t)n * LK + r) * U + u] = ok ? x[((int64_t)n * M + j) * U + u] : 0.0f;
}

__global__ __launch_bounds__(kGaBlock) void gather_backward_kernel(
    const float* __restrict__ grad_out, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ lengths, int64_t total, int64_t M, int U, int64_t LK, int K,
    float* __restrict__ grad_x) {
  const int64_t e = (int64_t)blockIdx.x * kGaBlock + threadIdx.x;
  if (e >= total) return;
  const int64_t r = e / U;
  const int u = (int)(e - r * U);
  const int64_t n = r / LK;
  const int k = (int)(r % K);
  const int64_t j = idx[r];
  bool ok = j >= 0 && j < M;
  if (lengths != nullptr) ok = ok && (int64_t)k < lengths[n];
  if (ok) {
    // Adding +-0.0 to a sum that started at +0.0 never changes it, so zero gradients (every
    // masked / padded row upstream) skip the atomic: padded rows all carry idx 0 and would
    // otherwise serialise ~1e5 atomics on one address.
    const float gv = grad_out[e];
    if (gv != 0.0f) atomicAdd(grad_x + (n * M + j) * U + u, gv);
  }
}

// addend of entry (l, k) -> row j: grad_out[n, l, k, :]
template <int C>
struct GatherGradSrc {
  static constexpr int kChannels = C;
  struct Regs {
    float v[C];
  };
  const float* grad_out;
  const int64_t* lengths;  // may be null
  int L, K;
  __device__ int rows(int) const { return L; }
  __device__ int kmax(int n) const { return lengths != nullptr ? (int)min((int64_t)K, max((int64_t)0, lengths[n])) : K; }
  __device__ void issue(int n, int e, int, int, int, Regs& r) const {
    const float* __restrict__ g = grad_out + ((int64_t)n * L * K + e) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) r.v[c] = g[c];
  }
  __device__ float value(const Regs& r, int c) const { return r.v[c]; }
};

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_gather_neighbors(const float* x, const int64_t* idx, const int64_t* lengths,
                                         int64_t N, int64_t M, int64_t U, int64_t L, int64_t K,
                                         float* out, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && M >= 0 && U >= 1 && L >= 0 && K >= 0 && U < (1LL << 31) && K < (1LL << 31),
                   "gather_neighbors: bad sizes");
  const int64_t total = N * L * K * U;
  if (total == 0) return POINTOPS_OK;
  const int64_t LK = L * K;
  if (N < 65536 && M < (1LL << 31) && LK * U < (1LL << 31)) {
    hipStream_t stream = (hipStream_t)stream_;
    if (U <= 4) {
      const dim3 grid((unsigned)ceil_div(LK, kGaBlock), (unsigned)N), block(kGaBlock);
#define PO_ROWS(UU) \
  hipLaunchKernelGGL(gather_rows_kernel<UU>, grid, block, 0, stream, x, idx, lengths, (int)M, (int)LK, (int)K, out)
      if (U == 1) PO_ROWS(1);
      else if (U == 2) PO_ROWS(2);
      else if (U == 3) PO_ROWS(3);
      else PO_ROWS(4);
#undef PO_ROWS
    } else {
      const dim3 grid((unsigned)ceil_div(LK * U, kGaBlock), (unsigned)N), block(kGaBlock);
      hipLaunchKernelGGL(gather_elems32_kernel, grid, block, 0, stream, x, idx, lengths, (int)M, (int)U, (int)LK,
                         (int)K, out);
    }
    return check_launch("gather_neighbors");
  }
  const int64_t blocks = ceil_div(total, kGaBlock);
  POINTOPS_REQUIRE(blocks < (1LL << 31), "gather_neighbors: grid too large");
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)blocks), dim3(kGaBlock), 0, (hipStream_t)stream_, x,
                     idx, lengths, total, M, (int)U, L * K, (int)K, out);
  return check_launch("gather_neighbors");
}

extern "C" int pointops_gather_neighbors_backward(const float* grad_out, const int64_t* idx,
                                                  const int64_t* lengths, int64_t N, int64_t M,
                                                  int64_t U, int64_t L, int64_t K, float* grad_x,
                                                  void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && M >= 0 && U >= 1 && L >= 0 && K >= 0 && U < (1LL << 31) && K < (1LL << 31),
                   "gather_neighbors_backward: bad sizes");
  hipStream_t stream = (hipStream_t)stream_;
  const TiledPlan plan = tiled_plan(N, L, K, M, (int)U, "gather_bwd_mode", "gather_bwd_split");
  if (N * M * U > 0 && !(plan.tiled && plan.S == 1)) {
    if (hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)(N * M * U), stream) != hipSuccess)
      return check_launch("gather_neighbors_backward(memset)");
  }
  const int64_t total = N * L * K * U;
  if (total == 0) return POINTOPS_OK;
  if (plan.tiled) {
    const DivMagic dm = division_magic((unsigned)K);
#define PO_TILED(C)                                                                                          \
  do {                                                                                                       \
    const GatherGradSrc<C> src{grad_out, lengths, (int)L, (int)K};                                           \
    hipLaunchKernelGGL((tiled_scatter_kernel<GatherGradSrc<C>>), plan.grid, dim3(kTiledBlock), 0, stream,    \
                       src, idx, (int)N, (int)L, (int)M, (int)K, dm, plan.parts, plan.S, grad_x);            \
  } while (0)
    if (U == 1) PO_TILED(1);
    else if (U == 2) PO_TILED(2);
    else if (U == 3) PO_TILED(3);
    else PO_TILED(4);
#undef PO_TILED
    return check_launch("gather_neighbors_backward(tiled)");
  }
  const int64_t blocks = ceil_div(total, kGaBlock);
  POINTOPS_REQUIRE(blocks < (1LL << 31), "gather_neighbors_backward: grid too large");
  hipLaunchKernelGGL(gather_backward_kernel, dim3((unsigned)blocks), dim3(kGaBlock), 0, stream,
                     grad_out, idx, lengths, total, M, (int)U, L * K, (int)K, grad_x);
  return check_launch("gather_neighbors_backward");
}
