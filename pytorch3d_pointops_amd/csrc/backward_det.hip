// backward_det.hip -- DETERMINISTIC scatter sides of the backward passes (gfx950): grad_p2 of knn_points / ball_query
// and grad_x of knn_gather through an INVERTED neighbour table instead of fp32 atomics.
//
// The reference's parity target is its CPU path, whose grad_p2 is a sequential sum: target row idx[n][i][k] receives
// its addends in (i, k) order (csrc/knn/knn_cpu.cpp:100-125).  Here the table entries e = (n P1 + i) K + k are keyed by
// their target row n P2 + idx and sorted with a STABLE radix sort (rocPRIM device_radix_sort, AMD's own primitive
// library; keys only as wide as N P2 needs), so every target's entries sit together in increasing e = the CPU's order;
// one lane per (target row, coordinate) then adds them one after the other starting from +0.  The result does not
// depend on scheduling, and for grad_p2 it is BIT-EQUAL to the reference's CPU kernel (same expression per addend,
// same order of the fp32 additions).  Selected by the callers under torch.use_deterministic_algorithms(True); the
// default backward passes (knn_backward.hip, gather.hip: LDS tiles / device atomics) are about three times faster.
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace pointops {

constexpr int kDetBlock = 256;

struct DetWs {
  unsigned *keys_in, *keys_out, *vals_in, *vals_out;
  int *seg_start, *seg_end;  // per target row (+ the slot of masked entries)
  void* sort_tmp;
  size_t sort_tmp_bytes;
};

static inline size_t det_align(size_t x) { return (x + 255) & ~(size_t)255; }

static int det_key_bits(int64_t targets) {  // keys 0 .. targets (targets = the slot of masked entries)
  int b = 1;
  while (((int64_t)1 << b) <= targets) ++b;
  return b;
}

static size_t det_carve(DetWs* ws, char* base, int64_t T, int64_t targets) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += det_align(bytes);
    return p;
  };
  DetWs w;
  w.keys_in = (unsigned*)take(sizeof(unsigned) * (size_t)T);
  w.keys_out = (unsigned*)take(sizeof(unsigned) * (size_t)T);
  w.vals_in = (unsigned*)take(sizeof(unsigned) * (size_t)T);
  w.vals_out = (unsigned*)take(sizeof(unsigned) * (size_t)T);
  w.seg_start = (int*)take(sizeof(int) * (size_t)(targets + 1));
  w.seg_end = (int*)take(sizeof(int) * (size_t)(targets + 1));
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (unsigned*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr,
                                  (unsigned*)nullptr, (size_t)T, 0, (unsigned)det_key_bits(targets), (hipStream_t)0);
  w.sort_tmp_bytes = tmp;
  w.sort_tmp = take(tmp);
  if (ws) *ws = w;
  return off;
}

// keys of the table entries: target row n M + idx, or `masked` (= N M) for entries the forward masks:
// k >= min(lengths2[n], K) (lengths2 may be null), i >= lengths1[n] (lengths1 may be null), idx outside [0, M)
__global__ __launch_bounds__(kDetBlock) void det_keys_kernel(const int64_t* __restrict__ idx,
                                                             const int64_t* __restrict__ lengths1,
                                                             const int64_t* __restrict__ lengths2, int64_t T, int L,
                                                             int K, int M, unsigned masked,
                                                             unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;
  if (e >= T) return;
  const int64_t row = e / K;
  const int k = (int)(e - row * K);
  const int n = (int)(row / L);
  const int i = (int)(row - (int64_t)n * L);
  bool ok = true;
  if (lengths1 != nullptr) ok = i < lengths1[n];
  if (lengths2 != nullptr) {
    const int64_t l2 = lengths2[n];
    ok = ok && k < (l2 < K ? l2 : K);
  }
  const int64_t j = idx[e];
  ok = ok && j >= 0 && j < M;
  keys[e] = ok ? (unsigned)((int64_t)n * M + j) : masked;
  vals[e] = (unsigned)e;
}

// segment [seg_start[r], seg_end[r]) of the sorted entries for every target row r that has any (both arrays are
// zero-filled before: rows without entries get an empty segment)
__global__ __launch_bounds__(kDetBlock) void det_segments_kernel(const unsigned* __restrict__ keys, int64_t T,
                                                                 int* __restrict__ seg_start,
                                                                 int* __restrict__ seg_end) {
  const int64_t t = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;
  if (t >= T) return;
  const unsigned k = keys[t];
  if (t == 0 || keys[t - 1] != k) seg_start[k] = (int)t;
  if (t == T - 1 || keys[t + 1] != k) seg_end[k] = (int)(t + 1);
}

// grad_p2[n][j][d] = sum over the row's entries, in table order, of -1.0f * c_d(entry) (knn_cpu.cpp:110-121);
// one lane per (target row, coordinate)
template <int NORM>
__global__ __launch_bounds__(kDetBlock) void det_knn_p2_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ grad_dists,
    const unsigned* __restrict__ vals, const int* __restrict__ seg_start, const int* __restrict__ seg_end,
    int64_t rows, int P1, int P2, int D, int K, float* __restrict__ grad_p2) {
  const int64_t o = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;  // element of grad_p2
  if (o >= rows * D) return;
  const int64_t r = o / D;
  const int d = (int)(o - r * D);
  const float b = p2[o];
  float acc = 0.0f;
  const int t1 = seg_end[r];
  for (int t = seg_start[r]; t < t1; ++t) {
    const unsigned e = vals[t];
    const float g = grad_dists[e];
    const float a = p1[(int64_t)(e / (unsigned)K) * D + d];
    float diff;
    if (NORM == 1) diff = g * ((a > b) ? 1.0f : -1.0f);
    else diff = 2.0f * g * (a - b);
    acc = acc + -1.0f * diff;
  }
  grad_p2[o] = acc;
}

// grad_p1[n][i][d] = sum over k, in order, of c_d (one lane per element; any D)
template <int NORM>
__global__ __launch_bounds__(kDetBlock) void det_knn_p1_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, const int64_t* __restrict__ idx, const float* __restrict__ grad_dists,
    int64_t rows, int P1, int P2, int D, int K, float* __restrict__ grad_p1) {
  const int64_t o = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;
  if (o >= rows * D) return;
  const int64_t row = o / D;
  const int d = (int)(o - row * D);
  const int n = (int)(row / P1);
  const int i = (int)(row - (int64_t)n * P1);
  float acc = 0.0f;
  if (i < lengths1[n]) {
    const int64_t l2 = lengths2[n];
    const int kmax = (int)(l2 < K ? l2 : K);
    const float a = p1[o];
    for (int k = 0; k < kmax; ++k) {
      const int64_t j = idx[row * K + k];
      if (j < 0 || j >= P2) continue;
      const float b = p2[((int64_t)n * P2 + j) * D + d];
      const float g = grad_dists[row * K + k];
      float diff;
      if (NORM == 1) diff = g * ((a > b) ? 1.0f : -1.0f);
      else diff = 2.0f * g * (a - b);
      acc = acc + diff;
    }
  }
  grad_p1[o] = acc;
}

// grad_x[n][m][u] = sum over the row's entries, in table order, of grad_out[entry][u]
__global__ __launch_bounds__(kDetBlock) void det_gather_kernel(const float* __restrict__ grad_out,
                                                               const unsigned* __restrict__ vals,
                                                               const int* __restrict__ seg_start,
                                                               const int* __restrict__ seg_end, int64_t rows, int U,
                                                               float* __restrict__ grad_x) {
  const int64_t o = (int64_t)blockIdx.x * kDetBlock + threadIdx.x;
  if (o >= rows * U) return;
  const int64_t r = o / U;
  const int u = (int)(o - r * U);
  float acc = 0.0f;
  const int t1 = seg_end[r];
  for (int t = seg_start[r]; t < t1; ++t) acc = acc + grad_out[(int64_t)vals[t] * U + u];
  grad_x[o] = acc;
}

// keys, stable sort, segments; afterwards ws.vals_out / seg_start / seg_end describe the inverted table
static int det_invert(const DetWs& ws, const int64_t* idx, const int64_t* lengths1, const int64_t* lengths2, int64_t N,
                      int64_t L, int64_t K, int64_t M, hipStream_t stream) {
  const int64_t T = N * L * K, targets = N * M;
  if (hipMemsetAsync(ws.seg_start, 0, sizeof(int) * (size_t)(targets + 1), stream) != hipSuccess ||
      hipMemsetAsync(ws.seg_end, 0, sizeof(int) * (size_t)(targets + 1), stream) != hipSuccess)
    return check_launch("deterministic backward (memset)");
  if (T == 0) return POINTOPS_OK;
  const unsigned blocks = (unsigned)ceil_div(T, kDetBlock);
  hipLaunchKernelGGL(det_keys_kernel, dim3(blocks), dim3(kDetBlock), 0, stream, idx, lengths1, lengths2, T, (int)L,
                     (int)K, (int)M, (unsigned)targets, ws.keys_in, ws.vals_in);
  size_t tmp = ws.sort_tmp_bytes;
  if (rocprim::radix_sort_pairs(ws.sort_tmp, tmp, ws.keys_in, ws.keys_out, ws.vals_in, ws.vals_out, (size_t)T, 0,
                                (unsigned)det_key_bits(targets), stream) != hipSuccess)
    return check_launch("deterministic backward (sort)");
  hipLaunchKernelGGL(det_segments_kernel, dim3(blocks), dim3(kDetBlock), 0, stream, ws.keys_out, T, ws.seg_start,
                     ws.seg_end);
  return check_launch("deterministic backward (invert)");
}

}  // namespace pointops

using namespace pointops;

extern "C" size_t pointops_backward_det_workspace_bytes(int64_t N, int64_t L, int64_t K, int64_t M) {
  if (N <= 0 || M <= 0) return 256;
  return det_carve(nullptr, nullptr, N * L * K > 0 ? N * L * K : 1, N * M);
}

extern "C" int pointops_knn_points_backward_det(const float* p1, const float* p2, const int64_t* lengths1,
                                                const int64_t* lengths2, const int64_t* idxs,
                                                const float* grad_dists, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                                int64_t K, int norm, float* grad_p1, float* grad_p2, void* workspace,
                                                size_t workspace_bytes, void* stream_) {
  POINTOPS_REQUIRE(norm == 1 || norm == 2, "knn_points_backward(deterministic): norm must be 1 or 2");
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && K >= 0, "knn_points_backward(deterministic): bad sizes");
  POINTOPS_REQUIRE(N * P1 * K < (1LL << 31) && N * P2 < (1LL << 31) - 1 && N * P1 * D < (1LL << 40) && D < (1LL << 16),
                   "knn_points_backward(deterministic): the neighbour table must have fewer than 2^31 entries");
  if (N == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= pointops_backward_det_workspace_bytes(N, P1, K, P2),
                   "knn_points_backward(deterministic): workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  DetWs ws;
  det_carve(&ws, (char*)workspace, N * P1 * K > 0 ? N * P1 * K : 1, N * P2);
  const int rc = det_invert(ws, idxs, lengths1, lengths2, N, P1, K, P2, stream);
  if (rc != POINTOPS_OK) return rc;
  if (N * P2 * D > 0) {
    const unsigned b2 = (unsigned)ceil_div(N * P2 * D, kDetBlock);
    if (norm == 1)
      hipLaunchKernelGGL(det_knn_p2_kernel<1>, dim3(b2), dim3(kDetBlock), 0, stream, p1, p2, grad_dists, ws.vals_out,
                         ws.seg_start, ws.seg_end, N * P2, (int)P1, (int)P2, (int)D, (int)K, grad_p2);
    else
      hipLaunchKernelGGL(det_knn_p2_kernel<2>, dim3(b2), dim3(kDetBlock), 0, stream, p1, p2, grad_dists, ws.vals_out,
                         ws.seg_start, ws.seg_end, N * P2, (int)P1, (int)P2, (int)D, (int)K, grad_p2);
  }
  if (N * P1 * D > 0) {
    const unsigned b1 = (unsigned)ceil_div(N * P1 * D, kDetBlock);
    if (norm == 1)
      hipLaunchKernelGGL(det_knn_p1_kernel<1>, dim3(b1), dim3(kDetBlock), 0, stream, p1, p2, lengths1, lengths2, idxs,
                         grad_dists, N * P1, (int)P1, (int)P2, (int)D, (int)K, grad_p1);
    else
      hipLaunchKernelGGL(det_knn_p1_kernel<2>, dim3(b1), dim3(kDetBlock), 0, stream, p1, p2, lengths1, lengths2, idxs,
                         grad_dists, N * P1, (int)P1, (int)P2, (int)D, (int)K, grad_p1);
  }
  return check_launch("knn_points_backward(deterministic)");
}

extern "C" int pointops_gather_neighbors_backward_det(const float* grad_out, const int64_t* idx,
                                                      const int64_t* lengths, int64_t N, int64_t M, int64_t U,
                                                      int64_t L, int64_t K, float* grad_x, void* workspace,
                                                      size_t workspace_bytes, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && M >= 0 && U >= 1 && L >= 0 && K >= 0, "gather_neighbors_backward(deterministic): bad sizes");
  POINTOPS_REQUIRE(N * L * K < (1LL << 31) && N * M < (1LL << 31) - 1,
                   "gather_neighbors_backward(deterministic): the neighbour table must have fewer than 2^31 entries");
  if (N == 0 || M == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= pointops_backward_det_workspace_bytes(N, L, K, M),
                   "gather_neighbors_backward(deterministic): workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  DetWs ws;
  det_carve(&ws, (char*)workspace, N * L * K > 0 ? N * L * K : 1, N * M);
  // knn_gather masks k >= lengths[n] (the lengths of the gathered cloud); queries have no lengths here
  const int rc = det_invert(ws, idx, nullptr, lengths, N, L, K, M, stream);
  if (rc != POINTOPS_OK) return rc;
  const unsigned b = (unsigned)ceil_div(N * M * U, kDetBlock);
  hipLaunchKernelGGL(det_gather_kernel, dim3(b), dim3(kDetBlock), 0, stream, grad_out, ws.vals_out, ws.seg_start,
                     ws.seg_end, N * M, (int)U, grad_x);
  return check_launch("gather_neighbors_backward(deterministic)");
}
