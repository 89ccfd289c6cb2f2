// packed_padded.hip -- ragged packed <-> zero-padded row copies for gfx950.
//
// Replaces PackedToPadded / PaddedToPacked (reference:
// csrc/packed_to_padded_tensor/packed_to_padded_tensor.h:78-113; CPU semantics
// packed_to_padded_tensor_cpu.cpp:11-70).  Cloud b owns packed rows
// [first_idxs[b], first_idxs[b+1]) (last cloud: F).
//
// Pure bandwidth work: the kernels are indexed by flat ELEMENT (row*D + column)
// inside a cloud, so consecutive lanes touch consecutive floats of both the
// packed and the padded image whatever D is (the reference's CUDA kernels use one
// block per cloud and a lane per ROW with an inner loop over D: 12-byte strides at
// D=3).  Grid = (element chunks, clouds) so ragged batches still fill the chip.
// packed_to_padded writes its zero padding itself (no separate memset pass).
#include "common.h"

namespace pointops {

constexpr int kPpBlock = 256;
constexpr int kPpPerThread = 4;

__global__ __launch_bounds__(kPpBlock) void packed_to_padded_kernel(
    const float* __restrict__ packed, const int64_t* __restrict__ first_idxs, int64_t F, int B,
    int64_t max_size, int64_t D, float* __restrict__ padded) {
  const int b = blockIdx.y;
  int64_t start = first_idxs[b];
  int64_t end = (b + 1 < B) ? first_idxs[b + 1] : F;
  start = start < 0 ? 0 : (start > F ? F : start);  // memory safety on inconsistent input
  end = end > F ? F : end;
  int64_t num = end - start;
  if (num < 0) num = 0;
  if (num > max_size) num = max_size;  // reference leaves this to the caller
  const int64_t valid = num * D;       // elements copied
  const int64_t total = max_size * D;  // elements written (rest zero)
  const float* __restrict__ src = packed + start * D;
  float* __restrict__ dst = padded + (int64_t)b * total;
  const int64_t base = ((int64_t)blockIdx.x * kPpBlock * kPpPerThread) + threadIdx.x;
#pragma unroll
  for (int r = 0; r < kPpPerThread; ++r) {
    const int64_t e = base + (int64_t)r * kPpBlock;
    if (e < total) dst[e] = (e < valid) ? src[e] : 0.0f;
  }
}

__global__ __launch_bounds__(kPpBlock) void padded_to_packed_kernel(
    const float* __restrict__ padded, const int64_t* __restrict__ first_idxs, int64_t F, int B,
    int64_t max_size, int64_t D, float* __restrict__ packed) {
  const int b = blockIdx.y;
  int64_t start = first_idxs[b];
  int64_t end = (b + 1 < B) ? first_idxs[b + 1] : F;
  start = start < 0 ? 0 : (start > F ? F : start);  // memory safety on inconsistent input
  end = end > F ? F : end;
  int64_t num = end - start;
  if (num < 0) num = 0;
  if (num > max_size) num = max_size;
  const int64_t valid = num * D;
  const float* __restrict__ src = padded + (int64_t)b * max_size * D;
  float* __restrict__ dst = packed + start * D;
  const int64_t base = ((int64_t)blockIdx.x * kPpBlock * kPpPerThread) + threadIdx.x;
#pragma unroll
  for (int r = 0; r < kPpPerThread; ++r) {
    const int64_t e = base + (int64_t)r * kPpBlock;
    if (e < valid) dst[e] = src[e];
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_packed_to_padded(const float* packed, const int64_t* first_idxs, int64_t F,
                                         int64_t B, int64_t max_size, int64_t D, float* padded,
                                         void* stream_) {
  POINTOPS_REQUIRE(F >= 0 && B >= 0 && max_size >= 0 && D >= 1, "packed_to_padded: bad sizes");
  POINTOPS_REQUIRE(B < 65536, "packed_to_padded: batch must be < 65536");
  if (B == 0 || max_size == 0) return POINTOPS_OK;
  const int64_t chunks = ceil_div(max_size * D, (int64_t)kPpBlock * kPpPerThread);
  POINTOPS_REQUIRE(chunks < (1LL << 31), "packed_to_padded: grid too large");
  hipLaunchKernelGGL(packed_to_padded_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(kPpBlock), 0,
                     (hipStream_t)stream_, packed, first_idxs, F, (int)B, max_size, D, padded);
  return check_launch("packed_to_padded");
}

extern "C" int pointops_padded_to_packed(const float* padded, const int64_t* first_idxs, int64_t F,
                                         int64_t B, int64_t max_size, int64_t D, float* packed,
                                         void* stream_) {
  POINTOPS_REQUIRE(F >= 0 && B >= 0 && max_size >= 0 && D >= 1, "padded_to_packed: bad sizes");
  POINTOPS_REQUIRE(B < 65536, "padded_to_packed: batch must be < 65536");
  hipStream_t stream = (hipStream_t)stream_;
  if (F * D > 0) {
    // rows owned by no cloud stay zero (reference: at::zeros, ..._cpu.cpp:52-53)
    if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)(F * D), stream) != hipSuccess)
      return check_launch("padded_to_packed(memset)");
  }
  if (B == 0 || max_size == 0 || F == 0) return POINTOPS_OK;
  const int64_t chunks = ceil_div(max_size * D, (int64_t)kPpBlock * kPpPerThread);
  POINTOPS_REQUIRE(chunks < (1LL << 31), "padded_to_packed: grid too large");
  hipLaunchKernelGGL(padded_to_packed_kernel, dim3((unsigned)chunks, (unsigned)B), dim3(kPpBlock), 0,
                     stream, padded, first_idxs, F, (int)B, max_size, D, packed);
  return check_launch("padded_to_packed");
}
