// fps.hip -- iterative farthest point sampling for gfx950.
//
// Replaces FarthestPointSampling (reference:
// csrc/sample_farthest_points/sample_farthest_points.h:55-76) with the CPU path's
// semantics (sample_farthest_points_cpu.cpp:14-103): idx[n][0] = start_idxs[n];
// every further sample is the FIRST index with the largest running min-distance
// to the selected set (std::max_element), unfused fp32 distances, -1 padding
// beyond min(lengths[n], K[n]).
//
// v1 layout: one workgroup of 1024 lanes (16 wave64) per cloud; the running
// min-distance array lives in HBM/L2 (`min_dist_ws`, N*P floats), each lane owns
// the points p = tid, tid+1024, ... so no lane ever reads another lane's entry.
// Per iteration: update + per-lane argmax (strict > keeps the lowest index),
// wave argmax by xor-butterfly on (value, index) keys preferring the LOWER index on
// ties, one LDS exchange across the 16 waves, broadcast of the winner.
#include <float.h>

#include "common.h"

namespace pointops {

constexpr int kFpsBlock = 1024;
constexpr int kFpsWaves = kFpsBlock / kWave;

__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
  // larger value wins; on equal values the lower index wins (first maximum)
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

template <int DT>
__global__ __launch_bounds__(kFpsBlock) void fps_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ lengths,
    const int64_t* __restrict__ Ks, const int64_t* __restrict__ start_idxs, int P, int Drt,
    int max_K, int64_t* __restrict__ idxs, float* __restrict__ min_dist) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  int len = (int)lengths[n];
  if (len > P) len = P;
  int64_t kn64 = Ks[n];
  int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
  if (kn > max_K) kn = max_K;
  if (kn < 0) kn = 0;
  int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
  // -1 padding beyond the samples this cloud produces
  for (int k = (kn > 0 ? kn : 0) + tid; k < max_K; k += kFpsBlock) out[k] = -1;
  if (len <= 0 || kn <= 0) return;

  const float* __restrict__ pts = points + (int64_t)n * P * D;
  float* __restrict__ md = min_dist + (int64_t)n * P;
  for (int p = tid; p < len; p += kFpsBlock) md[p] = FLT_MAX;

  __shared__ float s_val[kFpsWaves];
  __shared__ int s_idx[kFpsWaves];
  __shared__ int s_last;

  int last = (int)start_idxs[n];
  if (last < 0 || last >= len) last = 0;  // guard (reference: undefined behaviour)
  if (tid == 0) out[0] = last;

  for (int k = 1; k < kn; ++k) {
    float best = -1.0f;
    int besti = 0x7fffffff;
    if constexpr (DT > 0) {
      float c[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) c[d] = pts[(int64_t)last * DT + d];  // wave-uniform
      for (int p = tid; p < len; p += kFpsBlock) {
        float acc;
        {
          const float diff = c[0] - pts[(int64_t)p * DT];
          acc = diff * diff;
        }
#pragma unroll
        for (int d = 1; d < DT; ++d) {
          const float diff = c[d] - pts[(int64_t)p * DT + d];
          acc = acc + diff * diff;
        }
        float m = md[p];
        if (acc < m) {
          m = acc;
          md[p] = m;
        }
        if (m > best) {
          best = m;
          besti = p;
        }
      }
    } else {
      const float* __restrict__ c = pts + (int64_t)last * D;
      for (int p = tid; p < len; p += kFpsBlock) {
        const float* __restrict__ b = pts + (int64_t)p * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float diff = c[d] - b[d];
          acc = acc + diff * diff;
        }
        float m = md[p];
        if (acc < m) {
          m = acc;
          md[p] = m;
        }
        if (m > best) {
          best = m;
          besti = p;
        }
      }
    }
    // wave64 argmax butterfly
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, kWave);
      const int oi = __shfl_xor(besti, off, kWave);
      argmax_combine(best, besti, ov, oi);
    }
    if (lane == 0) {
      s_val[wave] = best;
      s_idx[wave] = besti;
    }
    __syncthreads();
    if (wave == 0) {
      float v = lane < kFpsWaves ? s_val[lane] : -2.0f;
      int ix = lane < kFpsWaves ? s_idx[lane] : 0x7fffffff;
#pragma unroll
      for (int off = kFpsWaves / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, kWave);
        const int oi = __shfl_xor(ix, off, kWave);
        argmax_combine(v, ix, ov, oi);
      }
      if (lane == 0) {
        s_last = ix;
        out[k] = ix;
      }
    }
    __syncthreads();
    last = s_last;
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_sample_farthest_points(const float* points, const int64_t* lengths,
                                               const int64_t* K, const int64_t* start_idxs,
                                               int64_t N, int64_t P, int64_t D, int64_t max_K,
                                               int64_t* idxs, float* min_dist_ws, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && D >= 1 && max_K >= 0, "sample_farthest_points: bad sizes");
  POINTOPS_REQUIRE(P < (1LL << 31) && max_K < (1LL << 31) && D < (1LL << 16) && N < (1LL << 31),
                   "sample_farthest_points: sizes must fit int32");
  if (N == 0 || max_K == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(P == 0 || min_dist_ws != nullptr, "sample_farthest_points: workspace is null");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((unsigned)N), block(kFpsBlock);
#define PO_LAUNCH(DT)                                                                             \
  hipLaunchKernelGGL((fps_kernel<DT>), grid, block, 0, stream, points, lengths, K, start_idxs,     \
                     (int)P, (int)D, (int)max_K, idxs, min_dist_ws)
  switch (D) {
    case 2: PO_LAUNCH(2); break;
    case 3: PO_LAUNCH(3); break;
    default: PO_LAUNCH(0); break;
  }
#undef PO_LAUNCH
  return check_launch("sample_farthest_points");
}
