// ball_small.hip -- first-K-within-radius for FEW queries (gfx950): one wave per query.
//
// Semantics: the CPU path of the reference (csrc/ball_query/ball_query_cpu.cpp:12-54), as ball_query.hip.
//
// The scan of ball_query.hip gives every lane a query: 2 x 1024 queries are 32 waves on 1024 SIMDs, and a wave
// keeps scanning until the LAST of its 64 queries is full (59 us for B=2, N=1024, K=32, r=0.2; 173 us at N=4096).
// Here a wave owns one query (wave-uniform: scalar loads, SGPR operands) and tests 64 consecutive candidates per
// step, one per lane: the hits of a step are a ballot, a hit's output slot is the running count plus the number
// of hit lanes below it -- index order by construction --, and the wave stops at the step that fills the row.
// Padding (-1 / 0) is written by the same wave.  No workspace.
#include "common.h"
#include "debug.h"
#include "knn_grid.h"

#include <algorithm>

namespace pointops {

constexpr int kBsWaves = 4;   // waves per workgroup (independent of each other)
constexpr int kBsUnroll = 4;  // candidates per lane in flight

template <int DT>
__global__ __launch_bounds__(kBsWaves * 64) void ball_small_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int Drt, int K, float radius2, int qw, int waves_per_cloud,
    int total_waves, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  const int D = DT > 0 ? DT : Drt;
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kBsWaves + (threadIdx.x >> 6)));
  if (g >= total_waves) return;
  const int n = g / waves_per_cloud, t = g - n * waves_per_cloud;
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  const int len1 = (int)lengths1[n];
  const float* __restrict__ cloud = p2 + (int64_t)n * P2 * D;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int iend = min((t + 1) * qw, P1);
  for (int i = t * qw; i < iend; ++i) {  // wave-uniform
    const int64_t row = (int64_t)n * P1 + i;
    int64_t* __restrict__ orow_i = idxs + row * K;
    float* __restrict__ orow_d = dists + row * K;
    int count = 0;  // wave-uniform
    if (i < len1) {
      if constexpr (DT > 0) {
        float a[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) a[d] = p1[row * DT + d];  // wave-uniform address: scalar loads
        for (int j0 = 0; j0 < len2 && count < K; j0 += 64 * kBsUnroll) {
          float c[kBsUnroll][DT];
#pragma unroll
          for (int u = 0; u < kBsUnroll; ++u) {
            const int j = min(j0 + u * 64 + lane, len2 - 1);
#pragma unroll
            for (int d = 0; d < DT; ++d) c[u][d] = cloud[(int64_t)j * DT + d];
          }
#pragma unroll
          for (int u = 0; u < kBsUnroll; ++u) {
            const int j = j0 + u * 64 + lane;
            float acc;
            {
              const float diff = a[0] - c[u][0];
              acc = diff * diff;
            }
#pragma unroll
            for (int d = 1; d < DT; ++d) {
              const float diff = a[d] - c[u][d];
              acc = acc + diff * diff;
            }
            const bool hit = j < len2 && acc < radius2;
            const unsigned long long mask = __ballot(hit);
            const int pos = count + __popcll(mask & below);
            if (hit && pos < K) {
              orow_i[pos] = j;
              orow_d[pos] = acc;
            }
            count += __popcll(mask);
          }
        }
      } else {
        const float* __restrict__ a = p1 + row * D;
        for (int j0 = 0; j0 < len2 && count < K; j0 += 64) {
          const int j = j0 + lane;
          const float* __restrict__ b = cloud + (int64_t)min(j, len2 - 1) * D;
          float acc = 0.0f;
          for (int d = 0; d < D; ++d) {
            const float diff = a[d] - b[d];
            acc = acc + diff * diff;
          }
          const bool hit = j < len2 && acc < radius2;
          const unsigned long long mask = __ballot(hit);
          const int pos = count + __popcll(mask & below);
          if (hit && pos < K) {
            orow_i[pos] = j;
            orow_d[pos] = acc;
          }
          count += __popcll(mask);
        }
      }
    }
    for (int k = min(count, K) + lane; k < K; k += 64) {  // padding, also of rows >= lengths1[n]
      orow_i[k] = -1;
      orow_d[k] = 0.0f;
    }
  }
}

// Few queries: fewer query WAVES of the lane-per-query scan than the chip has SIMDs x 2.  POINTOPS_DEBUG
// ball_small=0 keeps that scan, =1 takes every batch (tests).
bool ball_small_applies(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K) {
  if (N * P1 >= (1LL << 31)) return false;
  const long knob = debug_knob("ball_small", -1);
  if (knob == 0) return false;
  if (knob == 1) return true;
  return N * ceil_div(P1, (int64_t)64) < 2048;
}

void launch_ball_small(const float* p1, const float* p2, const int64_t* lengths1, const int64_t* lengths2, int64_t N,
                       int64_t P1, int64_t P2, int64_t D, int64_t K, float radius2, int64_t* idxs, float* dists,
                       hipStream_t stream) {
  const int64_t tq = N * P1;
  const int qw = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(tq, (int64_t)16384), 1), 64);
  const int64_t wpc = ceil_div(P1, (int64_t)qw), total = N * wpc;
  const dim3 grid((unsigned)ceil_div(total, (int64_t)kBsWaves));
#define PO_BS(DT)                                                                                              \
  hipLaunchKernelGGL((ball_small_kernel<DT>), grid, dim3(kBsWaves * 64), 0, stream, p1, p2, lengths1, lengths2, \
                     (int)P1, (int)P2, (int)D, (int)K, radius2, qw, (int)wpc, (int)total, idxs, dists)
  switch (D) {
    case 1: PO_BS(1); break;
    case 2: PO_BS(2); break;
    case 3: PO_BS(3); break;
    case 4: PO_BS(4); break;
    default: PO_BS(0); break;
  }
#undef PO_BS
}

}  // namespace pointops
