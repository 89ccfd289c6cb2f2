// grid_refine.hip -- REFINED CELLS: the second level of the cell grid (gfx950).
//
// One cell size per cloud cannot serve a cloud whose density varies by orders of magnitude: a cell that holds
// thousands of points (a cluster, the dense side of a density gradient) makes each of its queries walk thousands
// of candidates in the lane search ("Non-uniform clouds").  After the level-0 counting sort
// every cell with more than refine_threshold() points gets a sub-grid of s^3 sub-cells, s = ceil(cbrt(count /
// target)) <= 32, over the ROBUST extent of its points (mean +- 2.2 sigma per dimension: a 1e-3 cluster inside a
// 0.05 cell is resolved by ONE level; outliers land in the clamped boundary sub-cells), and its records are
// re-sorted by sub-cell inside the cell's own range of the sorted array (level-0 searches never notice).  The box
// search (knn_grid_box.h) walks sub-cell runs of refined cells and whole runs of unrefined ones.
//
//   (detection)    grid_build's scan pass marks the over-full cells and hands out descriptors and sub_start tables;
//   refine_build   one workgroup per refined cell (persistent over the list): mean / sigma, sub-cell histogram in
//                  LDS, exclusive scan -> sub_start table, scatter into a scratch copy, copy back.
// Clouds without over-full cells cost one empty launch.
#include "grid.h"

namespace pointops {

constexpr int kRefineBlock = 1024;
constexpr int kRefineWgs = 8;  // workgroups per cloud walking the cloud's list of refined cells (each claims 128 KB
                               // of LDS: 256 of them are one round of the chip, and an empty list costs one round)

// fn(record, i) for the `count` records behind `src`, the workgroup's threads striding over them with FOUR loads in
// flight each (a cell of 75 000 points is 73 strides of one workgroup: one load at a time made every pass ~50 us)
template <typename Fn>
__device__ __forceinline__ void for_each4(const float4* __restrict__ src, int count, Fn fn) {
  for (int i0 = threadIdx.x; i0 < count; i0 += 4 * kRefineBlock) {
    float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = i0 + u * kRefineBlock;
      if (j < count) p[u] = src[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = i0 + u * kRefineBlock;
      if (j < count) fn(p[u], j);
    }
  }
}

template <int D>
__global__ __launch_bounds__(kRefineBlock) void grid_refine_build_kernel(GridWs ws, int P2) {
  extern __shared__ int s_hist[];  // s^3 counters of the cell being built
  __shared__ double s_red[kRefineBlock / kWave][6];
  __shared__ float s_lo[3], s_scale[3];
  __shared__ int s_wsum[kRefineBlock / kWave];
  const int n = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  int nref = ws.rcount[n];
  if (nref > ws.rdesc_cap) nref = ws.rdesc_cap;
  float4* __restrict__ rec = ws.sorted + (int64_t)n * (P2 + kSortedPad);
  float4* __restrict__ tmp = ws.sorted_tmp + (int64_t)n * P2;
  for (int r = blockIdx.x; r < nref; r += gridDim.x) {
    RefinedCell* __restrict__ dp = ws.rdesc + (int64_t)n * ws.rdesc_cap + r;
    const int start = dp->start, count = dp->count, s = dp->s;
    if (count <= 0) continue;  // (wave-uniform) descriptor without a table
    const int nsub = s * s * s;
    // 1. mean and sigma of the cell's points (fp64 sums: exactness is not needed, robustness is)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for_each4(rec + start, count, [&](const float4 p, int) {
      const double v[3] = {p.x, p.y, p.z};
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc[d] += v[d];
        acc[3 + d] += v[d] * v[d];
      }
    });
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, kWave);
    }
    __syncthreads();  // (previous cell's users of the shared arrays are done)
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) s_red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 3) {
      float lo = 0.0f, scale = 0.0f;
      if (tid < D) {
        double sx = 0, sxx = 0;
        for (int w = 0; w < kRefineBlock / kWave; ++w) {
          sx += s_red[w][tid];
          sxx += s_red[w][3 + tid];
        }
        const double mean = sx / count;
        const double var = sxx / count - mean * mean;
        const double sig = var > 0 ? sqrt(var) : 0.0;
        const double half = 2.2 * sig;
        lo = (float)(mean - half);
        const float width = (float)(2.0 * half);
        scale = (width > 0.0f && width <= FLT_MAX) ? (float)s / width : 0.0f;
        if (!(scale > 0.0f && scale <= FLT_MAX)) scale = 0.0f;  // degenerate dimension: every point in sub-cell 0
        if (!(fabsf(lo) <= FLT_MAX)) {
          lo = 0.0f;
          scale = 0.0f;
        }
      }
      s_lo[tid] = lo;
      s_scale[tid] = scale;
      dp->lo[tid] = lo;
      dp->scale[tid] = scale;
    }
    for (int b = tid; b < nsub; b += kRefineBlock) s_hist[b] = 0;
    __syncthreads();
    const float lx = s_lo[0], ly = s_lo[1], lz = s_lo[2], kx = s_scale[0], ky = s_scale[1], kz = s_scale[2];
    auto sub_index = [&](const float4 p) {
      const int ix = sub_of(p.x, lx, kx, s);
      const int iy = D > 1 ? sub_of(p.y, ly, ky, s) : 0;
      const int iz = D > 2 ? sub_of(p.z, lz, kz, s) : 0;
      return (iz * s + iy) * s + ix;
    };
    // 2. histogram
    for_each4(rec + start, count, [&](const float4 p, int) { atomicAdd(&s_hist[sub_index(p)], 1); });
    __syncthreads();
    // 3. exclusive scan of nsub counters (each thread a contiguous slice) -> table in the pool, cursors in LDS
    int* __restrict__ table = ws.pool + (int64_t)n * ws.pool_cap + dp->pool_off;
    const int per = (nsub + kRefineBlock - 1) / kRefineBlock;
    const int b0 = tid * per, b1 = min(b0 + per, nsub);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += s_hist[b];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int v = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += v;
    }
    if (lane == kWave - 1) s_wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wsum[w];
    for (int b = b0; b < b1; ++b) {
      const int cnt = s_hist[b];
      table[b] = run;
      s_hist[b] = run;  // scatter cursor
      run += cnt;
    }
    if (tid == kRefineBlock - 1) table[nsub] = count;
    __syncthreads();
    // 4. scatter into the scratch copy, then back into the cell's range
    for_each4(rec + start, count, [&](const float4 p, int) { tmp[start + atomicAdd(&s_hist[sub_index(p)], 1)] = p; });
    __threadfence_block();
    __syncthreads();
    for_each4(tmp + start, count, [&](const float4 p, int i) { rec[start + i] = p; });
  }
}

int grid_refine(const KnnArgs& a, const GridWs& ws) {
  const size_t lds = sizeof(int) * (size_t)kRefineMaxS * kRefineMaxS * kRefineMaxS;  // 128 KB: one workgroup per CU
  // (a workgroup claims 128 KB of LDS = one CU: up to 256 of them chip-wide, at least kRefineWgs per cloud)
  const int64_t per_cloud = 256 / (a.N > 0 ? a.N : 1);
  const dim3 grid((unsigned)(per_cloud < kRefineWgs ? kRefineWgs : (per_cloud > 64 ? 64 : per_cloud)), (unsigned)a.N);
#define PO_REFINE(DD)                                                                                            \
  {                                                                                                              \
    static bool attr = false;                                                                                    \
    if (!attr) {                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(grid_refine_build_kernel<DD>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)               \
        return check_launch("grid refine attribute");                                                            \
      attr = true;                                                                                               \
    }                                                                                                            \
    hipLaunchKernelGGL((grid_refine_build_kernel<DD>), grid, dim3(kRefineBlock), lds, a.stream, ws, a.P2);       \
  }
  switch (a.D) {
    case 1: PO_REFINE(1); break;
    case 2: PO_REFINE(2); break;
    default: PO_REFINE(3); break;
  }
#undef PO_REFINE
  return check_launch("grid refine");
}

}  // namespace pointops
