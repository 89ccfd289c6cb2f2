// knn_grid_wsort.hip -- exact grid-pruned KNN for LONG lists, 64 < K <= 128 (gfx950).
//
// The lane searches of knn_grid_search.h keep a query's list in one lane's registers: 64 keys are the end of that road.
// Here a WAVE owns a query: the ~27 x 0.4 K candidates of the 3x3x3 cell cube around the query's cell (cells sized for
// 0.4 K points, as for the shorter lists) are spread over the 64 lanes, 32 keys each -- 2048 64-bit (dist, idx) keys in
// 64 VGPRs per lane -- and ONE bitonic sort of the whole wave (keys through the FP64 pipe: sort_net.h) puts the K
// smallest into the first lanes, in output order.  The flip form of the network needs no per-element direction: the
// first step of every merge compares element i with i ^ (k - 1), all later ones i with i ^ j, always "smaller index
// keeps the minimum"; in-lane steps (j < 32) are register pairs, cross-lane steps exchange a register with lane ^ m.
// Certification is the lane search's: K-th distance < the rigorous bound of everything outside the cube.  Queries whose
// cube holds more than 2048 records, or whose K-th neighbour is not certified, go to the list of the all-pairs scan
// (knn_wide.hip) -- ~1 % of a uniform cloud.  Round 2 sent every K > 64 to that scan: ~9e10 pairs for one
// 300 000-point cloud.
#include "debug.h"
#include "grid.h"
#include "knn_common.h"
#include "sort_net.h"

namespace pointops {

constexpr int kWsKeys = 32;   // keys per lane
constexpr int kWsCap = kWsKeys * kGridWave;
static_assert(kWsCap == 1 << 11, "ws_sort is written for 2048 keys");
constexpr int kWsBlock = 256;  // four waves = four queries per workgroup pass

// cross-lane compare-exchange of one register with lane ^ m: the lower lane keeps the minimum
__device__ __forceinline__ double ws_cross(double v, int m, bool keep_min) {
  const int hi = __shfl_xor(__double2hiint(v), m, kGridWave);
  const int lo = __shfl_xor(__double2loint(v), m, kGridWave);
  const double o = __hiloint2double(hi, lo);
  const double mn = kmin(v, o), mx = kmax(v, o);
  return keep_min ? mn : mx;
}

// Ascending bitonic sort of the wave's 2048 keys; element i lives in lane i >> 5, register i & 31.  Every step is its own
// template instance (a loop over the levels is "too large to unroll", and a run-time level indexes the registers
// dynamically: the whole array then lives in scratch).
template <int K>  // flip step of the merge of blocks of K: i with i ^ (K - 1)
__device__ __forceinline__ void ws_flip(double (&a)[kWsKeys], int lane) {
  if constexpr (K <= kWsKeys) {
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) {
      constexpr int dummy = 0;
      (void)dummy;
      const int v = u ^ (K - 1);
      if (v > u) key_ce(a[u], a[v], true);
    }
  } else {
    constexpr int m = (K >> 5) - 1;  // lane mask; the register index mirrors: u <-> 31 - u
    const bool keep_min = (lane & ((m + 1) >> 1)) == 0;  // the lower lane of the pair (top bit of m clear)
    double t[kWsKeys];
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) {
      // my register u meets register 31 - u of lane ^ m: both lanes send their register 31 - u
      const int hi = __shfl_xor(__double2hiint(a[kWsKeys - 1 - u]), m, kGridWave);
      const int lo = __shfl_xor(__double2loint(a[kWsKeys - 1 - u]), m, kGridWave);
      const double o = __hiloint2double(hi, lo);
      t[u] = keep_min ? kmin(a[u], o) : kmax(a[u], o);
    }
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) a[u] = t[u];
  }
}

template <int J>  // merge step: i with i ^ J
__device__ __forceinline__ void ws_step(double (&a)[kWsKeys], int lane) {
  if constexpr (J >= kWsKeys) {
    constexpr int m = J >> 5;
    const bool keep_min = (lane & m) == 0;
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) a[u] = ws_cross(a[u], m, keep_min);
  } else {
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) {
      const int v = u ^ J;
      if (v > u) key_ce(a[u], a[v], true);
    }
  }
}

template <int LJ>
__device__ __forceinline__ void ws_merge(double (&a)[kWsKeys], int lane) {
  if constexpr (LJ >= 0) {
    ws_step<(1 << LJ)>(a, lane);
    ws_merge<LJ - 1>(a, lane);
  }
}

template <int LK>
__device__ __forceinline__ void ws_levels(double (&a)[kWsKeys], int lane) {
  if constexpr (LK <= 11) {
    ws_flip<(1 << LK)>(a, lane);
    ws_merge<LK - 2>(a, lane);
    ws_levels<LK + 1>(a, lane);
  }
}

__device__ __forceinline__ void ws_sort(double (&a)[kWsKeys], int lane) { ws_levels<1>(a, lane); }

constexpr int kWsMaxRows = 25;  // rows of a radius-2 cube
constexpr int kWsMaxStream = 8 * kWsCap;  // records of a cube beyond which the query goes to the all-pairs list

// (two waves per SIMD: 64 key registers + 64 for the flip step's exchange; with the default bound the compiler held the
// kernel to 160 registers and put 68 dwords into scratch)
template <int D, int NORM>
__global__ __launch_bounds__(kWsBlock, 2) void knn_grid_wsort_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const float* __restrict__ edges,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted, int* __restrict__ fb2_count,
    int* __restrict__ fb2_list, int cell_cap, int P1, int P2, int K, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  // the wave's candidate STREAM: record numbers of the cube's rows one after the other; lane l then takes stream
  // positions l, l + 64, ... (written row by row with coalesced stores, read back 32 per lane)
  __shared__ int s_stream[kWsBlock / kGridWave][kWsCap];
  const int n = blockIdx.y;
  const GridCloud g = clouds[n];
  if (!g.use_grid) return;  // (the build pass has listed this cloud's queries for the all-pairs scan)
  const int lane = threadIdx.x & (kGridWave - 1);
  const int wslot = threadIdx.x / kGridWave;
  const int wave = blockIdx.x * (kWsBlock / kGridWave) + wslot;
  const int waves = gridDim.x * (kWsBlock / kGridWave);
  const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
  const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
  const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + kSortedPad);
  const int kvalid = g.len2 < K ? g.len2 : K;
  int* const stream = s_stream[wslot];

  for (int i = wave; i < g.len1; i += waves) {  // wave-uniform
    const int64_t row = (int64_t)n * P1 + i;
    float qx, qy, qz;
    load_point3<D>(p1 + row * D, qx, qy, qz);
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    double a[kWsKeys];
#pragma unroll
    for (int u = 0; u < kWsKeys; ++u) a[u] = TopKF64<1>::empty();  // (written on every path: a conditionally written
                                                                   // array goes through scratch)
    bool ok = false;
    // radius 1, then -- for the queries at the edges and corners of a cloud, whose cubes are clipped and whose K-th
    // neighbour lies beyond the first ring -- radius 2, while the cube's records fit the sort
    for (int r = 1; r <= 2 && !ok; ++r) {
      const int X0 = max(cx - r, 0), X1 = min(cx + r, g.G[0] - 1);
      const int Y0 = max(cy - r, 0), Y1 = min(cy + r, g.G[1] - 1);
      const int Z0 = max(cz - r, 0), Z1 = min(cz + r, g.G[2] - 1);
      bool whole;
      const float lb = box_lower_bound<NORM>(g, ed, qx, qy, qz, X0, X1, Y0, Y1, Z0, Z1, whole);
      const int ny = Y1 - Y0 + 1, nrows = ny * (Z1 - Z0 + 1);  // <= 25
      int src = 0, len = 0;
      if (lane < nrows) {
        const int y = Y0 + lane % ny, z = Z0 + lane / ny;
        const int rowbase = (z * g.G[1] + y) * g.G[0];
        src = cstart[rowbase + X0];
        len = cstart[rowbase + X1 + 1] - src;
      }
      int inc = len;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const int v = __shfl_up(inc, o, kGridWave);
        if (lane >= o) inc += v;
      }
      const int T = __shfl(inc, kWsMaxRows - 1, kGridWave);  // (lanes >= nrows carry the total on)
      if (T > kWsMaxStream) break;  // (a cluster: the all-pairs list is cheaper than dozens of sorts)
      // The stream is consumed in CHUNKS: the first fills all 2048 key slots; every further one (cubes with more
      // records than the sort holds: faces of a cloud at radius 2, dense regions) replaces the upper half -- the 1024
      // largest keys of the previous sort, lanes 32..63 -- and the wave sorts again: the lower half always holds the
      // 1024 smallest keys so far, and K <= 128 of them are wanted.
      for (int c0 = 0; c0 < T || c0 == 0; c0 += (c0 == 0 ? kWsCap : kWsCap / 2)) {
        const int cap = c0 == 0 ? kWsCap : kWsCap / 2;
        for (int rr = 0; rr < nrows; ++rr) {  // wave-uniform: the chunk's part of every row, coalesced stores
          const int s0 = __shfl(src, rr, kGridWave), l0 = __shfl(len, rr, kGridWave), p0 = __shfl(inc - len, rr, kGridWave);
          const int lo = max(p0, c0), hi = min(p0 + l0, c0 + cap);
          for (int l = lo + lane; l < hi; l += kGridWave) stream[l - c0] = s0 + (l - p0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (one wave writes and reads its own stream: LDS is in order)
        __builtin_amdgcn_wave_barrier();
        // slot (lane, u) of the keys takes chunk position u * 64 + lane (first chunk) or u * 32 + lane - 32 (lanes >= 32)
        const bool loads = c0 == 0 || lane >= kGridWave / 2;
        const int t0 = c0 == 0 ? lane : lane - kGridWave / 2, tstep = c0 == 0 ? kGridWave : kGridWave / 2;
#pragma unroll
        for (int u0 = 0; u0 < kWsKeys; u0 += 8) {
          float4 c[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int t = (u0 + u) * tstep + t0;
            c[u] = sp[(loads && c0 + t < T) ? stream[t] : P2];  // (past the stream: the cloud's NaN sentinel record)
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int t = (u0 + u) * tstep + t0;
            const float dd = point_dist<D, NORM>(qx, qy, qz, c[u]);
            const double key = c0 + t < T ? TopKF64<1>::make(dd, __float_as_int(c[u].w)) : TopKF64<1>::empty();
            a[u0 + u] = loads ? key : a[u0 + u];
          }
        }
        __builtin_amdgcn_wave_barrier();
        ws_sort(a, lane);
      }
      // the K-th best: element K - 1 = lane (K - 1) >> 5, register (K - 1) & 31
      double kth = a[0];
#pragma unroll
      for (int u = 1; u < kWsKeys; ++u) kth = (u == ((K - 1) & (kWsKeys - 1))) ? a[u] : kth;
      const unsigned kth_bits = (unsigned)__shfl(__double2hiint(kth), (K - 1) >> 5, kGridWave);
      const bool full = kvalid == K && kth_bits < 0x7f800000u;
      ok = whole || (full && __uint_as_float(kth_bits) < lb);
    }
    if (ok) {
      if (lane * kWsKeys < K) {
        int64_t* __restrict__ oi = idxs + row * K + lane * kWsKeys;
        float* __restrict__ od = dists + row * K + lane * kWsKeys;
#pragma unroll
        for (int u = 0; u < kWsKeys; ++u) {
          const int k = lane * kWsKeys + u;
          if (k < K) {
            const bool has = k < kvalid;
            oi[u] = has ? (int64_t)__double2loint(a[u]) : 0;
            od[u] = has ? __int_as_float(__double2hiint(a[u])) : 0.0f;
          }
        }
      }
    } else if (lane == 0) {
      const int p = atomicAdd(fb2_count + n, 1);
      fb2_list[(int64_t)n * P1 + p] = i;
    }
  }
}

void grid_search_wsort(const KnnArgs& a, const GridWs& ws, int norm) {
  int64_t wx = ceil_div(a.P1, kWsBlock / kGridWave);
  const int64_t cap = (int64_t)(65536 / (a.N > 0 ? a.N : 1));  // ~16 resident waves per SIMD-slot's worth of queries per cloud
  wx = wx > cap ? (cap < 64 ? 64 : cap) : wx;
  const dim3 grid((unsigned)wx, (unsigned)a.N);
#define PO_WS(DD, NN)                                                                                               \
  hipLaunchKernelGGL((knn_grid_wsort_kernel<DD, NN>), grid, dim3(kWsBlock), 0, a.stream, a.p1, (const GridCloud*)ws.cloud, \
                     (const float*)ws.edges, (const int*)ws.cell_start, (const float4*)ws.sorted, ws.fb2_count,      \
                     ws.fb2_list, ws.cell_cap, a.P1, a.P2, a.K, a.idxs, a.dists)
  if (norm == 1) {
    if (a.D == 1) PO_WS(1, 1);
    else if (a.D == 2) PO_WS(2, 1);
    else PO_WS(3, 1);
  } else {
    if (a.D == 1) PO_WS(1, 2);
    else if (a.D == 2) PO_WS(2, 2);
    else PO_WS(3, 2);
  }
#undef PO_WS
}

}  // namespace pointops
