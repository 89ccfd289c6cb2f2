// knn_grid.hip -- EXACT grid-pruned K nearest neighbours for D <= 3 on gfx950.
//
// Same results, bit for bit, as the all-pairs scan (knn.hip) and therefore as the
// reference CPU kernel (csrc/knn/knn_cpu.cpp:13-69): every candidate distance is the
// same unfused fp32 expression, the selection key is the same lexicographic
// (dist, idx), and a query is only answered from a pruned candidate set when a
// RIGOROUS fp32 lower bound proves that no unvisited point can enter its top-K;
// every other query is handed to the brute-force scan.  This is SURVEY.md section 8
// row f1 -- the only route from the VALU floor of the all-pairs scan (~16 ms at
// B=32, N=M=65536, K=16) towards the HBM floor (72 us).
//
// Passes (all clouds of the batch in every launch, no host synchronisation):
//   0 grid_bbox      bounding box of every p2 cloud (ordered-uint atomicMin/Max);
//   1 grid_setup     per cloud: cubic cell size h for ~0.4 K points per cell (>= 1), G = cells
//                    per dimension, and per-dimension EDGE TABLES
//                    E_d[c] = min{ x in [lo,hi] : cell_d(x) >= c }, found by bisection over
//                    the ordered fp32 bit patterns of the (monotone) cell function
//                    itself -- no error analysis of the binning arithmetic is needed;
//   2 grid_bin<cnt>  histograms: p2 points per cell, p1 queries per BLOCK of B^3 cells
//                    (LDS-aggregated, one global atomic per non-empty bin and tile); zero
//                    rows for padded queries;
//   3 grid_scan_*    chunked exclusive scans -> cell_start / blk_start;
//   4 grid_bin<sct>  counting-sort p2 into (x,y,z,idx) float4 records, queries into
//                    per-block lists;
//   5 knn_grid       persistent wave64 workgroups walk (block, 64-query chunk) slots: one
//                    query per lane; the records of the block's cells plus a one-cell halo
//                    form a flat stream of contiguous runs that is staged tile by tile in
//                    LDS and read back with wave-uniform (broadcast) ds_reads; candidates
//                    that beat a lane's threshold are parked in per-lane LDS queues and merged
//                    into the sorted register lists by sorting networks.  Afterwards each
//                    lane checks kth_dist < LB, LB = min over the region's faces of the bound
//                    below; on failure the query id goes to the fallback list;
//   6 knn_grid_wave  wave-per-query search of a cell cube that doubles until certified;
//   7 knn_reg_kernel whole-cloud scan (knn.hip) for the queries pass 6 gave up on and for
//                    clouds without a usable grid.
//
// Lower bound.  Let the visited region be cells [X0..X1]x[Y0..Y1]x[Z0..Z1].  A point in
// an unvisited cell has, in some dimension d, cell_d < X0 or cell_d > X1.  cell_d is
// monotone in the coordinate, so p_d <= prev(E_d[X0]) =: f or p_d >= E_d[X1+1] =: f.  fp32
// subtraction and multiplication by itself are monotone, so the COMPUTED |q_d - p_d| is
// >= fl(|q_d - f|) and the computed square >= fl(fl(|q_d - f|)^2); adding the other
// (non-negative) terms and rounding cannot go below that.  Hence the computed distance
// of every unvisited point is >= LB, and `kth < LB` (strict) also rules out ties.
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "knn_common.h"
#include "sort_net.h"
#include "knn_grid.h"

namespace pointops {

constexpr int kGMax = 1024;          // cells per dimension cap (edge table size)
constexpr int kEdgeStride = kGMax + 2;
constexpr int kSetupBlock = 1024;  // 3 x 1026 edge bisections per cloud
constexpr int kScanBlock = 1024;
constexpr int kGridWave = 64;
constexpr int kNumXcd = 8;  // MI355X: 8 XCDs x 32 CUs, private 4 MB L2 each
constexpr int kBinLdsBinsSetup = 40000;  // = kBinLdsBins of the binning pass (defined with it below)

struct GridCloud {
  float lo[3];
  float inv_h;
  int G[3];
  int NB[3];
  int ncell, nblock;
  int len1, len2;
  int use_grid;
  int B;
};

struct GridWs {
  GridCloud* cloud;   // N
  int* block_prefix;  // N + 1
  float* edges;       // N * 3 * kEdgeStride
  int* cell_count;    // N * cell_cap   histogram, then scatter cursor
  int* cell_start;    // N * (cell_cap + 1)
  float4* sorted;     // N * (P2 + 1)   (x, y, z, idx bits); record P2 of every cloud is a NaN sentinel that
                      //                exhausted lanes of the lane-private searches keep loading (never a candidate)
  int* blk_count;     // N * cell_cap
  int* blk_start;     // N * (cell_cap + 1)
  int* qlist;         // N * P1         query ids grouped by block
  int* fb_count;      // N          queries the block search could not certify
  int* fb_list;       // N * P1
  unsigned* fb_kth;   // N * P1     estimated KC-th distance (fp32 bits) of an uncertified query: picks the quad pass's cube
  int* fb2_count;     // N          queries the expanding search gave up on (whole-cloud scan)
  int* fb2_list;      // N * P1
  int* fb3_count;     // N          queries the radius-2 quad search could not certify (expanding search)
  int* fb3_list;      // N * P1
  unsigned* bbox;     // N * 8: ordered-uint keys of min x,y,z (atomicMin) and max x,y,z (atomicMax)
  int* scan_partial;  // N * 2 * ceil(cell_cap / 4096): per-chunk sums / offsets of the two scans
  int* rank1;         // N * P1     rank of a query / point inside its bin (many-bin clouds only)
  int* rank2;         // N * P2
  int* grid_flag;     // N          1 = the cloud was searched through its grid (ball query: scan only the list)
  int cell_cap;
  int ball;           // 0 = KNN (pad rows with idx 0), 1 = ball query (pad with idx -1; clouds without a
                      //     usable grid are left to the scan kernel instead of the query list)
};

// ---------------------------------------------------------------------------
// monotone cell function and ordered fp32 keys
// ---------------------------------------------------------------------------
__device__ __forceinline__ int cell_of(float x, float lo, float inv_h, int G) {
  const float t = (x - lo) * inv_h;  // unfused; monotone non-decreasing in x
  int c = (t < (float)G) ? (int)t : G - 1;
  if (!(t >= 0.0f)) c = 0;  // below the box, or NaN
  return c;
}
__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ float prev_float(float x) { return funkey(fkey(x) - 1u); }

// smallest x in [lo, hi] with cell_of(x) >= c, +inf if none
__device__ float edge_bisect(int c, float lo, float hi, float inv_h, int G) {
  if (c <= 0) return lo;
  if (cell_of(hi, lo, inv_h, G) < c) return __builtin_inff();
  unsigned a = fkey(lo), b = fkey(hi);
  while (a < b) {
    const unsigned m = a + (b - a) / 2u;
    if (cell_of(funkey(m), lo, inv_h, G) >= c) b = m;
    else a = m + 1u;
  }
  return funkey(a);
}

// ---------------------------------------------------------------------------
// pass 1: per-cloud grid parameters + edge tables
// ---------------------------------------------------------------------------
// pass 0: bounding boxes, all CUs.  fp32 min/max through order-preserving uint keys and
// atomicMin / atomicMax (keys pre-set by grid_bbox_init_kernel).
constexpr int kBboxBlock = 256;
constexpr int kBboxPerThread = 8;  // 2048 points per workgroup: enough workgroups to cover the latency of a 25 MB read

__global__ void grid_bbox_init_kernel(unsigned* __restrict__ bbox, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * 8) bbox[i] = ((i & 7) < 3) ? 0xffffffffu : 0u;  // [0..2] running min, [3..5] running max
}

__global__ __launch_bounds__(kBboxBlock) void grid_bbox_kernel(const float* __restrict__ p2,
                                                             const int64_t* __restrict__ lengths2, int P2, int D,
                                                             unsigned* __restrict__ bbox) {
  const int n = blockIdx.y;
  int len2 = (int)lengths2[n];
  len2 = len2 < 0 ? 0 : (len2 > P2 ? P2 : len2);
  const int j0 = blockIdx.x * (kBboxBlock * kBboxPerThread);
  if (j0 >= len2) return;
  float mn[3], mx[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    mn[d] = __builtin_inff();
    mx[d] = -__builtin_inff();
  }
  const float* __restrict__ base = p2 + (int64_t)n * P2 * D;
#pragma unroll 4
  for (int r = 0; r < kBboxPerThread; ++r) {
    const int j = j0 + r * kBboxBlock + threadIdx.x;
    if (j < len2) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (d < D) {
          const float v = base[(int64_t)j * D + d];
          mn[d] = fminf(mn[d], v);
          mx[d] = fmaxf(mx[d], v);
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, kWave));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, kWave));
    }
  }
  // one atomic per workgroup and bound (the 6 keys of a cloud are hot addresses)
  __shared__ float s_mn[kBboxBlock / kWave][3], s_mx[kBboxBlock / kWave][3];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      s_mn[wave][d] = mn[d];
      s_mx[wave][d] = mx[d];
    }
  }
  __syncthreads();
  if (threadIdx.x < 3 && (int)threadIdx.x < D) {
    const int d = threadIdx.x;
    float a = s_mn[0][d], b = s_mx[0][d];
#pragma unroll
    for (int w = 1; w < kBboxBlock / kWave; ++w) {
      a = fminf(a, s_mn[w][d]);
      b = fmaxf(b, s_mx[w][d]);
    }
    atomicMin(bbox + n * 8 + d, fkey(a));
    atomicMax(bbox + n * 8 + 3 + d, fkey(b));
  }
}

__global__ __launch_bounds__(kSetupBlock) void grid_setup_kernel(
    const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int D, float c_target, int B, float h_min,
    float ball_radius, int ball_K, float ball_factor, GridWs ws) {
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  int len2 = (int)lengths2[n];
  len2 = len2 < 0 ? 0 : (len2 > P2 ? P2 : len2);
  int len1 = (int)lengths1[n];
  len1 = len1 < 0 ? 0 : (len1 > P1 ? P1 : len1);
  __shared__ GridCloud s_g;
  __shared__ float s_hi[3];
  if (tid == 0) {
    GridCloud g;
    float lo[3], hi[3], e[3];
    bool finite = len2 > 0;
    for (int d = 0; d < 3; ++d) {
      float a = 0.0f, b = 0.0f;
      if (d < D && len2 > 0) {
        a = funkey(ws.bbox[n * 8 + d]);
        b = funkey(ws.bbox[n * 8 + 3 + d]);
      }
      if (d >= D) a = b = 0.0f;  // padded dimensions
      lo[d] = a;
      hi[d] = b;
      e[d] = b - a;
      if (!(fabsf(a) <= FLT_MAX) || !(fabsf(b) <= FLT_MAX) || !(e[d] <= FLT_MAX)) finite = false;
    }
    // cubic cells of edge h with ~c_target points each over the non-degenerate dimensions
    bool active[3] = {e[0] > 0.0f, e[1] > 0.0f, e[2] > 0.0f};
    float h = 0.0f;
    const float target_cells = fmaxf(1.0f, (float)len2 / c_target);
    if (finite) {
      for (int it = 0; it < 4; ++it) {
        int k = 0;
        double vol = 1.0;
        for (int d = 0; d < 3; ++d)
          if (active[d]) {
            ++k;
            vol *= (double)e[d];
          }
        if (k == 0) break;
        h = (float)pow(vol / (double)target_cells, 1.0 / (double)k);
        if (h < h_min) h = h_min;  // ball query: one cell beyond the query's own must cover the radius
        bool changed = false;
        for (int d = 0; d < 3; ++d)
          if (active[d] && !(e[d] >= h)) {
            active[d] = false;
            changed = true;
          }
        if (!changed) break;
      }
    }
    const bool any_active = active[0] || active[1] || active[2];
    bool ok = finite && (!any_active || (h > 0.0f && h <= FLT_MAX));
    if (ok && ws.ball) {
      // Ball query: the index-order scan stops after ~len2 * min(1, K / E) candidates per query
      // (E = expected points inside the ball), the grid visits ~27 cells >= 6.4 E candidates at a
      // higher cost each: take the grid only where it wins, factor * E max(E, K) < K len2.
      int k = 0;
      double vol = 1.0;
      for (int d = 0; d < 3; ++d)
        if (active[d]) {
          ++k;
          vol *= (double)e[d];
        }
      const double r = (double)ball_radius;
      const double ball = k == 3 ? 4.18879 * r * r * r : k == 2 ? 3.14159 * r * r : k == 1 ? 2.0 * r : 1.0;
      const double E = k == 0 ? (double)len2 : fmin((double)len2, (double)len2 * ball / vol);
      if (!((double)ball_factor * E * fmax(E, (double)ball_K) < (double)ball_K * (double)len2)) ok = false;
    }
    float inv_h = 1.0f;
    int G[3] = {1, 1, 1};
    if (ok && any_active) {
      for (int it = 0; it < 64; ++it) {
        inv_h = 1.0f / h;
        if (!(inv_h > 0.0f && inv_h <= FLT_MAX)) {
          ok = false;
          break;
        }
        long long cells = 1;
        for (int d = 0; d < 3; ++d) {
          G[d] = 1;
          if (active[d]) {
            const float t = e[d] * inv_h;  // same expression as cell_of(hi)
            G[d] = (t < (float)kGMax) ? (int)t + 1 : kGMax;
            if (G[d] < 1) G[d] = 1;
          }
          cells *= G[d];
        }
        // A histogram that fits the binning pass's LDS table is ~3x cheaper to build than one that
        // needs a global atomic per point: when the cell count is within 2x of the table, grow h a
        // little (cells ~ h^-3) until it fits.
        if (cells <= (long long)ws.cell_cap &&
            !(cells > (long long)kBinLdsBinsSetup && cells <= 2LL * kBinLdsBinsSetup))
          break;
        h *= cells > (long long)ws.cell_cap ? 1.2599211f : 1.04f;  // halve the cell count / nudge
      }
      if ((long long)G[0] * G[1] * G[2] > (long long)ws.cell_cap) ok = false;
    }
    for (int d = 0; d < 3; ++d) {
      g.lo[d] = lo[d];
      g.G[d] = G[d];
      g.NB[d] = (G[d] + B - 1) / B;
      s_hi[d] = hi[d];
    }
    g.inv_h = inv_h;
    g.ncell = G[0] * G[1] * G[2];
    g.nblock = g.NB[0] * g.NB[1] * g.NB[2];
    g.len1 = len1;
    g.len2 = len2;
    g.use_grid = ok ? 1 : 0;
    g.B = B;
    s_g = g;
    ws.cloud[n] = g;
    const float qnan = __uint_as_float(0x7fc00000u);
    ws.sorted[(int64_t)n * (P2 + 1) + P2] = make_float4(qnan, qnan, qnan, 0.0f);
    ws.grid_flag[n] = g.use_grid;
    ws.fb_count[n] = 0;
    ws.fb2_count[n] = 0;
    ws.fb3_count[n] = 0;
  }
  __syncthreads();
  if (s_g.use_grid) {
    float* __restrict__ ed = ws.edges + (int64_t)n * 3 * kEdgeStride;
    for (int t = tid; t < 3 * kEdgeStride; t += kSetupBlock) {
      const int d = t / kEdgeStride, c = t - d * kEdgeStride;
      const int G = s_g.G[d];
      ed[t] = (c <= G) ? edge_bisect(c, s_g.lo[d], s_hi[d], s_g.inv_h, G) : __builtin_inff();
    }
  }
}

// Work decode of the persistent search kernel.  The unit of work is one CHUNK of <= 64 queries
// of one block, so that a block holding thousands of queries (dense cluster in a uniform grid)
// is spread over many workgroups instead of being walked chunk after chunk by one wave.
// Chunk k of block b lives in slot  (blk_start[b] >> 6) + b + k : the slot ranges of
// consecutive blocks never overlap (floor((s+q)/64) - floor(s/64) + 1 >= ceil(q/64)) and a
// cloud needs at most len1/64 + nblock + 1 slots; empty slots are skipped.
__global__ void grid_prefix_kernel(GridWs ws, int N, int lane_mode) {  // one wave
  const int lane = threadIdx.x;
  int acc = 0;
  if (lane == 0) ws.block_prefix[0] = 0;
  for (int n0 = 0; n0 < N; n0 += kWave) {
    const int n = n0 + lane;
    int items = 0;
    if (n < N) {
      const GridCloud g = ws.cloud[n];
      if (g.use_grid) items = lane_mode ? (g.len1 + kGridWave - 1) / kGridWave : (g.len1 / kGridWave + g.nblock + 1);
    }
    int inc = items;  // inclusive wave scan
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int v = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += v;
    }
    if (n < N) ws.block_prefix[n + 1] = acc + inc;
    acc += __shfl(inc, kWave - 1, kWave);
  }
}

template <int D>
__device__ __forceinline__ void load_point3(const float* __restrict__ p, float& x, float& y, float& z) {
  x = p[0];
  y = D > 1 ? p[1] : 0.0f;
  z = D > 2 ? p[2] : 0.0f;
}

__device__ __forceinline__ void point_cells(const GridCloud& g, float x, float y, float z, int& cx, int& cy,
                                            int& cz) {
  cx = cell_of(x, g.lo[0], g.inv_h, g.G[0]);
  cy = cell_of(y, g.lo[1], g.inv_h, g.G[1]);
  cz = cell_of(z, g.lo[2], g.inv_h, g.G[2]);
}

// ---------------------------------------------------------------------------
// pass 2 / 4: histogram and counting-sort scatter (SCATTER = false / true), for the
// points of p2 by cell (IS_QUERY = false) and the queries of p1 by block (true).
//
// Scattered device-scope atomics run at only ~2e10/s chip-wide (they execute at the
// memory side, one 64-byte request each), so a workgroup first bins its tile of
// 1024 x 8 points in an LDS histogram (fast LDS atomics, which also hand every point
// its rank inside the (tile, bin) group) and then touches each non-empty global
// counter ONCE: count pass  global[bin] += n_tile ;  scatter pass  base = start[bin] +
// atomicAdd(cursor[bin], n_tile), position = base + rank.  Clouds with more bins than
// the LDS table holds (kBinLdsBins) use one global atomic per point.
// ---------------------------------------------------------------------------
constexpr int kBinBlock = 1024;
#ifndef POINTOPS_BIN_PER_THREAD
#define POINTOPS_BIN_PER_THREAD 8  // tile of 8192 points: 4096 / 8192 / 16384 / 32768 measured 0.988 / 0.969 / 0.987 / 1.125 ms per cfg2 step (chamfer cfg4: 1.14 / 1.13 / 1.22 / 1.66 ms)
#endif
constexpr int kBinPerThread = POINTOPS_BIN_PER_THREAD;
constexpr int kBinTile = kBinBlock * kBinPerThread;
static_assert(kBinLdsBinsSetup == 40000, "keep in sync");
constexpr int kBinLdsBins = 40000;  // 156 KiB of LDS: the whole CU's LDS, one workgroup per CU

template <int D, bool SCATTER, bool IS_QUERY>
__global__ __launch_bounds__(kBinBlock) void grid_bin_kernel(const float* __restrict__ pts, int P, int K, GridWs ws,
                                                           int64_t* __restrict__ idxs, float* __restrict__ dists) {
  __shared__ int s_hist[kBinLdsBins];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const GridCloud g = ws.cloud[n];  // wave-uniform
  const int len = IS_QUERY ? g.len1 : g.len2;
  const int nbins = IS_QUERY ? g.nblock : g.ncell;
  const int64_t cbase = (int64_t)n * ws.cell_cap;
  int* __restrict__ gcount = (IS_QUERY ? ws.blk_count : ws.cell_count) + cbase;
  const int* __restrict__ gstart = (IS_QUERY ? ws.blk_start : ws.cell_start) + (int64_t)n * (ws.cell_cap + 1);
  int* __restrict__ grank = (IS_QUERY ? ws.rank1 : ws.rank2) + (int64_t)n * P;
  const int i0 = blockIdx.x * kBinTile + tid;
  if (blockIdx.x * kBinTile >= P) return;

  if (IS_QUERY && !SCATTER) {
    // rows that get no search: zeros for padded queries (knn_cpu.cpp:25-26); whole-cloud list
    // when this cloud has no usable grid
#pragma unroll 4
    for (int r = 0; r < kBinPerThread; ++r) {
      const int i = i0 + r * kBinBlock;
      if (i < P && i >= g.len1) {
        int64_t* __restrict__ zi = idxs + ((int64_t)n * P + i) * K;
        float* __restrict__ zd = dists + ((int64_t)n * P + i) * K;
        const int64_t pad = ws.ball ? -1 : 0;
        for (int k = 0; k < K; ++k) {
          zi[k] = pad;
          zd[k] = 0.0f;
        }
      } else if (i < g.len1 && !g.use_grid && !ws.ball) {
        const int pos = atomicAdd(ws.fb2_count + n, 1);
        ws.fb2_list[(int64_t)n * P + pos] = i;
      }
    }
  }
  if (!g.use_grid || blockIdx.x * kBinTile >= len) return;

  const bool use_lds = nbins <= kBinLdsBins;
  if (use_lds) {
    for (int b = tid; b < nbins; b += kBinBlock) s_hist[b] = 0;
    __syncthreads();
  }
  int bin[kBinPerThread], rank[kBinPerThread];
  float px[kBinPerThread], py[kBinPerThread], pz[kBinPerThread];
#pragma unroll
  for (int r = 0; r < kBinPerThread; ++r) {
    const int i = i0 + r * kBinBlock;
    bin[r] = -1;
    rank[r] = 0;
    if (i < len) {
      float x, y, z;
      load_point3<D>(pts + ((int64_t)n * P + i) * D, x, y, z);
      int cx, cy, cz;
      point_cells(g, x, y, z, cx, cy, cz);
      bin[r] = IS_QUERY ? ((cz / g.B) * g.NB[1] + (cy / g.B)) * g.NB[0] + (cx / g.B)
                        : (cz * g.G[1] + cy) * g.G[0] + cx;
      if (SCATTER && !IS_QUERY) {
        px[r] = x;
        py[r] = y;
        pz[r] = z;
      }
      if (use_lds) {
        rank[r] = atomicAdd(&s_hist[bin[r]], 1);  // LDS atomic: rank inside (tile, bin)
      } else if (!SCATTER) {
        // too many bins for the LDS table: one device atomic per point, whose return value is the
        // point's rank in its bin -- remembered, so that the scatter pass needs no second atomic
        grank[i] = atomicAdd(gcount + bin[r], 1);
      } else {
        rank[r] = gstart[bin[r]] + grank[i];  // final position
      }
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int b = tid; b < nbins; b += kBinBlock) {
      const int c = s_hist[b];
      if (c > 0) {
        if (!SCATTER) atomicAdd(gcount + b, c);
        else s_hist[b] = gstart[b] + atomicAdd(gcount + b, c);  // base of this tile's group
      }
    }
    if (SCATTER) __syncthreads();
  }
  if (SCATTER) {
#pragma unroll
    for (int r = 0; r < kBinPerThread; ++r) {
      if (bin[r] >= 0) {
        const int i = i0 + r * kBinBlock;
        const int pos = use_lds ? s_hist[bin[r]] + rank[r] : rank[r];
        if (IS_QUERY) ws.qlist[(int64_t)n * P + pos] = i;
        else ws.sorted[(int64_t)n * (P + 1) + pos] = make_float4(px[r], py[r], pz[r], __int_as_float(i));
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass 3: exclusive scans of the cell and block histograms, chunked over all CUs:
//   a) every (chunk, cloud) workgroup sums its 4096 counters -> partial[cloud][chunk]
//   b) one workgroup per cloud turns the partials into chunk offsets (+ grand total)
//   c) every (chunk, cloud) workgroup rescans its chunk from its offset, writes the starts
//      and resets the counters to 0 so they can serve as scatter cursors.
// (a single workgroup per cloud took 0.43 ms at 2e5 cells -- the K=1 / chamfer regime.)
// ---------------------------------------------------------------------------
constexpr int kScanChunk = 4096;  // counters per workgroup: 1024 lanes x int4

__device__ __forceinline__ int block_sum_1024(int v, int* s_red) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  int t = lane < kScanBlock / kWave ? s_red[lane] : 0;
#pragma unroll
  for (int off = kScanBlock / kWave / 2; off > 0; off >>= 1) t += __shfl_xor(t, off, kWave);
  __syncthreads();
  return t;  // every lane of every wave holds the block total
}

// which: 0 = cells, 1 = blocks
__device__ __forceinline__ void scan_arrays(const GridWs& ws, int n, int which, int*& count, int*& start,
                                            int& len) {
  const GridCloud g = ws.cloud[n];
  count = (which == 0 ? ws.cell_count : ws.blk_count) + (int64_t)n * ws.cell_cap;
  start = (which == 0 ? ws.cell_start : ws.blk_start) + (int64_t)n * (ws.cell_cap + 1);
  len = g.use_grid ? (which == 0 ? g.ncell : g.nblock) : 0;
}

__global__ __launch_bounds__(kScanBlock) void grid_scan_partial_kernel(GridWs ws, int chunks) {
  __shared__ int s_red[kScanBlock / kWave];
  const int n = blockIdx.y, which = blockIdx.z, chunk = blockIdx.x;
  int *count, *start, len;
  scan_arrays(ws, n, which, count, start, len);
  if (chunk * kScanChunk >= len) return;
  int v = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = chunk * kScanChunk + threadIdx.x * 4 + r;
    if (i < len) v += count[i];
  }
  const int tot = block_sum_1024(v, s_red);
  if (threadIdx.x == 0) ws.scan_partial[((int64_t)n * 2 + which) * chunks + chunk] = tot;
}

__global__ __launch_bounds__(kScanBlock) void grid_scan_offsets_kernel(GridWs ws, int chunks) {
  // chunks <= 1024 is guaranteed by the host (cell_cap <= 4M)
  __shared__ int s_red[kScanBlock / kWave];
  const int n = blockIdx.x, which = blockIdx.y;
  int *count, *start, len;
  scan_arrays(ws, n, which, count, start, len);
  const int used = (len + kScanChunk - 1) / kScanChunk;
  int* __restrict__ part = ws.scan_partial + ((int64_t)n * 2 + which) * chunks;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int v = tid < used ? part[tid] : 0;
  int inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int u = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += u;
  }
  if (lane == kWave - 1) s_red[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < kScanBlock / kWave ? s_red[lane] : 0;
    int winc = w;
#pragma unroll
    for (int off = 1; off < kScanBlock / kWave; off <<= 1) {
      const int u = __shfl_up(winc, off, kWave);
      if (lane >= off) winc += u;
    }
    if (lane < kScanBlock / kWave) s_red[lane] = winc - w;
  }
  __syncthreads();
  if (tid < used) part[tid] = s_red[wave] + inc - v;  // exclusive chunk offset
  if (len > 0 && tid == used - 1) start[len] = s_red[wave] + inc;  // grand total
  if (len == 0 && tid == 0 && ws.cloud[n].use_grid) start[0] = 0;
}

__global__ __launch_bounds__(kScanBlock) void grid_scan_apply_kernel(GridWs ws, int chunks) {
  __shared__ int s_red[kScanBlock / kWave];
  const int n = blockIdx.y, which = blockIdx.z, chunk = blockIdx.x;
  int *count, *start, len;
  scan_arrays(ws, n, which, count, start, len);
  if (chunk * kScanChunk >= len) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int i0 = chunk * kScanChunk + tid * 4;
  int c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = (i0 + r < len) ? count[i0 + r] : 0;
  const int sum = c[0] + c[1] + c[2] + c[3];
  int inc = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int u = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += u;
  }
  if (lane == kWave - 1) s_red[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < kScanBlock / kWave ? s_red[lane] : 0;
    int winc = w;
#pragma unroll
    for (int off = 1; off < kScanBlock / kWave; off <<= 1) {
      const int u = __shfl_up(winc, off, kWave);
      if (lane >= off) winc += u;
    }
    if (lane < kScanBlock / kWave) s_red[lane] = winc - w;
  }
  __syncthreads();
  int run = ws.scan_partial[((int64_t)n * 2 + which) * chunks + chunk] + s_red[wave] + inc - sum;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (i0 + r < len) {
      start[i0 + r] = run;
      run += c[r];
      count[i0 + r] = 0;
    }
  }
}

template <int NORM>
__device__ __forceinline__ float face_bound(float t) {  // t = fl(|q - face|) >= 0
  return NORM == 1 ? t : t * t;
}

constexpr int kMaxRows = 100;  // (B + 2)^2 with B <= 8

template <int D, int KC, int NORM>
__global__ __launch_bounds__(kGridWave) void knn_grid_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds,
    const int* __restrict__ block_prefix, const float* __restrict__ edges,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted,
    const int* __restrict__ blk_start, const int* __restrict__ qlist, int* __restrict__ fb_count,
    int* __restrict__ fb_list, int cell_cap, int P1, int P2, int K, int N, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  // Per-lane candidate queues (KC >= 8).  Some lane of the wave wants almost every candidate
  // (64 queries spread over the block), so a direct sorted insert makes the whole wave walk the
  // ~100-instruction insert for one or two active lanes ~260 times per chunk.  Instead a
  // candidate that beats the lane's (stale) threshold is parked with one ds_write_b64, and
  // when some lane's queue is nearly full ALL lanes merge their queues into their lists with a
  // branch-free network: bitonic-sort the queue, take min(list[i], queue[KC-1-i]) -- the KC
  // smallest of the union as a bitonic sequence -- and bitonic-merge.  Cost per flush is fixed
  // (~120 compare-exchanges at KC=16) and independent of how unevenly the lanes filled.
  constexpr bool kUseQueue = KC >= 8 && (KC & (KC - 1)) == 0;
  constexpr int kQueueCap = KC < 16 ? KC : 16;
  constexpr int kSub = 4;  // candidates handled between two queue-full checks
  __shared__ unsigned long long s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  __shared__ float4 s_tile[2][kGridWave];
  __shared__ int s_rowsrc[kMaxRows];  // first record of the row's run in `sorted`
  __shared__ int s_rowoff[kMaxRows + 1];  // position of the row in the flat stream

  const int lane = threadIdx.x;
  const int total = block_prefix[N];
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    // cloud of this item: largest n with block_prefix[n] <= item (uniform binary search)
    int lo_n = 0, hi_n = N;
    while (hi_n - lo_n > 1) {
      const int mid = (lo_n + hi_n) >> 1;
      if (block_prefix[mid] <= item) lo_n = mid;
      else hi_n = mid;
    }
    const int n = lo_n;
    const int slot = item - block_prefix[n];
    const GridCloud g = clouds[n];
    const int* __restrict__ bstart = blk_start + (int64_t)n * (cell_cap + 1);
    // block of this slot: largest b with (bstart[b] >> 6) + b <= slot  (monotone in b)
    int lo_b = 0, hi_b = g.nblock;
    while (hi_b - lo_b > 1) {
      const int mid = (lo_b + hi_b) >> 1;
      if ((bstart[mid] >> 6) + mid <= slot) lo_b = mid;
      else hi_b = mid;
    }
    const int b = lo_b;
    const int qe = bstart[b + 1];
    const int qs = bstart[b] + kGridWave * (slot - ((bstart[b] >> 6) + b));  // first query of this chunk
    if (qs >= qe) continue;  // empty slot

    const int bx = b % g.NB[0], by = (b / g.NB[0]) % g.NB[1], bz = b / (g.NB[0] * g.NB[1]);
    const int X0 = max(bx * g.B - 1, 0), X1 = min(bx * g.B + g.B, g.G[0] - 1);
    const int Y0 = max(by * g.B - 1, 0), Y1 = min(by * g.B + g.B, g.G[1] - 1);
    const int Z0 = max(bz * g.B - 1, 0), Z1 = min(bz * g.B + g.B, g.G[2] - 1);
    const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
    // faces of the visited region (wave-uniform); a missing face = the grid boundary
    const bool hx0 = X0 > 0, hx1 = X1 < g.G[0] - 1;
    const bool hy0 = Y0 > 0, hy1 = Y1 < g.G[1] - 1;
    const bool hz0 = Z0 > 0, hz1 = Z1 < g.G[2] - 1;
    const float fx0 = hx0 ? prev_float(ed[X0]) : 0.0f, fx1 = hx1 ? ed[X1 + 1] : 0.0f;
    const float fy0 = hy0 ? prev_float(ed[kEdgeStride + Y0]) : 0.0f, fy1 = hy1 ? ed[kEdgeStride + Y1 + 1] : 0.0f;
    const float fz0 = hz0 ? prev_float(ed[2 * kEdgeStride + Z0]) : 0.0f,
                fz1 = hz1 ? ed[2 * kEdgeStride + Z1 + 1] : 0.0f;
    const bool whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);

    // row table of the region's flat candidate stream
    const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
    const int ny = Y1 - Y0 + 1, nrows = ny * (Z1 - Z0 + 1);
    __syncthreads();  // previous item's readers are done with the tables
    // Rows are visited NEAR-FIRST: the block's own cells, then the halo.  After the block's own
    // points the per-lane thresholds are almost final, so halo candidates rarely pass.
    const int zb0 = bz * g.B, nbz = min(zb0 + g.B - 1, g.G[2] - 1) - zb0 + 1;
    const int yb0 = by * g.B, nby = min(yb0 + g.B - 1, g.G[1] - 1) - yb0 + 1;
    for (int r = lane; r < nrows; r += kGridWave) {  // nrows <= 100
      const int iz = r / ny, iy = r % ny;
      const int z = iz < nbz ? zb0 + iz : ((Z0 < zb0 && iz == nbz) ? Z0 : Z1);
      const int y = iy < nby ? yb0 + iy : ((Y0 < yb0 && iy == nby) ? Y0 : Y1);
      const int rowbase = (z * g.G[1] + y) * g.G[0];
      const int s = cstart[rowbase + X0], e = cstart[rowbase + X1 + 1];
      s_rowsrc[r] = s;
      s_rowoff[r + 1] = e - s;  // lengths first, scanned below
    }
    __syncthreads();
    if (lane == 0) {
      int acc = 0;
      s_rowoff[0] = 0;
      for (int r = 0; r < nrows; ++r) {
        acc += s_rowoff[r + 1];
        s_rowoff[r + 1] = acc;
      }
    }
    __syncthreads();
    const int T = s_rowoff[nrows];  // records in the region
    const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + 1);  // + 1: the cloud's NaN sentinel record
    const int* __restrict__ ql = qlist + (int64_t)n * P1;

    // record t of the flat stream -> its address (per lane); `rh` = a row at or before t's row
    auto fetch = [&](int t, int rh) -> float4 {
      // beyond the stream: NaN coordinates -> NaN distance, whose bit pattern exceeds every
      // threshold (<= 0x7f800000), so tail records need no bounds check in the scan
      const float qnan = __uint_as_float(0x7fc00000u);
      float4 v = make_float4(qnan, qnan, qnan, 0.f);
      if (t < T) {
        int r = rh;
        while (s_rowoff[r + 1] <= t) ++r;  // t < T = rowoff[nrows] bounds the walk
        v = sp[s_rowsrc[r] + (t - s_rowoff[r])];
      }
      return v;
    };

    {
      const int c0 = qs;
      const bool active = c0 + lane < qe;
      const int qi = active ? ql[c0 + lane] : 0;
      float qx = 0.0f, qy = 0.0f, qz = 0.0f;
      if (active) load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, qx, qy, qz);

      TopKLex<KC> top;
      top.init();
      unsigned thr = 0x7f800000u;  // distance bits a candidate must not exceed (stale between flushes)
      int qn = 0;                  // entries in this lane's queue
      auto flush = [&]() {
        unsigned long long qk[kQueueCap];
#pragma unroll
        for (int t = 0; t < kQueueCap; ++t) {
          const unsigned long long v = s_queue[t * kGridWave + lane];
          qk[t] = t < qn ? v : TopKLex<KC>::kEmpty;
        }
        bitonic_sort<kQueueCap>(qk);
#pragma unroll
        for (int t = 0; t < kQueueCap; ++t) {  // list slot KC-1-t meets queue entry t
          const unsigned long long a = top.key[KC - 1 - t];
          top.key[KC - 1 - t] = qk[t] < a ? qk[t] : a;
        }
        bitonic_merge<KC>(top.key);
        qn = 0;
        thr = top.worst_bits();
      };
      float4 nxt = fetch(lane, 0);
      int buf = 0;
      int rhint = 0;  // wave-uniform: row containing the first record of the NEXT tile
      for (int t0 = 0; t0 < T; t0 += kGridWave) {
        __syncthreads();  // tile `buf` no longer read (two tiles ago)
        s_tile[buf][lane] = nxt;
        if (t0 + kGridWave < T) {
          while (s_rowoff[rhint + 1] <= t0 + kGridWave) ++rhint;
        }
        nxt = fetch(t0 + kGridWave + lane, rhint);  // next tile's loads fly during this tile's scan
        __syncthreads();
        const int cnt = min(kGridWave, T - t0);
        for (int tb = 0; tb < cnt; tb += kSub) {
          float4 cc[kSub];
#pragma unroll
          for (int u = 0; u < kSub; ++u) cc[u] = s_tile[buf][(tb + u) & (kGridWave - 1)];  // broadcast reads
#pragma unroll
          for (int u = 0; u < kSub; ++u) {
            const float4 c = cc[u];
            float d;
            if (NORM == 1) {
              d = __builtin_fabsf(qx - c.x);
              if (D > 1) d = d + __builtin_fabsf(qy - c.y);
              if (D > 2) d = d + __builtin_fabsf(qz - c.z);
            } else {
              const float dx = qx - c.x;
              d = dx * dx;
              if (D > 1) {
                const float dy = qy - c.y;
                d = d + dy * dy;
              }
              if (D > 2) {
                const float dz = qz - c.z;
                d = d + dz * dz;
              }
            }
            if (kUseQueue) {
              if (__float_as_uint(d) <= thr) {
                s_queue[qn * kGridWave + lane] = TopKLex<KC>::make(d, __float_as_int(c.w));
                ++qn;
              }
            } else if (__float_as_uint(d) <= top.worst_bits()) {
              const unsigned long long key = TopKLex<KC>::make(d, __float_as_int(c.w));
              if (key < top.key[KC - 1]) top.insert(key);
            }
          }
          if (kUseQueue) {
            if (__any(qn > kQueueCap - kSub)) flush();
          }
        }
        buf ^= 1;
      }
      if (kUseQueue) flush();
      // Acceptance: the KC-th best (KC >= K: conservative) against the rigorous lower bound
      // of every point that was not visited.
      const unsigned kth_bits = top.kth_bits(K);  // the K-th best, not the list's last slot
      float lb = __builtin_inff();
      if (hx0) lb = fminf(lb, face_bound<NORM>(qx - fx0));
      if (hx1) lb = fminf(lb, face_bound<NORM>(fx1 - qx));
      if (hy0) lb = fminf(lb, face_bound<NORM>(qy - fy0));
      if (hy1) lb = fminf(lb, face_bound<NORM>(fy1 - qy));
      if (hz0) lb = fminf(lb, face_bound<NORM>(qz - fz0));
      if (hz1) lb = fminf(lb, face_bound<NORM>(fz1 - qz));
      const bool full = kth_bits < 0x7f800000u;
      const bool ok = whole || (full && __uint_as_float(kth_bits) < lb);
      if (active) {
        if (ok) {
          const int64_t row = (int64_t)n * P1 + qi;
          write_row<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
        } else {
          const int pos = atomicAdd(fb_count + n, 1);
          fb_list[(int64_t)n * P1 + pos] = qi;
        }
      }
    }
  }
}

// Candidate threshold seeded from the certification bound lb: only candidates with d < lb can
// appear in a certified answer (certification needs the KC-th best below lb), so the search may
// ignore the rest from the first record on.  Distances are non-negative, so bit order = value order.
__device__ __forceinline__ unsigned seed_threshold(float lb, bool whole) {
  const unsigned b = __float_as_uint(lb);
  return whole ? 0x7f800000u : (b > 0u ? b - 1u : 0u);
}

// ---------------------------------------------------------------------------
// pass 5 (lane-private form): one query per lane, EVERY lane walks only the 3x3x3 cell cube
// around its own cell (9 contiguous runs of the sorted array), fetched with per-lane 16-byte
// loads, four in flight.  The queries are sorted by cell (block edge 1), so the 64 lanes of a
// wave sit in ~10 neighbouring cells and their runs overlap in L1/L2.  Compared with the
// block-shared broadcast form above this visits ~27 c instead of 64 c candidates per query and
// fills all 64 lanes of every wave (a block of 2^3 cells holds ~50 queries), at the price of
// per-lane addressing; the selection core (stale-threshold queues merged by sorting networks)
// and the certification bound are the same, evaluated on the lane's own cube.
// ---------------------------------------------------------------------------
constexpr int kLaneRows = 9;
// Pipeline forms of the lane search, selectable at compile time (tools/build_variant.py):
//  PIPE = 1 (default)  in-place: a candidate's registers are reloaded with the next group's record as soon as
//                its distance and index are taken -- no buffer copies (K=1 0.46 -> 0.45, K=8 0.69 -> 0.67,
//                K=32 1.94 -> 1.89 ms at cfg2 size; K=16 unchanged); PIPE = 0: two buffers, group g+1 copied over g;
//  PINGPONG = 1  two candidate buffers with the loop body written twice (saves the 16 v_mov_b64 that
//                copy group g+1 over group g): 1.08 -> 1.24 ms at cfg2 (K=16), 0.76 -> 0.82 (K=8),
//                2.35 -> 2.26 (K=32) -- the doubled flush code costs more than the copies;
//  SWITCH = 0  plain per-record loop over runs; 1 = the loop behind one wave-uniform branch (no change);
//           2 = branch-free, one switch test per record (1.025 -> 1.011 ms); 3 (default) = one look at the
//           next run per GROUP of four records (-> 0.99 ms; K=32: 2.12 -> 1.94 ms).
#ifndef POINTOPS_LANE_PIPE
#define POINTOPS_LANE_PIPE 1
#endif
#ifndef POINTOPS_LANE_PINGPONG
#define POINTOPS_LANE_PINGPONG 0
#endif
#ifndef POINTOPS_LANE_FETCH32
#define POINTOPS_LANE_FETCH32 8
#endif
#ifndef POINTOPS_LANE_FETCH
#define POINTOPS_LANE_FETCH 8
#endif
#ifndef POINTOPS_LANE_SWITCH
#define POINTOPS_LANE_SWITCH 3
#endif

template <int D, int KC, int NORM>
__global__ __launch_bounds__(kGridWave) void knn_grid_lane_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const int* __restrict__ chunk_prefix,
    const float* __restrict__ edges, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
    const int* __restrict__ qlist, int* __restrict__ fb_count, int* __restrict__ fb_list,
    unsigned* __restrict__ fb_kth, int cell_cap, int P1, int P2, int K, int N, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  constexpr bool kUseQueue = KC >= 8 && (KC & (KC - 1)) == 0;
  constexpr int kQueueCap = KC < 16 ? KC : 16;
  constexpr int kSub = 4;
  // gathers per lane and pipeline stage (processed kSub at a time).  With the in-place pipeline eight win or
  // tie everywhere (cfg2 size: K=8 0.67 -> 0.64, K=4 0.54 -> 0.52, K=32 2.07 -> 1.88 ms, K=16 unchanged); with the
  // two-buffer pipeline they only paid for the 32-slot lists.
  constexpr int kFetch = KC >= 32 ? POINTOPS_LANE_FETCH32 : POINTOPS_LANE_FETCH;
  __shared__ unsigned long long s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  __shared__ int2 s_rows[kLaneRows + 1][kGridWave];  // per-lane (first record, end) of its 9 runs; row 9 = empty

  const int lane = threadIdx.x;
  const int total = chunk_prefix[N];
  // XCD-aware item order: workgroup b runs on XCD b % 8 (round-robin dispatch), and the chunks are
  // sorted by (cloud, cell).  Each XCD walks its own contiguous eighth of the chunk list, so the
  // ~1000 chunks it has in flight belong to one or two clouds whose sorted records (1 MB at 65536
  // points) stay in that XCD's 4 MB L2, instead of every XCD touching every cloud in flight.
  const int xcd = blockIdx.x % kNumXcd, per_xcd = (total + kNumXcd - 1) / kNumXcd;
  for (int j = blockIdx.x / kNumXcd; j < per_xcd; j += gridDim.x / kNumXcd) {
    const int item = xcd * per_xcd + j;
    if (item >= total) break;
    int lo_n = 0, hi_n = N;
    while (hi_n - lo_n > 1) {
      const int mid = (lo_n + hi_n) >> 1;
      if (chunk_prefix[mid] <= item) lo_n = mid;
      else hi_n = mid;
    }
    const int n = lo_n;
    const GridCloud g = clouds[n];
    const int c0 = (item - chunk_prefix[n]) * kGridWave;
    const bool active = c0 + lane < g.len1;
    const int qi = active ? qlist[(int64_t)n * P1 + c0 + lane] : 0;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, qx, qy, qz);
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    const int X0 = max(cx - 1, 0), X1 = min(cx + 1, g.G[0] - 1);
    const int Y0 = max(cy - 1, 0), Y1 = min(cy + 1, g.G[1] - 1);
    const int Z0 = max(cz - 1, 0), Z1 = min(cz + 1, g.G[2] - 1);
    const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
    const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + 1);  // + 1: the cloud's NaN sentinel record

    // rigorous lower bound of every point outside the lane's cube (certification), known
    // before the walk: it also seeds the candidate threshold, since a query whose KC-th best
    // is not below it is uncertified whatever the candidates beyond it are
    const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
    const bool hx0 = X0 > 0, hx1 = X1 < g.G[0] - 1;
    const bool hy0 = Y0 > 0, hy1 = Y1 < g.G[1] - 1;
    const bool hz0 = Z0 > 0, hz1 = Z1 < g.G[2] - 1;
    float lb = __builtin_inff();
    if (hx0) lb = fminf(lb, face_bound<NORM>(qx - prev_float(ed[X0])));
    if (hx1) lb = fminf(lb, face_bound<NORM>(ed[X1 + 1] - qx));
    if (hy0) lb = fminf(lb, face_bound<NORM>(qy - prev_float(ed[kEdgeStride + Y0])));
    if (hy1) lb = fminf(lb, face_bound<NORM>(ed[kEdgeStride + Y1 + 1] - qy));
    if (hz0) lb = fminf(lb, face_bound<NORM>(qz - prev_float(ed[2 * kEdgeStride + Z0])));
    if (hz1) lb = fminf(lb, face_bound<NORM>(ed[2 * kEdgeStride + Z1 + 1] - qz));
    const bool whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);
    const unsigned thr0 = seed_threshold(lb, whole);

    // the lane's 9 runs, own row first (near-first order tightens the thresholds early)
#pragma unroll
    for (int r = 0; r < kLaneRows; ++r) {
      constexpr int kDz[kLaneRows] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
      constexpr int kDy[kLaneRows] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
      const int z = cz + kDz[r], y = cy + kDy[r];
      int2 se = make_int2(0, 0);
      if (active && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
        const int rowbase = (z * g.G[1] + y) * g.G[0];
        se.x = cstart[rowbase + X0];
        se.y = cstart[rowbase + X1 + 1];
      }
      s_rows[r][lane] = se;
    }
    s_rows[kLaneRows][lane] = make_int2(0, 0);
    int r = 0;
    int cur = s_rows[0][lane].x, end = s_rows[0][lane].y;
    auto next_record = [&]() __attribute__((always_inline)) -> int {  // index of the lane's next record, -1 when exhausted
      // run switches are rare per lane (9 per ~170 records): keep them behind ONE wave-uniform
      // branch so that the common step is branch-free (compare, two selects, add)
#if POINTOPS_LANE_SWITCH == 2
      // branch-free: at most ONE run switch per call (some lane of the wave switches on ~95 % of the
      // calls, so a divergent loop here runs almost always, with one or two lanes); a lane whose new
      // run is empty hands out the sentinel once and switches again on its next call
      const bool need = cur >= end && r < kLaneRows - 1;
      r += need ? 1 : 0;
      const int2 se = s_rows[r][lane];
      cur = need ? se.x : cur;
      end = need ? se.y : end;
      const bool ok = cur < end;
      const int a = ok ? cur : P2;
      cur += ok ? 1 : 0;
      return a;
#elif POINTOPS_LANE_SWITCH == 1
      if (__any(cur >= end && r < kLaneRows - 1)) {
        while (cur >= end && r < kLaneRows - 1) {
          ++r;
          const int2 se = s_rows[r][lane];
          cur = se.x;
          end = se.y;
        }
      }
      const bool ok = cur < end;
      const int a = ok ? cur : P2;
      cur += ok ? 1 : 0;
      return a;
#else
      while (cur >= end && r < kLaneRows - 1) {
        ++r;
        const int2 se = s_rows[r][lane];
        cur = se.x;
        end = se.y;
      }
      return cur < end ? cur++ : P2;  // exhausted: the NaN sentinel record
#endif
    };

    TopKLex<KC> top;
    top.init();
    unsigned thr = thr0;
    int qn = 0;
    auto flush = [&]() __attribute__((always_inline)) {
      unsigned long long qk[kQueueCap];
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned long long v = s_queue[t * kGridWave + lane];
        qk[t] = t < qn ? v : TopKLex<KC>::kEmpty;
      }
      bitonic_sort<kQueueCap>(qk);
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned long long a = top.key[KC - 1 - t];
        top.key[KC - 1 - t] = qk[t] < a ? qk[t] : a;
      }
      bitonic_merge<KC>(top.key);
      qn = 0;
      thr = min(top.worst_bits(), thr0);
    };

    // software pipeline: the loads of group g+1 are issued before group g is processed
    // four record indices at a time: ONE look at the next run per group instead of a run-switch test per
    // record; a group may straddle into the next run once (a shorter next run hands out sentinels for
    // the rest of the group, the following group moves on)
    auto next_group = [&](int* a) __attribute__((always_inline)) {
      const int2 nx = s_rows[r + 1][lane];
      const int left = end - cur;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        const int over = u - left;
        const int b = nx.x + over;
        a[u] = over < 0 ? cur + u : (b < nx.y ? b : P2);
      }
      const bool sw = left < kSub;
      const int ncur = nx.x + (kSub - left);
      cur = sw ? min(ncur, nx.y) : cur + kSub;
      end = sw ? nx.y : end;
      r += (sw && r < kLaneRows - 1) ? 1 : 0;
    };
    (void)next_record;
    (void)next_group;
    auto fetch = [&](float4 (&c)[kFetch]) __attribute__((always_inline)) -> bool {
      int a[kFetch];
#if POINTOPS_LANE_SWITCH == 3
#pragma unroll
      for (int u0 = 0; u0 < kFetch; u0 += kSub) next_group(a + u0);
#else
#pragma unroll
      for (int u = 0; u < kFetch; ++u) a[u] = next_record();
#endif
#pragma unroll
      for (int u = 0; u < kFetch; ++u) c[u] = sp[a[u]];  // unconditional: exhausted lanes read the sentinel
      // false once the lane has nothing left (a run switch may hand out the sentinel BEFORE real records)
      bool real = false;
#pragma unroll
      for (int u = 0; u < kFetch; ++u) real = real || a[u] != P2;
      return real || cur < end || r < kLaneRows - 1;
    };
    // two buffers in ping-pong (the loop body is written twice) so that no group is copied
    auto process = [&](const float4 (&c)[kFetch]) __attribute__((always_inline)) {
#pragma unroll
     for (int u0 = 0; u0 < kFetch; u0 += kSub) {
#pragma unroll
      for (int u = u0; u < u0 + kSub; ++u) {
        float d;
        if (NORM == 1) {
          d = __builtin_fabsf(qx - c[u].x);
          if (D > 1) d = d + __builtin_fabsf(qy - c[u].y);
          if (D > 2) d = d + __builtin_fabsf(qz - c[u].z);
        } else {
          const float dx = qx - c[u].x;
          d = dx * dx;
          if (D > 1) {
            const float dy = qy - c[u].y;
            d = d + dy * dy;
          }
          if (D > 2) {
            const float dz = qz - c[u].z;
            d = d + dz * dz;
          }
        }
        if (kUseQueue) {
          if (__float_as_uint(d) <= thr) {
            s_queue[qn * kGridWave + lane] = TopKLex<KC>::make(d, __float_as_int(c[u].w));
            ++qn;
          }
        } else if (__float_as_uint(d) <= min(top.worst_bits(), thr0)) {
          const unsigned long long key = TopKLex<KC>::make(d, __float_as_int(c[u].w));
          if (key < top.key[KC - 1]) top.insert(key);
        }
      }
      if (kUseQueue) {
        if (__any(qn > kQueueCap - kSub)) flush();
      }
     }
    };
#if POINTOPS_LANE_PIPE == 1
    // In-place pipeline: a candidate's registers are reloaded with the record of the NEXT group as soon
    // as its distance and index have been taken, so a group's gathers fly during the rest of the
    // previous group (threshold tests, pushes, flush) and nothing is copied between buffers.
    auto indices = [&](int* a) __attribute__((always_inline)) -> bool {
#pragma unroll
      for (int u0 = 0; u0 < kFetch; u0 += kSub) next_group(a + u0);
      bool real = false;
#pragma unroll
      for (int u = 0; u < kFetch; ++u) real = real || a[u] != P2;
      return real || cur < end || r < kLaneRows - 1;
    };
    float4 c[kFetch];
    int a0[kFetch];
    bool more = indices(a0);
#pragma unroll
    for (int u = 0; u < kFetch; ++u) c[u] = sp[a0[u]];
    while (__any(more)) {
      int an[kFetch];
      const bool more_next = indices(an);
#pragma unroll
      for (int u0 = 0; u0 < kFetch; u0 += kSub) {
        float dd[kSub];
        int ii[kSub];
#pragma unroll
        for (int u = u0; u < u0 + kSub; ++u) {
          float d;
          if (NORM == 1) {
            d = __builtin_fabsf(qx - c[u].x);
            if (D > 1) d = d + __builtin_fabsf(qy - c[u].y);
            if (D > 2) d = d + __builtin_fabsf(qz - c[u].z);
          } else {
            const float dx = qx - c[u].x;
            d = dx * dx;
            if (D > 1) {
              const float dy = qy - c[u].y;
              d = d + dy * dy;
            }
            if (D > 2) {
              const float dz = qz - c[u].z;
              d = d + dz * dz;
            }
          }
          dd[u - u0] = d;
          ii[u - u0] = __float_as_int(c[u].w);
          c[u] = sp[an[u]];  // next group's record into the same registers
        }
#pragma unroll
        for (int t = 0; t < kSub; ++t) {
          if (kUseQueue) {
            if (__float_as_uint(dd[t]) <= thr) {
              s_queue[qn * kGridWave + lane] = TopKLex<KC>::make(dd[t], ii[t]);
              ++qn;
            }
          } else if (__float_as_uint(dd[t]) <= min(top.worst_bits(), thr0)) {
            const unsigned long long key = TopKLex<KC>::make(dd[t], ii[t]);
            if (key < top.key[KC - 1]) top.insert(key);
          }
        }
        if (kUseQueue) {
          if (__any(qn > kQueueCap - kSub)) flush();
        }
      }
      more = more_next;
    }
    (void)process;
    (void)fetch;
#elif POINTOPS_LANE_PINGPONG
    float4 ca[kFetch], cb[kFetch];
    bool more = fetch(ca);
    while (__any(more)) {
      more = fetch(cb);  // group g+1 in flight while group g is processed
      process(ca);
      if (!__any(more)) break;
      more = fetch(ca);
      process(cb);
    }
#else
    float4 c[kFetch];
    bool more = fetch(c);
    while (__any(more)) {
      float4 nxt[kFetch];
      const bool more_next = fetch(nxt);  // group g+1 in flight while group g is processed
      process(c);
#pragma unroll
      for (int u = 0; u < kFetch; ++u) c[u] = nxt[u];
      more = more_next;
    }
#endif
    if (kUseQueue) flush();

    const unsigned kth_bits = top.kth_bits(K);  // the K-th best, not the list's last slot
    const bool full = kth_bits < 0x7f800000u;
    const bool ok = whole || (full && __uint_as_float(kth_bits) < lb);
    if (active) {
      if (ok) {
        const int64_t row = (int64_t)n * P1 + qi;
        write_row<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
      } else {
        const int pos = atomicAdd(fb_count + n, 1);
        fb_list[(int64_t)n * P1 + pos] = qi;
        // The seeded threshold admitted only the m < KC candidates below lb, so the KC-th best itself is
        // unknown; hand the quad pass an ESTIMATE from the density they imply (m points inside radius
        // sqrt(lb) -> K points inside sqrt(lb) (K/m)^(1/3)), 30 % up.  It only picks the cube to search.
        int m = 0;
#pragma unroll
        for (int t = 0; t < KC; ++t) m += (unsigned)(top.key[t] >> 32) < 0x7f800000u ? 1 : 0;
        const float est = m > 0 ? lb * __powf(fmaxf((float)K / (float)m, 1.0f), NORM == 1 ? 0.33333f : 0.66667f) * 1.3f
                                : __builtin_inff();
        fb_kth[(int64_t)n * P1 + pos] = __float_as_uint(est);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass 5b: radius-2 search for the queries pass 5 could not certify (~1 % of a cloud).
// FOUR lanes share a query: the 25 (y, z) rows of the 5x5x5 cell cube around the query's
// cell are dealt round-robin (nearest rows first) to the quad's lanes, each lane walks its
// <= 7 contiguous runs exactly like pass 5 (per-lane 16-byte gathers, eight in flight per
// pipeline stage, stale-threshold queue, sorting-network merges), and two quad-permute
// exchange steps merge the four sorted lists, after which every lane of the quad holds the
// cube's KC best.  (Kernel time at cfg2 with 4 / 8 / 16 lanes per query: 83 / 96 / 97-107 us -- the pass
// is throughput-bound, not bound by one wave's chain, so fewer, longer lanes win; 1.16 ms per step
// against 1.23 ms when these queries went straight to the expanding wave search.)  The same rigorous face bound
// decides; what is still uncertified (far-away queries) goes to the expanding wave search.
// ---------------------------------------------------------------------------
#ifndef POINTOPS_QUAD_LANES
#define POINTOPS_QUAD_LANES 4
#endif
#ifndef POINTOPS_QUAD_FETCH
#define POINTOPS_QUAD_FETCH 8
#endif
constexpr int kQuadLanes = POINTOPS_QUAD_LANES;  // lanes per query: 4, 8 or 16
constexpr int kQuadRows = (25 + kQuadLanes - 1) / kQuadLanes;
constexpr int kQuadQueries = kGridWave / kQuadLanes;
constexpr int kQuadFetch = POINTOPS_QUAD_FETCH;  // gathers in flight per lane and pipeline stage: 4 or 8
__constant__ signed char kQuadDy[32] = {0, 0, 0, -1, 1, -1, -1, 1, 1, 0, 0, -2, 2, -1, 1, -1, 1, -2, -2, 2, 2, -2, -2, 2, 2, 0, 0, 0, 0, 0, 0, 0};
__constant__ signed char kQuadDz[32] = {0, -1, 1, 0, 0, -1, 1, -1, 1, -2, 2, 0, 0, -2, -2, 2, 2, -1, 1, -1, 1, -2, 2, -2, 2, 0, 0, 0, 0, 0, 0, 0};

// DPP controls: quad_perm [1,0,3,2] (lane ^ 1), quad_perm [2,3,0,1] (lane ^ 2), row_half_mirror
// (lane -> 7 - lane within 8), row_mirror (lane -> 15 - lane within 16): after steps 1..s every
// lane of a 2^s group has met a lane of the other half, which already held that half's result
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// merge the partner lane's ascending list into mine: both lanes end with the KC smallest of the union
template <int KC, int CTRL>
__device__ __forceinline__ void dpp_merge(TopKLex<KC>& top) {
  unsigned long long o[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) o[t] = dpp_u64<CTRL>(top.key[t]);
  if constexpr ((KC & (KC - 1)) == 0 && KC >= 2) {
#pragma unroll
    for (int t = 0; t < KC; ++t) {  // min(a[t], o[KC-1-t]) is bitonic and holds the KC smallest
      const unsigned long long b = o[KC - 1 - t];
      top.key[t] = b < top.key[t] ? b : top.key[t];
    }
    bitonic_merge<KC>(top.key);
  } else {
#pragma unroll
    for (int t = 0; t < KC; ++t) {
      if (o[t] < top.key[KC - 1]) top.insert(o[t]);
    }
  }
}

template <int D, int KC, int NORM>
__global__ __launch_bounds__(kGridWave) void knn_grid_quad_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const float* __restrict__ edges,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted, const int* __restrict__ fb_count,
    const int* __restrict__ fb_list, const unsigned* __restrict__ fb_kth, int* __restrict__ fb3_count,
    int* __restrict__ fb3_list, int cell_cap, int P1, int P2, int K, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  constexpr bool kUseQueue = KC >= 8 && (KC & (KC - 1)) == 0;
  constexpr int kQueueCap = KC < 16 ? KC : 16;
  constexpr int kSub = 4;
  static_assert(kQuadFetch % kSub == 0, "fetch groups are processed four candidates at a time");
  __shared__ unsigned long long s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  __shared__ int2 s_rows[kQuadRows][kGridWave];

  const int n = blockIdx.y;
  const int cnt = fb_count[n];
  const int lane = threadIdx.x;
  const int sub = lane & (kQuadLanes - 1);
  const GridCloud g = clouds[n];
  const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
  const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + 1);  // + 1: the cloud's NaN sentinel record
  const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;

  for (int base = blockIdx.x * kQuadQueries; base < cnt; base += gridDim.x * kQuadQueries) {
    const int w = base + lane / kQuadLanes;
    const bool active = w < cnt;
    const int qi = active ? fb_list[(int64_t)n * P1 + w] : 0;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, qx, qy, qz);
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    // The cube grows by one cell only past the faces of the 3x3x3 cube that the lane search's KC-th best
    // reached (a superset search can only lower it, so the other faces stay certified): typically one
    // face -> 36 cells instead of 125.  Whatever cube is searched is certified against ITS faces below.
    const float kth3 = __uint_as_float(active ? fb_kth[(int64_t)n * P1 + w] : 0x7f800000u);
    auto reach = [&](bool has, float bound) { return (has && !(kth3 < bound)) ? 2 : 1; };
    const int ex0 = reach(cx - 1 > 0, face_bound<NORM>(qx - prev_float(ed[max(cx - 1, 0)])));
    const int ex1 = reach(cx + 1 < g.G[0] - 1, face_bound<NORM>(ed[min(cx + 2, g.G[0])] - qx));
    const int ey0 = reach(cy - 1 > 0, face_bound<NORM>(qy - prev_float(ed[kEdgeStride + max(cy - 1, 0)])));
    const int ey1 = reach(cy + 1 < g.G[1] - 1, face_bound<NORM>(ed[kEdgeStride + min(cy + 2, g.G[1])] - qy));
    const int ez0 = reach(cz - 1 > 0, face_bound<NORM>(qz - prev_float(ed[2 * kEdgeStride + max(cz - 1, 0)])));
    const int ez1 = reach(cz + 1 < g.G[2] - 1, face_bound<NORM>(ed[2 * kEdgeStride + min(cz + 2, g.G[2])] - qz));
    const int X0 = max(cx - ex0, 0), X1 = min(cx + ex1, g.G[0] - 1);
    const int Y0 = max(cy - ey0, 0), Y1 = min(cy + ey1, g.G[1] - 1);
    const int Z0 = max(cz - ez0, 0), Z1 = min(cz + ez1, g.G[2] - 1);

    const bool hx0 = X0 > 0, hx1 = X1 < g.G[0] - 1;
    const bool hy0 = Y0 > 0, hy1 = Y1 < g.G[1] - 1;
    const bool hz0 = Z0 > 0, hz1 = Z1 < g.G[2] - 1;
    float lb = __builtin_inff();
    if (hx0) lb = fminf(lb, face_bound<NORM>(qx - prev_float(ed[X0])));
    if (hx1) lb = fminf(lb, face_bound<NORM>(ed[X1 + 1] - qx));
    if (hy0) lb = fminf(lb, face_bound<NORM>(qy - prev_float(ed[kEdgeStride + Y0])));
    if (hy1) lb = fminf(lb, face_bound<NORM>(ed[kEdgeStride + Y1 + 1] - qy));
    if (hz0) lb = fminf(lb, face_bound<NORM>(qz - prev_float(ed[2 * kEdgeStride + Z0])));
    if (hz1) lb = fminf(lb, face_bound<NORM>(ed[2 * kEdgeStride + Z1 + 1] - qz));
    const bool whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);
    const unsigned thr0 = seed_threshold(lb, whole);

#pragma unroll
    for (int j = 0; j < kQuadRows; ++j) {
      const int rr = sub + kQuadLanes * j;  // table entries >= 25 do not exist
      const int z = cz + kQuadDz[rr], y = cy + kQuadDy[rr];
      int2 se = make_int2(0, 0);
      if (active && rr < 25 && z >= Z0 && z <= Z1 && y >= Y0 && y <= Y1) {
        const int rowbase = (z * g.G[1] + y) * g.G[0];
        se.x = cstart[rowbase + X0];
        se.y = cstart[rowbase + X1 + 1];
      }
      s_rows[j][lane] = se;
    }
    int r = 0;
    int cur = s_rows[0][lane].x, end = s_rows[0][lane].y;
    auto next_record = [&]() -> int {
      while (cur >= end && r < kQuadRows - 1) {
        ++r;
        const int2 se = s_rows[r][lane];
        cur = se.x;
        end = se.y;
      }
      return cur < end ? cur++ : P2;  // exhausted: the NaN sentinel record
    };

    TopKLex<KC> top;
    top.init();
    unsigned thr = thr0;
    int qn = 0;
    auto flush = [&]() {
      unsigned long long qk[kQueueCap];
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned long long v = s_queue[t * kGridWave + lane];
        qk[t] = t < qn ? v : TopKLex<KC>::kEmpty;
      }
      bitonic_sort<kQueueCap>(qk);
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned long long a = top.key[KC - 1 - t];
        top.key[KC - 1 - t] = qk[t] < a ? qk[t] : a;
      }
      bitonic_merge<KC>(top.key);
      qn = 0;
      thr = min(top.worst_bits(), thr0);
    };
    auto fetch = [&](float4 (&c)[kQuadFetch]) -> bool {
      int a[kQuadFetch];
#pragma unroll
      for (int u = 0; u < kQuadFetch; ++u) a[u] = next_record();
#pragma unroll
      for (int u = 0; u < kQuadFetch; ++u) c[u] = sp[a[u]];  // unconditional: exhausted lanes read the sentinel
      return a[0] != P2;
    };
    float4 c[kQuadFetch];
    bool more = fetch(c);
    while (__any(more)) {
      float4 nxt[kQuadFetch];
      const bool more_next = fetch(nxt);
#pragma unroll
      for (int u0 = 0; u0 < kQuadFetch; u0 += kSub) {
#pragma unroll
        for (int u = u0; u < u0 + kSub; ++u) {
          float d;
          if (NORM == 1) {
            d = __builtin_fabsf(qx - c[u].x);
            if (D > 1) d = d + __builtin_fabsf(qy - c[u].y);
            if (D > 2) d = d + __builtin_fabsf(qz - c[u].z);
          } else {
            const float dx = qx - c[u].x;
            d = dx * dx;
            if (D > 1) {
              const float dy = qy - c[u].y;
              d = d + dy * dy;
            }
            if (D > 2) {
              const float dz = qz - c[u].z;
              d = d + dz * dz;
            }
          }
          if (kUseQueue) {
            if (__float_as_uint(d) <= thr) {
              s_queue[qn * kGridWave + lane] = TopKLex<KC>::make(d, __float_as_int(c[u].w));
              ++qn;
            }
          } else if (__float_as_uint(d) <= min(top.worst_bits(), thr0)) {
            const unsigned long long key = TopKLex<KC>::make(d, __float_as_int(c[u].w));
            if (key < top.key[KC - 1]) top.insert(key);
          }
        }
        if (kUseQueue) {
          if (__any(qn > kQueueCap - kSub)) flush();
        }
      }
#pragma unroll
      for (int u = 0; u < kQuadFetch; ++u) c[u] = nxt[u];
      more = more_next;
    }
    if (kUseQueue) flush();

    // the group's sorted lists -> one, held by every lane of the group
    dpp_merge<KC, kDppXor1>(top);
    dpp_merge<KC, kDppXor2>(top);
    if (kQuadLanes >= 8) dpp_merge<KC, kDppHalfMirror>(top);
    if (kQuadLanes >= 16) dpp_merge<KC, kDppMirror>(top);

    const unsigned kth_bits = top.kth_bits(K);  // the K-th best, not the list's last slot
    const bool full = kth_bits < 0x7f800000u;
    const bool ok = whole || (full && __uint_as_float(kth_bits) < lb);
    if (active && sub == 0) {
      if (ok) {
        const int64_t row = (int64_t)n * P1 + qi;
        write_row<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
      } else {
        const int pos = atomicAdd(fb3_count + n, 1);
        fb3_list[(int64_t)n * P1 + pos] = qi;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass 6: wave-per-query EXPANDING search for the queries pass 5 could not certify
// (typically < 1 % of a cloud).  The wave's 64 lanes split the candidate records of
// the cube of cells [c - r, c + r]^3 around the query's cell (coalesced 16-byte
// loads along each row's contiguous run), keep a private lexicographic top-K each,
// and K rounds of a wave-wide 64-bit min extract the K global minima.  The same
// rigorous face bound decides; on failure r doubles, until the cube is the whole
// grid (always exact) or more than kWaveRegionCap records were scanned, in which
// case the query goes to the whole-cloud lane-per-query scan (far-away queries).
// ---------------------------------------------------------------------------
constexpr int kWaveKernelBlock = 256;
constexpr int kWaveKernelWgsPerCloud = 64;
constexpr int kWaveRegionCap = 16384;
constexpr int kWaveRows = 96;  // (2r+1)^2 rows for r = 2 (25) and r = 4 (81)

template <int D, int KC, int NORM>
__global__ __launch_bounds__(kWaveKernelBlock) void knn_grid_wave_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const float* __restrict__ edges,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted, const int* __restrict__ fb_count,
    const int* __restrict__ fb_list, int* __restrict__ fb2_count, int* __restrict__ fb2_list, int cell_cap,
    int P1, int P2, int K, int r_start, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  __shared__ int s_rowsrc[kWaveKernelBlock / kWave][kWaveRows];
  __shared__ int s_rowoff[kWaveKernelBlock / kWave][kWaveRows + 1];
  const int n = blockIdx.y;
  const int cnt = fb_count[n];
  if (cnt == 0) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int wslot = threadIdx.x / kWave;
  const int wave = blockIdx.x * (kWaveKernelBlock / kWave) + threadIdx.x / kWave;
  constexpr int kWavesPerCloud = kWaveKernelWgsPerCloud * (kWaveKernelBlock / kWave);
  const GridCloud g = clouds[n];
  const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
  const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
  const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + 1);  // + 1: the cloud's NaN sentinel record
  const int kvalid = g.len2 < K ? g.len2 : K;

  for (int w = wave; w < cnt; w += kWavesPerCloud) {
    const int qi = fb_list[(int64_t)n * P1 + w];  // wave-uniform
    const int64_t row = (int64_t)n * P1 + qi;
    float qx, qy, qz;
    load_point3<D>(p1 + row * D, qx, qy, qz);
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    bool done = false;
    for (int r = r_start; !done; r *= 2) {
      const int X0 = max(cx - r, 0), X1 = min(cx + r, g.G[0] - 1);
      const int Y0 = max(cy - r, 0), Y1 = min(cy + r, g.G[1] - 1);
      const int Z0 = max(cz - r, 0), Z1 = min(cz + r, g.G[2] - 1);
      const bool hx0 = X0 > 0, hx1 = X1 < g.G[0] - 1;
      const bool hy0 = Y0 > 0, hy1 = Y1 < g.G[1] - 1;
      const bool hz0 = Z0 > 0, hz1 = Z1 < g.G[2] - 1;
      const bool whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);
      TopKLex<KC> top;
      top.init();
      auto consider = [&](const float4 c) {
        float d;
        if (NORM == 1) {
          d = __builtin_fabsf(qx - c.x);
          if (D > 1) d = d + __builtin_fabsf(qy - c.y);
          if (D > 2) d = d + __builtin_fabsf(qz - c.z);
        } else {
          const float dx = qx - c.x;
          d = dx * dx;
          if (D > 1) {
            const float dy = qy - c.y;
            d = d + dy * dy;
          }
          if (D > 2) {
            const float dz = qz - c.z;
            d = d + dz * dz;
          }
        }
        if (__float_as_uint(d) <= top.worst_bits()) {
          const unsigned long long key = TopKLex<KC>::make(d, __float_as_int(c.w));
          if (key < top.key[KC - 1]) top.insert(key);
        }
      };
      int scanned = 0;
      bool giveup = false;
      const int ny = Y1 - Y0 + 1, nrows = ny * (Z1 - Z0 + 1);
      if (nrows <= kWaveRows) {
        // Small cubes (r = 2, 4): latency-bound if walked row by row (two dependent scalar loads
        // per row, then one load per lane).  Instead the lanes fetch all row bounds at once,
        // a wave scan turns them into a flat record stream, and every lane then owns the
        // records lane, lane+64, ... with four loads in flight.
        int* __restrict__ rs = s_rowsrc[wslot];
        int* __restrict__ ro = s_rowoff[wslot];
        for (int r0 = 0; r0 < nrows; r0 += kWave) {
          const int rr = r0 + lane;
          int len_r = 0, src = 0;
          if (rr < nrows) {
            const int z = Z0 + rr / ny, y = Y0 + rr % ny;
            const int rowbase = (z * g.G[1] + y) * g.G[0];
            src = cstart[rowbase + X0];
            len_r = cstart[rowbase + X1 + 1] - src;
          }
          int inc = len_r;  // inclusive wave scan
#pragma unroll
          for (int off = 1; off < kWave; off <<= 1) {
            const int v = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += v;
          }
          if (rr < nrows) {
            rs[rr] = src;
            ro[rr + 1] = scanned + inc;
          }
          scanned += __shfl(inc, kWave - 1, kWave);
        }
        if (lane == 0) ro[0] = 0;
        const int T = scanned;
        if (!whole && T > kWaveRegionCap) {
          giveup = true;
        } else {
          int rrow = 0;
          for (int t0 = lane; t0 < T; t0 += 4 * kWave) {
            float4 c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int t = t0 + u * kWave;
              const float qnan = __uint_as_float(0x7fc00000u);
              c[u] = make_float4(qnan, qnan, qnan, 0.f);
              if (t < T) {
                while (ro[rrow + 1] <= t) ++rrow;
                c[u] = sp[rs[rrow] + (t - ro[rrow])];
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) consider(c[u]);
          }
        }
      } else {
        for (int z = Z0; z <= Z1 && !giveup; ++z) {
          for (int y = Y0; y <= Y1; ++y) {
            const int rowbase = (z * g.G[1] + y) * g.G[0];
            const int s = cstart[rowbase + X0], e = cstart[rowbase + X1 + 1];
            for (int j = s + lane; j < e; j += kWave) consider(sp[j]);
            scanned += e - s;
          }
          if (!whole && scanned > kWaveRegionCap) giveup = true;
        }
      }
      if (giveup) {
        if (lane == 0) {
          const int pos = atomicAdd(fb2_count + n, 1);
          fb2_list[(int64_t)n * P1 + pos] = qi;
        }
        break;
      }
      // K rounds: wave-wide lexicographic minimum of the list heads; the (unique) winner pops
      unsigned long long mine = TopKLex<KC>::kEmpty, kth = TopKLex<KC>::kEmpty;
      for (int k = 0; k < kvalid; ++k) {
        unsigned long long m = top.key[0];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
          const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(m >> 32), off, kWave);
          const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)m, off, kWave);
          const unsigned long long o = ((unsigned long long)hi << 32) | lo;
          m = o < m ? o : m;
        }
        if (lane == k) mine = m;
        kth = m;
        if (top.key[0] == m) {
#pragma unroll
          for (int s2 = 0; s2 + 1 < KC; ++s2) top.key[s2] = top.key[s2 + 1];
          top.key[KC - 1] = TopKLex<KC>::kEmpty;
        }
      }
      float lb = __builtin_inff();
      if (hx0) lb = fminf(lb, face_bound<NORM>(qx - prev_float(ed[X0])));
      if (hx1) lb = fminf(lb, face_bound<NORM>(ed[X1 + 1] - qx));
      if (hy0) lb = fminf(lb, face_bound<NORM>(qy - prev_float(ed[kEdgeStride + Y0])));
      if (hy1) lb = fminf(lb, face_bound<NORM>(ed[kEdgeStride + Y1 + 1] - qy));
      if (hz0) lb = fminf(lb, face_bound<NORM>(qz - prev_float(ed[2 * kEdgeStride + Z0])));
      if (hz1) lb = fminf(lb, face_bound<NORM>(ed[2 * kEdgeStride + Z1 + 1] - qz));
      const unsigned kth_bits = (unsigned)(kth >> 32);
      const bool full = kvalid == K && kth_bits < 0x7f800000u;
      if (whole || (full && __uint_as_float(kth_bits) < lb)) {
        if (lane < K) {
          const bool ok = lane < kvalid;
          idxs[row * K + lane] = ok ? (int64_t)(int)(unsigned)mine : 0;
          dists[row * K + lane] = ok ? __uint_as_float((unsigned)(mine >> 32)) : 0.0f;
        }
        done = true;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// BALL QUERY through the same grid (ball_query.hip owns the operator; reference semantics
// ball_query_cpu.cpp:12-54: the first K points in INDEX order with dist2 < radius2).  Cells are at
// least 1.001 radius wide, so the 3x3x3 cube around the query's cell contains its ball; that is
// not assumed but CERTIFIED per query with the face bound of the KNN search (lb >= radius2: no
// unvisited point can pass `dist2 < radius2`), anything else goes to the index-order scan.  One
// lane per query walks its nine runs exactly like knn_grid_lane_kernel; a hit pushes its INDEX
// into the lane's LDS queue, queues are merged into a sorted register list of the KC smallest
// indices by the 32-bit sorting networks (v_min_u32 / v_max_u32 per compare-exchange), and a full
// list prunes by its largest index.  The list is the output order; distances are recomputed from
// the chosen points with the scan kernel's expression.
// ---------------------------------------------------------------------------
template <int D, int KC>
__global__ __launch_bounds__(kGridWave) void ball_grid_lane_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const GridCloud* __restrict__ clouds,
    const int* __restrict__ chunk_prefix, const float* __restrict__ edges, const int* __restrict__ cell_start,
    const float4* __restrict__ sorted, const int* __restrict__ qlist, int* __restrict__ fb_count,
    int* __restrict__ fb_list, int cell_cap, int P1, int P2, int K, int N, float radius2,
    int64_t* __restrict__ idxs, float* __restrict__ dists) {
  constexpr int kQueueCap = KC < 16 ? KC : 16;
  constexpr int kSub = 4;
  constexpr unsigned kNone = 0xffffffffu;
  __shared__ unsigned s_queue[kQueueCap * kGridWave];
  __shared__ int2 s_rows[kLaneRows + 1][kGridWave];  // row 9 = empty

  const int lane = threadIdx.x;
  const int total = chunk_prefix[N];
  // XCD-aware item order: workgroup b runs on XCD b % 8 (round-robin dispatch), and the chunks are
  // sorted by (cloud, cell).  Each XCD walks its own contiguous eighth of the chunk list, so the
  // ~1000 chunks it has in flight belong to one or two clouds whose sorted records (1 MB at 65536
  // points) stay in that XCD's 4 MB L2, instead of every XCD touching every cloud in flight.
  const int xcd = blockIdx.x % kNumXcd, per_xcd = (total + kNumXcd - 1) / kNumXcd;
  for (int j = blockIdx.x / kNumXcd; j < per_xcd; j += gridDim.x / kNumXcd) {
    const int item = xcd * per_xcd + j;
    if (item >= total) break;
    int lo_n = 0, hi_n = N;
    while (hi_n - lo_n > 1) {
      const int mid = (lo_n + hi_n) >> 1;
      if (chunk_prefix[mid] <= item) lo_n = mid;
      else hi_n = mid;
    }
    const int n = lo_n;
    const GridCloud g = clouds[n];
    const int c0 = (item - chunk_prefix[n]) * kGridWave;
    const bool active = c0 + lane < g.len1;
    const int qi = active ? qlist[(int64_t)n * P1 + c0 + lane] : 0;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, qx, qy, qz);
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    const int X0 = max(cx - 1, 0), X1 = min(cx + 1, g.G[0] - 1);
    const int Y0 = max(cy - 1, 0), Y1 = min(cy + 1, g.G[1] - 1);
    const int Z0 = max(cz - 1, 0), Z1 = min(cz + 1, g.G[2] - 1);
    const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
    const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + 1);  // + 1: the cloud's NaN sentinel record

    // certification first: a lane whose cube cannot be proven to contain its ball does not walk
    const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
    const bool hx0 = X0 > 0, hx1 = X1 < g.G[0] - 1;
    const bool hy0 = Y0 > 0, hy1 = Y1 < g.G[1] - 1;
    const bool hz0 = Z0 > 0, hz1 = Z1 < g.G[2] - 1;
    float lb = __builtin_inff();
    if (hx0) lb = fminf(lb, face_bound<2>(qx - prev_float(ed[X0])));
    if (hx1) lb = fminf(lb, face_bound<2>(ed[X1 + 1] - qx));
    if (hy0) lb = fminf(lb, face_bound<2>(qy - prev_float(ed[kEdgeStride + Y0])));
    if (hy1) lb = fminf(lb, face_bound<2>(ed[kEdgeStride + Y1 + 1] - qy));
    if (hz0) lb = fminf(lb, face_bound<2>(qz - prev_float(ed[2 * kEdgeStride + Z0])));
    if (hz1) lb = fminf(lb, face_bound<2>(ed[2 * kEdgeStride + Z1 + 1] - qz));
    const bool whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);
    const bool ok = whole || lb >= radius2;  // every unvisited point has computed dist2 >= lb
    const bool walk = active && ok;

#pragma unroll
    for (int r = 0; r < kLaneRows; ++r) {
      constexpr int kDz[kLaneRows] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
      constexpr int kDy[kLaneRows] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
      const int z = cz + kDz[r], y = cy + kDy[r];
      int2 se = make_int2(0, 0);
      if (walk && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
        const int rowbase = (z * g.G[1] + y) * g.G[0];
        se.x = cstart[rowbase + X0];
        se.y = cstart[rowbase + X1 + 1];
      }
      s_rows[r][lane] = se;
    }
    s_rows[kLaneRows][lane] = make_int2(0, 0);
    int r = 0;
    int cur = s_rows[0][lane].x, end = s_rows[0][lane].y;
    auto next_group = [&](int* a) __attribute__((always_inline)) {  // see knn_grid_lane_kernel
      const int2 nx = s_rows[r + 1][lane];
      const int left = end - cur;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        const int over = u - left;
        const int b = nx.x + over;
        a[u] = over < 0 ? cur + u : (b < nx.y ? b : P2);
      }
      const bool sw = left < kSub;
      const int ncur = nx.x + (kSub - left);
      cur = sw ? min(ncur, nx.y) : cur + kSub;
      end = sw ? nx.y : end;
      r += (sw && r < kLaneRows - 1) ? 1 : 0;
    };

    unsigned top[KC];  // ascending indices, kNone = empty
#pragma unroll
    for (int t = 0; t < KC; ++t) top[t] = kNone;
    unsigned thr = kNone;  // an index must be below the list's largest to matter (stale between flushes)
    int qn = 0;
    auto flush = [&]() {
      unsigned qk[kQueueCap];
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned v = s_queue[t * kGridWave + lane];
        qk[t] = t < qn ? v : kNone;
      }
      bitonic_sort<kQueueCap>(qk);
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) {
        const unsigned a = top[KC - 1 - t];
        top[KC - 1 - t] = qk[t] < a ? qk[t] : a;
      }
      bitonic_merge<KC>(top);
      qn = 0;
      thr = top[KC - 1];
    };
    auto fetch = [&](float4 (&c)[kSub]) __attribute__((always_inline)) -> bool {
      int a[kSub];
      next_group(a);
      bool real = false;
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        c[u] = sp[a[u]];  // unconditional: exhausted lanes read the sentinel
        real = real || a[u] != P2;
      }
      return real || cur < end || r < kLaneRows - 1;
    };
    float4 c[kSub];
    bool more = fetch(c);
    while (__any(more)) {
      float4 nxt[kSub];
      const bool more_next = fetch(nxt);
#pragma unroll
      for (int u = 0; u < kSub; ++u) {
        const float dx = qx - c[u].x;
        float d = dx * dx;
        if (D > 1) {
          const float dy = qy - c[u].y;
          d = d + dy * dy;
        }
        if (D > 2) {
          const float dz = qz - c[u].z;
          d = d + dz * dz;
        }
        const unsigned j = __float_as_uint(c[u].w);
        if (d < radius2 && j < thr) {
          s_queue[qn * kGridWave + lane] = j;
          ++qn;
        }
      }
      if (__any(qn > kQueueCap - kSub)) flush();
#pragma unroll
      for (int u = 0; u < kSub; ++u) c[u] = nxt[u];
      more = more_next;
    }
    flush();

    if (active) {
      if (ok) {
        const int64_t row = (int64_t)n * P1 + qi;
        int64_t* __restrict__ oi = idxs + row * K;
        float* __restrict__ od = dists + row * K;
        const float* __restrict__ pts = p2 + (int64_t)n * P2 * D;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          if (k < K) {
            const unsigned j = top[k];
            const bool hit = j != kNone;
            float d = 0.0f;
            if (hit) {
              float bx, by, bz;
              load_point3<D>(pts + (int64_t)j * D, bx, by, bz);
              const float dx = qx - bx;
              d = dx * dx;
              if (D > 1) {
                const float dy = qy - by;
                d = d + dy * dy;
              }
              if (D > 2) {
                const float dz = qz - bz;
                d = d + dz * dz;
              }
            }
            oi[k] = hit ? (int64_t)j : -1;
            od[k] = d;
          }
        }
      } else {
        const int pos = atomicAdd(fb_count + n, 1);
        fb_list[(int64_t)n * P1 + pos] = qi;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static bool grid_lane_mode();

// list capacity of the search kernels.  24 only exists for the block form: in the lane-private form a
// 24-slot list has no sorting network (direct inserts: 2.9 ms at cfg2 size) and loses to the 32-slot
// queue/network variant (2.4 ms).
static int grid_kc(int K) {
  return K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : (K <= 24 && !grid_lane_mode()) ? 24 : 32;
}

static void grid_tuning(int K, float* c_target, int* B) {
  // the search keeps the KC >= K best but certifies the K-th, so the cells are sized for K points:
  // measured optimum at B=32, N=65536 (profiles/r01_grid_tuning.txt): 0.4 K for the queue/network
  // variants (KC >= 8; fewer candidates per cube at ~1 % uncertified queries), 0.625 K for the
  // direct-insert variants (re-swept after the walk got cheaper: K=4 0.53 -> 0.51 ms)
  const int kc = grid_kc(K);
  float c = (kc >= 8 ? 0.4f : 0.625f) * (float)K;
  if (const char* e = getenv("POINTOPS_GRID_C_SCALE")) c *= (float)atof(e);  // tuning experiments only
  if (c < 1.0f) c = 1.0f;
  int b = (int)lround(cbrt(64.0 / (double)c));  // ~64 queries (one wave) per block
  if (b < 1) b = 1;
  if (b > 8) b = 8;
  *c_target = c;
  *B = b;
}

static int grid_cell_cap(int64_t P2, float c_target) {
  const int64_t cells = (int64_t)ceil((double)P2 / (double)c_target);
  return (int)(2 * cells + 64);
}

static size_t carve(GridWs* ws, char* base, int64_t N, int64_t P1, int64_t P2, float c) {
  const int cap = grid_cell_cap(P2, c);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align_up(bytes);
    return p;
  };
  GridWs w;
  w.cell_cap = cap;
  w.cloud = (GridCloud*)take(sizeof(GridCloud) * (size_t)N);
  w.block_prefix = (int*)take(sizeof(int) * (size_t)(N + 1));
  w.edges = (float*)take(sizeof(float) * (size_t)N * 3 * kEdgeStride);
  w.cell_count = (int*)take(sizeof(int) * (size_t)N * cap);
  w.blk_count = (int*)take(sizeof(int) * (size_t)N * cap);  // adjacent to cell_count: one memset
  w.cell_start = (int*)take(sizeof(int) * (size_t)N * (cap + 1));
  w.blk_start = (int*)take(sizeof(int) * (size_t)N * (cap + 1));
  w.sorted = (float4*)take(sizeof(float4) * (size_t)N * (size_t)(P2 + 1));
  w.qlist = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.fb_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.fb_kth = (unsigned*)take(sizeof(unsigned) * (size_t)N * (size_t)P1);
  w.fb2_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb2_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.fb3_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb3_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.bbox = (unsigned*)take(sizeof(unsigned) * (size_t)N * 8);
  w.scan_partial = (int*)take(sizeof(int) * (size_t)N * 2 * (size_t)((cap + kScanChunk - 1) / kScanChunk));
  w.grid_flag = (int*)take(sizeof(int) * (size_t)N);
  w.rank1 = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.rank2 = (int*)take(sizeof(int) * (size_t)N * (size_t)P2);
  w.ball = 0;
  if (ws) *ws = w;
  return off;
}

static float knn_cell_target(int K) {
  float c;
  int B;
  grid_tuning(K, &c, &B);
  return c;
}

size_t knn_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K) {
  return carve(nullptr, nullptr, N, P1, P2, knn_cell_target((int)K));
}

template <int D, int KC, int NORM>
static void launch_grid_search(const KnnArgs& a, const GridWs& ws, int wgs) {
  hipLaunchKernelGGL((knn_grid_kernel<D, KC, NORM>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1,
                     (const GridCloud*)ws.cloud, (const int*)ws.block_prefix, (const float*)ws.edges,
                     (const int*)ws.cell_start, (const float4*)ws.sorted, (const int*)ws.blk_start,
                     (const int*)ws.qlist, ws.fb_count, ws.fb_list, ws.cell_cap, a.P1, a.P2, a.K, (int)a.N,
                     a.idxs, a.dists);
}

static bool grid_lane_mode() {
  // POINTOPS_GRID_MODE=block selects the block-shared broadcast search (A/B measurements)
  const char* e = getenv("POINTOPS_GRID_MODE");
  return !(e && e[0] == 'b');
}

template <int D, int KC, int NORM>
static void launch_grid_lane(const KnnArgs& a, const GridWs& ws, int wgs) {
  hipLaunchKernelGGL((knn_grid_lane_kernel<D, KC, NORM>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1,
                     (const GridCloud*)ws.cloud, (const int*)ws.block_prefix, (const float*)ws.edges,
                     (const int*)ws.cell_start, (const float4*)ws.sorted, (const int*)ws.qlist, ws.fb_count,
                     ws.fb_list, ws.fb_kth, ws.cell_cap, a.P1, a.P2, a.K, (int)a.N, a.idxs, a.dists);
}

static bool grid_quad_mode(int P1) {
  // The quad pass has a ~90 us floor (one wave walking ~200 candidates per lane), which only pays
  // when a cloud sends it hundreds of queries: measured 1.16 vs 1.23 ms at 32 x 65536 queries, but
  // 0.33 vs 0.24 ms at 32 x 4096, where the expanding wave search takes the uncertified queries
  // directly.  POINTOPS_GRID_QUAD=0/1 forces the choice (A/B measurements).
  if (const char* e = getenv("POINTOPS_GRID_QUAD")) return e[0] != '0';
  return P1 >= 32768;
}

template <int D, int KC, int NORM>
static void launch_grid_wave(const KnnArgs& a, const GridWs& ws) {
  const bool quad = grid_quad_mode(a.P1);
  if (quad) {
    int64_t wx = a.P1 / (32 * kQuadQueries);  // a few % of a cloud arrive here
    wx = wx < 8 ? 8 : wx > 4096 ? 4096 : wx;
    hipLaunchKernelGGL((knn_grid_quad_kernel<D, KC, NORM>), dim3((unsigned)wx, (unsigned)a.N), dim3(kGridWave), 0,
                       a.stream, a.p1, (const GridCloud*)ws.cloud, (const float*)ws.edges, (const int*)ws.cell_start,
                       (const float4*)ws.sorted, (const int*)ws.fb_count, (const int*)ws.fb_list,
                       (const unsigned*)ws.fb_kth, ws.fb3_count, ws.fb3_list, ws.cell_cap, a.P1, a.P2, a.K, a.idxs,
                       a.dists);
  }
  hipLaunchKernelGGL((knn_grid_wave_kernel<D, KC, NORM>), dim3(kWaveKernelWgsPerCloud, (unsigned)a.N),
                     dim3(kWaveKernelBlock), 0, a.stream, a.p1, (const GridCloud*)ws.cloud, (const float*)ws.edges,
                     (const int*)ws.cell_start, (const float4*)ws.sorted,
                     (const int*)(quad ? ws.fb3_count : ws.fb_count), (const int*)(quad ? ws.fb3_list : ws.fb_list),
                     ws.fb2_count, ws.fb2_list, ws.cell_cap, a.P1, a.P2, a.K, quad ? 4 : 2, a.idxs, a.dists);
}

template <int D, int NORM>
static void dispatch_grid_k(const KnnArgs& a, const GridWs& ws, int wgs) {
  const int K = a.K;
  if (K <= 1) { if (grid_lane_mode()) launch_grid_lane<D, 1, NORM>(a, ws, wgs); else launch_grid_search<D, 1, NORM>(a, ws, wgs); launch_grid_wave<D, 1, NORM>(a, ws); }
  else if (K <= 2) { if (grid_lane_mode()) launch_grid_lane<D, 2, NORM>(a, ws, wgs); else launch_grid_search<D, 2, NORM>(a, ws, wgs); launch_grid_wave<D, 2, NORM>(a, ws); }
  else if (K <= 4) { if (grid_lane_mode()) launch_grid_lane<D, 4, NORM>(a, ws, wgs); else launch_grid_search<D, 4, NORM>(a, ws, wgs); launch_grid_wave<D, 4, NORM>(a, ws); }
  else if (K <= 8) { if (grid_lane_mode()) launch_grid_lane<D, 8, NORM>(a, ws, wgs); else launch_grid_search<D, 8, NORM>(a, ws, wgs); launch_grid_wave<D, 8, NORM>(a, ws); }
  else if (K <= 16) { if (grid_lane_mode()) launch_grid_lane<D, 16, NORM>(a, ws, wgs); else launch_grid_search<D, 16, NORM>(a, ws, wgs); launch_grid_wave<D, 16, NORM>(a, ws); }
  else if (K <= 24 && !grid_lane_mode()) { launch_grid_search<D, 24, NORM>(a, ws, wgs); launch_grid_wave<D, 24, NORM>(a, ws); }
  else { if (grid_lane_mode()) launch_grid_lane<D, 32, NORM>(a, ws, wgs); else launch_grid_search<D, 32, NORM>(a, ws, wgs); launch_grid_wave<D, 32, NORM>(a, ws); }
}

template <int D>
static void build_d(const KnnArgs& a, const GridWs& ws) {
  const dim3 g2((unsigned)ceil_div(a.P2, kBinTile), (unsigned)a.N), g1((unsigned)ceil_div(a.P1, kBinTile), (unsigned)a.N);
  hipLaunchKernelGGL((grid_bin_kernel<D, false, false>), g2, dim3(kBinBlock), 0, a.stream, a.p2, a.P2, a.K, ws, a.idxs,
                     a.dists);
  hipLaunchKernelGGL((grid_bin_kernel<D, false, true>), g1, dim3(kBinBlock), 0, a.stream, a.p1, a.P1, a.K, ws, a.idxs,
                     a.dists);
  const int chunks = (ws.cell_cap + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(grid_scan_partial_kernel, dim3((unsigned)chunks, (unsigned)a.N, 2), dim3(kScanBlock), 0, a.stream,
                     ws, chunks);
  hipLaunchKernelGGL(grid_scan_offsets_kernel, dim3((unsigned)a.N, 2), dim3(kScanBlock), 0, a.stream, ws, chunks);
  hipLaunchKernelGGL(grid_scan_apply_kernel, dim3((unsigned)chunks, (unsigned)a.N, 2), dim3(kScanBlock), 0, a.stream,
                     ws, chunks);
  hipLaunchKernelGGL((grid_bin_kernel<D, true, false>), g2, dim3(kBinBlock), 0, a.stream, a.p2, a.P2, a.K, ws, a.idxs,
                     a.dists);
  hipLaunchKernelGGL((grid_bin_kernel<D, true, true>), g1, dim3(kBinBlock), 0, a.stream, a.p1, a.P1, a.K, ws, a.idxs,
                     a.dists);
}

template <int D>
static void run_d(const KnnArgs& a, int norm, const GridWs& ws, int wgs) {
  build_d<D>(a, ws);
  if (norm == 1) dispatch_grid_k<D, 1>(a, ws, wgs);
  else dispatch_grid_k<D, 2>(a, ws, wgs);
}

int knn_grid_run(const KnnArgs& a, int norm, void* workspace) {
  POINTOPS_REQUIRE(a.N < 65536, "knn_points_idx(grid): batch must be < 65536");
  POINTOPS_REQUIRE(a.P2 <= (1 << 20), "knn_points_idx(grid): P2 must be <= 2^20");
  GridWs ws;
  float c;
  int B;
  grid_tuning(a.K, &c, &B);
  carve(&ws, (char*)workspace, a.N, a.P1, a.P2, c);
  const bool lane_mode = grid_lane_mode();
  if (lane_mode) B = 1;  // queries sorted by cell
  // histogram buffers (cell_count and blk_count are adjacent) start at zero
  const size_t zero_bytes = (size_t)((char*)ws.cell_start - (char*)ws.cell_count);
  if (hipMemsetAsync(ws.cell_count, 0, zero_bytes, a.stream) != hipSuccess) return check_launch("knn grid memset");
  hipLaunchKernelGGL(grid_bbox_init_kernel, dim3((unsigned)ceil_div(a.N * 8, 256)), dim3(256), 0, a.stream, ws.bbox,
                     (int)a.N);
  hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)ceil_div(a.P2, kBboxBlock * kBboxPerThread), (unsigned)a.N),
                     dim3(kBboxBlock), 0, a.stream, a.p2, a.l2, a.P2, a.D, ws.bbox);
  hipLaunchKernelGGL(grid_setup_kernel, dim3((unsigned)a.N), dim3(kSetupBlock), 0, a.stream, a.p2, a.l1, a.l2,
                     a.P1, a.P2, a.D, c, B, 0.0f, 0.0f, 0, 0.0f, ws);
  hipLaunchKernelGGL(grid_prefix_kernel, dim3(1), dim3(64), 0, a.stream, ws, (int)a.N, lane_mode ? 1 : 0);
  const int wgs = 256 * 32;  // one wave64 per workgroup, up to 32 waves per CU resident
  switch (a.D) {
    case 1: run_d<1>(a, norm, ws, wgs); break;
    case 2: run_d<2>(a, norm, ws, wgs); break;
    default: run_d<3>(a, norm, ws, wgs); break;
  }
  int rc = check_launch("knn_points_idx(grid)");
  if (rc != POINTOPS_OK) return rc;
  // exact fallback: whole-cloud scan for the queries the bound could not certify
  KnnArgs fa = a;
  fa.qlist = ws.fb2_list;
  fa.qcount = ws.fb2_count;
  launch_knn_bruteforce(fa, norm);
  return check_launch("knn_points_idx(grid fallback)");
}

// ---------------------------------------------------------------------------
// ball query host side
// ---------------------------------------------------------------------------
constexpr float kBallCellTarget = 2.0f;  // density floor of the cell size; the radius usually decides

size_t ball_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2) {
  return carve(nullptr, nullptr, N, P1, P2, kBallCellTarget);
}

template <int D>
static void ball_run_d(const KnnArgs& a, float radius2, const GridWs& ws, int wgs) {
  build_d<D>(a, ws);
#define PO_BALL(KC)                                                                                              \
  hipLaunchKernelGGL((ball_grid_lane_kernel<D, KC>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1, a.p2, \
                     (const GridCloud*)ws.cloud, (const int*)ws.block_prefix, (const float*)ws.edges,            \
                     (const int*)ws.cell_start, (const float4*)ws.sorted, (const int*)ws.qlist, ws.fb2_count,    \
                     ws.fb2_list, ws.cell_cap, a.P1, a.P2, a.K, (int)a.N, radius2, a.idxs, a.dists)
  if (a.K <= 8) PO_BALL(8);
  else if (a.K <= 16) PO_BALL(16);
  else if (a.K <= 32) PO_BALL(32);
  else PO_BALL(64);
#undef PO_BALL
}

// Builds the grids and answers every query it can certify.  On return (stream order) flag[n] = 1 for
// the clouds that were searched through their grid -- for those only the qcount[n] queries of
// qlist[n * P1 ..] are left -- and 0 for the clouds the index-order scan has to do in full.
int ball_grid_run(const KnnArgs& a, float radius, void* workspace, const int** flag, const int** qcount,
                  const int** qlist) {
  POINTOPS_REQUIRE(a.N < 65536 && a.P2 <= (1 << 20) && a.K <= 64 && a.D <= 3, "ball_query(grid): unsupported shape");
  GridWs ws;
  carve(&ws, (char*)workspace, a.N, a.P1, a.P2, kBallCellTarget);
  ws.ball = 1;
  const size_t zero_bytes = (size_t)((char*)ws.cell_start - (char*)ws.cell_count);
  if (hipMemsetAsync(ws.cell_count, 0, zero_bytes, a.stream) != hipSuccess) return check_launch("ball grid memset");
  hipLaunchKernelGGL(grid_bbox_init_kernel, dim3((unsigned)ceil_div(a.N * 8, 256)), dim3(256), 0, a.stream, ws.bbox,
                     (int)a.N);
  hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)ceil_div(a.P2, kBboxBlock * kBboxPerThread), (unsigned)a.N),
                     dim3(kBboxBlock), 0, a.stream, a.p2, a.l2, a.P2, a.D, ws.bbox);
  const float h_min = fabsf(radius) * 1.001f;
  float factor = 5.0f;  // measured crossover (grid wins where K len2 / (E max(E, K)) > ~4-7), profiles/r01_ball_crossover.txt
  if (const char* e = getenv("POINTOPS_BALL_FACTOR")) factor = (float)atof(e);  // tuning experiments only
  hipLaunchKernelGGL(grid_setup_kernel, dim3((unsigned)a.N), dim3(kSetupBlock), 0, a.stream, a.p2, a.l1, a.l2,
                     a.P1, a.P2, a.D, kBallCellTarget, 1, h_min, fabsf(radius), a.K, factor, ws);
  hipLaunchKernelGGL(grid_prefix_kernel, dim3(1), dim3(64), 0, a.stream, ws, (int)a.N, 1);
  const int wgs = 256 * 32;
  const float radius2 = radius * radius;  // fp32 product (ball_query_cpu.cpp:26)
  switch (a.D) {
    case 1: ball_run_d<1>(a, radius2, ws, wgs); break;
    case 2: ball_run_d<2>(a, radius2, ws, wgs); break;
    default: ball_run_d<3>(a, radius2, ws, wgs); break;
  }
  *flag = ws.grid_flag;
  *qcount = ws.fb2_count;
  *qlist = ws.fb2_list;
  return check_launch("ball_query(grid)");
}

}  // namespace pointops

extern "C" int pointops_knn_grid_fallback_counts(const void* workspace, int64_t N, int64_t P1, int64_t P2,
                                                 int64_t K, int32_t* counts, void* stream) {
  using namespace pointops;
  POINTOPS_REQUIRE(workspace != nullptr && counts != nullptr && N > 0, "knn_grid_fallback_counts: bad arguments");
  GridWs ws;
  carve(&ws, (char*)workspace, N, P1, P2, knn_cell_target((int)K));
  if (hipMemcpyAsync(counts + N, ws.fb2_count, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice,
                     (hipStream_t)stream) != hipSuccess)
    return check_launch("knn_grid_fallback_counts");
  if (hipMemcpyAsync(counts, ws.fb_count, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice,
                     (hipStream_t)stream) != hipSuccess)
    return check_launch("knn_grid_fallback_counts");
  return POINTOPS_OK;
}
