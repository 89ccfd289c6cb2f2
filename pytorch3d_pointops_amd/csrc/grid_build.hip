// grid_build.hip -- build passes of the exact cell-grid searches (gfx950); see knn_grid.hip for the
// algorithm.  All clouds of the batch in every launch, no host synchronisation:
//   0 grid_bbox      bounding box of every p2 cloud (ordered-uint atomicMin/Max);
//   1 grid_setup     per cloud: cubic cell size h for ~c_target points per cell, G = cells per dimension,
//                    and per-dimension EDGE TABLES E_d[c] = min{ x in [lo,hi] : cell_d(x) >= c }, found by
//                    bisection over the ordered fp32 bit patterns of the (monotone) cell function itself --
//                    no error analysis of the binning arithmetic is needed;
//   2 grid_bin<cnt>  histograms: p2 points per cell, p1 queries per cell (LDS-aggregated, one global
//                    atomic per non-empty bin and tile); zero rows for padded queries;
//   3 grid_scan_*    chunked exclusive scans -> cell_start / qcell_start;
//   4 grid_bin<sct>  counting-sort p2 into (x,y,z,idx) float4 records, query ids into per-cell lists.
// When the queries ARE the points (same buffer, same lengths: self-KNN, ball query of a cloud on itself,
// get_point_covariances) the query passes are skipped: the point sort is the query order.
#include <stdlib.h>

#include "grid.h"

namespace pointops {

constexpr int kSetupBlock = 1024;  // 3 x 1026 edge bisections per cloud
constexpr int kScanBlock = 1024;
constexpr int kBinLdsBins = 40000;  // 156 KiB of LDS: the whole CU's LDS, one workgroup per CU

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
    const int64_t* __restrict__ lengths2, int P1, int P2, int D, float c_target, float h_min,
    float ball_radius, int ball_K, float ball_factor, int same, GridWs ws) {
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
            !(cells > (long long)kBinLdsBins && cells <= 2LL * kBinLdsBins))
          break;
        h *= cells > (long long)ws.cell_cap ? 1.2599211f : 1.04f;  // halve the cell count / nudge
      }
      if ((long long)G[0] * G[1] * G[2] > (long long)ws.cell_cap) ok = false;
    }
    for (int d = 0; d < 3; ++d) {
      g.lo[d] = lo[d];
      g.G[d] = G[d];
      s_hi[d] = hi[d];
    }
    g.inv_h = inv_h;
    g.ncell = G[0] * G[1] * G[2];
    g.len1 = len1;
    g.len2 = len2;
    g.use_grid = ok ? 1 : 0;
    g.same = same;
    s_g = g;
    ws.cloud[n] = g;
    const float qnan = __uint_as_float(0x7fc00000u);
    ws.sorted[(int64_t)n * (P2 + kSortedPad) + P2] = make_float4(qnan, qnan, qnan, 0.0f);
    ws.grid_flag[n] = g.use_grid;
    ws.fb_count[n] = 0;
    ws.fb2_count[n] = 0;
    ws.fb3_count[n] = 0;
    ws.rcount[n] = 0;
    ws.pool_top[n] = 0;
    ws.box_count[n] = 0;
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


// Chunk prefix of the lane searches: the unit of work is one CHUNK of 64 consecutive entries of a cloud's
// cell-sorted query order; chunk_prefix[n] = chunks of the clouds before cloud n.
__global__ void grid_prefix_kernel(GridWs ws, int N) {  // one wave
  const int lane = threadIdx.x;
  int acc = 0;
  if (lane == 0) ws.chunk_prefix[0] = 0;
  for (int n0 = 0; n0 < N; n0 += kWave) {
    const int n = n0 + lane;
    int items = 0;
    if (n < N) {
      const GridCloud g = ws.cloud[n];
      if (g.use_grid) items = (g.len1 + kGridWave - 1) / kGridWave;
    }
    int inc = items;  // inclusive wave scan
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int v = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += v;
    }
    if (n < N) ws.chunk_prefix[n + 1] = acc + inc;
    acc += __shfl(inc, kWave - 1, kWave);
  }
}

// ---------------------------------------------------------------------------
// pass 2 / 4: histogram and counting-sort scatter (SCATTER = false / true), for the
// points of p2 by cell (IS_QUERY = false) and the queries of p1 by cell (true).
//
// Scattered device-scope atomics run at only ~2e10/s chip-wide (they execute at the
// memory side, one 64-byte request each), so a workgroup first bins its tile of
// 1024 x 8 points in an LDS histogram (fast LDS atomics, which also hand every point
// its rank inside the (tile, bin) group) and then touches each non-empty global
// counter ONCE: count pass  global[bin] += n_tile ;  scatter pass  base = start[bin] +
// atomicAdd(cursor[bin], n_tile), position = base + rank.  Clouds with more bins than
// the LDS table holds (kBinLdsBins) spend one device atomic per point, or -- tiles with crowded cells -- hash the
// tile's bins into the same LDS (count pass), and keep every point's final rank for the scatter pass.
// PAD_ROWS: this launch also writes the rows that get no search (zeros / -1 for padded queries) and lists
// the queries of clouds without a usable grid for the whole-cloud scan; it is the query count pass, or the
// point count pass when the queries are the points.
// ---------------------------------------------------------------------------
constexpr int kBinBlock = 1024;
constexpr int kBinPerThread = 8;  // tile of 8192 points: 4096 / 8192 / 16384 / 32768 measured 0.988 / 0.969 / 0.987 / 1.125 ms per cfg2 step (chamfer cfg4: 1.14 / 1.13 / 1.22 / 1.66 ms)
constexpr int kBinTile = kBinBlock * kBinPerThread;
constexpr int kBinHashBits = 14, kBinHashSlots = 1 << kBinHashBits;  // hash table of the many-bin count pass
constexpr int kBinCrowdedPairs = 8;
static_assert(2 * kBinHashSlots <= kBinLdsBins && kBinHashSlots >= 2 * kBinTile, "hash table lives in s_hist");

// One launch bins BOTH sets: blockIdx.z = 0 the points of p2, 1 the queries of p1 (QUERIES = false: points only,
// the self-query case).
template <int D, bool SCATTER, bool QUERIES>
__global__ __launch_bounds__(kBinBlock) void grid_bin_kernel(const float* __restrict__ p2, int P2,
                                                           const float* __restrict__ p1, int P1, int K, GridWs ws,
                                                           int64_t* __restrict__ idxs, float* __restrict__ dists) {
  __shared__ int s_hist[kBinLdsBins];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const bool IS_QUERY = QUERIES && blockIdx.z == 1;   // (workgroup-uniform)
  const bool PAD_ROWS = QUERIES ? IS_QUERY : true;    // the pass over the query index space also pads rows
  const float* __restrict__ pts = IS_QUERY ? p1 : p2;
  const int P = IS_QUERY ? P1 : P2;
  const GridCloud g = ws.cloud[n];  // wave-uniform
  const int len = IS_QUERY ? g.len1 : g.len2;
  const int nbins = g.ncell;
  const int64_t cbase = (int64_t)n * ws.cell_cap;
  int* __restrict__ gcount = (IS_QUERY ? ws.qcell_count : ws.cell_count) + cbase;
  const int* __restrict__ gstart = (IS_QUERY ? ws.qcell_start : ws.cell_start) + (int64_t)n * (ws.cell_cap + 1);
  int* __restrict__ grank = (IS_QUERY ? ws.rank1 : ws.rank2) + (int64_t)n * P;
  const int i0 = blockIdx.x * kBinTile + tid;
  if (blockIdx.x * kBinTile >= P) return;

  if (PAD_ROWS && !SCATTER) {
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
      bin[r] = (cz * g.G[1] + cy) * g.G[0] + cx;
      if (SCATTER && !IS_QUERY) {
        px[r] = x;
        py[r] = y;
        pz[r] = z;
      }
      if (use_lds) {
        rank[r] = atomicAdd(&s_hist[bin[r]], 1);  // LDS atomic: rank inside (tile, bin)
      } else if (SCATTER) {
        rank[r] = gstart[bin[r]] + grank[i];  // final position
      }
    }
  }
  if (!use_lds && !SCATTER) {
    // Too many bins for a direct LDS table.  A tile of a cloud without crowded cells spends one device atomic per
    // point (nearly every point of the tile has its own bin; the pass runs at the chip's atomic rate).  A CROWDED
    // tile -- a cluster that puts half a cloud into one cell serialises tens of thousands of atomics on one address:
    // 1.6 ms of the K=1 count pass on 8 x 150 000 points -- goes through an LDS HASH table instead (bin -> count,
    // open addressing; <= 8192 distinct bins in 16384 slots) that hands out the rank inside (tile, bin), and spends
    // one device atomic per distinct (tile, bin).  Crowded = at least kBinCrowdedPairs of 36 736 sampled pairs of
    // the tile's points share a bin (pairs at index distances 1 and 1024 k: periodic interleavings of a cluster with
    // the rest of the cloud do not hide from all of them); a cell with a share f of the points gives 36 736 f^2 of
    // them, a uniform cloud of 2e5 cells 0.2.
    __shared__ int s_pairs;
    if (tid == 0) s_pairs = 0;
    __syncthreads();
    int pairs = 0;  // pairs of equal bins among this lane's 8 points (index distances 1024 k) and towards lane + 1
#pragma unroll
    for (int r = 0; r < kBinPerThread; ++r) {
      const int nb = __shfl_down(bin[r], 1, kWave);
      pairs += (bin[r] >= 0 && nb == bin[r] && (tid & (kWave - 1)) != kWave - 1) ? 1 : 0;
#pragma unroll
      for (int q = r + 1; q < kBinPerThread; ++q) pairs += (bin[r] >= 0 && bin[q] == bin[r]) ? 1 : 0;
    }
    if (pairs > 0) atomicAdd(&s_pairs, pairs);
    __syncthreads();
    const bool crowded = s_pairs >= kBinCrowdedPairs;  // (workgroup-uniform)
    if (!crowded) {
      // the point's rank in its bin, remembered so that the scatter pass needs no atomic at all
#pragma unroll
      for (int r = 0; r < kBinPerThread; ++r)
        if (bin[r] >= 0) rank[r] = atomicAdd(gcount + bin[r], 1);
#pragma unroll
      for (int r = 0; r < kBinPerThread; ++r)
        if (bin[r] >= 0) grank[i0 + r * kBinBlock] = rank[r];
    } else {
      for (int b = tid; b < kBinHashSlots; b += kBinBlock) {
        s_hist[b] = -1;                 // keys
        s_hist[kBinHashSlots + b] = 0;  // counts, then group bases
      }
      __syncthreads();
      int slot[kBinPerThread];
#pragma unroll
      for (int r = 0; r < kBinPerThread; ++r) {
        if (bin[r] < 0) continue;
        int h = (int)(((unsigned)bin[r] * 2654435761u) >> (32 - kBinHashBits));
        for (;;) {
          const int old = atomicCAS(&s_hist[h], -1, bin[r]);
          if (old == -1 || old == bin[r]) break;
          h = (h + 1) & (kBinHashSlots - 1);
        }
        slot[r] = h;
        rank[r] = atomicAdd(&s_hist[kBinHashSlots + h], 1);
      }
      __syncthreads();
      // (two loops: all of a thread's device atomics are in flight together; one loop waited for each return)
      constexpr int kSlotsPerThread = kBinHashSlots / kBinBlock;
      int base[kSlotsPerThread];
#pragma unroll
      for (int s = 0; s < kSlotsPerThread; ++s) {
        const int h = tid + s * kBinBlock;
        const int key = s_hist[h];
        base[s] = key >= 0 ? atomicAdd(gcount + key, s_hist[kBinHashSlots + h]) : 0;  // base of the group
      }
#pragma unroll
      for (int s = 0; s < kSlotsPerThread; ++s) s_hist[kBinHashSlots + tid + s * kBinBlock] = base[s];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kBinPerThread; ++r)
        if (bin[r] >= 0) grank[i0 + r * kBinBlock] = s_hist[kBinHashSlots + slot[r]] + rank[r];
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
        else ws.sorted[(int64_t)n * (P + kSortedPad) + pos] = make_float4(px[r], py[r], pz[r], __int_as_float(i));
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

// which: 0 = points per cell, 1 = queries per cell
__device__ __forceinline__ void scan_arrays(const GridWs& ws, int n, int which, int*& count, int*& start,
                                            int& len) {
  const GridCloud g = ws.cloud[n];
  count = (which == 0 ? ws.cell_count : ws.qcell_count) + (int64_t)n * ws.cell_cap;
  start = (which == 0 ? ws.cell_start : ws.qcell_start) + (int64_t)n * (ws.cell_cap + 1);
  len = g.use_grid ? g.ncell : 0;
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

// `refine` != 0: the pass over the point cells also marks the over-full ones for refinement (grid_refine.hip):
// a descriptor and a sub_start table from the cloud's pool, the sub-grid itself is built by refine_build.
__global__ __launch_bounds__(kScanBlock) void grid_scan_apply_kernel(GridWs ws, int chunks, int refine) {
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
      if (which == 0 && refine >= 0) {
        int ref = -1;
        if (refine > 0 && c[r] > refine_threshold(ws.c_target)) {
          int s = (int)ceilf(cbrtf((float)c[r] / ws.c_target));
          s = s < 2 ? 2 : (s > kRefineMaxS ? kRefineMaxS : s);
          const int cells = s * s * s + 1;
          const int idx = atomicAdd(ws.rcount + n, 1);
          if (idx < ws.rdesc_cap) {
            RefinedCell d{};  // count == 0: a descriptor without a table (pool exhausted), skipped by refine_build
            const int off = atomicAdd(ws.pool_top + n, cells);
            if (off + cells <= ws.pool_cap) {
              d.start = run;
              d.count = c[r];
              d.s = s;
              d.pool_off = off;
              ref = idx;
            }
            ws.rdesc[(int64_t)n * ws.rdesc_cap + idx] = d;
          }
        }
        ws.refine_ref[(int64_t)n * ws.cell_cap + i0 + r] = ref;
      }
      run += c[r];
      count[i0 + r] = 0;
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static int grid_cell_cap(int64_t P2, float c_target) {
  const int64_t cells = (int64_t)ceil((double)P2 / (double)c_target);
  return (int)(2 * cells + 64);
}

size_t grid_carve(GridWs* ws, char* base, int64_t N, int64_t P1, int64_t P2, float c) {
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
  w.chunk_prefix = (int*)take(sizeof(int) * (size_t)(N + 1));
  w.edges = (float*)take(sizeof(float) * (size_t)N * 3 * kEdgeStride);
  w.cell_count = (int*)take(sizeof(int) * (size_t)N * cap);
  w.qcell_count = (int*)take(sizeof(int) * (size_t)N * cap);  // adjacent to cell_count: one memset
  w.cell_start = (int*)take(sizeof(int) * (size_t)N * (cap + 1));
  w.qcell_start = (int*)take(sizeof(int) * (size_t)N * (cap + 1));
  w.sorted = (float4*)take(sizeof(float4) * (size_t)N * (size_t)(P2 + kSortedPad));
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
  w.rdesc_cap = (int)(P2 / 64 + 1);  // a refined cell holds more than refine_threshold() >= 64 points
  w.pool_cap = (int)(4 * P2 + 64);   // sum of (s^3 + 1) <= sum of (8 count / c + 9) over refined cells
  w.refine_ref = (int*)take(sizeof(int) * (size_t)N * cap);
  w.rdesc = (RefinedCell*)take(sizeof(RefinedCell) * (size_t)N * (size_t)w.rdesc_cap);
  w.rcount = (int*)take(sizeof(int) * (size_t)N);
  w.pool = (int*)take(sizeof(int) * (size_t)N * (size_t)w.pool_cap);
  w.pool_top = (int*)take(sizeof(int) * (size_t)N);
  w.sorted_tmp = (float4*)take(sizeof(float4) * (size_t)N * (size_t)P2);
  w.box_count = (int*)take(sizeof(int) * (size_t)N);
  w.box_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.c_target = c;
  w.ball = 0;
  if (ws) *ws = w;
  return off;
}

template <int D>
static void build_d(const KnnArgs& a, const GridWs& ws, bool same, int refine) {
  const int chunks = (ws.cell_cap + kScanChunk - 1) / kScanChunk;
  const unsigned which = same ? 1u : 2u;  // points only / points and queries
  const dim3 gb((unsigned)ceil_div(same ? a.P2 : (a.P2 > a.P1 ? a.P2 : a.P1), kBinTile), (unsigned)a.N, which);
#define PO_BIN(SCT, QRY)                                                                                          \
  hipLaunchKernelGGL((grid_bin_kernel<D, SCT, QRY>), gb, dim3(kBinBlock), 0, a.stream, a.p2, a.P2, a.p1, a.P1, a.K, \
                     ws, a.idxs, a.dists)
  if (same) PO_BIN(false, false);
  else PO_BIN(false, true);
  hipLaunchKernelGGL(grid_scan_partial_kernel, dim3((unsigned)chunks, (unsigned)a.N, which), dim3(kScanBlock), 0,
                     a.stream, ws, chunks);
  hipLaunchKernelGGL(grid_scan_offsets_kernel, dim3((unsigned)a.N, which), dim3(kScanBlock), 0, a.stream, ws, chunks);
  hipLaunchKernelGGL(grid_scan_apply_kernel, dim3((unsigned)chunks, (unsigned)a.N, which), dim3(kScanBlock), 0,
                     a.stream, ws, chunks, refine);
  if (same) PO_BIN(true, false);
  else PO_BIN(true, true);
#undef PO_BIN
}

int grid_build(const KnnArgs& a, const GridWs& ws, const GridBuild& b) {
  // histogram buffers (cell_count and qcell_count are adjacent) start at zero
  const size_t zero_bytes = b.same ? (size_t)((char*)ws.qcell_count - (char*)ws.cell_count)
                                   : (size_t)((char*)ws.cell_start - (char*)ws.cell_count);
  if (hipMemsetAsync(ws.cell_count, 0, zero_bytes, a.stream) != hipSuccess) return check_launch("grid memset");
  hipLaunchKernelGGL(grid_bbox_init_kernel, dim3((unsigned)ceil_div(a.N * 8, 256)), dim3(256), 0, a.stream, ws.bbox,
                     (int)a.N);
  hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)ceil_div(a.P2, kBboxBlock * kBboxPerThread), (unsigned)a.N),
                     dim3(kBboxBlock), 0, a.stream, a.p2, a.l2, a.P2, a.D, ws.bbox);
  hipLaunchKernelGGL(grid_setup_kernel, dim3((unsigned)a.N), dim3(kSetupBlock), 0, a.stream, a.p2, a.l1, a.l2, a.P1,
                     a.P2, a.D, b.c_target, b.h_min, b.ball_radius, b.ball_K, b.ball_factor, b.same ? 1 : 0, ws);
  hipLaunchKernelGGL(grid_prefix_kernel, dim3(1), dim3(64), 0, a.stream, ws, (int)a.N);
  switch (a.D) {
    case 1: build_d<1>(a, ws, b.same, b.refine); break;
    case 2: build_d<2>(a, ws, b.same, b.refine); break;
    default: build_d<3>(a, ws, b.same, b.refine); break;
  }
  return check_launch("grid build");
}

}  // namespace pointops
