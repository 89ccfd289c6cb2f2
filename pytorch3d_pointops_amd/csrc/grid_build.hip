// grid_build.hip -- build passes of the exact cell-grid searches (gfx950); see knn_grid.hip for the
// algorithm.  All clouds of the batch in every launch, no host synchronisation:
//   0 grid_bbox      bounding box of every p2 cloud (one min / max slot per workgroup, reduced by grid_setup);
//   1 grid_setup     per cloud: cubic cell size h for ~c_target points per cell, G = cells per dimension,
//                    and per-dimension EDGE TABLES E_d[c] = min{ x in [lo,hi] : cell_d(x) >= c }, found by
//                    bisection over the ordered fp32 bit patterns of the (monotone) cell function itself --
//                    no error analysis of the binning arithmetic is needed;
//   2 grid_partition<count>   entries per MICRO-BIN (cell id >> mshift, <= 1024 per cloud): p2 points and p1 queries,
//                    LDS histogram per tile, one device atomic per (tile, micro-bin); the last tile groups consecutive
//                    micro-bins into BINS of ~1000-2000 entries and lists the crowded ones; zero rows for padded queries;
//   3 grid_partition<scatter> (x,y,z,idx) float4 records of points and of queries, grouped by bin (records of crowded
//                    bins also get their rank inside their cell);
//   4 grid_sort      one workgroup per bin: cells counted, scanned and sorted in LDS -> cell_start, the
//                    point records by cell, the query records by cell (the searches read their queries
//                    coalesced, in cell order), the refined-cell marks; crowded bins by slices.
// When the queries ARE the points (same buffer, same lengths: self-KNN, ball query of a cloud on itself,
// get_point_covariances) the query passes are skipped: the point sort is the query order.
#include <stdlib.h>

#include "grid.h"

namespace pointops {

constexpr int kSetupBlock = 1024;  // 3 x 1026 edge bisections per cloud
constexpr int kCoarsePoints = 1024;  // entries a bin of the two-level sort aims at (512 / 2048: cfg2 sort pass 48 / 46 us
                                     // instead of 36 us)

__device__ __forceinline__ int64_t coarse_row(int n, int set) { return ((int64_t)n * 2 + set) * (kCoarseMax + 1); }

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
// pass 0: bounding boxes, all CUs: every workgroup reduces its 2048 points and leaves min / max in its own slot
// (no atomics, nothing to initialise); grid_setup reduces the cloud's slots.
constexpr int kBboxBlock = 256;
constexpr int kBboxPerThread = 8;  // 2048 points per workgroup: enough workgroups to cover the latency of a 25 MB read
constexpr int kBboxTile = kBboxBlock * kBboxPerThread;

__global__ __launch_bounds__(kBboxBlock) void grid_bbox_kernel(const float* __restrict__ p2,
                                                             const int64_t* __restrict__ lengths2, int P2, int D,
                                                             float* __restrict__ bbox_part) {
  const int n = blockIdx.y;
  int len2 = (int)lengths2[n];
  len2 = len2 < 0 ? 0 : (len2 > P2 ? P2 : len2);
  const int j0 = blockIdx.x * kBboxTile;
  if (j0 >= len2) return;  // (grid_setup reads the slots of the first ceil(len2 / tile) workgroups only)
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
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    float a = s_mn[0][d], b = s_mx[0][d];
#pragma unroll
    for (int w = 1; w < kBboxBlock / kWave; ++w) {
      a = fminf(a, s_mn[w][d]);
      b = fmaxf(b, s_mx[w][d]);
    }
    float* __restrict__ slot = bbox_part + ((int64_t)n * gridDim.x + blockIdx.x) * 6;
    slot[d] = a;      // (+inf / -inf in dimensions >= D: grid_setup ignores them)
    slot[3 + d] = b;
  }
}

__global__ __launch_bounds__(kSetupBlock) void grid_setup_kernel(
    const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int D, float c_target, float h_min,
    float ball_radius, int ball_K, float ball_factor, int same, int bbox_slots, GridWs ws) {
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  int len2 = (int)lengths2[n];
  len2 = len2 < 0 ? 0 : (len2 > P2 ? P2 : len2);
  int len1 = (int)lengths1[n];
  len1 = len1 < 0 ? 0 : (len1 > P1 ? P1 : len1);
  __shared__ GridCloud s_g;
  __shared__ float s_hi[3];
  __shared__ float s_box[kSetupBlock / kWave][6];
  {  // bounding box of the cloud: min / max over the slots of grid_bbox_kernel
    const int slots = (len2 + kBboxTile - 1) / kBboxTile;
    float v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = k < 3 ? __builtin_inff() : -__builtin_inff();
    for (int t = tid; t < slots; t += kSetupBlock) {
      const float* __restrict__ slot = ws.bbox_part + ((int64_t)n * bbox_slots + t) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = k < 3 ? fminf(v[k], slot[k]) : fmaxf(v[k], slot[k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) {
        const float u = __shfl_xor(v[k], off, kWave);
        v[k] = k < 3 ? fminf(v[k], u) : fmaxf(v[k], u);
      }
    }
    if ((tid & (kWave - 1)) == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) s_box[tid / kWave][k] = v[k];
    }
    __syncthreads();
  }
  if (tid == 0) {
    GridCloud g;
    float lo[3], hi[3], e[3];
    bool finite = len2 > 0;
    for (int d = 0; d < 3; ++d) {
      float a = 0.0f, b = 0.0f;
      if (d < D && len2 > 0) {
        a = s_box[0][d];
        b = s_box[0][3 + d];
        for (int w = 1; w < kSetupBlock / kWave; ++w) {
          a = fminf(a, s_box[w][d]);
          b = fmaxf(b, s_box[w][3 + d]);
        }
      }
      ws.bbox[n * 8 + d] = fkey(a);  // (ball query: the coarse query order of the scan-mode clouds reads these)
      ws.bbox[n * 8 + 3 + d] = fkey(b);
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
      // at most kGMax cells per dimension: a long 1-D cloud would otherwise clamp nearly all of its points into the
      // last cell (exact, but one cell = the whole cloud)
      for (int d = 0; d < 3; ++d)
        if (active[d]) h = fmaxf(h, e[d] * (1.0f / ((float)kGMax - 0.5f)));
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
        // (the two-level sort bins at most kCoarseMax micro-bins of 2^kFineLogMax cells: beyond that, bigger cells)
        if (cells <= (long long)ws.cell_cap && cells <= ((long long)kCoarseMax << kFineLogMax)) break;
        h *= 1.2599211f;  // halve the cell count
      }
      if ((long long)G[0] * G[1] * G[2] > (long long)ws.cell_cap ||
          (long long)G[0] * G[1] * G[2] > ((long long)kCoarseMax << kFineLogMax))
        ok = false;
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
    // micro-bins of the two-level sort: cell id >> mshift, at most kCoarseMax of them
    g.mshift = 0;
    while (((g.ncell - 1) >> g.mshift) + 1 > kCoarseMax) ++g.mshift;
    g.nmicro = ((g.ncell - 1) >> g.mshift) + 1;
    if (g.mshift > kFineLogMax) ok = false;
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
  for (int t = tid; t < 2 * (kCoarseMax + 1); t += kSetupBlock) {  // counters and cursors of the partition passes
    ws.coarse_count[coarse_row(n, 0) + t] = 0;
    ws.coarse_cursor[coarse_row(n, 0) + t] = 0;
  }
  if (tid < 2) {
    ws.coarse_ticket[n * 2 + tid] = 0;
    ws.crowded_count[n * 2 + tid] = 0;
    ws.nbins[n * 2 + tid] = 0;
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
// cell-sorted query order; chunk_prefix[n] = chunks of the clouds before cloud n.  One wave (of the count launch).
__device__ void grid_chunk_prefix(const GridWs& ws, int N) {
  const int lane = threadIdx.x & (kWave - 1);
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
// passes 2-4: two-level counting sort of the points of p2 (set 0) and the queries of p1 (set 1) by cell.
//
// The inputs arrive in storage order, so a tile of a few thousand points touches about as many distinct cells:
// a direct counting sort costs one scattered device-scope atomic per point (they execute at the memory side,
// ~2e10/s chip-wide: 105 us for the 2.4 M points and queries of the K=1 / chamfer case) and a 16-byte store at a
// random position of a megabyte-sized array (4x write amplification).  Instead:
//   PARTITION  count: tiles of 16384 entries count MICRO-BINS (cell id >> mshift, <= 1024 per cloud) in LDS and add
//              each non-empty one to the cloud's counters with ONE device atomic; the last tile to finish groups
//              consecutive micro-bins into BINS of ~1000-2000 entries and <= 4096 cells (so the bins follow the data: a
//              surface or a cluster occupies few micro-bins) and lists the crowded ones;
//              scatter: tiles of 4096 entries write their records grouped by bin (runs of tens of records);
//   SORT       one workgroup per bin: histogram of the bin's cells in LDS, exclusive scan -> the bin's slice
//              of cell_start (and the refined-cell bookkeeping), then the records move to their final place inside the
//              bin's own compact range of the sorted array (second read from L2).
// No per-cell global counters, no global scans, no memset.  PAD_ROWS: the count launch also writes the rows that get
// no search (zeros / -1 for padded queries) and lists the queries of clouds without a usable grid for the whole-cloud
// scan.
// ---------------------------------------------------------------------------
constexpr int kPartBlock = 1024;
constexpr int kCountPerThread = 16;   // count launch: tiles of 16384 entries (half the device atomics of 8192)
constexpr int kScatterPerThread = 4;  // scatter launch: tiles of 4096 entries, the records stay in registers (cfg2: 42.6 /
                                      // 29.6 / 30.9 us for tiles of 8192 / 4096 / 2048; count tiles of 8192 / 32768: 22 / 31.5 us)
constexpr int kSortBlock = 256;  // (128 / 512 / 1024 threads: 43 / 89 / 157 us against 36 / 62 / 62 us for the cfg2 sort pass)
constexpr int kHashBits = 12, kHashSlots = 1 << kHashBits, kHashProbes = 8;  // partition pass: cells of crowded bins
constexpr int kCrowded = 8192;       // records from which a bin is CROWDED: sorted by slices (grid_sort_kernel)
constexpr int kCrowdedSlice = 4096;  // records per slice
constexpr int kFineMax = 1 << kFineLogMax;  // cells per bin cap: the LDS counters of one sort workgroup
static_assert(kCoarseMax == kPartBlock, "the count launch's last tile scans one micro-bin per thread");

// atomicAdd(&counter[slot], 1) of every live lane of a wave; when all of them name the SAME counter (a bin whose
// records sit in one cell: 64 LDS atomics on one address serialise) they spend one atomic.  Returns what the lane's
// own atomicAdd would have (a rank unique per counter).  Call it from wave-uniform control flow only.
__device__ __forceinline__ int wave_add(int* __restrict__ counter, int slot, bool live) {
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned long long todo = __ballot(live);
  if (todo == 0ull) return 0;
  const int leader = __ffsll((long long)todo) - 1;
  const int s0 = __shfl(slot, leader, kWave);
  if ((__ballot(live && slot == s0)) == todo) {  // (wave-uniform)
    int base = 0;
    if (lane == leader) base = atomicAdd(&counter[s0], __popcll(todo));
    base = __shfl(base, leader, kWave);
    return base + __popcll(todo & ((1ull << lane) - 1ull));
  }
  return live ? atomicAdd(&counter[slot], 1) : 0;
}

// One launch handles BOTH sets: blockIdx.z = 0 the points of p2, 1 the queries of p1 (QUERIES = false: points only,
// the self-query case).
template <int D, bool SCATTER, bool QUERIES>
__global__ __launch_bounds__(kPartBlock) void grid_partition_kernel(const float* __restrict__ p2, int P2,
                                                                 const float* __restrict__ p1, int P1, int K,
                                                                 GridWs ws, int zbase, int64_t* __restrict__ idxs,
                                                                 float* __restrict__ dists) {
  constexpr int kPartPerThread = SCATTER ? kScatterPerThread : kCountPerThread;
  constexpr int kPartTile = kPartBlock * kPartPerThread;
  __shared__ int s_hist[kCoarseMax];   // entries of this tile per micro-bin / per bin; then the base of the tile's group
  __shared__ int s_start[kCoarseMax];  // SCATTER: starts of the bins (count launch, last tile: of the micro-bins)
  __shared__ int s_group[SCATTER ? kCoarseMax : 1];  // SCATTER: micro-bin -> bin
  __shared__ int s_wsum[kPartBlock / kWave];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const bool IS_QUERY = QUERIES && blockIdx.z + zbase == 1;   // (workgroup-uniform; zbase = 1: a queries-only launch)
  const bool PAD_ROWS = QUERIES ? IS_QUERY : true;    // the pass over the query index space also pads rows
  const float* __restrict__ pts = IS_QUERY ? p1 : p2;
  const int P = IS_QUERY ? P1 : P2;
  const GridCloud g = ws.cloud[n];  // wave-uniform
  const int set = IS_QUERY ? 1 : 0;
  const int len = IS_QUERY ? g.len1 : g.len2;
  const int mshift = g.mshift, nmicro = g.nmicro;
  int* __restrict__ gcount = ws.coarse_count + coarse_row(n, set);    // per micro-bin
  int* __restrict__ gcursor = ws.coarse_cursor + coarse_row(n, set);  // per bin
  int* __restrict__ gstart = ws.coarse_start + coarse_row(n, set);    // per bin
  int* __restrict__ gbinof = ws.bin_of + coarse_row(n, set);          // micro-bin -> bin
  int* __restrict__ gfirst = ws.bin_first + coarse_row(n, set);       // bin -> its first micro-bin
  if (!SCATTER && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid < kWave)
    grid_chunk_prefix(ws, (int)gridDim.y);  // (one wave of the whole launch)
  const int i0 = blockIdx.x * kPartTile + tid;
  if (blockIdx.x * kPartTile >= P) return;

  if (PAD_ROWS && !SCATTER) {
    // rows that get no search: zeros for padded queries (knn_cpu.cpp:25-26); whole-cloud list
    // when this cloud has no usable grid
#pragma unroll 4
    for (int r = 0; r < kPartPerThread; ++r) {
      const int i = i0 + r * kPartBlock;
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
  if (!g.use_grid || blockIdx.x * kPartTile >= len) return;

  const int nb = SCATTER ? ws.nbins[n * 2 + set] : nmicro;  // counters of this launch: bins / micro-bins
  for (int b = tid; b < nb; b += kPartBlock) s_hist[b] = 0;
  if (SCATTER) {
    for (int m = tid; m < nmicro; m += kPartBlock) s_group[m] = gbinof[m];
    for (int b = tid; b < nb; b += kPartBlock) s_start[b] = gstart[b];
  }
  __syncthreads();
  int bin[kPartPerThread], key[kPartPerThread], rank[kPartPerThread];
  float px[kPartPerThread], py[kPartPerThread], pz[kPartPerThread];
#pragma unroll
  for (int r = 0; r < kPartPerThread; ++r) {
    const int i = i0 + r * kPartBlock;
    bin[r] = -1;
    key[r] = 0;
    rank[r] = 0;
    if (i < len) {
      float x, y, z;
      load_point3<D>(pts + ((int64_t)n * P + i) * D, x, y, z);
      int cx, cy, cz;
      point_cells(g, x, y, z, cx, cy, cz);
      bin[r] = (cz * g.G[1] + cy) * g.G[0] + cx;
      key[r] = bin[r] >> mshift;
      if (SCATTER) {
        key[r] = s_group[key[r]];
        px[r] = x;
        py[r] = y;
        pz[r] = z;
      }
      rank[r] = atomicAdd(&s_hist[key[r]], 1);  // LDS atomic: rank inside (tile, counter)
    }
  }
  __syncthreads();
  for (int b = tid; b < nb; b += kPartBlock) {
    const int c = s_hist[b];
    if (c > 0) {
      if (!SCATTER) {
        const int before = atomicAdd(gcount + b, c);
        asm volatile("" ::"v"(before));  // a RETURNING atomic: performed before this wave passes the barrier below
      } else {
        s_hist[b] = s_start[b] + atomicAdd(gcursor + b, c);  // base of this tile's group
      }
    }
  }
  int* __restrict__ fine = ws.fine_count + (int64_t)(n * 2 + set) * ws.cell_cap;
  int* __restrict__ clist = ws.crowded_list + (int64_t)(n * 2 + set) * kCrowdedMax;
  if (!SCATTER) {
    // The LAST tile of the (cloud, set) to finish lists the CROWDED bins (see grid_sort_kernel) and zeroes their
    // per-cell counters for the scatter launch.
    // (No agent-scope fence: on this chip it writes back the whole L2.  The counters are only ever touched by
    // device-scope atomics, which execute at the memory side; a tile's atomics have returned before its ticket is
    // drawn, and the last tile reads the counters with device-scope atomic loads.)
    __shared__ int s_last, s_ncrowd;
    __syncthreads();
    if (tid == 0) {
      const int tiles = (len + kPartTile - 1) / kPartTile;
      s_last = atomicAdd(ws.coarse_ticket + n * 2 + set, 1) == tiles - 1;
      s_ncrowd = 0;
    }
    __syncthreads();
    if (s_last) {  // (workgroup-uniform)
      // BINS: consecutive micro-bins whose first record falls into the same block of kCoarsePoints records of the
      // cloud's cell order and that lie in the same block of kFineMax cells.
      const int lane = tid & (kWave - 1), wave = tid / kWave;
      auto scan1 = [&](int v, int& total) {  // exclusive prefix of v over the workgroup (one value per thread)
        int inc = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
          const int u = __shfl_up(inc, off, kWave);
          if (lane >= off) inc += u;
        }
        __syncthreads();  // (s_wsum free again)
        if (lane == kWave - 1) s_wsum[wave] = inc;
        __syncthreads();
        int excl = inc - v;
        total = 0;
        for (int w = 0; w < kPartBlock / kWave; ++w) {
          if (w < wave) excl += s_wsum[w];
          total += s_wsum[w];
        }
        return excl;
      };
      const int m = tid;
      const int v = m < nmicro ? __hip_atomic_load(gcount + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      int total;
      const int e = scan1(v, total);  // records before the micro-bin
      s_start[m] = e;
      __syncthreads();
      const int span = kFineMax >> mshift;  // micro-bins a bin may cover
      int o = 0;                            // micro-bin m opens a new bin
      if (m < nmicro) {
        o = 1;
        if (m > 0) {
          const int ep = s_start[m - 1];
          o = ((e / kCoarsePoints) != (ep / kCoarsePoints) || (m / span) != ((m - 1) / span)) ? 1 : 0;
        }
      }
      int nbins;
      const int b = scan1(o, nbins) + o - 1;  // bin of the micro-bin
      if (m < nmicro) gbinof[m] = b;
      if (o) {
        gfirst[b] = m;
        gstart[b] = e;
        s_hist[b] = e;  // (the tile's histogram is no longer needed: bin starts, for the crowded list)
      }
      if (tid == 0) {
        gfirst[nbins] = nmicro;
        gstart[nbins] = total;
        ws.nbins[n * 2 + set] = nbins;
      }
      __syncthreads();
      for (int b = tid; b < nbins; b += kPartBlock) {
        const int c = (b + 1 < nbins ? s_hist[b + 1] : total) - s_hist[b];
        if (c > kCrowded) {
          const int k = atomicAdd(&s_ncrowd, 1);
          if (k < kCrowdedMax) clist[k] = b;
        }
      }
      __syncthreads();
      const int nc = min(s_ncrowd, kCrowdedMax);
      if (tid == 0) ws.crowded_count[n * 2 + set] = nc;
      for (int k = 0; k < nc; ++k) {  // (clist and gfirst: this workgroup's own writes, behind the barrier)
        const int b = clist[k];
        const int f0 = gfirst[b] << mshift;
        const int nf = min((gfirst[b + 1] - gfirst[b]) << mshift, g.ncell - f0);
        for (int f = tid; f < nf; f += kPartBlock) fine[f0 + f] = 0;
      }
    }
  }
  if (SCATTER) {
    __shared__ unsigned s_crowd[kCoarseMax / 32];  // bitmap of the crowded bins
    const int nc = ws.crowded_count[n * 2 + set];  // (workgroup-uniform; 0 unless the cloud is skewed)
    if (nc > 0) {
      for (int w = tid; w < kCoarseMax / 32; w += kPartBlock) s_crowd[w] = 0u;
      __syncthreads();
      if (tid < nc) {
        const int b = clist[tid];
        atomicOr(&s_crowd[b >> 5], 1u << (b & 31));
      }
    }
    __syncthreads();
    int* __restrict__ grank = (IS_QUERY ? ws.qrank : ws.prank) + (int64_t)n * P;
    int pos[kPartPerThread];
#pragma unroll
    for (int r = 0; r < kPartPerThread; ++r) {
      pos[r] = -1;
      if (bin[r] >= 0) {
        const int i = i0 + r * kPartBlock;
        pos[r] = s_hist[key[r]] + rank[r];
        (IS_QUERY ? ws.qtmp : ws.sorted_tmp)[(int64_t)n * P + pos[r]] =
            make_float4(px[r], py[r], pz[r], __int_as_float(i));
      }
    }
    if (nc > 0) {
      // Records of crowded bins: rank inside the CELL, counted in fine_count.  The tile's records of one cell share
      // ONE device atomic through an LDS hash table (cell -> count, open addressing, at most kHashProbes probes; a
      // record that finds no slot spends its own atomic): a cell that holds a few per cent of a cloud would
      // otherwise serialise thousands of atomics on one address.
      __shared__ int s_hkey[kHashSlots], s_hcnt[kHashSlots];
      for (int h = tid; h < kHashSlots; h += kPartBlock) {
        s_hkey[h] = -1;
        s_hcnt[h] = 0;
      }
      __syncthreads();
      int slot[kPartPerThread];
#pragma unroll
      for (int r = 0; r < kPartPerThread; ++r) {
        slot[r] = -2;  // not a record of a crowded bin
        if (bin[r] < 0) continue;
        const int cb = key[r];
        if (((s_crowd[cb >> 5] >> (cb & 31)) & 1u) == 0u) continue;
        slot[r] = -1;  // no slot: own atomic
        int h = (int)(((unsigned)bin[r] * 2654435761u) >> (32 - kHashBits));
        for (int probe = 0; probe < kHashProbes; ++probe) {
          const int old = atomicCAS(&s_hkey[h], -1, bin[r]);
          if (old == -1 || old == bin[r]) {
            slot[r] = h;
            rank[r] = atomicAdd(&s_hcnt[h], 1);
            break;
          }
          h = (h + 1) & (kHashSlots - 1);
        }
      }
      __syncthreads();
      constexpr int kSlotsPerThread = kHashSlots / kPartBlock;
      int base[kSlotsPerThread];  // (two loops: a thread's device atomics are in flight together)
#pragma unroll
      for (int t = 0; t < kSlotsPerThread; ++t) {
        const int h = tid + t * kPartBlock;
        const int key = s_hkey[h];
        base[t] = key >= 0 ? atomicAdd(fine + key, s_hcnt[h]) : 0;
      }
#pragma unroll
      for (int t = 0; t < kSlotsPerThread; ++t) s_hcnt[tid + t * kPartBlock] = base[t];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kPartPerThread; ++r) {
        if (slot[r] >= 0) grank[pos[r]] = s_hcnt[slot[r]] + rank[r];
        else if (slot[r] == -1) grank[pos[r]] = atomicAdd(fine + bin[r], 1);
      }
    }
  }
}

// counter f of the sort pass lives at f + f / 32: a thread scanning a contiguous slice of counters does not keep
// hitting one LDS bank
__device__ __forceinline__ int fine_slot(int f) { return f + (f >> 5); }

// SORT pass: workgroup (b mod gridDim.x) of (cloud, set) sorts bin b by cell.  `refine` as in GridBuild.
// A CROWDED bin (more than kCrowded records: a cluster, the dense end of a density gradient) would keep one workgroup
// busy for hundreds of microseconds; the partition pass has already counted its records per cell and handed every
// record its rank inside its cell (fine_count / rank arrays), so its slices of kCrowdedSlice records are placed by as
// many workgroups as there are slices, without atomics: position = cell start (scan of the bin's cell counts, redone
// per slice from L2) + rank.
__global__ __launch_bounds__(kSortBlock) void grid_sort_kernel(GridWs ws, int P1, int P2, int refine, int zbase) {
  __shared__ int s_cnt[kFineMax + kFineMax / 32];
  __shared__ int s_wsum[kSortBlock / kWave];
  __shared__ int s_list[kCrowdedMax];
  const int n = blockIdx.y, set = blockIdx.z + zbase;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const GridCloud g = ws.cloud[n];
  const int len = set ? g.len1 : g.len2;
  if (!g.use_grid || len <= 0) return;
  const int mshift = g.mshift, nbin = ws.nbins[n * 2 + set];
  const int* __restrict__ bfirst = ws.bin_first + coarse_row(n, set);
  const int* __restrict__ cstart = ws.coarse_start + coarse_row(n, set);
  int* __restrict__ cell_start = ws.cell_start + (int64_t)n * (ws.cell_cap + 1);
  const float4* __restrict__ tmp = set ? ws.qtmp + (int64_t)n * P1 : ws.sorted_tmp + (int64_t)n * P2;
  float4* __restrict__ out = set ? ws.qsorted + (int64_t)n * P1 : ws.sorted + (int64_t)n * (P2 + kSortedPad);
  const int ncrowd = ws.crowded_count[n * 2 + set];  // (<= kCrowdedMax)
  if (ncrowd > 0) {
    if (tid < ncrowd) s_list[tid] = ws.crowded_list[(int64_t)(n * 2 + set) * kCrowdedMax + tid];
    __syncthreads();
  }
  constexpr int kKeep = 2048 / kSortBlock;  // records per thread a bin may have to stay in registers

  // counts in s_cnt[0, nf) -> exclusive cursors (relative to the bin); `publish`: cell_start and the refined-cell
  // marks of the bin's cells
  auto scan_cells = [&](int f0, int nf, int start, int cnt, bool publish) {
    const int per = (nf + kSortBlock - 1) / kSortBlock;
    const int a0 = min(tid * per, nf), a1 = min(a0 + per, nf);
    int sum = 0;
    for (int f = a0; f < a1; ++f) sum += s_cnt[fine_slot(f)];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int u = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += u;
    }
    if (lane == kWave - 1) s_wsum[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wsum[w];
    for (int f = a0; f < a1; ++f) {
      const int c = s_cnt[fine_slot(f)];
      s_cnt[fine_slot(f)] = run;
      run += c;
    }
    __syncthreads();
    if (publish && set == 0) {  // (coalesced: cell f of the bin by thread f)
      for (int f = tid; f < nf; f += kSortBlock) {
        const int e = s_cnt[fine_slot(f)];
        const int c = (f + 1 < nf ? s_cnt[fine_slot(f + 1)] : cnt) - e;
        cell_start[f0 + f] = start + e;
        if (refine >= 0) {
          int ref = -1;
          if (refine > 0 && c > refine_threshold(ws.c_target)) {
            int s = (int)ceilf(cbrtf((float)c / ws.c_target));
            s = s < 2 ? 2 : (s > kRefineMaxS ? kRefineMaxS : s);
            const int cells = s * s * s + 1;
            const int idx = atomicAdd(ws.rcount + n, 1);
            if (idx < ws.rdesc_cap) {
              // count == 0: a descriptor without a table (pool exhausted), skipped by refine_build.  (Field by field:
              // a RefinedCell temporary gave the kernel a private segment.)
              const int off = atomicAdd(ws.pool_top + n, cells);
              const bool fits = off + cells <= ws.pool_cap;
              RefinedCell* __restrict__ dp = ws.rdesc + (int64_t)n * ws.rdesc_cap + idx;
              dp->start = fits ? start + e : 0;
              dp->count = fits ? c : 0;
              dp->s = fits ? s : 0;
              dp->pool_off = fits ? off : 0;
              if (fits) ref = idx;
            }
          }
          ws.refine_ref[(int64_t)n * ws.cell_cap + f0 + f] = ref;
        }
      }
      if (f0 + nf == g.ncell && tid == 0) cell_start[g.ncell] = start + cnt;  // (the cloud's last bin)
    }
  };

  for (int b = blockIdx.x; b < nbin; b += gridDim.x) {
    const int start = cstart[b], cnt = cstart[b + 1] - start;
    if (cnt > kCrowded) {  // listed as crowded: done by slices below
      bool listed = false;
      for (int k = 0; k < ncrowd; ++k) listed = listed || s_list[k] == b;
      if (listed) continue;
    }
    const int f0 = bfirst[b] << mshift;
    const int nf = min((bfirst[b + 1] - bfirst[b]) << mshift, g.ncell - f0);
    for (int f = tid; f < nf; f += kSortBlock) s_cnt[fine_slot(f)] = 0;
    __syncthreads();
    auto fine_of = [&](const float4 p) {  // the same expression as the partition pass
      int cx, cy, cz;
      point_cells(g, p.x, p.y, p.z, cx, cy, cz);
      return (cz * g.G[1] + cy) * g.G[0] + cx - f0;
    };
    // 1. histogram of the bin's cells.  A bin of up to 2048 records (the usual case) stays in registers with
    //    the rank the LDS atomic handed out; a longer one is read twice.
    const bool keep = cnt <= kKeep * kSortBlock;  // (workgroup-uniform)
    float4 kp[kKeep];
    int kf[kKeep], krank[kKeep];
    if (keep) {
#pragma unroll
      for (int u = 0; u < kKeep; ++u) {
        const int j = tid + u * kSortBlock;
        kf[u] = -1;
        if (j < cnt) {
          kp[u] = tmp[start + j];
          kf[u] = fine_of(kp[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < kKeep; ++u)
        if (kf[u] >= 0) krank[u] = atomicAdd(&s_cnt[fine_slot(kf[u])], 1);
    } else {
      // (workgroup-uniform trip count: wave_add shuffles)
      for (int i0 = 0; i0 < cnt; i0 += 4 * kSortBlock) {
        const int j = i0 + tid;
        const bool l0 = j < cnt, l1 = j + kSortBlock < cnt, l2 = j + 2 * kSortBlock < cnt, l3 = j + 3 * kSortBlock < cnt;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 p0 = l0 ? tmp[start + j] : z;
        const float4 p1 = l1 ? tmp[start + j + kSortBlock] : z;
        const float4 p2 = l2 ? tmp[start + j + 2 * kSortBlock] : z;
        const float4 p3 = l3 ? tmp[start + j + 3 * kSortBlock] : z;
        wave_add(s_cnt, fine_slot(l0 ? fine_of(p0) : 0), l0);
        wave_add(s_cnt, fine_slot(l1 ? fine_of(p1) : 0), l1);
        wave_add(s_cnt, fine_slot(l2 ? fine_of(p2) : 0), l2);
        wave_add(s_cnt, fine_slot(l3 ? fine_of(p3) : 0), l3);
      }
    }
    __syncthreads();
    // 2. exclusive scan -> cell_start, refined-cell marks, cursors
    scan_cells(f0, nf, start, cnt, true);
    __syncthreads();
    // 3. records to their final places
    if (keep) {
#pragma unroll
      for (int u = 0; u < kKeep; ++u)
        if (kf[u] >= 0) out[start + s_cnt[fine_slot(kf[u])] + krank[u]] = kp[u];
    } else {
      for (int i0 = 0; i0 < cnt; i0 += 4 * kSortBlock) {  // (second read of the bin: L2)
        const int j = i0 + tid;
        const bool l0 = j < cnt, l1 = j + kSortBlock < cnt, l2 = j + 2 * kSortBlock < cnt, l3 = j + 3 * kSortBlock < cnt;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 p0 = l0 ? tmp[start + j] : z;
        const float4 p1 = l1 ? tmp[start + j + kSortBlock] : z;
        const float4 p2 = l2 ? tmp[start + j + 2 * kSortBlock] : z;
        const float4 p3 = l3 ? tmp[start + j + 3 * kSortBlock] : z;
        const int q0 = wave_add(s_cnt, fine_slot(l0 ? fine_of(p0) : 0), l0);
        const int q1 = wave_add(s_cnt, fine_slot(l1 ? fine_of(p1) : 0), l1);
        const int q2 = wave_add(s_cnt, fine_slot(l2 ? fine_of(p2) : 0), l2);
        const int q3 = wave_add(s_cnt, fine_slot(l3 ? fine_of(p3) : 0), l3);
        if (l0) out[start + q0] = p0;
        if (l1) out[start + q1] = p1;
        if (l2) out[start + q2] = p2;
        if (l3) out[start + q3] = p3;
      }
    }
    __syncthreads();
  }

  // crowded bins, by slices
  const int* __restrict__ fine = ws.fine_count + (int64_t)(n * 2 + set) * ws.cell_cap;
  const int* __restrict__ rank = set ? ws.qrank + (int64_t)n * P1 : ws.prank + (int64_t)n * P2;
  for (int k = 0; k < ncrowd; ++k) {
    const int b = s_list[k];
    const int start = cstart[b], cnt = cstart[b + 1] - start;
    const int f0 = bfirst[b] << mshift;
    const int nf = min((bfirst[b + 1] - bfirst[b]) << mshift, g.ncell - f0);
    const int slices = (cnt + kCrowdedSlice - 1) / kCrowdedSlice;
    for (int sl = blockIdx.x; sl < slices; sl += gridDim.x) {
      for (int f = tid; f < nf; f += kSortBlock) s_cnt[fine_slot(f)] = fine[f0 + f];
      __syncthreads();
      scan_cells(f0, nf, start, cnt, sl == 0);
      const int j1 = min(cnt, (sl + 1) * kCrowdedSlice);
      for (int j = sl * kCrowdedSlice + tid; j < j1; j += kSortBlock) {
        const float4 p = tmp[start + j];
        int cx, cy, cz;
        point_cells(g, p.x, p.y, p.z, cx, cy, cz);
        const int f = (cz * g.G[1] + cy) * g.G[0] + cx - f0;
        out[start + s_cnt[fine_slot(f)] + rank[start + j]] = p;
      }
      __syncthreads();
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

size_t grid_carve(GridWs* ws, char* base, int64_t N, int64_t P1, int64_t P2, float c, bool ball) {
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
  w.cell_start = (int*)take(sizeof(int) * (size_t)N * (cap + 1));
  w.coarse_count = (int*)take(sizeof(int) * (size_t)N * 2 * (kCoarseMax + 1));
  w.coarse_cursor = (int*)take(sizeof(int) * (size_t)N * 2 * (kCoarseMax + 1));
  w.coarse_start = (int*)take(sizeof(int) * (size_t)N * 2 * (kCoarseMax + 1));
  w.bin_of = (int*)take(sizeof(int) * (size_t)N * 2 * (kCoarseMax + 1));
  w.bin_first = (int*)take(sizeof(int) * (size_t)N * 2 * (kCoarseMax + 1));
  w.nbins = (int*)take(sizeof(int) * (size_t)N * 2);
  w.coarse_ticket = (int*)take(sizeof(int) * (size_t)N * 2);
  w.crowded_count = (int*)take(sizeof(int) * (size_t)N * 2);
  w.crowded_list = (int*)take(sizeof(int) * (size_t)N * 2 * kCrowdedMax);
  w.fine_count = (int*)take(sizeof(int) * (size_t)N * 2 * cap);
  w.prank = (int*)take(sizeof(int) * (size_t)N * (size_t)P2);
  w.qrank = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.order_table = (int*)take(ball ? sizeof(int) * (size_t)N * (size_t)((P1 + 2047) / 2048) * kOrderBins : 0);
  w.sorted = (float4*)take(sizeof(float4) * (size_t)N * (size_t)(P2 + kSortedPad));
  w.qsorted = (float4*)take(sizeof(float4) * (size_t)N * (size_t)P1);
  w.fb_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.fb_kth = (unsigned*)take(sizeof(unsigned) * (size_t)N * (size_t)P1);
  w.fb2_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb2_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.fb3_count = (int*)take(sizeof(int) * (size_t)N);
  w.fb3_list = (int*)take(sizeof(int) * (size_t)N * (size_t)P1);
  w.bbox = (unsigned*)take(sizeof(unsigned) * (size_t)N * 8);
  w.bbox_part = (float*)take(sizeof(float) * (size_t)N * 6 * (size_t)((P2 + 2047) / 2048));
  w.grid_flag = (int*)take(sizeof(int) * (size_t)N);
  w.qtmp = (float4*)take(sizeof(float4) * (size_t)N * (size_t)P1);
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

// sets: 0 = points only (self-query), 1 = queries only (the point side is already built: grid_build_queries), 2 = both
template <int D>
static void build_d(const KnnArgs& a, const GridWs& ws, int sets_mode, int refine) {
  const bool same = sets_mode == 0;
  const unsigned sets = sets_mode == 2 ? 2u : 1u;
  const int zbase = sets_mode == 1 ? 1 : 0;
  const int64_t pmax = same ? a.P2 : sets_mode == 1 ? a.P1 : (a.P2 > a.P1 ? a.P2 : a.P1);
#define PO_PART(SCT, QRY)                                                                                         \
  hipLaunchKernelGGL((grid_partition_kernel<D, SCT, QRY>),                                                        \
                     dim3((unsigned)ceil_div(pmax, kPartBlock * (SCT ? kScatterPerThread : kCountPerThread)),     \
                          (unsigned)a.N, sets),                                                                   \
                     dim3(kPartBlock), 0, a.stream, a.p2, a.P2, a.p1, a.P1, a.K, ws, zbase, a.idxs, a.dists)
  if (same) {
    PO_PART(false, false);
    PO_PART(true, false);
  } else {
    PO_PART(false, true);
    PO_PART(true, true);
  }
#undef PO_PART
  // sort workgroups per (cloud, set): enough of them to fill the chip when the batch is small
  // (a cloud of P entries has at most P / kCoarsePoints + cells / kFineMax + 2 bins)
  const int64_t bins = pmax / kCoarsePoints + ws.cell_cap / kFineMax + 2;
  int64_t wgs = 8192 / (a.N * (int64_t)sets);
  wgs = wgs < 16 ? 16 : (wgs > kCoarseMax ? kCoarseMax : wgs);
  wgs = wgs > bins ? bins : wgs;
  hipLaunchKernelGGL(grid_sort_kernel, dim3((unsigned)wgs, (unsigned)a.N, sets), dim3(kSortBlock), 0, a.stream, ws,
                     (int)a.P1, (int)a.P2, refine, zbase);
}

// ---------------------------------------------------------------------------
// REUSE of a built grid (pointops_knn_points_idx_reuse): the point side of the workspace -- cloud geometry, edge
// tables, cell_start, sorted records, refined cells -- is still valid; a new query set only needs its own sort, an
// unchanged one nothing but fresh counters and the padded rows of the new output tensors.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kSetupBlock) void grid_requery_kernel(const int64_t* __restrict__ lengths1, int P1, int same,
                                                                GridWs ws) {
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    int len1 = (int)lengths1[n];
    len1 = len1 < 0 ? 0 : (len1 > P1 ? P1 : len1);
    ws.cloud[n].len1 = len1;
    ws.cloud[n].same = same;
    ws.fb_count[n] = 0;
    ws.fb2_count[n] = 0;
    ws.fb3_count[n] = 0;
    ws.box_count[n] = 0;
    ws.coarse_ticket[n * 2 + 1] = 0;
    ws.crowded_count[n * 2 + 1] = 0;
    ws.nbins[n * 2 + 1] = 0;
  }
  for (int t = tid; t < kCoarseMax + 1; t += kSetupBlock) {  // counters and cursors of the query set's partition passes
    ws.coarse_count[coarse_row(n, 1) + t] = 0;
    ws.coarse_cursor[coarse_row(n, 1) + t] = 0;
  }
}

// what the count launch does besides counting, for calls that skip it: the chunk prefix, the rows that get no search
// (padded queries) and the whole-cloud list of clouds without a usable grid
__global__ __launch_bounds__(kPartBlock) void grid_pad_prefix_kernel(GridWs ws, int P1, int K, int64_t* __restrict__ idxs,
                                                                  float* __restrict__ dists) {
  const int n = blockIdx.y, tid = threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid < kWave) grid_chunk_prefix(ws, (int)gridDim.y);
  const GridCloud g = ws.cloud[n];
  if (g.len1 >= P1 && g.use_grid) return;  // (the usual case: nothing to pad, nothing to list)
  constexpr int kTile = kPartBlock * kCountPerThread;
  for (int r = 0; r < kCountPerThread; ++r) {
    const int i = blockIdx.x * kTile + r * kPartBlock + tid;
    if (i < P1 && i >= g.len1) {
      int64_t* __restrict__ zi = idxs + ((int64_t)n * P1 + i) * K;
      float* __restrict__ zd = dists + ((int64_t)n * P1 + i) * K;
      const int64_t pad = ws.ball ? -1 : 0;
      for (int k = 0; k < K; ++k) {
        zi[k] = pad;
        zd[k] = 0.0f;
      }
    } else if (i < g.len1 && !g.use_grid && !ws.ball) {
      const int pos = atomicAdd(ws.fb2_count + n, 1);
      ws.fb2_list[(int64_t)n * P1 + pos] = i;
    }
  }
}

// level 1: a new query set against the built point side; level 2: the same query set again
int grid_build_queries(const KnnArgs& a, const GridWs& ws, bool same, int level) {
  hipLaunchKernelGGL(grid_requery_kernel, dim3((unsigned)a.N), dim3(kSetupBlock), 0, a.stream, a.l1, a.P1, same ? 1 : 0,
                     ws);
  if (same || level >= 2) {
    hipLaunchKernelGGL(grid_pad_prefix_kernel, dim3((unsigned)ceil_div(a.P1, kPartBlock * kCountPerThread), (unsigned)a.N),
                       dim3(kPartBlock), 0, a.stream, ws, a.P1, a.K, a.idxs, a.dists);
  } else {
    switch (a.D) {
      case 1: build_d<1>(a, ws, 1, -1); break;
      case 2: build_d<2>(a, ws, 1, -1); break;
      default: build_d<3>(a, ws, 1, -1); break;
    }
  }
  return check_launch("grid build (queries)");
}

int grid_build(const KnnArgs& a, const GridWs& ws, const GridBuild& b) {
  const int bbox_slots = (int)ceil_div(a.P2, kBboxTile);
  hipLaunchKernelGGL(grid_bbox_kernel, dim3((unsigned)bbox_slots, (unsigned)a.N), dim3(kBboxBlock), 0, a.stream, a.p2,
                     a.l2, a.P2, a.D, ws.bbox_part);
  hipLaunchKernelGGL(grid_setup_kernel, dim3((unsigned)a.N), dim3(kSetupBlock), 0, a.stream, a.p2, a.l1, a.l2, a.P1,
                     a.P2, a.D, b.c_target, b.h_min, b.ball_radius, b.ball_K, b.ball_factor, b.same ? 1 : 0, bbox_slots,
                     ws);
  switch (a.D) {
    case 1: build_d<1>(a, ws, b.same ? 0 : 2, b.refine); break;
    case 2: build_d<2>(a, ws, b.same ? 0 : 2, b.refine); break;
    default: build_d<3>(a, ws, b.same ? 0 : 2, b.refine); break;
  }
  return check_launch("grid build");
}

}  // namespace pointops
