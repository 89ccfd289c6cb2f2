// ball_grid.hip -- BALL QUERY through the cell grid of the exact KNN search (ball_query.hip owns the
// operator; reference semantics ball_query_cpu.cpp:12-54: the first K points in INDEX order with
// dist2 < radius2).  Cells are at least 1.001 radius wide, so the 3x3x3 cube around the query's cell
// contains its ball; that is not assumed but CERTIFIED per query with the face bound of the KNN search
// (lb >= radius2: no unvisited point can pass `dist2 < radius2`), anything else goes to the index-order
// scan.  One lane per query walks its runs exactly like knn_grid_lane_kernel (knn_grid_search.h: groups
// of consecutive records behind one 32-bit offset, tail of a run's last group masked by one compare,
// per-lane list of non-empty runs); a hit pushes its INDEX into the lane's LDS queue, queues are merged
// into a sorted register list of the KC smallest indices by 32-bit sorting networks (v_min_u32 /
// v_max_u32 per compare-exchange), and a full list prunes by its largest index.  The list is the output
// order; distances are recomputed from the chosen points with the scan kernel's expression.
#include "debug.h"
#include "grid.h"
#include "knn_grid_search.h"

namespace pointops {

template <int D, int KC>
__global__ __launch_bounds__(kGridWave) void ball_grid_lane_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const GridCloud* __restrict__ clouds,
    const int* __restrict__ chunk_prefix, const float* __restrict__ edges, const int* __restrict__ cell_start,
    const float4* __restrict__ sorted, const float4* __restrict__ qsorted, int* __restrict__ fb_count,
    int* __restrict__ fb_list, int cell_cap, int P1, int P2, int K, int N, float radius2,
    int64_t* __restrict__ idxs, float* __restrict__ dists) {
  constexpr int kQueueCap = KC < 16 ? KC : 16;
  constexpr int kSub = 4;
  constexpr int G = 8;
  constexpr int kGroupBytes = G * 16;
  constexpr unsigned kNone = 0xffffffffu;
  __shared__ unsigned s_queue[kQueueCap * kGridWave];
  __shared__ int2 s_rows[kLaneRows + 1][kGridWave];

  const int lane = threadIdx.x;
  const int total = chunk_prefix[N];
  int2* const rows = &s_rows[0][0];
#pragma unroll
  for (int t = 0; t < kQueueCap; ++t) s_queue[t * kGridWave + lane] = kNone;
  // XCD-aware item order (see knn_grid_lane_kernel)
  const int xcd = blockIdx.x % kNumXcd, per_xcd = (total + kNumXcd - 1) / kNumXcd;
  for (int j = blockIdx.x / kNumXcd; j < per_xcd; j += gridDim.x / kNumXcd) {
    const int item = xcd * per_xcd + j;
    if (item >= total) break;
    const int n = item_cloud(chunk_prefix, N, item, (P1 + kGridWave - 1) / kGridWave);
    const GridCloud g = clouds[n];
    const int c0 = (item - chunk_prefix[n]) * kGridWave;
    const bool active = c0 + lane < g.len1;
    const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + kSortedPad);
    int qi = 0;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (active) {  // query records in cell order (the point records themselves when the queries are the points)
      const float4 q = (g.same ? sp : qsorted + (int64_t)n * P1)[c0 + lane];
      qx = q.x;
      qy = q.y;
      qz = q.z;
      qi = __float_as_int(q.w);
    }
    int cx, cy, cz;
    point_cells(g, qx, qy, qz, cx, cy, cz);
    const int X0 = max(cx - 1, 0), X1 = min(cx + 1, g.G[0] - 1);
    const int Y0 = max(cy - 1, 0), Y1 = min(cy + 1, g.G[1] - 1);
    const int Z0 = max(cz - 1, 0), Z1 = min(cz + 1, g.G[2] - 1);
    const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);

    // certification first: a lane whose cube cannot be proven to contain its ball does not walk
    bool whole;
    const float lb = box_lower_bound<2>(g, edges + (int64_t)n * 3 * kEdgeStride, qx, qy, qz, X0, X1, Y0, Y1, Z0, Z1,
                                        whole);
    const bool ok = whole || lb >= radius2;  // every unvisited point has computed dist2 >= lb
    const bool walk = active && ok;

    {
      int cnt = 0;
#pragma unroll
      for (int r = 0; r < kLaneRows; ++r) {
        constexpr int kDz[kLaneRows] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        constexpr int kDy[kLaneRows] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        const int z = cz + kDz[r], y = cy + kDy[r];
        if (walk && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
          const int rowbase = (z * g.G[1] + y) * g.G[0];
          const int s = cstart[rowbase + X0], e = cstart[rowbase + X1 + 1];
          if (e > s) {
            rows[lane + cnt] = make_int2(s, e);
            cnt += kGridWave;
          }
        }
      }
#pragma unroll
      for (int r = 0; r <= kLaneRows; ++r) {
        if (r * kGridWave >= cnt) s_rows[r][lane] = make_int2(0, 0);
      }
    }
    int rowi = lane + kGridWave;
    const int rowlast = lane + kLaneRows * kGridWave;
    unsigned off;
    int rem;
    {
      const int2 se = rows[lane];
      off = (unsigned)se.x * 16u;
      rem = (se.y - se.x) * 16;
    }
    auto advance = [&]() __attribute__((always_inline)) {
      rem -= kGroupBytes;
      off += kGroupBytes;
      if (rem <= 0) {
        const int2 se = rows[rowi];
        rowi = min(rowi + kGridWave, rowlast);
        off = (unsigned)se.x * 16u;
        rem = (se.y - se.x) * 16;
      }
    };
    const char* __restrict__ spb = (const char*)sp;
    auto record = [&](unsigned o, int u) __attribute__((always_inline)) -> float4 {
      return *(const float4*)(spb + o + (unsigned)(16 * u));
    };

    unsigned top[KC];  // ascending indices, kNone = empty
#pragma unroll
    for (int t = 0; t < KC; ++t) top[t] = kNone;
    unsigned thr = kNone;  // an index must be below the list's largest to matter (stale between flushes)
    int qn = lane;
    auto flush = [&]() __attribute__((always_inline)) {
      unsigned qk[kQueueCap];
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) qk[t] = s_queue[t * kGridWave + lane];  // free slots hold kNone
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) s_queue[t * kGridWave + lane] = kNone;
      bitonic_sort<kQueueCap>(qk);
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) top[KC - 1 - t] = min(top[KC - 1 - t], qk[t]);
      bitonic_merge<KC>(top);
      qn = lane;
      thr = top[KC - 1];
    };

    float4 c[G];
    int crem = rem;
#pragma unroll
    for (int u = 0; u < G; ++u) c[u] = record(off, u);
    while (__any(crem > 0)) {
      advance();
#pragma unroll
      for (int u0 = 0; u0 < G; u0 += kSub) {
        float dd[kSub];
        unsigned jj[kSub];
#pragma unroll
        for (int u = u0; u < u0 + kSub; ++u) {
          dd[u - u0] = point_dist<D, 2>(qx, qy, qz, c[u]);
          jj[u - u0] = __float_as_uint(c[u].w);
          c[u] = record(off, u);
        }
#pragma unroll
        for (int t = 0; t < kSub; ++t) {
          if (16 * (u0 + t) < crem && dd[t] < radius2 && jj[t] < thr) {
            s_queue[qn] = jj[t];
            qn += kGridWave;
          }
        }
        if (__any(qn > lane + (kQueueCap - kSub) * kGridWave)) flush();
      }
      crem = rem;
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
            const unsigned jx = top[k];
            const bool hit = jx != kNone;
            float d = 0.0f;
            if (hit) {
              float4 b;
              load_point3<D>(pts + (int64_t)jx * D, b.x, b.y, b.z);
              d = point_dist<D, 2>(qx, qy, qz, b);
            }
            oi[k] = hit ? (int64_t)jx : -1;
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
// Query ORDER for the clouds the index-order scan keeps (dense balls).  A wave of the scan kernel runs until
// ALL of its 64 queries have their K hits, and how long a query scans is set by how much of its ball lies
// inside the cloud: ~1/8 of the ball for a query in a corner of a uniform cube, so its scan is ~8x as long as
// an interior one, and with queries in storage order nearly every wave holds such a query.  Sorting the
// query ids by a coarse 16x16x16 cell of the cloud's bounding box puts queries with similar clipping -- and
// similar hit patterns -- in the same wave.  The scan kernel then takes these clouds in its list mode
// (qlist = all queries, coarse-cell order); results do not depend on the order.
// ---------------------------------------------------------------------------
constexpr int kOrderBlock = 256, kOrderPerThread = 8;

template <int D>
__device__ __forceinline__ int order_bin(const float* __restrict__ q, const unsigned* __restrict__ bbox) {
  int bin = 0;
#pragma unroll
  for (int d = D - 1; d >= 0; --d) {
    const float lo = funkey(bbox[d]), hi = funkey(bbox[3 + d]);
    const float t = (q[d] - lo) * ((float)kOrderG / fmaxf(hi - lo, FLT_MIN));
    int c = t < (float)kOrderG ? (int)t : kOrderG - 1;
    if (!(t >= 0.0f)) c = 0;
    bin = bin * kOrderG + c;
  }
  return bin;
}

static_assert(kOrderBlock * kOrderPerThread == 2048, "order_table is sized for tiles of 2048 queries");

// Counting sort of the query ids by coarse cell WITHOUT device atomics (round 2 drew every list position from a
// per-cell cursor in memory: 2 M atomics on 4096 hot counters per cloud, 139 us at cfg3):
//   count    every tile of 2048 queries writes its whole LDS histogram to order_table[n][tile][cell] (no zeroing pass);
//   scan     per cloud: per cell the exclusive prefix over the tiles, then the exclusive scan of the cell totals;
//            order_table[n][tile][cell] becomes the list position of the tile's first query of that cell;
//   scatter  the tile ranks its queries per cell in LDS again and stores the ids.
template <int D, bool SCATTER>
__global__ __launch_bounds__(kOrderBlock) void ball_order_kernel(const float* __restrict__ p1, int P1, GridWs ws) {
  const int n = blockIdx.y;
  const GridCloud g = ws.cloud[n];
  if (g.use_grid || g.len2 <= 0) return;  // grid clouds have their own lists; empty clouds: the scan pads
  __shared__ int s_bin[kOrderBins];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * (kOrderBlock * kOrderPerThread);
  if (i0 >= g.len1) return;
  int* __restrict__ table = ws.order_table + ((int64_t)n * gridDim.x + blockIdx.x) * kOrderBins;
  for (int b = tid; b < kOrderBins; b += kOrderBlock) s_bin[b] = 0;
  __syncthreads();
  const unsigned* __restrict__ bbox = ws.bbox + n * 8;
  int bin[kOrderPerThread], rank[kOrderPerThread];
#pragma unroll
  for (int r = 0; r < kOrderPerThread; ++r) {
    const int i = i0 + r * kOrderBlock + tid;
    bin[r] = -1;
    if (i < g.len1) {
      bin[r] = order_bin<D>(p1 + ((int64_t)n * P1 + i) * D, bbox);
      rank[r] = atomicAdd(&s_bin[bin[r]], 1);
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (int b = tid; b < kOrderBins; b += kOrderBlock) table[b] = s_bin[b];
  } else {
#pragma unroll
    for (int r = 0; r < kOrderPerThread; ++r)
      if (bin[r] >= 0) ws.fb2_list[(int64_t)n * P1 + table[bin[r]] + rank[r]] = i0 + r * kOrderBlock + tid;
  }
}

constexpr int kOrderScanBlock = 1024;
__global__ __launch_bounds__(kOrderScanBlock) void ball_order_scan_kernel(GridWs ws, int tiles_cap) {
  const int n = blockIdx.x;
  const GridCloud g = ws.cloud[n];
  if (g.use_grid || g.len2 <= 0) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int tiles = (g.len1 + 2047) / 2048;
  int* __restrict__ table = ws.order_table + (int64_t)n * tiles_cap * kOrderBins;
  constexpr int kPer = kOrderBins / kOrderScanBlock;  // consecutive cells per thread
  int tot[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) tot[u] = 0;
  static_assert(kPer == 4, "one int4 per thread");
  constexpr int kBatch = 16;  // tiles in flight (one load at a time: 64 dependent round trips, 91 us at cfg3)
  int4* __restrict__ col = (int4*)table + tid;  // this thread's four cells, tile t at col[t * (kOrderBins / 4)]
  for (int t0 = 0; t0 < tiles; t0 += kBatch) {  // per cell: exclusive prefix over the tiles
    int4 v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      v[u] = t0 + u < tiles ? col[(int64_t)(t0 + u) * (kOrderBins / 4)] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (t0 + u < tiles) col[(int64_t)(t0 + u) * (kOrderBins / 4)] = make_int4(tot[0], tot[1], tot[2], tot[3]);
      tot[0] += v[u].x;
      tot[1] += v[u].y;
      tot[2] += v[u].z;
      tot[3] += v[u].w;
    }
  }
  // exclusive scan of the cell totals over the workgroup: list positions in cell order.  (Measured and dropped: the
  // cells whose balls the bounding box clips -- the long scans -- first in the list: 662 -> 776 us for the scan kernel.)
  __shared__ int s_tot[kOrderScanBlock / kWave];
  const int mine = tot[0] + tot[1] + tot[2] + tot[3];
  int inc = mine;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int v = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += v;
  }
  if (lane == kWave - 1) s_tot[wave] = inc;
  __syncthreads();
  int run = inc - mine;
  for (int w = 0; w < wave; ++w) run += s_tot[w];
  int start[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    start[u] = run;
    run += tot[u];
  }
  for (int t0 = 0; t0 < tiles; t0 += kBatch) {
    int4 v[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      v[u] = t0 + u < tiles ? col[(int64_t)(t0 + u) * (kOrderBins / 4)] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (t0 + u < tiles)
        col[(int64_t)(t0 + u) * (kOrderBins / 4)] =
            make_int4(v[u].x + start[0], v[u].y + start[1], v[u].z + start[2], v[u].w + start[3]);
    }
  }
  if (tid == 0) {
    ws.fb2_count[n] = g.len1;  // the list holds every query of the cloud
    ws.grid_flag[n] = 1;       // "listed" (stream order: the scatter launch follows, the scan kernel after it)
  }
}

constexpr float kBallCellTarget = 2.0f;  // density floor of the cell size; the radius usually decides

size_t ball_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2) {
  return grid_carve(nullptr, nullptr, N, P1, P2, kBallCellTarget, true);
}

template <int D>
static void ball_run_d(const KnnArgs& a, float radius2, const GridWs& ws, int wgs) {
#define PO_BALL(KC)                                                                                              \
  hipLaunchKernelGGL((ball_grid_lane_kernel<D, KC>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1, a.p2, \
                     (const GridCloud*)ws.cloud, (const int*)ws.chunk_prefix, (const float*)ws.edges,            \
                     (const int*)ws.cell_start, (const float4*)ws.sorted, (const float4*)ws.qsorted, ws.fb2_count,    \
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
  POINTOPS_REQUIRE(a.N < 65536 && a.P2 <= knn_grid_max_points() && a.K <= 64 && a.D <= 3, "ball_query(grid): unsupported shape");
  GridWs ws;
  grid_carve(&ws, (char*)workspace, a.N, a.P1, a.P2, kBallCellTarget, true);
  ws.ball = 1;
  GridBuild b{};
  b.c_target = kBallCellTarget;
  b.h_min = fabsf(radius) * 1.001f;
  b.ball_radius = fabsf(radius);
  b.ball_K = a.K;
  // measured crossover (grid wins where K len2 / (E max(E, K)) > ~4-7), profiles/r01_ball_crossover.txt
  b.ball_factor = (float)debug_knob_f("ball_factor", 5.0);
  b.same = a.p1 == a.p2 && a.l1 == a.l2 && a.P1 == a.P2 && debug_knob("grid_same", 1) != 0;
  b.refine = -1;
  const int rc = grid_build(a, ws, b);
  if (rc != POINTOPS_OK) return rc;
  int64_t chunks = (int64_t)a.N * ceil_div(a.P1, kGridWave);  // one chunk of 64 queries per workgroup (see knn_grid_search.h)
  chunks = (chunks + 7) / 8 * 8;
  const int wgs = (int)(chunks < 2048 ? 2048 : (chunks > (1 << 20) ? (1 << 20) : chunks));
  const float radius2 = radius * radius;  // fp32 product (ball_query_cpu.cpp:26)
  switch (a.D) {
    case 1: ball_run_d<1>(a, radius2, ws, wgs); break;
    case 2: ball_run_d<2>(a, radius2, ws, wgs); break;
    default: ball_run_d<3>(a, radius2, ws, wgs); break;
  }
  // scan-mode clouds: all queries, ordered by coarse cell
  if (debug_knob("ball_order", 1) != 0) {
    const dim3 og((unsigned)ceil_div(a.P1, kOrderBlock * kOrderPerThread), (unsigned)a.N);
#define PO_ORDER(DD)                                                                                         \
  hipLaunchKernelGGL((ball_order_kernel<DD, false>), og, dim3(kOrderBlock), 0, a.stream, a.p1, a.P1, ws);     \
  hipLaunchKernelGGL(ball_order_scan_kernel, dim3((unsigned)a.N), dim3(kOrderScanBlock), 0, a.stream, ws,    \
                     (int)og.x);                                                                             \
  hipLaunchKernelGGL((ball_order_kernel<DD, true>), og, dim3(kOrderBlock), 0, a.stream, a.p1, a.P1, ws)
    switch (a.D) {
      case 1: PO_ORDER(1); break;
      case 2: PO_ORDER(2); break;
      default: PO_ORDER(3); break;
    }
#undef PO_ORDER
  }
  *flag = ws.grid_flag;
  *qcount = ws.fb2_count;
  *qlist = ws.fb2_list;
  return check_launch("ball_query(grid)");
}

}  // namespace pointops
