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
    const float4* __restrict__ sorted, const int* __restrict__ qlist, int* __restrict__ fb_count,
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
    const int n = item_cloud(chunk_prefix, N, item);
    const GridCloud g = clouds[n];
    const int c0 = (item - chunk_prefix[n]) * kGridWave;
    const bool active = c0 + lane < g.len1;
    const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + kSortedPad);
    int qi = 0;
    float qx = 0.0f, qy = 0.0f, qz = 0.0f;
    if (g.same) {
      if (active) {
        const float4 q = sp[c0 + lane];
        qx = q.x;
        qy = q.y;
        qz = q.z;
        qi = __float_as_int(q.w);
      }
    } else if (active) {
      qi = qlist[(int64_t)n * P1 + c0 + lane];
      load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, qx, qy, qz);
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

constexpr float kBallCellTarget = 2.0f;  // density floor of the cell size; the radius usually decides

size_t ball_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2) {
  return grid_carve(nullptr, nullptr, N, P1, P2, kBallCellTarget);
}

template <int D>
static void ball_run_d(const KnnArgs& a, float radius2, const GridWs& ws, int wgs) {
#define PO_BALL(KC)                                                                                              \
  hipLaunchKernelGGL((ball_grid_lane_kernel<D, KC>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1, a.p2, \
                     (const GridCloud*)ws.cloud, (const int*)ws.chunk_prefix, (const float*)ws.edges,            \
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
  grid_carve(&ws, (char*)workspace, a.N, a.P1, a.P2, kBallCellTarget);
  ws.ball = 1;
  GridBuild b{};
  b.c_target = kBallCellTarget;
  b.h_min = fabsf(radius) * 1.001f;
  b.ball_radius = fabsf(radius);
  b.ball_K = a.K;
  // measured crossover (grid wins where K len2 / (E max(E, K)) > ~4-7), profiles/r01_ball_crossover.txt
  b.ball_factor = (float)debug_knob_f("ball_factor", 5.0);
  b.same = a.p1 == a.p2 && a.l1 == a.l2 && a.P1 == a.P2 && debug_knob("grid_same", 1) != 0;
  const int rc = grid_build(a, ws, b);
  if (rc != POINTOPS_OK) return rc;
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
