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
//
// Passes (all clouds of the batch in every launch, no host synchronisation):
//   0-4 grid_build.hip: bounding boxes, cell size + exact edge tables, two-level counting sort of the points and of
//       the queries into (x,y,z,idx) float4 records by cell (count / scatter by bin / per-bin sort);
//   4b  grid_refine.hip: sub-grids of over-full cells (multi-scale clouds);
//   5   knn_grid_lane_kernel (knn_grid_search.h): one query per lane over the 3x3x3 cell cube around its
//       cell; afterwards each lane checks kth_dist < LB, LB = the bound below over the cube's faces; on
//       failure the query id goes to the fallback list;
//   5b  knn_grid_quad_kernel: four lanes per uncertified query, cube grown where an estimate says so;
//   5c  knn_grid_box_kernel (knn_grid_box.h): box search over refined cells for over-full neighbourhoods (and for
//       the uncertified queries of the 64-slot lists, which have no quad pass);
//   6   knn_grid_wave_kernel: wave-per-query search of a cell cube that doubles until certified;
//   7   knn_reg_kernel (knn.hip): whole-cloud scan for the queries pass 6 gave up on and for clouds
//       without a usable grid.
//
// Lower bound.  Let the visited region be cells [X0..X1]x[Y0..Y1]x[Z0..Z1].  A point in
// an unvisited cell has, in some dimension d, cell_d < X0 or cell_d > X1.  cell_d is
// monotone in the coordinate, so p_d <= prev(E_d[X0]) =: f or p_d >= E_d[X1+1] =: f.  fp32
// subtraction and multiplication by itself are monotone, so the COMPUTED |q_d - p_d| is
// >= fl(|q_d - f|) and the computed square >= fl(fl(|q_d - f|)^2); adding the other
// (non-negative) terms and rounding cannot go below that.  Hence the computed distance
// of every unvisited point is >= LB, and `kth < LB` (strict) also rules out ties.
#include "debug.h"
#include "grid.h"
#include "knn_grid_search.h"

namespace pointops {

// list capacity of the search kernels (sorting networks exist for powers of two)
static int grid_kc(int K) { return K <= 1 ? 1 : K <= 2 ? 2 : K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 64; }

static float knn_cell_target(int K) {
  // the search keeps the KC >= K best but certifies the K-th, so the cells are sized for K points:
  // measured optimum at B=32, N=65536 (profiles/r02_grid_tuning.txt)
  const int kc = grid_kc(K);
  float c = (kc >= 8 ? 0.4f : 0.625f) * (float)K;
  c *= (float)debug_knob_f("grid_c_scale", 1.0);  // tuning experiments only
  return c < 1.0f ? 1.0f : c;
}

size_t knn_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K) {
  return grid_carve(nullptr, nullptr, N < 32768 ? N : 32768, P1, P2, knn_cell_target((int)K));  // (kGridBatchSlice)
}

static bool grid_quad_mode(int64_t queries, int kc) {
  // The quad pass has a ~60 us floor (one wave walking ~200 candidates per lane), which only pays when the BATCH sends
  // it thousands of queries; below that the expanding wave search takes the uncertified queries directly.  Measured
  // with / without the pass (ms): 64 x 16384 queries K=16 0.412 / 0.461, 32 x 32768 K=8 0.303 / 0.320, but 8 x 32768
  // K=16 0.206 / 0.186 and 64 x 8192 K=8 0.241 / 0.220; 32-slot lists: 64 x 8192 0.468 / 0.555, 16 x 16384
  // 0.307 / 0.319, 32 x 4096 0.271 / 0.252.  Debug knob grid_quad=0/1 forces the choice (tests of both paths).
  const long k = debug_knob("grid_quad", -1);
  return k >= 0 ? k != 0 : queries >= (kc >= 32 ? (1 << 18) : (1 << 20));
}

// clouds per launch: the build passes and the fallback searches put the cloud on blockIdx.y (< 65536); a bigger
// batch is searched in slices of this many clouds through the same workspace, one after the other
constexpr int64_t kGridBatchSlice = 32768;

int64_t knn_grid_max_points() { return kGridMaxPointsBig; }

static int knn_grid_run_slice(const KnnArgs& a, int norm, void* workspace, int reuse) {
  GridWs ws;
  const float c = knn_cell_target(a.K);
  grid_carve(&ws, (char*)workspace, a.N, a.P1, a.P2, c);
  GridBuild b{};
  b.c_target = c;
  // the queries are the points (self-KNN): their cell sort is the query order
  b.same = a.p1 == a.p2 && a.l1 == a.l2 && a.P1 == a.P2 && debug_knob("grid_same", 1) != 0;
  b.refine = (debug_knob("grid_refine", 1) != 0 && a.K <= 64) ? 1 : 0;  // (the long-list search walks whole cells)
  int rc;
  if (reuse > 0) {  // the point side of the workspace is the previous call's (the caller vouches for it)
    if ((rc = grid_build_queries(a, ws, b.same, reuse)) != POINTOPS_OK) return rc;
  } else {
    if ((rc = grid_build(a, ws, b)) != POINTOPS_OK) return rc;
    if (b.refine && (rc = grid_refine(a, ws)) != POINTOPS_OK) return rc;
  }
  if (a.K > 64) {  // long lists: a wave per query sorts its cube's candidates (knn_grid_wsort.hip)
    grid_search_wsort(a, ws, norm);
    rc = check_launch("knn_points_idx(grid, long lists)");
    if (rc != POINTOPS_OK) return rc;
    KnnArgs fa = a;
    fa.qlist = ws.fb2_list;
    fa.qcount = ws.fb2_count;
    if ((rc = launch_knn_wide(fa, norm, nullptr)) != POINTOPS_OK) return rc;
    return check_launch("knn_points_idx(grid fallback)");
  }
  const int kc = grid_kc(a.K);
  const bool quad = kc <= 32 && grid_quad_mode(a.N * (int64_t)a.P1, kc);  // (64-slot lists: four of them do not fit a quad's registers)
  // run words of the lane / box searches: 21-bit record indices, or 24-bit ones for the biggest clouds
  const bool big = a.P2 > kGridMaxPoints || debug_knob("grid_big", 0) != 0;
  switch (a.D) {
    case 1: (big ? grid_search_d1w : grid_search_d1)(a, ws, norm, kc, quad); break;
    case 2: (big ? grid_search_d2w : grid_search_d2)(a, ws, norm, kc, quad); break;
    default: (big ? grid_search_d3w : grid_search_d3)(a, ws, norm, kc, quad); break;
  }
  rc = check_launch("knn_points_idx(grid)");
  if (rc != POINTOPS_OK) return rc;
  // exact fallback: whole-cloud scan for the queries the bound could not certify
  KnnArgs fa = a;
  fa.qlist = ws.fb2_list;
  fa.qcount = ws.fb2_count;
  if (a.K <= 32) launch_knn_bruteforce(fa, norm);
  else if ((rc = launch_knn_wide(fa, norm, nullptr)) != POINTOPS_OK) return rc;  // 64-key lists (knn_wide.hip)
  return check_launch("knn_points_idx(grid fallback)");
}

int knn_grid_run(const KnnArgs& a, int norm, void* workspace, int reuse) {
  POINTOPS_REQUIRE(a.P2 <= kGridMaxPointsBig, "knn_points_idx(grid): P2 must be <= 2^24 - 16");
  if (a.N <= kGridBatchSlice) return knn_grid_run_slice(a, norm, workspace, reuse);
  // more clouds than a launch takes: slices share the workspace (stream order), so nothing of it outlives the call
  for (int64_t n0 = 0; n0 < a.N; n0 += kGridBatchSlice) {
    KnnArgs s = a;
    const bool same_pts = a.p1 == a.p2, same_len = a.l1 == a.l2;
    s.N = a.N - n0 < kGridBatchSlice ? a.N - n0 : kGridBatchSlice;
    s.p1 = a.p1 + n0 * (int64_t)a.P1 * a.D;
    s.p2 = same_pts ? s.p1 : a.p2 + n0 * (int64_t)a.P2 * a.D;
    s.l1 = a.l1 + n0;
    s.l2 = same_len ? s.l1 : a.l2 + n0;
    s.idxs = a.idxs + n0 * (int64_t)a.P1 * a.K;
    s.dists = a.dists + n0 * (int64_t)a.P1 * a.K;
    const int rc = knn_grid_run_slice(s, norm, workspace, 0);
    if (rc != POINTOPS_OK) return rc;
  }
  return POINTOPS_OK;
}

}  // namespace pointops

extern "C" int pointops_knn_grid_fallback_counts(const void* workspace, int64_t N, int64_t P1, int64_t P2,
                                                 int64_t K, int32_t* counts, void* stream) {
  using namespace pointops;
  POINTOPS_REQUIRE(workspace != nullptr && counts != nullptr && N > 0 && N <= 32768, "knn_grid_fallback_counts: bad arguments");
  GridWs ws;
  grid_carve(&ws, (char*)workspace, N, P1, P2, knn_cell_target((int)K));
  if (hipMemcpyAsync(counts + N, ws.fb2_count, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice,
                     (hipStream_t)stream) != hipSuccess)
    return check_launch("knn_grid_fallback_counts");
  if (hipMemcpyAsync(counts, ws.fb_count, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice,
                     (hipStream_t)stream) != hipSuccess)
    return check_launch("knn_grid_fallback_counts");
  return POINTOPS_OK;
}

// grid geometry + fallback counters of the last knn_points_idx call that used `workspace`:
// stats (N, 14) int32 = G[0], G[1], G[2], ncell, use_grid, uncertified after the lane pass, after the quad + box passes,
// sent to the whole-cloud scan, deferred to the box search, refined cells; build: bins of the point sort / of the query
// sort, crowded bins (listed ones) of the point sort / of the query sort
namespace pointops {
__global__ void grid_stats_kernel(GridWs ws, int N, int32_t* __restrict__ stats) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const GridCloud g = ws.cloud[n];
  int32_t* s = stats + (int64_t)n * 14;
  s[0] = g.G[0];
  s[1] = g.G[1];
  s[2] = g.G[2];
  s[3] = g.ncell;
  s[4] = g.use_grid;
  s[5] = ws.fb_count[n];
  s[6] = ws.fb3_count[n];
  s[7] = ws.fb2_count[n];
  s[8] = ws.box_count[n];
  s[9] = ws.rcount[n];
  s[10] = ws.nbins[n * 2];
  s[11] = g.same ? 0 : ws.nbins[n * 2 + 1];
  s[12] = ws.crowded_count[n * 2];
  s[13] = g.same ? 0 : ws.crowded_count[n * 2 + 1];
}
}  // namespace pointops

extern "C" int pointops_knn_grid_stats(const void* workspace, int64_t N, int64_t P1, int64_t P2, int64_t K,
                                       int32_t* stats, void* stream) {
  using namespace pointops;
  POINTOPS_REQUIRE(workspace != nullptr && stats != nullptr && N > 0 && N <= 32768, "knn_grid_stats: bad arguments");
  GridWs ws;
  grid_carve(&ws, (char*)workspace, N, P1, P2, knn_cell_target((int)K));
  hipLaunchKernelGGL(grid_stats_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, ws, (int)N,
                     stats);
  return check_launch("knn_grid_stats");
}
