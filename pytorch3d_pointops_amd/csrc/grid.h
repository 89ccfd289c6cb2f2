// grid.h -- the per-cloud cell grid shared by the exact grid-pruned searches (gfx950):
// KNN (knn_grid.hip, knn_grid_search.h), ball query (ball_grid.hip) and the build passes
// (grid_build.hip).  See knn_grid.hip for the algorithm and the exactness argument.
#pragma once
#include <float.h>
#include <math.h>

#include "knn_grid.h"

namespace pointops {

constexpr int kGMax = 1024;          // cells per dimension cap (edge table size)
constexpr int kEdgeStride = kGMax + 2;
constexpr int kGridWave = 64;
constexpr int kNumXcd = 8;           // MI355X: 8 XCDs x 32 CUs, private 4 MB L2 each
constexpr int kCoarseMax = 1024;     // micro-bins (and bins) per cloud and set of the two-level sort
constexpr int kCrowdedMax = 64;      // crowded coarse bins listed per cloud and set (further ones: one workgroup each)
constexpr int kFineLogMax = 12;      // cells per bin <= 4096
constexpr int kOrderG = 16, kOrderBins = kOrderG * kOrderG * kOrderG;  // ball query: coarse cells of the scan-mode query order
constexpr int kSortedPad = 32;       // records of padding behind every cloud's sorted array: record P2 is a NaN
                                     // sentinel (never a candidate); group loads of the lane searches may run
                                     // up to 7 records past a run's end (G x stride - 1 in the strided walk of the radius-2 pass)
                                     // and stay inside the cloud's array

struct GridCloud {
  float lo[3];
  float inv_h;
  int G[3];
  int ncell;
  int len1, len2;
  int use_grid;
  int same;  // 1 = the queries ARE the points (p1 == p2, lengths1 == lengths2): the point sort is the query order
  int mshift, nmicro;  // two-level sort (grid_build.hip): micro-bin = cell id >> mshift, nmicro <= kCoarseMax of them
};

// A REFINED cell: a level-0 cell holding far more points than the target gets its own s x s x s sub-grid over the
// robust extent (mean +- 2.2 sigma per dimension) of ITS points; the cell's records are re-sorted by sub-cell inside
// the cell's range of the sorted array.  sub_d(x) = clamp(int((x - lo_d) * scale_d), 0, s - 1): monotone, total.
struct RefinedCell {
  int start, count;  // the cell's range of the cloud's sorted records
  int s;             // sub-cells per dimension
  int pool_off;      // sub_start table (s^3 + 1 ints, relative to `start`) inside the cloud's pool
  float lo[3], scale[3];
};

struct GridWs {
  GridCloud* cloud;   // N
  int* chunk_prefix;  // N + 1     64-query chunks of the clouds before cloud n
  float* edges;       // N * 3 * kEdgeStride
  int* cell_start;    // N * (cell_cap + 1)
  int* coarse_count;  // N * 2 * (kCoarseMax + 1)   entries per micro-bin (set 0 points, 1 queries)
  int* bin_of;        // N * 2 * (kCoarseMax + 1)   micro-bin -> bin (consecutive micro-bins with ~kCoarsePoints entries)
  int* bin_first;     // N * 2 * (kCoarseMax + 1)   bin -> its first micro-bin; [nbins] = nmicro
  int* nbins;         // N * 2
  int* coarse_cursor; // N * 2 * (kCoarseMax + 1)   per bin: entries handed out so far
  int* coarse_start;  // N * 2 * (kCoarseMax + 1)   per bin: first entry; [nbins] = entries of the set
  float4* sorted;     // N * (P2 + kSortedPad)   (x, y, z, idx bits) by cell
  int* coarse_ticket; // N * 2           tiles of the count launch that have finished
  int* crowded_count; // N * 2           crowded coarse bins (<= kCrowdedMax), listed in
  int* crowded_list;  // N * 2 * kCrowdedMax
  int* fine_count;    // N * 2 * cell_cap   records per cell, maintained for the cells of crowded bins only
  int* prank;         // N * P2          rank of a record inside its cell (records of crowded bins only)
  int* qrank;         // N * P1
  float4* qtmp;       // N * P1         query records grouped by coarse bin
  float4* qsorted;    // N * P1         query records (x, y, z, idx bits) by cell: the query order of the lane searches
  int* order_table;   // N * ceil(P1 / 2048) * kOrderBins   ball query: coarse query order of the scan-mode clouds
                      // (ball_grid.hip): queries per (2048-query tile, coarse cell), then their first list position
  int* fb_count;      // N          queries the lane search could not certify
  int* fb_list;       // N * P1
  unsigned* fb_kth;   // N * P1     estimated KC-th distance (fp32 bits) of an uncertified query: picks the quad pass's cube
  int* fb2_count;     // N          queries the expanding search gave up on (whole-cloud scan)
  int* fb2_list;      // N * P1
  int* fb3_count;     // N          queries the radius-2 quad search could not certify (expanding search)
  int* fb3_list;      // N * P1
  unsigned* bbox;     // N * 8: ordered-uint keys of min x,y,z and max x,y,z
  float* bbox_part;   // N * ceil(P2 / 2048) * 6: min / max of every 2048-point tile
  int* grid_flag;     // N          1 = the cloud was searched through its grid (ball query: scan only the list)
  // refined cells and the box search (grid_refine.hip, knn_grid_box.h)
  int* refine_ref;    // N * cell_cap   per cell: index of its RefinedCell, or -1
  RefinedCell* rdesc; // N * rdesc_cap
  int* rcount;        // N          refined cells of the cloud (may exceed rdesc_cap: clamp)
  int* pool;          // N * pool_cap   sub_start tables
  int* pool_top;      // N
  float4* sorted_tmp; // N * P2     records grouped by coarse bin; later the scratch of the in-cell re-sort
  int* box_count;     // N          queries deferred to the box search
  int* box_list;      // N * P1
  int rdesc_cap, pool_cap;
  float c_target;     // points per cell the grid was sized for
  int cell_cap;
  int ball;           // 0 = KNN (pad rows with idx 0), 1 = ball query (pad with idx -1; clouds without a
                      //     usable grid are left to the scan kernel instead of the query list)
};

// parameters of one grid build
struct GridBuild {
  float c_target;      // points per cell the cell size aims at
  float h_min;         // lower bound of the cell edge (ball query: 1.001 radius), 0 = none
  float ball_radius;   // ball query only: grid-or-scan decision per cloud on the device
  int ball_K;
  float ball_factor;
  bool same;           // p1 and p2 are the same buffer with the same lengths: sort once
  int refine;          // -1: no refined-cell bookkeeping (ball query); 0: every cell marked unrefined; 1: over-full
                       // cells get a descriptor + table during the scan (grid_refine() then builds their sub-grids)
};

constexpr int kRefineMaxS = 32;  // sub-cells per dimension cap (32^3 counters = the LDS of one workgroup)

// A query leaves the lane search for the box search when its 3x3x3 cube holds more than kDeferFactor x
// refine_threshold() records (~4x what a uniform cloud gives it); a cell is refined when it holds more than
// refine_threshold() points -- a quarter of that limit, so a cube is only ever over-full through cells that are
// refined or through many moderately full ones (which the box search, with its smaller box, walks whole).
// (Refining from 8x the target on cost the sphere-surface cloud 0.15 ms of sub-sorts that no query used.)
__host__ __device__ inline int refine_threshold(float c_target) {
  const float r = 28.0f * c_target;
  return r < 64.0f ? 64 : (int)r;
}
constexpr int kDeferFactor = 4;

__device__ __forceinline__ int sub_of(float x, float lo, float scale, int s) {
  const float t = (x - lo) * scale;  // unfused; monotone non-decreasing in x
  int c = (t < (float)s) ? (int)t : s - 1;
  if (!(t >= 0.0f)) c = 0;
  return c;
}

// workspace layout (grid_build.hip)
size_t grid_carve(GridWs* ws, char* base, int64_t N, int64_t P1, int64_t P2, float c_target, bool ball = false);
// bbox, cell size, edge tables, counting sorts of points and queries, chunk prefix; stream-ordered
int grid_build(const KnnArgs& a, const GridWs& ws, const GridBuild& b);
// the query side only, against a point side built by an earlier call into the same workspace (level 1: new queries,
// level 2: the same queries); stream-ordered
int grid_build_queries(const KnnArgs& a, const GridWs& ws, bool same, int level);
// build the sub-grids of the cells grid_build marked (grid_refine.hip); stream-ordered behind grid_build
int grid_refine(const KnnArgs& a, const GridWs& ws);

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

template <int NORM>
__device__ __forceinline__ float face_bound(float t) {  // t = fl(|q - face|) >= 0
  return NORM == 1 ? t : t * t;
}

template <int D, int NORM>
__device__ __forceinline__ float point_dist(float qx, float qy, float qz, const float4 c) {
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
  return d;
}

// Rigorous lower bound of the computed distance of every point outside the cell box
// [X0..X1] x [Y0..Y1] x [Z0..Z1] (see the header of knn_grid.hip); +inf when the box is the whole grid.
template <int NORM>
__device__ __forceinline__ float box_lower_bound(const GridCloud& g, const float* __restrict__ ed, float qx, float qy,
                                                 float qz, int X0, int X1, int Y0, int Y1, int Z0, int Z1,
                                                 bool& whole) {
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
  whole = !(hx0 || hx1 || hy0 || hy1 || hz0 || hz1);
  return lb;
}

// Candidate threshold seeded from the certification bound lb: only candidates with d < lb can
// appear in a certified answer (certification needs the KC-th best below lb), so the search may
// ignore the rest from the first record on.  Distances are non-negative, so bit order = value order.
__device__ __forceinline__ unsigned seed_threshold(float lb, bool whole) {
  const unsigned b = __float_as_uint(lb);
  return whole ? 0x7f800000u : (b > 0u ? b - 1u : 0u);
}

// cloud of a chunk item: largest n with prefix[n] <= item (wave-uniform).  `per_cloud` = chunks of a FULL cloud: in a
// batch of full clouds item / per_cloud is the answer, checked with two independent loads; ragged batches fall back to the
// binary search (log2 N dependent loads at the head of every workgroup).
__device__ __forceinline__ int item_cloud(const int* __restrict__ prefix, int N, int item, int per_cloud) {
  const int guess = min(item / per_cloud, N - 1);
  if (prefix[guess] <= item && item < prefix[guess + 1]) return guess;
  int lo_n = 0, hi_n = N;
  while (hi_n - lo_n > 1) {
    const int mid = (lo_n + hi_n) >> 1;
    if (prefix[mid] <= item) lo_n = mid;
    else hi_n = mid;
  }
  return lo_n;
}

}  // namespace pointops
