// knn_grid_search.h -- search kernels of the exact grid-pruned KNN (gfx950); the algorithm, the
// exactness argument and the pass list are in knn_grid.hip.  Templates only: the translation units
// knn_grid_d*.hip instantiate them per point dimension so that they compile in parallel.
#pragma once
#include "debug.h"
#include "grid.h"
#include "knn_common.h"
#include "sort_net.h"

#include <type_traits>

namespace pointops {

// Write one output row from a TopKF64 list: the first min(K, len2) entries, zeros after (knn_cpu.cpp:25-26).
template <int KC>
__device__ __forceinline__ void write_row_f64(const TopKF64<KC>& top, int K, int len2, int64_t* __restrict__ orow_i,
                                              float* __restrict__ orow_d) {
  const int kvalid = len2 < K ? len2 : K;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    if (k < K) {
      const bool ok = k < kvalid;
      orow_i[k] = ok ? (int64_t)top.idx_at(k) : 0;
      orow_d[k] = ok ? top.dist_at(k) : 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------
// pass 5 (lane-private search): one query per lane, EVERY lane walks only the 3x3x3 cell cube
// around its own cell -- up to 9 contiguous runs of the cell-sorted records (a row of three
// x-cells is contiguous).  The queries are processed in cell order, so the 64 lanes of a wave sit
// in ~10 neighbouring cells and their runs overlap in L1/L2.
//
// Walk.  A lane consumes its runs in GROUPS of G consecutive records: one 32-bit byte offset per
// group, the G gathers are `global_load_dwordx4 v, v_off, s[base] offset:16u` (no per-record address
// arithmetic), a record's registers are reloaded with the NEXT group's record as soon as its distance
// is taken (in-place software pipeline).  A group never straddles two runs: the tail of a run's last
// group lies beyond the run (real records of other cells, or the pad behind the cloud) and is masked
// by ONE compare per record (`16u < remaining bytes`).  Moving to the next run is per lane, under an
// exec mask, from a per-lane list of the NON-EMPTY runs in LDS.  (Round 1 let groups straddle runs with
// per-record selects: ~26 VALU per record, a third of them v_cndmask_b32 on a shared VCC, which costs
// ~22 cycles for every consumer after the first -- tools/valu_microbench.hip.  Now ~11.)
//
// Selection.  KC >= 8: candidates that beat the lane's (stale) threshold are parked in a per-lane LDS
// queue; when ANY lane's queue is nearly full all lanes merge queue and list with branch-free sorting
// networks on 64-bit (dist, idx) keys driven through the FP64 pipe (sort_net.h: a compare-exchange is
// v_min_f64 + v_max_f64).  Queue slots are re-filled with the empty key after a flush, so reading the
// queue needs no per-slot validity select.  KC <= 4: branch-free sorted insert (2 KC - 1 FP64 min/max)
// under the exec mask of the lanes whose candidate passes.
//
// The certification bound is known before the walk and seeds the threshold (seed_threshold()).
// ---------------------------------------------------------------------------
constexpr int kLaneRows = 9;

template <int KC>
struct LaneCfg {
  static constexpr bool kUseQueue = KC >= 8;
  static constexpr int kQueueCap = 16;                  // keys a flush sorts (KC = 8 merges the 8 smallest)
  // per-lane LDS queue slots: 15 of them + the 2.5 KB run table = 10 KB per wave, SIXTEEN waves per CU (the VGPR limit of
  // the K <= 16 kernels) instead of fifteen: cfg2 0.715 -> 0.703 ms per step (14 slots: 0.707)
  static constexpr int kQueueLds = KC <= 16 ? 15 : 16;
#ifndef POINTOPS_LANE_GROUP
#define POINTOPS_LANE_GROUP 4  // 8 measured 0.776 vs 0.755 ms at cfg2 (K=16), 1.46 vs 1.31 ms at K=32: fewer masked slots
#endif                         // (a run of ~19 records wastes 1.5 of 4 against 3.5 of 8) beat the fewer, wider groups
  static constexpr int kGroup = KC >= 8 ? POINTOPS_LANE_GROUP : 4;  // records per group
  static constexpr int kSub = kGroup;                   // candidates between two queue-full checks (whole groups:
                                                        // a check inside the group needs two code paths that merge, and
                                                        // the compiler then rotates the record registers with copies
                                                        // behind an s_waitcnt vmcnt(0))
};

// ---------------------------------------------------------------------------
// The walk + selection core shared by the lane search and the box search: the calling lane's runs are the
// packed words rows[lane + r * 64], r = 0 .. ROWS (first record << kRunBits | length; zero-terminated, slot ROWS
// is always zero); on return `top` holds the KC best of those records (see the comment above the lane kernel).
// `s_queue` is the wave's 16 x 64 queue, every slot holding the empty key on entry and on return.
// ---------------------------------------------------------------------------
// Run words: first record << RB | length.  RB = 11 (runs of up to 2047 records, clouds of up to 2^21 - 16 points) is what
// every kernel below was tuned with; RB = 8 (runs of up to 255 records -- a longer run sends its query to the box
// search, which splits runs -- clouds of up to 2^24 - 16 points) exists so that big clouds do not fall back to the
// all-pairs scan (knn_grid_d*w.hip instantiate it).
constexpr int kRunBitsStd = 11, kRunBitsBig = 8;
constexpr int64_t kGridMaxPoints = (1LL << (32 - kRunBitsStd)) - 16, kGridMaxPointsBig = (1LL << (32 - kRunBitsBig)) - 16;

// STRIDE > 1: the lane takes every STRIDE-th record of its runs (the caller has shifted each run's start by the lane's
// rank among the STRIDE lanes that share the runs and shortened its length by as much): record u of a group sits
// 16 STRIDE u bytes behind the group's offset, and the STRIDE lanes' gathers fall into the same 64-byte segments.
template <int D, int KC, int NORM, int ROWS, int RB, int STRIDE = 1>
__device__ __forceinline__ void lane_walk(const char* __restrict__ spb, const unsigned* rows, int lane,
                                          double* s_queue, float qx, float qy, float qz, unsigned thr0,
                                          TopKF64<KC>& top) {
  using Cfg = LaneCfg<KC>;
  constexpr bool kUseQueue = Cfg::kUseQueue;
  constexpr int kQueueCap = Cfg::kQueueCap;
  constexpr int kQueueLds = Cfg::kQueueLds;
  constexpr int kSub = Cfg::kSub;
  constexpr int G = Cfg::kGroup;
  constexpr int kGroupBytes = G * 16 * STRIDE;
  static_assert(G * STRIDE <= kSortedPad, "a masked tail reads up to G * STRIDE records past its run");
  constexpr int kRunBits = RB, kRunMax = (1 << RB) - 1;
  double* const qbase = s_queue + lane;
  int rowi = lane + 2 * kGridWave;  // entry of `rows` after the prefetched one
  const int rowlast = lane + ROWS * kGridWave;
  unsigned off;  // byte offset of the lane's next group inside the cloud's record array
  int rem;       // bytes of the current run from `off` on (<= 0: the run is used up)
  unsigned nse;  // the lane's next run, read from LDS one switch ahead (its latency stays off the walk)
  {
    const unsigned se = rows[lane];
    off = (se >> kRunBits) * 16u;
    rem = (int)(se & (unsigned)kRunMax) * 16;
    nse = rows[lane + kGridWave];
  }
  auto advance = [&]() __attribute__((always_inline)) {
    rem -= kGroupBytes;
    off += kGroupBytes;
    if (rem <= 0) {  // next run of this lane (exec-masked; some lane switches in most iterations)
      off = (nse >> kRunBits) * 16u;
      rem = (int)(nse & (unsigned)kRunMax) * 16;
      nse = rows[rowi];
      rowi = min(rowi + kGridWave, rowlast);
    }
  };
  auto record = [&](unsigned o, int u) __attribute__((always_inline)) -> float4 {
    return *(const float4*)(spb + o + (unsigned)(16 * STRIDE * u));  // saddr + 32-bit voffset + immediate
  };

  unsigned thr = thr0;
  int qn = lane;  // next free slot of the lane's queue (element index into s_queue)

  float4 c[G];
  int crem = rem;  // validity of the group held in c[]: record u is part of the run iff 16 u < crem
#pragma unroll
  for (int u = 0; u < G; ++u) c[u] = record(off, u);
  // kSub records of the group in c[], starting at u0 (compile-time): distances, then reload the registers
  // with the NEXT group's records, then the threshold tests
  auto part = [&](auto u0c) __attribute__((always_inline)) {
    constexpr int u0 = decltype(u0c)::value;
    float dd[kSub];
    int ii[kSub];
#pragma unroll
    for (int u = u0; u < u0 + kSub; ++u) {
      dd[u - u0] = point_dist<D, NORM>(qx, qy, qz, c[u]);
      ii[u - u0] = __float_as_int(c[u].w);
    }
    // the reloads stay BEHIND the distances (hoisted above them, old and new records overlap in lifetime
    // and the compiler copies four float4 per part to rotate the registers)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = u0; u < u0 + kSub; ++u) c[u] = record(off, u);  // next group's records into the same registers
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < kSub; ++t) {
      const bool valid = 16 * STRIDE * (u0 + t) < crem;
      if (kUseQueue) {
        if (valid && __float_as_uint(dd[t]) <= thr) {
          s_queue[qn] = TopKF64<KC>::make(dd[t], ii[t]);
          qn += kGridWave;
        }
      } else if (valid && __float_as_uint(dd[t]) <= min(top.worst_bits(), thr0)) {
        top.insert(TopKF64<KC>::make(dd[t], ii[t]));
      }
    }
  };
  if constexpr (kUseQueue) {
    // The sorted list lives in 2 KC registers that only a flush touches: the walk is an INNER loop that
    // never names them (a flush inside the walk loop made the compiler shuffle the whole list between
    // two register sets on every iteration), left whenever some lane's queue could overflow in the next group.
    static_assert(G == kSub && kQueueCap >= 2 * kSub, "queue geometry");
    bool more = __any(crem > 0);
    while (more) {
      bool full;
      do {
        advance();  // (off, rem) now describe the NEXT group
        part(std::integral_constant<int, 0>{});
        crem = rem;
        more = __any(crem > 0);
        full = __any(qn > lane + (kQueueLds - kSub) * kGridWave);
      } while (!full && more);
      // merge: queue -> sorted network -> list
      double qk[kQueueCap];
#pragma unroll
      for (int t = 0; t < kQueueCap; ++t) qk[t] = t < kQueueLds ? qbase[t * kGridWave] : TopKF64<KC>::empty();
#pragma unroll
      for (int t = 0; t < kQueueLds; ++t) qbase[t * kGridWave] = TopKF64<KC>::empty();  // (free slots hold the empty key)
      bitonic_sort<kQueueCap>(qk);
      constexpr int kMeet = KC < kQueueCap ? KC : kQueueCap;
#pragma unroll
      for (int t = 0; t < kMeet; ++t) top.key[KC - 1 - t] = kmin(top.key[KC - 1 - t], qk[t]);  // list slot KC-1-t meets queue entry t
      bitonic_merge<KC>(top.key);
      qn = lane;
      thr = min(top.worst_bits(), thr0);
    }
  } else {
    static_assert(G == kSub, "direct-insert variants take one part per group");
    while (__any(crem > 0)) {
      advance();
      part(std::integral_constant<int, 0>{});
      crem = rem;
    }
  }
}

// (second launch bound = waves per SIMD the compiler has to leave room for: the 64-slot list needs 261 registers
// without it, five more than two waves allow)
template <int D, int KC, int NORM, int RB>
__global__ __launch_bounds__(kGridWave, KC > 32 ? 2 : 1) void knn_grid_lane_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const int* __restrict__ chunk_prefix,
    const float* __restrict__ edges, const int* __restrict__ cell_start, const float4* __restrict__ sorted,
    const float4* __restrict__ qsorted, int* __restrict__ fb_count, int* __restrict__ fb_list,
    unsigned* __restrict__ fb_kth, int* __restrict__ box_count, int* __restrict__ box_list, int defer_limit,
    int uncertified_to_box, int cell_cap, int P1, int P2, int K, int N, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  using Cfg = LaneCfg<KC>;
  constexpr bool kUseQueue = Cfg::kUseQueue;
  constexpr int kQueueCap = Cfg::kQueueLds;
  constexpr int kSub = Cfg::kSub;
  constexpr int G = Cfg::kGroup;
  constexpr int kRunBits = RB, kRunMax = (1 << RB) - 1;
  static_assert(G % kSub == 0 && G <= kSortedPad, "group geometry");
  __shared__ double s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  // per-lane list of its non-empty runs, one word each: first record << 11 | length (<= 2047; a lane with a longer
  // run -- an over-full cell -- leaves its query to the fallback passes), then zeros.  Packed so that a wave needs
  // 2.5 KB instead of 5 KB of LDS: with the 8 KB queue that is 15 instead of 12 waves per CU.
  __shared__ unsigned s_rows[kLaneRows + 1][kGridWave];

  const int lane = threadIdx.x;
  const int total = chunk_prefix[N];
  if (kUseQueue) {
#pragma unroll
    for (int t = 0; t < kQueueCap; ++t) s_queue[t * kGridWave + lane] = TopKF64<KC>::empty();
  }
  unsigned* const rows = &s_rows[0][0];
  // XCD-aware item order: workgroup b runs on XCD b % 8 (round-robin dispatch), and the chunks are
  // sorted by (cloud, cell).  Each XCD walks its own contiguous eighth of the chunk list, so the
  // ~1000 chunks it has in flight belong to one or two clouds whose sorted records (1 MB at 65536
  // points) stay in that XCD's 4 MB L2, instead of every XCD touching every cloud in flight.
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

    // rigorous lower bound of every point outside the lane's cube (certification), known before the
    // walk: it also seeds the candidate threshold
    bool whole;
    const float lb = box_lower_bound<NORM>(g, edges + (int64_t)n * 3 * kEdgeStride, qx, qy, qz, X0, X1, Y0, Y1, Z0,
                                           Z1, whole);
    const unsigned thr0 = seed_threshold(lb, whole);

    // the lane's non-empty runs, own row first (near-first order tightens the thresholds early)
    // A lane whose cube is over-full (a run longer than the packed length field, or more than `defer_limit`
    // records in all) does not walk: its query goes to the box search over the refined cells (knn_grid_box.h).
    bool overlong = false;
    {
      int cnt = 0;  // rows written so far, as an element offset into s_rows
      int total = 0;
#pragma unroll
      for (int r = 0; r < kLaneRows; ++r) {
        constexpr int kDz[kLaneRows] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
        constexpr int kDy[kLaneRows] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
        const int z = cz + kDz[r], y = cy + kDy[r];
        if (active && z >= 0 && z < g.G[2] && y >= 0 && y < g.G[1]) {
          const int rowbase = (z * g.G[1] + y) * g.G[0];
          const int s = cstart[rowbase + X0], e = cstart[rowbase + X1 + 1];
          if (e > s) {
            overlong = overlong || e - s > kRunMax;
            total += e - s;
            rows[lane + cnt] = ((unsigned)s << kRunBits) | (unsigned)min(e - s, kRunMax);
            cnt += kGridWave;
          }
        }
      }
      overlong = overlong || total > defer_limit;
      if (overlong) cnt = 0;  // this lane does not walk
      // terminators: every slot from the lane's count on (a finished lane keeps reading 0)
#pragma unroll
      for (int r = 0; r <= kLaneRows; ++r) {
        if (r * kGridWave >= cnt) s_rows[r][lane] = 0u;
      }
    }
    TopKF64<KC> top;
    top.init();
    lane_walk<D, KC, NORM, kLaneRows, RB>((const char*)sp, rows, lane, s_queue, qx, qy, qz, thr0, top);

    const unsigned kth_bits = top.kth_bits(K);  // the K-th best, not the list's last slot
    const bool full = kth_bits < 0x7f800000u;
    const bool ok = !overlong && (whole || (full && __uint_as_float(kth_bits) < lb));
    if (active) {
      if (ok) {
        const int64_t row = (int64_t)n * P1 + qi;
        write_row_f64<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
      } else if (overlong || uncertified_to_box) {
        // (long lists have no quad pass: in big batches their uncertified queries take the box search -- one lane per
        // query, a box sized from the local density -- instead of the wave-per-query search: 1.5 ms of the 5.1 ms at
        // K=64, cfg2 size; see long_lists_to_box)
        const int pos = atomicAdd(box_count + n, 1);
        box_list[(int64_t)n * P1 + pos] = qi;
      } else {
        const int pos = atomicAdd(fb_count + n, 1);
        fb_list[(int64_t)n * P1 + pos] = qi;
        // The seeded threshold admitted only the m < KC candidates below lb, so the KC-th best itself is
        // unknown; hand the quad pass an ESTIMATE from the density they imply (m points inside radius
        // sqrt(lb) -> K points inside sqrt(lb) (K/m)^(1/3)), 30 % up.  It only picks the cube to search.
        int m = 0;
#pragma unroll
        for (int t = 0; t < KC; ++t) m += (unsigned)__double2hiint(top.key[t]) < 0x7f800000u ? 1 : 0;
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
// <= 7 contiguous runs with the lane search's own walk (lane_walk: packed run words, groups of four
// records behind one offset, stale-threshold queue, sorting-network merges), and two quad-permute exchange
// steps merge the four sorted lists, after which every lane of the quad holds the cube's KC best.  (Kernel time at cfg2 with 4 / 8 / 16
// lanes per query: 83 / 96 / 97-107 us in round 1, 63 / 67 / 69 us with this round's kernel (mirror DPP steps for
// the wider merges) -- the pass is throughput-bound, not bound by one wave's chain,
// so fewer, longer lanes win.)  The cube grows from 3x3x3 only past the faces that an ESTIMATE of the
// KC-th distance reaches.  The same rigorous face bound decides; what is still uncertified (far-away
// queries) goes to the expanding wave search.
// ---------------------------------------------------------------------------
#ifndef POINTOPS_QUAD_LANES
#define POINTOPS_QUAD_LANES 4
#endif
constexpr int kQuadLanes = POINTOPS_QUAD_LANES;  // lanes per query (4, or 8: a third merge step over row_half_mirror)
static_assert(kQuadLanes == 4 || kQuadLanes == 8, "lanes per query");
constexpr int kQuadRows = 25;  // every lane of the quad lists every row of the cube and takes every fourth record of it
constexpr int kQuadQueries = kGridWave / kQuadLanes;
constexpr int kQuadMaxRecords = 1024;  // per lane of the quad
__device__ constexpr signed char kQuadDy[32] = {0, 0, 0, -1, 1, -1, -1, 1, 1, 0, 0, -2, 2, -1, 1, -1, 1, -2, -2, 2, 2, -2, -2, 2, 2, 0, 0, 0, 0, 0, 0, 0};
__device__ constexpr signed char kQuadDz[32] = {0, -1, 1, 0, 0, -1, 1, -1, 1, -2, 2, 0, 0, -2, -2, 2, 2, -1, 1, -1, 1, -2, 2, -2, 2, 0, 0, 0, 0, 0, 0, 0};

// DPP controls: quad_perm [1,0,3,2] (lane ^ 1), quad_perm [2,3,0,1] (lane ^ 2)
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141;  // (row_half_mirror: lane i <-> 7 - i of its eight)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// merge the partner lane's ascending list into mine: both lanes end with the KC smallest of the union
template <int KC, int CTRL>
__device__ __forceinline__ void dpp_merge(TopKF64<KC>& top) {
  double o[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) o[t] = dpp_f64<CTRL>(top.key[t]);
  if constexpr (KC >= 2) {
#pragma unroll
    for (int t = 0; t < KC; ++t) top.key[t] = kmin(top.key[t], o[KC - 1 - t]);  // bitonic, holds the KC smallest
    bitonic_merge<KC>(top.key);
  } else {
    top.key[0] = kmin(top.key[0], o[0]);
  }
}

// (second launch bound: the 32-slot kernel needs 259 registers without it, three more than two waves per SIMD allow:
// 186 -> 138 us at the cfg2 size)
template <int D, int KC, int NORM, int RB>
__global__ __launch_bounds__(kGridWave, KC >= 32 ? 2 : 1) void knn_grid_quad_kernel(
    const float* __restrict__ p1, const GridCloud* __restrict__ clouds, const float* __restrict__ edges,
    const int* __restrict__ cell_start, const float4* __restrict__ sorted, const int* __restrict__ fb_count,
    const int* __restrict__ fb_list, const unsigned* __restrict__ fb_kth, int* __restrict__ fb3_count,
    int* __restrict__ fb3_list, int* __restrict__ box_count, int* __restrict__ box_list, int cell_cap, int P1, int P2,
    int K, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  constexpr bool kUseQueue = LaneCfg<KC>::kUseQueue;
  constexpr int kQueueCap = LaneCfg<KC>::kQueueLds;
  constexpr int kRunBits = RB, kRunMax = (1 << RB) - 1;
  __shared__ double s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  __shared__ unsigned s_rows[kQuadRows + 1][kGridWave];  // the lane's non-empty runs as packed words, zero-terminated
  unsigned* const rows = &s_rows[0][0];

  const int n = blockIdx.y;
  const int cnt = fb_count[n];
  const int lane = threadIdx.x;
  const int sub = lane & (kQuadLanes - 1);
  const GridCloud g = clouds[n];
  const int* __restrict__ cstart = cell_start + (int64_t)n * (cell_cap + 1);
  const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + kSortedPad);  // record P2: the cloud's NaN sentinel
  const float* __restrict__ ed = edges + (int64_t)n * 3 * kEdgeStride;
  double* const qbase = s_queue + lane;
  if (kUseQueue) {
#pragma unroll
    for (int t = 0; t < kQueueCap; ++t) qbase[t * kGridWave] = TopKF64<KC>::empty();
  }

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
    bool whole;
    const float lb = box_lower_bound<NORM>(g, ed, qx, qy, qz, X0, X1, Y0, Y1, Z0, Z1, whole);
    const unsigned thr0 = seed_threshold(lb, whole);

    // The quad's lanes SHARE every row of the cube: lane `sub` takes the records sub, sub + 4, sub + 8, ... of each of
    // the <= 25 contiguous runs (nearest rows first), so the four gathers of a quad fall into the same 64-byte segment and
    // the quad's lanes finish together.  Until round 3 the rows were DEALT to the lanes: the 64 lanes of a wave then
    // gathered from 64 different lines per instruction, and the vector L1 serves about one such lane-request per clock
    // and CU -- 87 000 cycles of walk per wave, whatever the cursor logic, the selection or the lanes per query cost
    // (profiles/r03_lane_experiments.md, r03_quad_stamps.txt).  The walk and the selection are the lane search's
    // (lane_walk with a record stride of four: packed run words, groups behind one offset, masked tails, stale-threshold
    // queue).
    bool overlong = false;
    {
      int cnt = 0;  // rows written so far, as an element offset into s_rows
      int total = 0;  // records of the whole cube
#pragma unroll
      for (int rr = 0; rr < kQuadRows; ++rr) {
        const int z = cz + kQuadDz[rr], y = cy + kQuadDy[rr];
        if (active && z >= Z0 && z <= Z1 && y >= Y0 && y <= Y1) {
          const int rowbase = (z * g.G[1] + y) * g.G[0];
          const int s = cstart[rowbase + X0], e = cstart[rowbase + X1 + 1];
          overlong = overlong || e - s > kRunMax;
          total += e - s;
          if (e - s > sub) {  // (the lane's share of the row: records s + sub, s + sub + 4, ... below e)
            rows[lane + cnt] = ((unsigned)(s + sub) << kRunBits) | (unsigned)min(e - s - sub, kRunMax);
            cnt += kGridWave;
          }
        }
      }
      // a quad whose cube holds an over-full cell (or a run longer than the packed length field) does not walk it:
      // the box search (which sees the refined cell's inside and splits runs) takes the query
      overlong = overlong || total > kQuadLanes * kQuadMaxRecords;  // (the same for the four lanes: no exchange needed)
      if (overlong) cnt = 0;
#pragma unroll
      for (int r = 0; r <= kQuadRows; ++r) {
        if (r * kGridWave >= cnt) s_rows[r][lane] = 0u;
      }
    }
    const bool big = overlong;
    TopKF64<KC> top;
    top.init();
    lane_walk<D, KC, NORM, kQuadRows, RB, kQuadLanes>((const char*)sp, rows, lane, s_queue, qx, qy, qz, thr0, top);

    // the group's sorted lists -> one, held by every lane of the group
    dpp_merge<KC, kDppXor1>(top);
    dpp_merge<KC, kDppXor2>(top);
    if constexpr (kQuadLanes == 8) dpp_merge<KC, kDppHalfMirror>(top);  // the other quad of the group

    const unsigned kth_bits = top.kth_bits(K);  // the K-th best, not the list's last slot
    const bool full = kth_bits < 0x7f800000u;
    const bool ok = !big && (whole || (full && __uint_as_float(kth_bits) < lb));
    if (active && sub == 0) {
      if (ok) {
        const int64_t row = (int64_t)n * P1 + qi;
        write_row_f64<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
      } else if (big) {  // an over-full cell in the cube: the box search knows its inside
        const int pos = atomicAdd(box_count + n, 1);
        box_list[(int64_t)n * P1 + pos] = qi;
      } else {
        const int pos = atomicAdd(fb3_count + n, 1);
        fb3_list[(int64_t)n * P1 + pos] = qi;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// pass 6: wave-per-query EXPANDING search for the queries the lane / quad passes could not certify
// (typically < 0.1 % of a cloud).  The wave's 64 lanes split the candidate records of
// the cube of cells [c - r, c + r]^3 around the query's cell (coalesced 16-byte
// loads along each row's contiguous run), keep a private lexicographic top-K each,
// and K rounds of a wave-wide 64-bit min extract the K global minima.  The same
// rigorous face bound decides; on failure r doubles, until the cube is the whole
// grid (always exact) or more than kWaveRegionCap records were scanned, in which
// case the query goes to the whole-cloud lane-per-query scan (far-away queries).
// ---------------------------------------------------------------------------
constexpr int kWaveKernelBlock = 256;
constexpr int kWaveKernelWgsPerCloud = 256;  // x 4 waves: the pass is latency-bound per query, so one query per wave
// (Waves own a fixed stride of the list.  Drawing the next query from a per-cloud counter was measured and dropped:
// 1 024 waves per cloud on one address -- 17 -> 243 us on the slab cloud, whose 10 000 queries here are cheap.)
                                             // wherever a cloud sends it up to ~1000 (64: 2.1 -> ms at 761 queries per cloud)
constexpr int kWaveRegionCap = 1 << 22;  // records: in effect the wave search always finishes (a cube that holds an
                                         // over-full cell is still 64 lanes on one coalesced stream; the whole-cloud
                                         // scan it used to give up to after 16 384 records runs ONE lane per query)
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
  const float4* __restrict__ sp = sorted + (int64_t)n * (P2 + kSortedPad);
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
      bool whole;
      const float lb = box_lower_bound<NORM>(g, ed, qx, qy, qz, X0, X1, Y0, Y1, Z0, Z1, whole);
      TopKF64<KC> top;
      top.init();
      auto consider = [&](const float4 c) {
        const float d = point_dist<D, NORM>(qx, qy, qz, c);
        if (__float_as_uint(d) <= top.worst_bits()) top.insert(TopKF64<KC>::make(d, __float_as_int(c.w)));
      };
      int scanned = 0;
      bool giveup = false;
      const int ny = Y1 - Y0 + 1, nrows = ny * (Z1 - Z0 + 1);
      // Row by row the scan is latency-bound (two dependent scalar loads per row, then one load per lane).
      // Instead the lanes fetch the bounds of up to kWaveRows rows at once, a wave scan turns them into a flat
      // record stream, and every lane owns the records lane, lane + 64, ... with eight loads in flight; bigger
      // cubes (r >= 8) repeat that per block of rows.
      int* __restrict__ rs = s_rowsrc[wslot];
      int* __restrict__ ro = s_rowoff[wslot];
      for (int rb = 0; rb < nrows && !giveup; rb += kWaveRows) {
        const int nblk = min(kWaveRows, nrows - rb);
        int T = 0;
        for (int r0 = 0; r0 < nblk; r0 += kWave) {
          const int rr = r0 + lane;
          int len_r = 0, src = 0;
          if (rr < nblk) {
            const int z = Z0 + (rb + rr) / ny, y = Y0 + (rb + rr) % ny;
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
          if (rr < nblk) {
            rs[rr] = src;
            ro[rr + 1] = T + inc;
          }
          T += __shfl(inc, kWave - 1, kWave);
        }
        if (lane == 0) ro[0] = 0;
        scanned += T;
        if (!whole && scanned > kWaveRegionCap) {
          giveup = true;
          break;
        }
        int rrow = 0;
        constexpr int kInFlight = 8;  // loads per lane and step
        for (int t0 = lane; t0 < T; t0 += kInFlight * kWave) {
          float4 c[kInFlight];
#pragma unroll
          for (int u = 0; u < kInFlight; ++u) {
            const int t = t0 + u * kWave;
            const float qnan = __uint_as_float(0x7fc00000u);
            c[u] = make_float4(qnan, qnan, qnan, 0.f);
            if (t < T) {
              while (ro[rrow + 1] <= t) ++rrow;
              c[u] = sp[rs[rrow] + (t - ro[rrow])];
            }
          }
#pragma unroll
          for (int u = 0; u < kInFlight; ++u) consider(c[u]);
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
      double mine = TopKF64<KC>::empty(), kth = TopKF64<KC>::empty();
      for (int k = 0; k < kvalid; ++k) {
        double m = top.key[0];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
          const int hi = __shfl_xor(__double2hiint(m), off, kWave);
          const int lo = __shfl_xor(__double2loint(m), off, kWave);
          m = kmin(m, __hiloint2double(hi, lo));
        }
        if (lane == k) mine = m;
        kth = m;
        if (__double_as_longlong(top.key[0]) == __double_as_longlong(m)) {
#pragma unroll
          for (int s2 = 0; s2 + 1 < KC; ++s2) top.key[s2] = top.key[s2 + 1];
          top.key[KC - 1] = TopKF64<KC>::empty();
        }
      }
      const unsigned kth_bits = (unsigned)__double2hiint(kth);
      const bool full = kvalid == K && kth_bits < 0x7f800000u;
      if (whole || (full && __uint_as_float(kth_bits) < lb)) {
        if (lane < K) {
          const bool ok = lane < kvalid;
          idxs[row * K + lane] = ok ? (int64_t)__double2loint(mine) : 0;
          dists[row * K + lane] = ok ? __int_as_float(__double2hiint(mine)) : 0.0f;
        }
        done = true;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// launches of one (D, NORM): lane search, then the exact fallbacks for what it could not certify
// ---------------------------------------------------------------------------
template <int D, int KC, int NORM, int RB>
static void launch_grid_box(const KnnArgs& a, const GridWs& ws, bool quad);  // knn_grid_box.h

// 64-slot lists have no quad pass: their uncertified queries take the box search when the batch sends it enough of them
// to fill the chip, else the wave search directly (K=64, ms with / without: 32 x 65536 queries 3.09 / 3.91, 64 x 16384
// 2.69 / 2.66, 8 x 65536 1.16 / 1.10, 8 x 32768 0.87 / 0.69, 2 x 65536 0.74 / 0.38); debug knob grid_long_box=0/1 forces
// the choice
static inline bool long_lists_to_box(const KnnArgs& a) {
  const long k = debug_knob("grid_long_box", -1);
  return k >= 0 ? k != 0 : a.N * (int64_t)a.P1 >= (3LL << 19);
}

template <int D, int KC, int NORM, int RB>
static void launch_grid_passes(const KnnArgs& a, const GridWs& ws, bool quad) {
  // One wave64 per workgroup and ONE chunk of 64 queries per workgroup where the launch allows it (a multiple of 8: the
  // XCD-aware order): the hardware's workgroup dispatcher then balances the chunks (cfg2, ms per step: 3 840 resident
  // workgroups looping over their share 0.870, 8 192 0.757, 32 768 = one per chunk 0.731).
  int64_t chunks = (int64_t)a.N * ceil_div(a.P1, kGridWave);
  chunks = (chunks + 7) / 8 * 8;
  const int wgs = (int)(chunks < 2048 ? 2048 : (chunks > (1 << 20) ? (1 << 20) : chunks));
  hipLaunchKernelGGL((knn_grid_lane_kernel<D, KC, NORM, RB>), dim3((unsigned)wgs), dim3(kGridWave), 0, a.stream, a.p1,
                     (const GridCloud*)ws.cloud, (const int*)ws.chunk_prefix, (const float*)ws.edges,
                     (const int*)ws.cell_start, (const float4*)ws.sorted, (const float4*)ws.qsorted, ws.fb_count,
                     ws.fb_list, ws.fb_kth, ws.box_count, ws.box_list, kDeferFactor * refine_threshold(ws.c_target),
                     (KC > 32 && long_lists_to_box(a)) ? 1 : 0, ws.cell_cap, a.P1, a.P2, a.K, (int)a.N, a.idxs, a.dists);
  if constexpr (KC <= 32) if (quad) {
    int64_t wx = a.P1 / (32 * kQuadQueries);  // a few % of a cloud arrive here
    wx = wx < 8 ? 8 : wx > 4096 ? 4096 : wx;
    hipLaunchKernelGGL((knn_grid_quad_kernel<D, KC, NORM, RB>), dim3((unsigned)wx, (unsigned)a.N), dim3(kGridWave), 0,
                       a.stream, a.p1, (const GridCloud*)ws.cloud, (const float*)ws.edges, (const int*)ws.cell_start,
                       (const float4*)ws.sorted, (const int*)ws.fb_count, (const int*)ws.fb_list,
                       (const unsigned*)ws.fb_kth, ws.fb3_count, ws.fb3_list, ws.box_count, ws.box_list, ws.cell_cap, a.P1,
                       a.P2, a.K, a.idxs, a.dists);
  }
  launch_grid_box<D, KC, NORM, RB>(a, ws, quad);  // over-full neighbourhoods (appends what it cannot certify)
  hipLaunchKernelGGL((knn_grid_wave_kernel<D, KC, NORM>), dim3(kWaveKernelWgsPerCloud, (unsigned)a.N),
                     dim3(kWaveKernelBlock), 0, a.stream, a.p1, (const GridCloud*)ws.cloud, (const float*)ws.edges,
                     (const int*)ws.cell_start, (const float4*)ws.sorted,
                     (const int*)(quad ? ws.fb3_count : ws.fb_count), (const int*)(quad ? ws.fb3_list : ws.fb_list),
                     ws.fb2_count, ws.fb2_list, ws.cell_cap, a.P1, a.P2, a.K, quad ? 4 : 2, a.idxs, a.dists);
}

template <int D, int NORM, int RB>
void grid_search_dispatch(const KnnArgs& a, const GridWs& ws, int kc, bool quad) {
  switch (kc) {
    case 1: launch_grid_passes<D, 1, NORM, RB>(a, ws, quad); break;
    case 2: launch_grid_passes<D, 2, NORM, RB>(a, ws, quad); break;
    case 4: launch_grid_passes<D, 4, NORM, RB>(a, ws, quad); break;
    case 8: launch_grid_passes<D, 8, NORM, RB>(a, ws, quad); break;
    case 16: launch_grid_passes<D, 16, NORM, RB>(a, ws, quad); break;
    case 32: launch_grid_passes<D, 32, NORM, RB>(a, ws, quad); break;
    default: launch_grid_passes<D, 64, NORM, RB>(a, ws, false); break;
  }
}

// one translation unit per point dimension and run-word geometry (knn_grid_d1/2/3.hip: clouds of up to
// kGridMaxPoints points; knn_grid_d1/2/3w.hip: bigger ones)
void grid_search_d1(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);
void grid_search_d2(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);
void grid_search_d3(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);
void grid_search_d1w(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);
void grid_search_d2w(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);
void grid_search_d3w(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad);

}  // namespace pointops
