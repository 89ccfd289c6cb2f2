// knn_grid_box.h -- BOX SEARCH: the exact KNN search of the queries that left the lane search because their
// neighbourhood is over-full (their own cell is refined, or their 3x3x3 cube holds many times the expected
// candidates); gfx950.  See grid_refine.hip for the refined cells and knn_grid.hip for the passes around this one.
//
// A query picks a radius r from the density AROUND IT (its sub-cell if its cell is refined, else its cell: r =
// the radius expected to hold ~2.5 K points), forms the box [q - r, q + r] in fp32 and visits EVERY (sub-)cell that
// can hold a point of the box: coarse cells cell_d(a_d) .. cell_d(b_d) per dimension, and inside a refined cell the
// sub-cells sub_d(a_d) .. sub_d(b_d) -- both index functions are monotone and total, so the enumeration is complete
// by construction and needs no edge tables.  An unvisited point u therefore has u_d < a_d or u_d > b_d in some
// dimension, and since fp32 subtraction is monotone its COMPUTED |q_d - u_d| is >= fl(q_d - a_d) resp.
// fl(b_d - q_d): the certification bound is the minimum of those six face terms (squared for the L2 norm) -- the
// same argument as the cube faces of the lane search.  kth < bound (strict) => the list is exact; otherwise the
// radius grows (x 1.6, up to five attempts), then the query goes to the expanding wave search.
// A refined cell that the box covers in many sub-rows is taken as ONE run (visiting more points is always allowed).
// The walk and the selection are the lane search's (lane_walk).
#pragma once
#include "knn_grid_search.h"

namespace pointops {

constexpr int kBoxRows = 47;      // run slots per lane (+ 1 terminator): 48 words x 64 lanes = 12 KB (23 slots: more waves, but
                                  // twice the overflows to the wave search on the u^4 cloud: 6.4 -> 8.4 ms)
constexpr int kBoxAttempts = 5;      // radii r0, 1.6 r0, ... ; r0 = the density estimate inside a refined cell, 0.35 of it next
                                     // to one (the estimate of a sparse cell is far beyond the near face of a dense
                                     // neighbour: small steps find the radius where the box clips a corner of it)
constexpr int kBoxSubRowsMax = 36;  // sub-rows of one refined cell a box may enumerate before taking the cell whole (a query
                                    // at a corner of a cluster needs its third radius: 5-6 sub-cells per dimension)
constexpr int kBoxMaxRecords = 3072;  // a lane never walks more than this: a bigger candidate set (a box that covers a
                                      // dense cell from outside) belongs to the wave-per-query search, whose 64 lanes
                                      // share it -- one lane walking 32 k records holds its whole wave for milliseconds

template <int D, int KC, int NORM, int RB>
__global__ __launch_bounds__(kGridWave) void knn_grid_box_kernel(
    const float* __restrict__ p1, GridWs ws, int P1, int P2, int K, int* __restrict__ out_count,
    int* __restrict__ out_list, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  constexpr bool kUseQueue = LaneCfg<KC>::kUseQueue;
  constexpr int kQueueCap = LaneCfg<KC>::kQueueLds;
  constexpr int kRunBits = RB, kRunMax = (1 << RB) - 1;
  __shared__ double s_queue[kUseQueue ? kQueueCap * kGridWave : 1];
  __shared__ unsigned s_rows[kBoxRows + 1][kGridWave];
  const int n = blockIdx.y;
  const int cnt = ws.box_count[n];
  if (cnt == 0) return;
  const int lane = threadIdx.x;
  const GridCloud g = ws.cloud[n];
  const int* __restrict__ cstart = ws.cell_start + (int64_t)n * (ws.cell_cap + 1);
  const int* __restrict__ rref = ws.refine_ref + (int64_t)n * ws.cell_cap;
  const RefinedCell* __restrict__ rdesc = ws.rdesc + (int64_t)n * ws.rdesc_cap;
  const int* __restrict__ pool = ws.pool + (int64_t)n * ws.pool_cap;
  const float4* __restrict__ sp = ws.sorted + (int64_t)n * (P2 + kSortedPad);
  unsigned* const rows = &s_rows[0][0];
  if (kUseQueue) {
#pragma unroll
    for (int t = 0; t < kQueueCap; ++t) s_queue[t * kGridWave + lane] = TopKF64<KC>::empty();
  }
  const float h = 1.0f / g.inv_h;

  for (int base = blockIdx.x * kGridWave; base < cnt; base += gridDim.x * kGridWave) {
    const bool active = base + lane < cnt;
    const int qi = active ? ws.box_list[(int64_t)n * P1 + base + lane] : 0;
    float q[3] = {0.0f, 0.0f, 0.0f};
    if (active) load_point3<D>(p1 + ((int64_t)n * P1 + qi) * D, q[0], q[1], q[2]);
    int cc[3];
    point_cells(g, q[0], q[1], q[2], cc[0], cc[1], cc[2]);
    // radius from the local density: points expected inside the ball ~ 2.5 K
    float r;
    {
      const int cell = (cc[2] * g.G[1] + cc[1]) * g.G[0] + cc[0];
      const int ref = active ? rref[cell] : -1;
      float vol = 1.0f, cntf;
      if (ref >= 0) {
        const RefinedCell d = rdesc[ref];
        int sc[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < D; ++k) {
          sc[k] = sub_of(q[k], d.lo[k], d.scale[k], d.s);
          vol *= d.scale[k] > 0.0f ? 1.0f / d.scale[k] : h;
        }
        const int sb = (sc[2] * d.s + sc[1]) * d.s + sc[0];
        const int m = pool[d.pool_off + sb + 1] - pool[d.pool_off + sb];
        cntf = (float)(m > 0 ? m : 1);
      } else {
        const int m = active ? cstart[cell + 1] - cstart[cell] : 1;
        cntf = (float)(m > 0 ? m : 1);
#pragma unroll
        for (int k = 0; k < D; ++k) vol *= h;
      }
      const float ball = 2.5f * (float)K * vol / cntf;  // volume expected to hold 2.5 K points
      r = D == 3 ? cbrtf(ball * 0.2387f) : D == 2 ? sqrtf(ball * 0.3183f) : ball * 0.5f;
      if (!(r > 0.0f) || !(r <= FLT_MAX)) r = h;
      if (ref < 0) r *= 0.35f;
    }

    bool done = !active;  // nothing (more) to do for this lane
    bool certified = false;
    TopKF64<KC> top;
    top.init();
    for (int attempt = 0; attempt < kBoxAttempts && __any(!done); ++attempt) {  // wave-uniform
      // ---- the lane's run list for this radius (a finished lane lists nothing)
      float lb = __builtin_inff();
      int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      float a[3] = {0.0f, 0.0f, 0.0f}, b[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < D; ++k) {
        a[k] = q[k] - r;
        b[k] = q[k] + r;
        lo[k] = cell_of(a[k], g.lo[k], g.inv_h, g.G[k]);
        hi[k] = cell_of(b[k], g.lo[k], g.inv_h, g.G[k]);
        lb = fminf(lb, fminf(face_bound<NORM>(q[k] - a[k]), face_bound<NORM>(b[k] - q[k])));
      }
      int nrun = 0;  // element offset (multiples of 64) of the next free slot
      int total = 0;
      bool overflow = false;
      auto push_run = [&](int s, int e) {
        total += e - s;
        if (total > kBoxMaxRecords) overflow = true;
        while (e > s && !overflow) {  // runs longer than the packed length field are split
          const int len = min(e - s, kRunMax);
          if (nrun >= kBoxRows * kGridWave) {
            overflow = true;
            break;
          }
          rows[lane + nrun] = ((unsigned)s << kRunBits) | (unsigned)len;
          nrun += kGridWave;
          s += len;
        }
      };
      if (!done) {
        for (int z = lo[2]; z <= hi[2] && !overflow; ++z) {
          for (int y = lo[1]; y <= hi[1] && !overflow; ++y) {
            const int rowbase = (z * g.G[1] + y) * g.G[0];
            for (int x = lo[0]; x <= hi[0] && !overflow; ++x) {
              const int cell = rowbase + x;
              const int cs = cstart[cell], ce = cstart[cell + 1];
              if (ce <= cs) continue;
              const int ref = rref[cell];
              if (ref < 0) {
                push_run(cs, ce);
                continue;
              }
              const RefinedCell d = rdesc[ref];
              int s0[3] = {0, 0, 0}, s1[3] = {0, 0, 0};
#pragma unroll
              for (int k = 0; k < D; ++k) {
                s0[k] = sub_of(a[k], d.lo[k], d.scale[k], d.s);
                s1[k] = sub_of(b[k], d.lo[k], d.scale[k], d.s);
              }
              const int nsr = (s1[1] - s0[1] + 1) * (s1[2] - s0[2] + 1);
              if (nsr > kBoxSubRowsMax) {
                push_run(cs, ce);  // most of the cell: take it whole
                continue;
              }
              const int* __restrict__ tab = pool + d.pool_off;
              for (int sz = s0[2]; sz <= s1[2]; ++sz) {
                for (int sy = s0[1]; sy <= s1[1]; ++sy) {
                  const int rb = (sz * d.s + sy) * d.s;
                  push_run(cs + tab[rb + s0[0]], cs + tab[rb + s1[0] + 1]);
                }
              }
            }
          }
        }
      }
      if (overflow || done) nrun = 0;
#pragma unroll
      for (int t = 0; t <= kBoxRows; ++t) {
        if (t * kGridWave >= nrun) s_rows[t][lane] = 0u;
      }
      // ---- walk + select (list restarted for every attempt), certify against the box faces
      top.init();
      const unsigned thr0 = seed_threshold(lb, false);
      lane_walk<D, KC, NORM, kBoxRows, RB>((const char*)sp, rows, lane, s_queue, q[0], q[1], q[2], thr0, top);
      if (!done && !overflow) {
        const unsigned kth_bits = top.kth_bits(K);
        if (kth_bits < 0x7f800000u && __uint_as_float(kth_bits) < lb) {
          const int64_t row = (int64_t)n * P1 + qi;
          write_row_f64<KC>(top, K, g.len2, idxs + row * K, dists + row * K);
          done = certified = true;
        }
      }
      if (overflow) done = true;  // too many runs for the list: leave it to the fallback below
      r *= 1.6f;
    }
    if (active && !certified) {  // expanding wave search (knn_grid_wave_kernel), through the list it reads
      const int pos = atomicAdd(out_count + n, 1);
      out_list[(int64_t)n * P1 + pos] = qi;
    }
  }
}

template <int D, int KC, int NORM, int RB>
static void launch_grid_box(const KnnArgs& a, const GridWs& ws, bool quad) {
  // one 64-query chunk of the box list per workgroup where the grid allows it: the lanes' work differs by orders of
  // magnitude on the clouds that need this pass (u^4 cloud 4.63 -> 4.10 ms, half_in_cluster 3.20 -> 2.53 ms against one
  // workgroup per 8 chunks; the 32 768 workgroups that find an empty list on a uniform cloud cost < 10 us).  Measured and
  // dropped, both with a pool of 256 entries per workgroup: lanes that draw a new query as soon as theirs is certified
  // (every round then costs the longest walk of a lane on its third radius: 4.10 -> 4.82 ms) and round-by-round
  // processing with the uncertified queries compacted into a pending list per radius (4.10 -> 5.10 ms: fewer, longer
  // workgroups and 4 KB more LDS cost more than the idle lanes of the late radii)
  int64_t wx = a.P1 / kGridWave;
  wx = wx < 8 ? 8 : wx > 1024 ? 1024 : wx;
  hipLaunchKernelGGL((knn_grid_box_kernel<D, KC, NORM, RB>), dim3((unsigned)wx, (unsigned)a.N), dim3(kGridWave), 0, a.stream,
                     a.p1, ws, a.P1, a.P2, a.K, quad ? ws.fb3_count : ws.fb_count, quad ? ws.fb3_list : ws.fb_list, a.idxs,
                     a.dists);
}

}  // namespace pointops
