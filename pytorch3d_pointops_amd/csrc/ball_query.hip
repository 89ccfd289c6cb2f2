// ball_query.hip -- first-K-within-radius neighbour search for gfx950.
//
// Replaces BallQuery (reference: csrc/ball_query/ball_query.h:62-93) with the CPU
// path's semantics (csrc/ball_query/ball_query_cpu.cpp:12-54): for each query the
// first K points of p2 in INDEX order with dist2 < radius*radius (strict, fp32
// product), idx padded with -1 and dists with 0 (also for rows >= lengths1[n]).
//
// One lane per query; p2 is streamed through the scalar path exactly as in
// knn.hip (wave-uniform address -> s_load -> SGPR operands).  A wave stops
// scanning as soon as all of its 64 queries are full (wave-uniform early exit via
// ballot), which on dense clouds is after ~1 % of p2: the
// op is then bound by writing the (N,P1,K) outputs.
//
// SPARSE balls (few points inside the radius: the scan would walk most of the cloud for every
// query) go through the cell grid of knn_grid.hip instead when the caller provides the
// workspace: `ball_grid_lane_kernel` answers every query whose 3x3x3 cell cube provably contains
// its ball; per cloud the choice between grid and scan is made ON THE DEVICE from the bounding box
// (expected points per ball), and this kernel then scans only what is left: whole clouds whose
// flag is 0, and the listed (uncertified) queries of the others.
#include "common.h"
#include "debug.h"
#include "knn_grid.h"

namespace pointops {

constexpr int kBqBlock = 256;
constexpr int kBqTile = 8;  // points per scalar-load group of the storage-order scan

template <int DT>
__global__ __launch_bounds__(kBqBlock) void ball_query_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2, int P1, int P2,
    int Drt, int K, float radius2, int tiles_per_cloud, const int* __restrict__ grid_flag,
    const int* __restrict__ qcount, const int* __restrict__ qlist, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  int i = tile * kBqBlock + threadIdx.x;
  bool in_range = i < P1;
  const bool listed = grid_flag != nullptr && grid_flag[n];  // (workgroup-uniform)
  if (listed) {  // this cloud's queries come from a list: uncertified by the grid, or coarse-cell order
    if (tile * kBqBlock >= qcount[n]) return;
    in_range = i < qcount[n];
    i = in_range ? qlist[(int64_t)n * P1 + i] : 0;
  }
  const int len1 = (int)lengths1[n];
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  const int64_t row = (int64_t)n * P1 + (in_range ? i : 0);
  int64_t* __restrict__ orow_i = idxs + row * K;
  float* __restrict__ orow_d = dists + row * K;
  const bool active = in_range && i < len1;
  int count = 0;
  const float* __restrict__ q = p2 + (int64_t)n * P2 * D;

  if constexpr (DT > 0) {
    float a[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) a[d] = active ? p1[row * DT + d] : 0.0f;
    auto dist_to = [&](const float* __restrict__ b) {
      float acc;
      {
        const float diff = a[0] - b[0];
        acc = diff * diff;
      }
#pragma unroll
      for (int d = 1; d < DT; ++d) {
        const float diff = a[d] - b[d];
        acc = acc + diff * diff;
      }
      return acc;
    };
    // Hits are rare per lane (a few %) but SOME lane of the wave hits on ~90 % of the candidates, so
    // a per-candidate `if (hit) store` runs its ~15-instruction store block with one or two active
    // lanes almost every iteration (66 cycles per wave-point measured vs ~26 of arithmetic).  Instead
    // a group of 32 candidates only accumulates a 32-bit hit mask per lane (branch-free), and one
    // expansion loop per group then writes the hits in bit (= index) order with every lane that still
    // has a bit active; its distance is recomputed from the same operands (bit-identical).
    const int room = active ? K : 0;
    if (listed) {
      const int lane = threadIdx.x & (kWave - 1);
      // Bounding box of the wave's active queries (exact min / max).  Its queries come in coarse-cell order
      // (ball_grid.hip) or through the grid's fallback list, so the box is small next to the cloud.
      float blo[DT], bhi[DT];
  #pragma unroll
      for (int d = 0; d < DT; ++d) {
        float mn = active ? a[d] : __builtin_inff(), mx = active ? a[d] : -__builtin_inff();
  #pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
          mn = fminf(mn, __shfl_xor(mn, off, kWave));
          mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
        }
        blo[d] = mn;
        bhi[d] = mx;
      }
      // The wave consumes p2 in TILES of 64 points, one per lane:
      //  1. every lane tests ITS point against the wave's box: LB = the scan's own distance expression on the
      //     per-dimension gaps fl(lo - c) / fl(c - hi) / 0.  fp32 subtraction, squaring and the sums are monotone,
      //     so every query's COMPUTED distance to the point is >= LB; LB >= radius2 proves no lane can hit;
      //  2. the surviving points (a ballot; ~10-15 % of a uniform cube at r = 0.2) are broadcast one by one
      //     (v_readlane) and tested by all lanes: a hit sets bit b of the lane's 64-bit tile mask;
      //  3. the lanes write their hits in bit (= index) order; the distance is recomputed from the same operands.
      // (Round 1 tested every point in every lane through the scalar path: ~35 cycles per wave-point for what is
      // now one vector instruction per 64 points plus ~50 cycles per surviving point.)
      int j0 = 0;
      float cn[DT];  // the NEXT tile's point of this lane, loaded one tile ahead
  #pragma unroll
      for (int d = 0; d < DT; ++d) cn[d] = lane < len2 ? q[(int64_t)lane * DT + d] : 0.0f;
      while (j0 < len2 && __any(count < room)) {
        const int jc = j0 + lane;
        float c[DT];
  #pragma unroll
        for (int d = 0; d < DT; ++d) c[d] = cn[d];
        {
          const int jn = jc + kWave;
  #pragma unroll
          for (int d = 0; d < DT; ++d) cn[d] = jn < len2 ? q[(int64_t)jn * DT + d] : 0.0f;
        }
        float lb;
        {
          float g0 = fmaxf(fmaxf(blo[0] - c[0], c[0] - bhi[0]), 0.0f);
          lb = g0 * g0;
  #pragma unroll
          for (int d = 1; d < DT; ++d) {
            const float gd = fmaxf(fmaxf(blo[d] - c[d], c[d] - bhi[d]), 0.0f);
            lb = lb + gd * gd;
          }
        }
        unsigned long long cand = __ballot(jc < len2 && lb < radius2);
        unsigned mlo = 0u, mhi = 0u;  // the lane's hits among the tile's points
        if (__popcll(cand) >= 48 && j0 + kWave <= len2) {
          // Most of the tile survives (queries in storage order, or a box as large as the cloud): the broadcast
          // loop would cost more than testing all 64 points through the scalar path (wave-uniform addresses ->
          // s_load, SGPR operands).  `mask = 2 mask + hit` is ONE v_addc_co_u32 whose carry-in is the lane's bit
          // of the compare; the first point ends up in the highest bit, hence the bit reversal.
          unsigned r0 = 0u, r1 = 0u;
          auto push_hit = [&](unsigned& mask, float acc) __attribute__((always_inline)) {
            asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "s"(radius2), "v"(acc) : "vcc");
          };
          for (int jj = 0; jj < 32; jj += 8) {
            float t[8 * DT];
  #pragma unroll
            for (int u = 0; u < 8 * DT; ++u) t[u] = q[(int64_t)(j0 + jj) * DT + u];
  #pragma unroll
            for (int u = 0; u < 8; ++u) push_hit(r0, dist_to(t + u * DT));
          }
          for (int jj = 32; jj < 64; jj += 8) {
            float t[8 * DT];
  #pragma unroll
            for (int u = 0; u < 8 * DT; ++u) t[u] = q[(int64_t)(j0 + jj) * DT + u];
  #pragma unroll
            for (int u = 0; u < 8; ++u) push_hit(r1, dist_to(t + u * DT));
          }
          mlo = __brev(r0);
          mhi = __brev(r1);
          cand = 0ull;
        }
        while (cand != 0ull) {  // wave-uniform
          const int b = __builtin_ctzll(cand);
          cand &= cand - 1ull;
          float pb[DT];
  #pragma unroll
          for (int d = 0; d < DT; ++d) pb[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[d]), b));
          const bool hit = dist_to(pb) < radius2;
          const unsigned bit = 1u << (b & 31);
          if (b < 32) mlo |= hit ? bit : 0u;
          else mhi |= hit ? bit : 0u;
        }
        if (count >= room) mlo = mhi = 0u;
        while (__any((mlo | mhi) != 0u)) {  // wave-uniform loop: every lane takes part in the shuffles
          const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
          const bool has = m != 0ull;
          const int b = has ? __builtin_ctzll(m) : 0;
          float ph[DT];
  #pragma unroll
          for (int d = 0; d < DT; ++d) ph[d] = __shfl(c[d], b, kWave);  // the tile still sits in the lanes
          if (has) {
            if (b < 32) mlo &= mlo - 1u;
            else mhi &= mhi - 1u;
            orow_i[count] = j0 + b;
            orow_d[count] = dist_to(ph);
            ++count;
            if (count >= room) mlo = mhi = 0u;
          }
        }
        j0 += kWave;
      }
    } else {
      // Queries in storage order (small calls without workspace): the wave's box is the cloud, nothing is
      // rejected, so every point goes through the scalar path (wave-uniform address -> s_load -> SGPR operands),
      // 32 points per round, hits collected as a per-lane bit mask and written by one expansion loop.
      int j = 0;
      // hit bit of one candidate into the lane's mask: `mask = 2 mask + (acc < radius2)` is ONE v_addc_co_u32 whose
      // carry-in is the lane's bit of the compare result (cmp + select + shift + or: four instructions otherwise).
      // The first candidate of a group therefore ends up in the HIGHEST used bit.
      auto push_hit = [&](unsigned& mask, float acc) __attribute__((always_inline)) {
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "s"(radius2), "v"(acc) : "vcc");
      };
      while (j < len2 && __any(count < room)) {
        unsigned mask = 0u;
        const int jg = j;
        const int g_end = (j + 32 < len2) ? j + 32 : len2;
        for (; j + kBqTile <= g_end; j += kBqTile) {
          float t[kBqTile * DT];
  #pragma unroll
          for (int u = 0; u < kBqTile * DT; ++u) t[u] = q[(int64_t)j * DT + u];  // wave-uniform -> s_load
  #pragma unroll
          for (int jj = 0; jj < kBqTile; ++jj) push_hit(mask, dist_to(t + jj * DT));
        }
        for (; j < g_end; ++j) {
          float t[DT];
  #pragma unroll
          for (int u = 0; u < DT; ++u) t[u] = q[(int64_t)j * DT + u];
          push_hit(mask, dist_to(t));
        }
        const int top = g_end - jg - 1;  // bit of the group's first candidate
        if (count >= room) mask = 0u;
        while (__any(mask != 0u)) {
          if (mask != 0u) {
            const int b = 31 - __builtin_clz(mask);  // highest set bit = lowest index
            mask &= ~(1u << b);
            const int jh = jg + (top - b);
            float pb[DT];
  #pragma unroll
            for (int d = 0; d < DT; ++d) pb[d] = q[(int64_t)jh * DT + d];  // per-lane gather (L2-resident)
            orow_i[count] = jh;
            orow_d[count] = dist_to(pb);
            ++count;
            if (count >= room) mask = 0u;
          }
        }
      }
    }
  } else {
    if (active) {
      const float* __restrict__ a = p1 + row * D;
      for (int j = 0; j < len2 && count < K; ++j) {
        const float* __restrict__ b = q + (int64_t)j * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float diff = a[d] - b[d];
          acc = acc + diff * diff;
        }
        if (acc < radius2) {
          orow_i[count] = j;
          orow_d[count] = acc;
          ++count;
        }
      }
    }
  }
  if (in_range) {
    for (int k = count; k < K; ++k) {
      orow_i[k] = -1;
      orow_d[k] = 0.0f;
    }
  }
}


// ---------------------------------------------------------------------------
// LISTED clouds with the hits STAGED IN LDS (D <= 4, K <= 64): one wave per workgroup, the tile scan of the
// kernel above, but a hit goes to the lane's row of an LDS image of the wave's 64 output rows (index as 32 bits +
// distance) instead of straight to memory, and every lane writes its finished row -- padding included -- with 16-byte
// stores.  The per-hit `orow_i[count] = j; orow_d[count] = d` of the kernel above is one 8-byte and one 4-byte store
// per lane into 64 different rows per instruction: measured at cfg3 (16 x 131072, r = 0.2, K = 32) WRITE_SIZE 3.2 GB
// for 0.8 GB of output, 70 % of the wave-cycles waiting for an issue slot behind the store queue
// (profiles/r03_ball_a_pmc.json).
// Row stride = 4 (2 ceil(K / 8) + 1) dwords: rows stay 16-byte aligned and the lanes' ds_read_b128 of their own rows
// fall into distinct banks.
// ---------------------------------------------------------------------------
__host__ __device__ inline int ball_stage_stride(int K) { return 4 * (2 * ((K + 7) / 8) + 1); }

#ifndef POINTOPS_BALL_LIST_WAVES
#define POINTOPS_BALL_LIST_WAVES 1
#endif
constexpr int kBqListWaves = POINTOPS_BALL_LIST_WAVES;  // independent waves per workgroup

template <int DT>
__global__ __launch_bounds__(kWave * kBqListWaves) void ball_query_list_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int K, float radius2, const int* __restrict__ grid_flag,
    const int* __restrict__ qcount, const int* __restrict__ qlist, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  extern __shared__ unsigned s_stage[];  // [64][stride] indices of the hits (distances are recomputed when the rows are
                                         // written: half the LDS, twice the waves per CU -- 5 resident of 8 possible
                                         // with the distances staged too, and the scan then waits on memory)
  const int n = blockIdx.y;
  const bool listed = grid_flag[n] != 0;  // (a cloud without a list -- empty, or ordering switched off: storage order)
  const int cnt = listed ? qcount[n] : P1;
  const int lane = threadIdx.x & (kWave - 1), wslot = threadIdx.x / kWave;
  const int base = (blockIdx.x * kBqListWaves + wslot) * kWave;
  if (base >= cnt) return;
  const bool in_range = base + lane < cnt;
  const int i = !in_range ? 0 : listed ? qlist[(int64_t)n * P1 + base + lane] : base + lane;
  const int len1 = (int)lengths1[n];
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  const int64_t row = (int64_t)n * P1 + i;
  const bool active = in_range && i < len1;
  const int stride = ball_stage_stride(K);
  unsigned* const my_i = s_stage + (wslot * kWave + lane) * stride;
  for (int k = 0; k < stride; k += 4)  // the row starts as padding: idx -1 (ball_query_cpu.cpp:20-21)
    *(uint4*)(my_i + k) = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  int count = 0;
  const float* __restrict__ q = p2 + (int64_t)n * P2 * DT;
  float a[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) a[d] = active ? p1[row * DT + d] : 0.0f;
  auto dist_to = [&](const float* __restrict__ b) {
    float acc;
    {
      const float diff = a[0] - b[0];
      acc = diff * diff;
    }
#pragma unroll
    for (int d = 1; d < DT; ++d) {
      const float diff = a[d] - b[d];
      acc = acc + diff * diff;
    }
    return acc;
  };
  const int room = active ? K : 0;
  // bounding box of the wave's active queries (exact min / max): see ball_query_kernel
  float blo[DT], bhi[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    float mn = active ? a[d] : __builtin_inff(), mx = active ? a[d] : -__builtin_inff();
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, off, kWave));
      mx = fmaxf(mx, __shfl_xor(mx, off, kWave));
    }
    blo[d] = mn;
    bhi[d] = mx;
  }
  int j0 = 0;
  float cn[DT];  // the NEXT tile's point of this lane, loaded one tile ahead
#pragma unroll
  for (int d = 0; d < DT; ++d) cn[d] = lane < len2 ? q[(int64_t)lane * DT + d] : 0.0f;
  while (j0 < len2 && __any(count < room)) {
    const int jc = j0 + lane;
    float c[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) c[d] = cn[d];
    {
      const int jn = jc + kWave;
#pragma unroll
      for (int d = 0; d < DT; ++d) cn[d] = jn < len2 ? q[(int64_t)jn * DT + d] : 0.0f;
    }
    float lb;
    {
      float g0 = fmaxf(fmaxf(blo[0] - c[0], c[0] - bhi[0]), 0.0f);
      lb = g0 * g0;
#pragma unroll
      for (int d = 1; d < DT; ++d) {
        const float gd = fmaxf(fmaxf(blo[d] - c[d], c[d] - bhi[d]), 0.0f);
        lb = lb + gd * gd;
      }
    }
    unsigned long long cand = __ballot(jc < len2 && lb < radius2);
    unsigned mlo = 0u, mhi = 0u;  // the lane's hits among the tile's points
    if (__popcll(cand) >= 48 && j0 + kWave <= len2) {  // most of the tile survives: all 64 points through the scalar path
      unsigned r0 = 0u, r1 = 0u;
      auto push_hit = [&](unsigned& mask, float acc) __attribute__((always_inline)) {
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "s"(radius2), "v"(acc) : "vcc");
      };
      for (int jj = 0; jj < 32; jj += 8) {
        float t[8 * DT];
#pragma unroll
        for (int u = 0; u < 8 * DT; ++u) t[u] = q[(int64_t)(j0 + jj) * DT + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) push_hit(r0, dist_to(t + u * DT));
      }
      for (int jj = 32; jj < 64; jj += 8) {
        float t[8 * DT];
#pragma unroll
        for (int u = 0; u < 8 * DT; ++u) t[u] = q[(int64_t)(j0 + jj) * DT + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) push_hit(r1, dist_to(t + u * DT));
      }
      mlo = __brev(r0);
      mhi = __brev(r1);
      cand = 0ull;
    }
    while (cand != 0ull) {  // wave-uniform
      const int b = __builtin_ctzll(cand);
      cand &= cand - 1ull;
      float pb[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) pb[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c[d]), b));
      const bool hit = dist_to(pb) < radius2;
      const unsigned bit = 1u << (b & 31);
      if (b < 32) mlo |= hit ? bit : 0u;
      else mhi |= hit ? bit : 0u;
    }
    if (count >= room) mlo = mhi = 0u;
    while (__any((mlo | mhi) != 0u)) {  // wave-uniform loop: every lane takes part in the shuffles
      const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
      const bool has = m != 0ull;
      const int b = has ? __builtin_ctzll(m) : 0;
      float ph[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) ph[d] = __shfl(c[d], b, kWave);  // the tile still sits in the lanes
      if (has) {
        if (b < 32) mlo &= mlo - 1u;
        else mhi &= mhi - 1u;
        my_i[count] = (unsigned)(j0 + b);
        ++count;
        if (count >= room) mlo = mhi = 0u;
      }
    }
    j0 += kWave;
  }
  // the lane's finished row: 16-byte pieces (two indices widened to int64, four distances recomputed from the chosen
  // points with the scan's expression -- the same operands, bit-identical); the caller guarantees K % 4 == 0, so
  // every row of the outputs is 16-byte aligned
  if (in_range) {
    int64_t* __restrict__ orow_i = idxs + row * K;
    float* __restrict__ orow_d = dists + row * K;
    for (int k = 0; k < K; k += 4) {
      const uint4 v = *(const uint4*)(my_i + k);
      const unsigned jv[4] = {v.x, v.y, v.z, v.w};
      float pt[4][DT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned jx = jv[u] == 0xffffffffu ? 0u : jv[u];  // (padding: any valid address; its distance is 0)
#pragma unroll
        for (int d = 0; d < DT; ++d) pt[u][d] = q[(int64_t)jx * DT + d];
      }
      float dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) dv[u] = jv[u] == 0xffffffffu ? 0.0f : dist_to(pt[u]);
      *(longlong2*)(orow_i + k) = make_longlong2((long long)(int)v.x, (long long)(int)v.y);
      *(longlong2*)(orow_i + k + 2) = make_longlong2((long long)(int)v.z, (long long)(int)v.w);
      *(float4*)(orow_d + k) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
  }
}

}  // namespace pointops

using namespace pointops;

// grid path considered (the per-cloud decision is made on the device): D <= 3, K <= 64, clouds large
// enough for the build passes to pay
static bool ball_grid_candidate(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K) {
  const long force = debug_knob("ball_grid", -1);  // 0 = never, 1 = whenever the shape allows (tests)
  if (force == 0) return false;
  // (its lane kernel keeps run bounds as (first, end) pairs: record indices up to the KNN grid's 2^24 - 16)
  if (force == 1) return D <= 3 && K <= 64 && N < 65536 && P2 <= knn_grid_max_points() && N > 0 && P1 > 0 && P2 > 0;
  return D <= 3 && K <= 64 && N < 65536 && P2 >= 4096 && P2 <= knn_grid_max_points() &&
         (double)N * (double)P1 * (double)P2 >= (double)(1LL << 27);
}

extern "C" size_t pointops_ball_query_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K) {
  if (!ball_grid_candidate(N, P1, P2, D, K)) return 0;
  return ball_grid_workspace_bytes(N, P1, P2);
}

extern "C" int pointops_ball_query(const float* p1, const float* p2, const int64_t* lengths1,
                                   const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2,
                                   int64_t D, int64_t K, float radius, int64_t* idxs, float* dists,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && K >= 0, "ball_query: bad sizes");
  POINTOPS_REQUIRE(P1 < (1LL << 31) && P2 < (1LL << 31) && K < (1LL << 31) && D < (1LL << 16),
                   "ball_query: sizes must fit int32");
  if (N == 0 || P1 == 0 || K == 0) return POINTOPS_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const float radius2 = radius * radius;  // fp32 product (ball_query_cpu.cpp:26)
  const int tiles = (int)ceil_div(P1, kBqBlock);
  POINTOPS_REQUIRE(N * tiles < (1LL << 31), "ball_query: grid too large");
  const int *flag = nullptr, *qcount = nullptr, *qlist = nullptr;
  const size_t need = pointops_ball_query_workspace_bytes(N, P1, P2, D, K);
  if (need > 0 && workspace != nullptr && workspace_bytes >= need) {
    KnnArgs a;
    a.p1 = p1; a.p2 = p2; a.l1 = lengths1; a.l2 = lengths2;
    a.P1 = (int)P1; a.P2 = (int)P2; a.D = (int)D; a.K = (int)K; a.N = N;
    a.tiles = tiles; a.qlist = nullptr; a.qcount = nullptr;
    a.idxs = idxs; a.dists = dists; a.stream = stream;
    const int rc = ball_grid_run(a, radius, workspace, &flag, &qcount, &qlist);
    if (rc != POINTOPS_OK) return rc;
  }
  if (flag == nullptr && ball_small_applies(N, P1, P2, D, K)) {  // few queries: one wave per query (ball_small.hip)
    launch_ball_small(p1, p2, lengths1, lengths2, N, P1, P2, D, K, radius2, idxs, dists, stream);
    return check_launch("ball_query(small)");
  }
  // listed clouds (every cloud once the lists exist) with hits staged in LDS: K a multiple of 4 up to 64
  const bool staged = flag != nullptr && D <= 4 && K <= 64 && K % 4 == 0 && debug_knob("ball_stage", 1) != 0;
  if (staged) {
    const size_t lds = (size_t)kBqListWaves * kWave * ball_stage_stride((int)K) * sizeof(unsigned);
    const dim3 lgrid((unsigned)ceil_div(P1, kWave * kBqListWaves), (unsigned)N);
#define PO_LIST(DT)                                                                                               \
  hipLaunchKernelGGL((ball_query_list_kernel<DT>), lgrid, dim3(kWave * kBqListWaves), lds, stream, p1, p2,       \
                     lengths1, lengths2,                                                                          \
                     (int)P1, (int)P2, (int)K, radius2, flag, qcount, qlist, idxs, dists)
    switch (D) {
      case 1: PO_LIST(1); break;
      case 2: PO_LIST(2); break;
      case 3: PO_LIST(3); break;
      default: PO_LIST(4); break;
    }
#undef PO_LIST
    return check_launch("ball_query(list)");
  }
  const dim3 grid((unsigned)(N * tiles)), block(kBqBlock);
#define PO_LAUNCH(DT)                                                                            \
  hipLaunchKernelGGL((ball_query_kernel<DT>), grid, block, 0, stream, p1, p2, lengths1, lengths2, \
                     (int)P1, (int)P2, (int)D, (int)K, radius2, tiles, flag, qcount, qlist, idxs, dists)
  switch (D) {
    case 1: PO_LAUNCH(1); break;
    case 2: PO_LAUNCH(2); break;
    case 3: PO_LAUNCH(3); break;
    case 4: PO_LAUNCH(4); break;
    default: PO_LAUNCH(0); break;
  }
#undef PO_LAUNCH
  return check_launch("ball_query");
}
