// knn_small.hip -- exact K nearest neighbours for FEW queries (gfx950): one wave per query.
//
// Semantics: the CPU path of the reference (csrc/knn/knn_cpu.cpp:13-69), as knn.hip.
//
// The brute-force scan of knn.hip gives every LANE a query and streams the cloud through the scalar
// path: 64 queries per wave, so a batch of 2 x 1024 queries (BASELINE.json configs[0]) is 32 waves
// on a chip with 1024 SIMDs; slicing the cloud (knn_split_count) bought 256 waves and a second
// launch that merged the slices' lists from global memory: 19 + 19 us for 2 M pairs.
// Here the roles are swapped: the QUERY is wave-uniform (scalar loads, SGPR operands) and the cloud is
// dealt over the lanes (lane l takes candidates l, l + 64, ...: coalesced rows), every lane keeps the
// sorted top-K of ITS candidates in registers (candidates arrive in increasing index order per lane,
// so the (dist, idx) order of TopK holds), and the K answers are then pulled out of the 64 list heads
// with K rounds of a wave-wide lexicographic minimum -- two v_min_u32 reductions over DPP
// (distance bits, then the index among the lanes that hold that distance); the winning lane pops its
// head.  Lane r keeps answer r, so a row leaves as one coalesced store per output.
// One launch, TQ waves instead of TQ / 64, no workspace.
#include "debug.h"
#include "knn_common.h"
#include "knn_grid.h"

#include <algorithm>

namespace pointops {

constexpr int kSmallWaves = 4;   // waves per workgroup (independent of each other)
constexpr int kSmallUnroll = 4;  // candidates per lane in flight

// minimum over the 64 lanes, returned wave-uniform.  quad_perm / row_ror make every lane of a row of 16 hold
// the row's minimum; row_bcast15 / row_bcast31 carry it up the rows, so lane 63 ends with the wave's.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define PO_DPP_MIN(ctrl, rows) \
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false))
  PO_DPP_MIN(0xB1, 0xf);   // quad_perm [1,0,3,2]
  PO_DPP_MIN(0x4E, 0xf);   // quad_perm [2,3,0,1]
  PO_DPP_MIN(0x124, 0xf);  // row_ror 4
  PO_DPP_MIN(0x128, 0xf);  // row_ror 8
  PO_DPP_MIN(0x142, 0xa);  // row_bcast15 into rows 1, 3
  PO_DPP_MIN(0x143, 0xc);  // row_bcast31 into rows 2, 3
#undef PO_DPP_MIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// K-th smallest DISTINCT value of v over the lanes (K rounds of a wave minimum): an upper bound of the K-th smallest
// value.  With v = the heads of the lanes' sorted lists it bounds the query's K-th distance from above (the K smallest
// heads are K different candidates), so later candidates beyond it can be dropped before they reach a list.
template <int KC>
__device__ __forceinline__ unsigned kth_head_bits(unsigned v, int K) {
  unsigned m = wave_min_u32(v);
  for (int r = 1; r < K; ++r) {  // wave-uniform
    v = v == m ? 0xffffffffu : v;
    const unsigned m2 = wave_min_u32(v);
    if (m2 == 0xffffffffu) break;  // fewer than K distinct heads
    m = m2;
  }
  return m;
}

// Q queries per wave share every loaded candidate
template <int D, int KC, int NORM, int Q>
__global__ __launch_bounds__(kSmallWaves * 64) void knn_small_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int K, int qw, int waves_per_cloud, int total_waves,
    int64_t* __restrict__ idxs, float* __restrict__ dists) {
  const int lane = threadIdx.x & 63;
  const int g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kSmallWaves + (threadIdx.x >> 6)));
  if (g >= total_waves) return;
  const int n = g / waves_per_cloud, t = g - n * waves_per_cloud;
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  if (len2 < 0) len2 = 0;
  const int len1 = min((int)lengths1[n], P1);
  const float* __restrict__ cloud = p2 + (int64_t)n * P2 * D;
  const int kvalid = len2 < K ? len2 : K;
  const int iend = min((t + 1) * qw * Q, P1);
  for (int i0 = t * qw * Q; i0 < iend; i0 += Q) {  // wave-uniform
    const int64_t row0 = (int64_t)n * P1 + i0;
    if (i0 < len1 && len2 > 0) {
      float a[Q][D];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int64_t row = row0 + (i0 + q < len1 ? q : 0);  // rows past the cloud's queries repeat the first (not written)
#pragma unroll
        for (int d = 0; d < D; ++d) a[q][d] = p1[row * D + d];  // wave-uniform address: scalar loads
      }
      TopK<KC> top[Q];
      float gate[Q];  // wave-uniform: candidates beyond it cannot be among the K nearest
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        top[q].init();
        gate[q] = __builtin_inff();
      }
      // candidates after which the gates are refreshed (doubling): a gate only bites once a lane has seen several
      // times K candidates, and a refresh costs ~10 K instructions per query
      int next_gate = max(64 * kSmallUnroll * 2, 128 * K);
      for (int j0 = 0; j0 < len2; j0 += 64 * kSmallUnroll) {
        float c[kSmallUnroll][D];
#pragma unroll
        for (int u = 0; u < kSmallUnroll; ++u) {
          const int j = min(j0 + u * 64 + lane, len2 - 1);
#pragma unroll
          for (int d = 0; d < D; ++d) c[u][d] = cloud[(int64_t)j * D + d];
        }
#pragma unroll
        for (int u = 0; u < kSmallUnroll; ++u) {
          const int j = j0 + u * 64 + lane;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            float dist = pair_dist<D, NORM>(a[q], c[u]);
            dist = j < len2 ? dist : __builtin_inff();
            if (dist < top[q].worst() && dist <= gate[q]) top[q].insert(dist, j);
          }
        }
        if (j0 + 64 * kSmallUnroll >= next_gate && j0 + 64 * kSmallUnroll < len2) {
          next_gate *= 2;
#pragma unroll
          for (int q = 0; q < Q; ++q)
            gate[q] = __uint_as_float(kth_head_bits<KC>(__float_as_uint(top[q].dk[0]), K));
        }
      }
      // the K smallest (dist, idx) of the 64 sorted lists, in order
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        unsigned rd = 0;
        int ri = 0;
        for (int r = 0; r < kvalid; ++r) {  // wave-uniform
          const unsigned db = __float_as_uint(top[q].dk[0]);
          const unsigned m = wave_min_u32(db);
          const unsigned ci = db == m ? (unsigned)top[q].ik[0] : 0xffffffffu;
          const unsigned mi = wave_min_u32(ci);
          if (ci == mi) {  // this lane's head is the answer (all lanes, with empty heads, once the lists run dry)
#pragma unroll
            for (int k = 0; k < KC - 1; ++k) {
              top[q].dk[k] = top[q].dk[k + 1];
              top[q].ik[k] = top[q].ik[k + 1];
            }
            top[q].dk[KC - 1] = __builtin_inff();
            top[q].ik[KC - 1] = 0;
          }
          if (lane == r) {
            rd = m;
            ri = (int)mi;
          }
        }
        if (i0 + q < len1 && lane < K) {  // slots >= min(K, len2): zeros (knn_cpu.cpp:25-26)
          idxs[(row0 + q) * K + lane] = (int64_t)ri;
          dists[(row0 + q) * K + lane] = __uint_as_float(rd);
        }
      }
    }
    // padded rows and empty clouds: zeros
#pragma unroll
    for (int q = 0; q < Q; ++q)
      if (i0 + q < P1 && (i0 + q >= len1 || len2 == 0) && lane < K) {
        idxs[(row0 + q) * K + lane] = 0;
        dists[(row0 + q) * K + lane] = 0.0f;
      }
  }
}

template <int D, int KC, int NORM, int Q>
static void launch_small_q(const KnnArgs& a, int qw) {
  const int64_t wpc = ceil_div((int64_t)a.P1, (int64_t)qw * Q), total = a.N * wpc;
  const dim3 grid((unsigned)ceil_div(total, (int64_t)kSmallWaves));
  hipLaunchKernelGGL((knn_small_kernel<D, KC, NORM, Q>), grid, dim3(kSmallWaves * 64), 0, a.stream, a.p1, a.p2, a.l1,
                     a.l2, a.P1, a.P2, a.K, qw, (int)wpc, (int)total, a.idxs, a.dists);
}

// queries per wave: FOUR for lists of one or two slots once that still leaves ~4 waves per SIMD (a quarter of the loads
// per pair: 16384 x 16384, K=1: 145 -> 99 us), else one -- longer lists spend their time in the inserts and the final
// extraction, which sharing does not touch (K=8, 4096 x 4096: 25 us alone, 44 us shared);
// beyond 16384 waves a wave takes several such groups in turn
template <int D, int KC, int NORM>
static void launch_small(const KnnArgs& a) {
  constexpr int QMAX = KC <= 2 ? 4 : 1;
  const int64_t tq = a.N * (int64_t)a.P1;
  const long knob = debug_knob("knn_small_q", 0);
  const bool wide = knob > 0 ? knob > 1 : tq >= 4096 * QMAX;
  const int Q = wide ? QMAX : 1;
  const int qw = (int)std::min<int64_t>(std::max<int64_t>(ceil_div(tq, (int64_t)16384 * Q), 1), 64);
  if (QMAX > 1 && wide) launch_small_q<D, KC, NORM, QMAX>(a, qw);
  else launch_small_q<D, KC, NORM, 1>(a, qw);
}

template <int D, int NORM>
static void small_k(const KnnArgs& a) {
  const int K = a.K;
  if (K <= 1) launch_small<D, 1, NORM>(a);
  else if (K <= 2) launch_small<D, 2, NORM>(a);
  else if (K <= 4) launch_small<D, 4, NORM>(a);
  else if (K <= 8) launch_small<D, 8, NORM>(a);
  else if (K <= 16) launch_small<D, 16, NORM>(a);
  else if (K <= 24) launch_small<D, 24, NORM>(a);
  else launch_small<D, 32, NORM>(a);
}

template <int NORM>
static void small_d(const KnnArgs& a) {
  switch (a.D) {
    case 1: small_k<1, NORM>(a); break;
    case 2: small_k<2, NORM>(a); break;
    case 3: small_k<3, NORM>(a); break;
    case 4: small_k<4, NORM>(a); break;
    case 5: small_k<5, NORM>(a); break;
    case 6: small_k<6, NORM>(a); break;
    case 7: small_k<7, NORM>(a); break;
    case 8: small_k<8, NORM>(a); break;
    default: break;
  }
}

// Few queries: fewer query WAVES of the lane-per-query scan than the chip has SIMDs x 2 (x 0.5 for lists beyond 16 slots,
// whose inserts and extraction cost more here): tools/knn_small_sweep.py, profiles/r03_knn_small_sweep.jsonl.
// POINTOPS_DEBUG knn_small=0 keeps the sliced scan, =1 takes every batch the kernel supports (tests).
bool knn_small_applies(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K) {
  if (D < 1 || D > 8 || K < 1 || K > 32 || N * P1 >= (1LL << 31)) return false;
  const long knob = debug_knob("knn_small", -1);
  if (knob == 0) return false;
  if (knob == 1) return true;
  return N * ceil_div(P1, (int64_t)64) < (K > 16 ? 512 : 2048);
}

void launch_knn_small(const KnnArgs& a, int norm) {
  if (norm == 1) small_d<1>(a);
  else small_d<2>(a);
}

}  // namespace pointops
