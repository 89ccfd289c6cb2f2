// knn_wide.hip -- exact brute-force KNN for the shapes the register kernel (knn.hip: D <= 8,
// K <= 32) does not cover: feature-space neighbours (any D, e.g. 64) and long lists (any K,
// e.g. 64).  Same semantics as everything else (reference CPU path knn_cpu.cpp:13-69: unfused
// fp32 sum in d order, (dist, idx) lexicographic lists, zeros for padding); the reference's
// CUDA counterpart is its generic V0/V1 kernels (csrc/knn/knn.cu:44-140), whose design this
// does not follow.
//
// One lane = one query.  Runtime D, so the workgroup's queries sit TRANSPOSED in LDS
// (s_q[d][lane]: conflict-free, D x 4 bytes per query) instead of a register array; p2 is
// streamed like in knn.hip through the scalar path -- four rows x eight coordinates per
// s_load_dwordx8 group from wave-uniform addresses, consumed as SGPR operands -- so one LDS read
// feeds 12 VALU instructions (sub, mul, add for four points).  The running list is either the
// sorted register array of knn_common.h (K <= 32: 256-lane workgroups) or, for longer lists, a
// per-lane sorted list in LDS (s_list[k][lane], insertion by shifting; 64-lane workgroups so
// that K x 8 bytes per lane fit).  Candidates arrive in increasing index order and displace
// only on strict <, which is exactly the (dist, idx) order.
#include "knn_common.h"
#include "knn_grid.h"
#include "sort_net.h"

#include <algorithm>

namespace pointops {

constexpr int kWideJ = 4;      // p2 rows per scalar-load group
constexpr int kWideChunk = 8;  // coordinates per s_load_dwordx8

// acc[jj] += |a - p2[j0+jj][:]| over all D coordinates, for J rows at once
template <int J, int NORM, int WG>
__device__ __forceinline__ void wide_dists(const float* __restrict__ s_q, int lane, const float* __restrict__ q,
                                           int64_t j0, int D, float (&acc)[J]) {
#pragma unroll
  for (int jj = 0; jj < J; ++jj) acc[jj] = 0.0f;  // 0 + t == t: same bits as starting from the first term
  int d = 0;
  for (; d + kWideChunk <= D; d += kWideChunk) {
    float t[J][kWideChunk];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
#pragma unroll
      for (int u = 0; u < kWideChunk; ++u) t[jj][u] = q[(j0 + jj) * D + d + u];
    }
#pragma unroll
    for (int u = 0; u < kWideChunk; ++u) {
      const float a = s_q[(d + u) * WG + lane];
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        const float diff = a - t[jj][u];
        acc[jj] = (NORM == 1) ? (acc[jj] + __builtin_fabsf(diff)) : (acc[jj] + diff * diff);
      }
    }
  }
  for (; d < D; ++d) {
    const float a = s_q[d * WG + lane];
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const float diff = a - q[(j0 + jj) * D + d];
      acc[jj] = (NORM == 1) ? (acc[jj] + __builtin_fabsf(diff)) : (acc[jj] + diff * diff);
    }
  }
}

// per-lane sorted list in LDS (K runtime): dist and idx planes, entry k of lane l at [k * WG + l]
template <int WG>
struct LdsList {
  float* dk;
  int* ik;
  int lane, K, cnt;
  float worst;
  __device__ __forceinline__ void init(float* d, int* i, int lane_, int K_) {
    dk = d, ik = i, lane = lane_, K = K_, cnt = 0;
    worst = __builtin_inff();
  }
  __device__ __forceinline__ void offer(float d, int j) {
    if (cnt < K || d < worst) {
      int pos = cnt < K ? cnt : K - 1;  // slot that is appended / freed
      while (pos > 0) {
        const float prev = dk[(pos - 1) * WG + lane];
        if (!(d < prev)) break;
        dk[pos * WG + lane] = prev;
        ik[pos * WG + lane] = ik[(pos - 1) * WG + lane];
        --pos;
      }
      dk[pos * WG + lane] = d;
      ik[pos * WG + lane] = j;
      if (cnt < K) ++cnt;
      if (cnt == K) worst = dk[(K - 1) * WG + lane];
    }
  }
};

constexpr int kLongKC = 64;     // lists of 33..64: 64-bit keys in registers, fed by a per-lane LDS queue
constexpr int kLongQueue = 16;  // queue slots per lane

// KC in (0, 32]: register list of KC >= K slots (insertion);  KC == 64: register list of 64 keys,
// candidates below the (stale) threshold queue up in LDS and are merged 16 at a time by sorting
// networks (knn_grid.hip's selection core);  KC == 0: LDS list of K slots (insertion by shifting)
template <int KC, int NORM, int WG>
__global__ __launch_bounds__(WG) void knn_wide_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int D, int K, int tiles_per_cloud, int S,
    unsigned long long* __restrict__ partial, const int* __restrict__ qlist, const int* __restrict__ qcount,
    int64_t* __restrict__ idxs, float* __restrict__ dists) {
  extern __shared__ float s_dyn[];
  float* __restrict__ s_q = s_dyn;  // [D][WG]
  const int n = blockIdx.x / tiles_per_cloud;  // wave-uniform
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int split = blockIdx.y;  // slice of p2 this workgroup scans (S > 1: partial lists, merged afterwards)
  const int lane = threadIdx.x;
  const int i0 = tile * WG, i = i0 + lane;
  int len1 = (int)lengths1[n];
  if (len1 > P1) len1 = P1;
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  if (len2 < 0) len2 = 0;

  // With `qlist` the workgroup's queries are qlist[n*P1 + i0 ..] (the exact fallback pass of the grid search
  // for lists longer than the register scan takes): gathered rows instead of a contiguous tile.
  const int* __restrict__ ql = qlist != nullptr ? qlist + (int64_t)n * P1 : nullptr;
  const int nlist = ql != nullptr ? qcount[n] : P1;
  if (i0 >= nlist) return;  // (wave-uniform)
  // query tile, transposed (coalesced global reads of the tile's contiguous rows)
  const int nq = min(WG, nlist - i0);
  const float* __restrict__ src = p1 + ((int64_t)n * P1 + i0) * D;
  for (int f = lane; f < nq * D; f += WG) {
    const int r = f / D, d = f - r * D;
    s_q[d * WG + r] = ql != nullptr ? p1[((int64_t)n * P1 + ql[i0 + r]) * D + d] : src[f];
  }
  for (int f = nq * D + lane; f < WG * D; f += WG) {  // lanes beyond the cloud: harmless zeros
    const int r = f / D, d = f - r * D;
    s_q[d * WG + r] = 0.0f;
  }
  __syncthreads();

  const float* __restrict__ q = p2 + (int64_t)n * P2 * D;
  const int qi = ql != nullptr ? (i < nlist ? ql[i] : 0) : i;  // output row of this lane
  const bool live = ql != nullptr ? i < nlist : i < len1;
  const bool owns_row = ql != nullptr ? i < nlist : i < P1;
  const int64_t row = (int64_t)n * P1 + qi;

  if constexpr (KC == kLongKC) {
    // 64-bit (dist, idx) keys ordered through the FP64 pipe (sort_net.h); free queue slots hold the empty key
    double* __restrict__ s_queue = reinterpret_cast<double*>(s_dyn + (size_t)D * WG);
#pragma unroll
    for (int t = 0; t < kLongQueue; ++t) s_queue[t * WG + lane] = TopKF64<KC>::empty();
    TopKF64<KC> top;
    top.init();
    unsigned thr = 0x7f800000u;  // distance bits a candidate must not exceed (stale between flushes)
    int qn = 0;
    auto flush = [&]() {
      double qk[kLongQueue];
#pragma unroll
      for (int t = 0; t < kLongQueue; ++t) qk[t] = s_queue[t * WG + lane];
#pragma unroll
      for (int t = 0; t < kLongQueue; ++t) s_queue[t * WG + lane] = TopKF64<KC>::empty();
      bitonic_sort<kLongQueue>(qk);
#pragma unroll
      for (int t = 0; t < kLongQueue; ++t) top.key[KC - 1 - t] = kmin(top.key[KC - 1 - t], qk[t]);  // list slot KC-1-t meets queue entry t
      bitonic_merge<KC>(top.key);
      qn = 0;
      thr = top.worst_bits();
    };
    auto offer = [&](float d, int j) {
      // an equal distance with a larger index (every later candidate) still passes `<=`; the
      // 64-bit key order then drops it at the merge, so the list stays in (dist, idx) order
      if (__float_as_uint(d) <= thr) {
        s_queue[qn * WG + lane] = TopKF64<KC>::make(d, j);
        ++qn;
      }
    };
    int j = 0;
    for (; j + kWideJ <= len2; j += kWideJ) {
      float acc[kWideJ];
      wide_dists<kWideJ, NORM, WG>(s_q, lane, q, j, D, acc);
#pragma unroll
      for (int jj = 0; jj < kWideJ; ++jj) offer(acc[jj], j + jj);
      if (__any(qn > kLongQueue - kWideJ)) flush();
    }
    for (; j < len2; ++j) {
      float acc[1];
      wide_dists<1, NORM, WG>(s_q, lane, q, j, D, acc);
      offer(acc[0], j);
      if (__any(qn > kLongQueue - kWideJ)) flush();
    }
    flush();
    if (owns_row) {
      const int kvalid = live ? min(K, len2) : 0;
      int64_t* __restrict__ oi = idxs + row * K;
      float* __restrict__ od = dists + row * K;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        if (k < K) {
          const bool ok = k < kvalid;
          oi[k] = ok ? (int64_t)top.idx_at(k) : 0;
          od[k] = ok ? top.dist_at(k) : 0.0f;
        }
      }
    }
  } else if constexpr (KC > 0) {
    TopK<KC> top;
    top.init();
    const int jbeg = (int)((int64_t)len2 * split / S), jend = (int)((int64_t)len2 * (split + 1) / S);
    int j = jbeg;
    for (; j + kWideJ <= jend; j += kWideJ) {
      float acc[kWideJ];
      wide_dists<kWideJ, NORM, WG>(s_q, lane, q, j, D, acc);
#pragma unroll
      for (int jj = 0; jj < kWideJ; ++jj)
        if (acc[jj] < top.worst()) top.insert(acc[jj], j + jj);
    }
    for (; j < jend; ++j) {
      float acc[1];
      wide_dists<1, NORM, WG>(s_q, lane, q, j, D, acc);
      if (acc[0] < top.worst()) top.insert(acc[0], j);
    }
    if (S == 1) {
      if (owns_row) write_row<KC>(top, K, live ? len2 : 0, idxs + row * K, dists + row * K);
    } else if (owns_row) {
      // partial list of this slice as (dist bits, idx) keys; empty slots order last
      unsigned long long* __restrict__ o = partial + (row * S + split) * K;
      const int have = min(K, jend - jbeg);
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (k < K) o[k] = k < have ? TopKLex<KC>::make(top.dk[k], top.ik[k]) : TopKLex<KC>::kEmpty;
    }
  } else {
    LdsList<WG> top;
    top.init(s_dyn + (size_t)D * WG, reinterpret_cast<int*>(s_dyn + (size_t)D * WG + (size_t)K * WG), lane, K);
    int j = 0;
    for (; j + kWideJ <= len2; j += kWideJ) {
      float acc[kWideJ];
      wide_dists<kWideJ, NORM, WG>(s_q, lane, q, j, D, acc);
#pragma unroll
      for (int jj = 0; jj < kWideJ; ++jj) top.offer(acc[jj], j + jj);
    }
    for (; j < len2; ++j) {
      float acc[1];
      wide_dists<1, NORM, WG>(s_q, lane, q, j, D, acc);
      top.offer(acc[0], j);
    }
    if (owns_row) {
      const int kvalid = live ? top.cnt : 0;  // = min(K, len2)
      int64_t* __restrict__ oi = idxs + row * K;
      float* __restrict__ od = dists + row * K;
      for (int k = 0; k < K; ++k) {
        const bool ok = k < kvalid;
        oi[k] = ok ? (int64_t)top.ik[k * WG + lane] : 0;
        od[k] = ok ? top.dk[k * WG + lane] : 0.0f;
      }
    }
  }
}

constexpr size_t kWideLdsMax = 160 * 1024;

template <int KC, int NORM, int WG>
static int launch_wide(const KnnArgs& a, size_t lds, int S, void* workspace) {
  auto kern = knn_wide_kernel<KC, NORM, WG>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kWideLdsMax) != hipSuccess)
      return check_launch("knn_points_idx(wide attribute)");
    attr_set = true;
  }
  const int tiles = (int)ceil_div(a.P1, WG);
  if (a.N * tiles >= (1LL << 31)) return POINTOPS_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * tiles), (unsigned)S), dim3(WG), lds, a.stream, a.p1, a.p2, a.l1, a.l2,
                     a.P1, a.P2, a.D, a.K, tiles, S, (unsigned long long*)workspace, a.qlist, a.qcount, a.idxs, a.dists);
  if (KC > 0 && KC <= 32 && S > 1) knn_merge_partials(a, S, workspace);
  return POINTOPS_OK;
}

// true when the wide kernel can take the shape (LDS budget); see knn.hip for the dispatch
bool knn_wide_supported(int64_t D, int64_t K) {
  if (K <= 32) return (size_t)D * 256 * 4 <= kWideLdsMax - 1024;
  if (K <= kLongKC) return ((size_t)D * 4 + (size_t)kLongQueue * 8) * 64 <= kWideLdsMax - 1024;
  return ((size_t)D * 4 + (size_t)K * 8) * 64 <= kWideLdsMax - 1024;
}

int launch_knn_wide(const KnnArgs& a, int norm, void* workspace) {
  const int K = a.K;
  if (K <= 32) {
    // 64-lane workgroups while the batch is small (4x the workgroups), 256 lanes otherwise
    const int S = knn_split_count(a.N, a.P1, a.P2, a.K);
    const bool small = a.N * ceil_div(a.P1, 256) < 2048;
    const size_t lds = (size_t)a.D * (small ? 64 : 256) * 4;
#define PO_WIDE(KC)                                                                                                 \
  return small ? (norm == 1 ? launch_wide<KC, 1, 64>(a, lds, S, workspace) : launch_wide<KC, 2, 64>(a, lds, S, workspace)) \
               : (norm == 1 ? launch_wide<KC, 1, 256>(a, lds, 1, workspace) : launch_wide<KC, 2, 256>(a, lds, 1, workspace))
    if (K <= 4) PO_WIDE(4);
    else if (K <= 8) PO_WIDE(8);
    else if (K <= 16) PO_WIDE(16);
    else if (K <= 24) PO_WIDE(24);
    else PO_WIDE(32);
#undef PO_WIDE
  }
  if (K <= kLongKC) {
    const size_t lds = ((size_t)a.D * 4 + (size_t)kLongQueue * 8) * 64;
    return norm == 1 ? launch_wide<kLongKC, 1, 64>(a, lds, 1, workspace) : launch_wide<kLongKC, 2, 64>(a, lds, 1, workspace);
  }
  const size_t lds = ((size_t)a.D * 4 + (size_t)K * 8) * 64;
  return norm == 1 ? launch_wide<0, 1, 64>(a, lds, 1, workspace) : launch_wide<0, 2, 64>(a, lds, 1, workspace);
}

}  // namespace pointops
