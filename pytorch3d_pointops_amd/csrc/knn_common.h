// knn_common.h -- device building blocks shared by the KNN kernels (gfx950).
#pragma once
#include "common.h"

namespace pointops {

constexpr int kKnnBlock = 256;  // lanes per workgroup of the brute-force scan
constexpr int kTileP2 = 8;      // p2 points fetched per scalar-load group

// ---------------------------------------------------------------------------
// distance (unfused fp32; accumulation order d = 0,1,2,... from the first term,
// identical to `dist = 0; dist += diff*diff` of knn_cpu.cpp:42-50 since 0 + t == t)
// ---------------------------------------------------------------------------
template <int D, int NORM>
__device__ __forceinline__ float pair_dist(const float (&a)[D], const float* __restrict__ b) {
  float acc;
  {
    const float diff = a[0] - b[0];
    acc = (NORM == 1) ? __builtin_fabsf(diff) : diff * diff;
  }
#pragma unroll
  for (int d = 1; d < D; ++d) {
    const float diff = a[d] - b[d];
    acc = (NORM == 1) ? (acc + __builtin_fabsf(diff)) : (acc + diff * diff);
  }
  return acc;
}

// ---------------------------------------------------------------------------
// Sorted (ascending) register top-K for candidates that arrive in INCREASING
// index order (brute-force scan).  insert() requires d < dk[KC-1]; a newcomer is
// placed after equal distances, which is exactly (dist, idx) order.
// ---------------------------------------------------------------------------
template <int KC>
struct TopK {
  float dk[KC];
  int ik[KC];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      dk[i] = __builtin_inff();
      ik[i] = 0;
    }
  }
  __device__ __forceinline__ float worst() const { return dk[KC - 1]; }
  __device__ __forceinline__ void insert(float d, int j) {
#pragma unroll
    for (int i = KC - 1; i > 0; --i) {
      const bool up = d < dk[i - 1];  // element i-1 moves up to slot i
      const bool here = d < dk[i];    // newcomer lands at or below slot i
      dk[i] = up ? dk[i - 1] : (here ? d : dk[i]);
      ik[i] = up ? ik[i - 1] : (here ? j : ik[i]);
    }
    if (d < dk[0]) {
      dk[0] = d;
      ik[0] = j;
    }
  }
  __device__ __forceinline__ float dist_at(int k) const { return dk[k]; }
  __device__ __forceinline__ int idx_at(int k) const { return ik[k]; }
};

// ---------------------------------------------------------------------------
// Sorted register top-K for candidates in ARBITRARY index order (grid search):
// 64-bit keys (fp32 distance bits << 32 | index).  Distances are >= +0, so their
// bit patterns order like unsigned integers and one unsigned 64-bit compare IS
// the lexicographic (dist, idx) compare of std::tuple<float,int>.
// ---------------------------------------------------------------------------
template <int KC>
struct TopKLex {
  unsigned long long key[KC];
  static constexpr unsigned long long kEmpty = 0x7f8000007fffffffULL;  // (+inf, INT_MAX)
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < KC; ++i) key[i] = kEmpty;
  }
  __device__ __forceinline__ static unsigned long long make(float d, int j) {
    return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
  }
  __device__ __forceinline__ unsigned worst_bits() const { return (unsigned)(key[KC - 1] >> 32); }
  // distance bits of the K-th best (1 <= K <= KC, runtime): what a query needs certified
  __device__ __forceinline__ unsigned kth_bits(int K) const {
    unsigned b = (unsigned)(key[KC - 1] >> 32);
    if (K < KC) {  // wave-uniform branch; a select chain (indexing the register array by K would go through scratch)
#pragma unroll
      for (int t = 0; t < KC - 1; ++t) b = (t == K - 1) ? (unsigned)(key[t] >> 32) : b;
    }
    return b;
  }
  // requires k < key[KC-1]
  __device__ __forceinline__ void insert(unsigned long long k) {
    bool lt_hi = true;  // k < key[i] for the slot above the current one
#pragma unroll
    for (int i = KC - 1; i > 0; --i) {
      const bool lt_lo = k < key[i - 1];
      key[i] = lt_lo ? key[i - 1] : (lt_hi ? k : key[i]);
      lt_hi = lt_lo;
    }
    if (lt_hi) key[0] = k;
  }
  __device__ __forceinline__ float dist_at(int k) const { return __uint_as_float((unsigned)(key[k] >> 32)); }
  __device__ __forceinline__ int idx_at(int k) const { return (int)(unsigned)key[k]; }
};

// ---------------------------------------------------------------------------
// Brute-force scan of one whole cloud for the calling lane's query `a`.
// `q` is a wave-uniform pointer: the loads become s_load_dwordx8/x16 and the
// points are consumed as SGPR operands.
// ---------------------------------------------------------------------------
template <int D, int KC, int NORM>
__device__ __forceinline__ void scan_cloud(const float (&a)[D], const float* __restrict__ q, int jbeg, int len2,
                                           TopK<KC>& top) {  // rows [jbeg, len2) of the cloud at q
  int j = jbeg;
  for (; j + kTileP2 <= len2; j += kTileP2) {
    float t[kTileP2 * D];
#pragma unroll
    for (int u = 0; u < kTileP2 * D; ++u) t[u] = q[(int64_t)j * D + u];
#pragma unroll
    for (int jj = 0; jj < kTileP2; ++jj) {
      const float dist = pair_dist<D, NORM>(a, t + jj * D);
      if (dist < top.worst()) top.insert(dist, j + jj);
    }
  }
  for (; j < len2; ++j) {
    float t[D];
#pragma unroll
    for (int u = 0; u < D; ++u) t[u] = q[(int64_t)j * D + u];
    const float dist = pair_dist<D, NORM>(a, t);
    if (dist < top.worst()) top.insert(dist, j);
  }
}

// Write one output row: the first min(K, len2) entries of the list, zeros after
// (knn_cpu.cpp:25-26 pre-fill).
template <int KC, typename TOP>
__device__ __forceinline__ void write_row(const TOP& top, int K, int len2, int64_t* __restrict__ orow_i,
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

}  // namespace pointops
