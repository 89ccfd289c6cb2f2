// sort_net.h -- register sorting networks on 64-bit (dist bits, idx) keys, shared by the grid
// search (knn_grid.hip) and the long-list brute-force scan (knn_wide.hip).
#pragma once
#include "common.h"

namespace pointops {

// ---------------------------------------------------------------------------
// register sorting networks on 64-bit (dist bits, idx) keys
// ---------------------------------------------------------------------------
__device__ __forceinline__ void key_ce(unsigned long long& a, unsigned long long& b, bool asc) {
  // compare-exchange: afterwards a <= b when asc, a >= b otherwise.
  // One 64-bit compare, one mask, four bit-selects (v_bfi_b32).  Written with an opaque mask
  // because `sw ? b : a` / `sw ? a : b` are re-canonicalised by the compiler into umin/umax
  // and lowered as TWO v_cmp_*_u64 (each with its own s_nop hazard pad) + 4 v_cndmask.
  unsigned m = (asc ? (b < a) : (a < b)) ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(m));
  const unsigned alo = (unsigned)a, ahi = (unsigned)(a >> 32), blo = (unsigned)b, bhi = (unsigned)(b >> 32);
  const unsigned lo_lo = (m & blo) | (~m & alo), lo_hi = (m & bhi) | (~m & ahi);
  const unsigned hi_lo = (m & alo) | (~m & blo), hi_hi = (m & ahi) | (~m & bhi);
  a = ((unsigned long long)lo_hi << 32) | lo_lo;
  b = ((unsigned long long)hi_hi << 32) | hi_lo;
}

// Sorting networks for the queue (ascending).  16 inputs: the 60-comparator, 10-layer network
// (optimal size; checked exhaustively with the 0-1 principle, tools/verify_sort_networks.py);
// 8 inputs: Batcher's odd-even merge sort, 19 comparators (optimal).
template <int N>
struct SortNet {  // other sizes are never executed (the queue exists only for KC >= 8)
  static constexpr int kSize = 0;
  static constexpr unsigned char kA[1] = {0};
  static constexpr unsigned char kB[1] = {0};
};
template <>
struct SortNet<16> {
  static constexpr int kSize = 60;
  static constexpr unsigned char kA[60] = {0, 1, 2,  3,  4, 5, 7,  9,  0, 1, 2, 3, 6,  8,  10, 11, 0, 2, 4, 6,
                                           7, 10, 12, 14, 0, 1, 4,  5,  6, 8, 12, 13, 1, 3,  4,  5,  8, 9, 13, 1,
                                           2, 5,  7,  9,  11, 2, 3, 9,  11, 3, 6, 7,  10, 3, 5,  7,  9,  11, 6, 8};
  static constexpr unsigned char kB[60] = {13, 12, 15, 14, 8,  6,  11, 10, 5,  7,  9,  4,  13, 14, 15, 12, 1, 3, 5, 8,
                                           9,  11, 13, 15, 2,  3,  10, 11, 7,  9,  14, 15, 2,  12, 6,  7,  10, 11, 14, 4,
                                           6,  8,  10, 13, 14, 4,  6,  12, 13, 5,  8,  9,  12, 4,  6,  8,  10, 12, 7, 9};
};
template <>
struct SortNet<8> {
  static constexpr int kSize = 19;
  static constexpr unsigned char kA[19] = {0, 2, 0, 1, 1, 4, 6, 4, 5, 5, 0, 2, 2, 1, 3, 3, 1, 3, 5};
  static constexpr unsigned char kB[19] = {1, 3, 2, 3, 2, 5, 7, 6, 7, 6, 4, 6, 4, 5, 7, 5, 2, 4, 6};
};

template <int N>
__device__ __forceinline__ void bitonic_sort(unsigned long long (&a)[N]) {  // ascending
#pragma unroll
  for (int i = 0; i < SortNet<N>::kSize; ++i) key_ce(a[SortNet<N>::kA[i]], a[SortNet<N>::kB[i]], true);
}

template <int N>
__device__ __forceinline__ void bitonic_merge(unsigned long long (&a)[N]) {  // bitonic -> ascending
#pragma unroll
  for (int j = N >> 1; j > 0; j >>= 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = i ^ j;
      if (l > i) key_ce(a[i], a[l], true);
    }
  }
}


// 32-bit keys (ball query: point indices): a compare-exchange is v_min_u32 + v_max_u32
__device__ __forceinline__ void key_ce(unsigned& a, unsigned& b, bool) {
  const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}
template <int N>
__device__ __forceinline__ void bitonic_sort(unsigned (&a)[N]) {  // ascending
#pragma unroll
  for (int i = 0; i < SortNet<N>::kSize; ++i) key_ce(a[SortNet<N>::kA[i]], a[SortNet<N>::kB[i]], true);
}
template <int N>
__device__ __forceinline__ void bitonic_merge(unsigned (&a)[N]) {  // bitonic -> ascending
#pragma unroll
  for (int j = N >> 1; j > 0; j >>= 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = i ^ j;
      if (l > i) key_ce(a[i], a[l], true);
    }
  }
}

// ---------------------------------------------------------------------------
// 64-bit keys through the FP64 pipe.  A key (fp32 distance bits << 32 | index) with distance >= +0 and
// not NaN is the bit pattern of a non-negative, non-NaN double (the top 12 bits are 0x7f8 at most, never
// 0x7ff), and for such doubles value order == unsigned bit-pattern order (denormal doubles included: FP64
// denormals are never flushed on gfx9).  v_min_f64 / v_max_f64 return one operand unchanged, so a
// compare-exchange is TWO instructions (10-11 cycles per wave on a SIMD, tools/ce_microbench.hip)
// instead of v_cmp_lt_u64 + mask + 4 v_bfi_b32 (27 cycles); exactness against an integer sort is
// checked by the same tool over denormal / zero / +inf / tie-heavy key sets.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double kmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double kmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void key_ce(double& a, double& b, bool) {
  const double lo = kmin(a, b), hi = kmax(a, b);
  a = lo;
  b = hi;
}
template <int N>
__device__ __forceinline__ void bitonic_sort(double (&a)[N]) {  // ascending
#pragma unroll
  for (int i = 0; i < SortNet<N>::kSize; ++i) key_ce(a[SortNet<N>::kA[i]], a[SortNet<N>::kB[i]], true);
}
template <int N>
__device__ __forceinline__ void bitonic_merge(double (&a)[N]) {  // bitonic -> ascending
#pragma unroll
  for (int j = N >> 1; j > 0; j >>= 1) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int l = i ^ j;
      if (l > i) key_ce(a[i], a[l], true);
    }
  }
}

// Sorted (ascending) register top-K on such keys, candidates in ARBITRARY index order (grid searches).
template <int KC>
struct TopKF64 {
  double key[KC];
  __device__ __forceinline__ static double empty() { return __hiloint2double(0x7f800000, 0x7fffffff); }  // (+inf, INT_MAX)
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < KC; ++i) key[i] = empty();
  }
  __device__ __forceinline__ static double make(float d, int j) { return __hiloint2double(__float_as_int(d), j); }
  __device__ __forceinline__ unsigned worst_bits() const { return (unsigned)__double2hiint(key[KC - 1]); }
  // distance bits of the K-th best (1 <= K <= KC, runtime): what a query needs certified
  __device__ __forceinline__ unsigned kth_bits(int K) const {
    unsigned b = (unsigned)__double2hiint(key[KC - 1]);
    if (K < KC) {  // wave-uniform branch; a select chain (indexing the register array by K would go through scratch)
#pragma unroll
      for (int t = 0; t < KC - 1; ++t) b = (t == K - 1) ? (unsigned)__double2hiint(key[t]) : b;
    }
    return b;
  }
  // branch-free sorted insert: slot i takes min(key[i], max(key[i-1], k)); 2 KC - 1 FP64 min/max
  __device__ __forceinline__ void insert(double k) {
#pragma unroll
    for (int i = KC - 1; i > 0; --i) key[i] = kmin(key[i], kmax(key[i - 1], k));
    key[0] = kmin(key[0], k);
  }
  __device__ __forceinline__ float dist_at(int k) const { return __int_as_float(__double2hiint(key[k])); }
  __device__ __forceinline__ int idx_at(int k) const { return __double2loint(key[k]); }
};

}  // namespace pointops
