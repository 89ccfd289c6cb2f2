// knn.hip -- exact brute-force K nearest neighbours for gfx950 (MI355X).
//
// Replaces KNearestNeighborIdx (reference: csrc/knn/knn.h:59-80) with the CPU
// path's semantics (csrc/knn/knn_cpu.cpp:13-69): the K smallest (dist, idx)
// pairs per query in ascending lexicographic order, unfused fp32 arithmetic,
// zeros for padded rows/slots.  This is NOT the reference's CUDA design (fixed
// 256x256 grid, every thread streaming p2 from global memory, MinK that evicts
// the lower index on ties): see DESIGN.md section "KNN".
//
// Design (one lane = one query point, wave64):
//   * the query point lives in VGPRs; every lane of a wave consumes the SAME p2
//     point, so p2 is fetched through the scalar path (s_load_dwordx8/x16 from a
//     wave-uniform address into SGPRs, served by the scalar cache / L2) and used
//     directly as the SGPR operand of v_sub_f32 -- no LDS traffic, no VGPRs, no
//     per-lane address arithmetic for the streamed cloud;
//   * the running top-K is a sorted register array (K compile-time).  A candidate
//     is compared once against the K-th best (strict <, so an equal-distance
//     newcomer -- which always has the larger index because p2 is scanned in
//     index order -- never displaces: exactly std::priority_queue<tuple> order);
//     the insert shifts with v_cndmask chains and places the newcomer AFTER equal
//     keys, so the list is always in (dist, idx) order and is emitted as is: the
//     reference's separate sort + gather pass (functions/knn.py:77-89) is not needed;
//   * grid = clouds x ceil(P1/256) workgroups of 256 lanes (>> 256 CUs at the
//     bench sizes); all workgroups of a cloud re-stream the same 12*P2 bytes,
//     which stay L2/Infinity-Cache resident (786 KB per cloud at P2=65536).
#include "debug.h"
#include "knn_common.h"

#include <algorithm>
#include <cstdlib>
#include "knn_grid.h"

namespace pointops {

// ---------------------------------------------------------------------------
// Register kernel: D in [1,8], K <= KC <= 32.  One lane per query, whole-cloud scan.
// With `qlist` != nullptr the lanes of workgroup (n, tile) take their queries from
// qlist[n*P1 + tile*256 + lane] for lane < qcount[n] (the exact fallback pass of the
// grid search, knn_grid.hip); otherwise lane = query row and padded rows are zeroed.
// ---------------------------------------------------------------------------
template <int D, int KC, int NORM>
__global__ __launch_bounds__(kKnnBlock) void knn_reg_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2, int P1, int P2,
    int K, int tiles_per_cloud, const int* __restrict__ qlist, const int* __restrict__ qcount, int S,
    unsigned long long* __restrict__ partial, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  const int n = blockIdx.x / tiles_per_cloud;  // wave-uniform
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int split = blockIdx.y;  // slice of p2 (S > 1: partial lists, merged by knn_merge_kernel)
  int i = tile * (int)blockDim.x + threadIdx.x;  // 256 lanes, or 64 for small batches
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  if (len2 < 0) len2 = 0;
  if (qlist != nullptr) {
    if (i >= qcount[n]) return;
    i = qlist[(int64_t)n * P1 + i];
  } else {
    if (i >= P1) return;
    if (i >= (int)lengths1[n]) {  // padded query row: zeros (knn_cpu.cpp:25-26)
      if (S > 1) return;          // the merge pass writes the row
      int64_t* __restrict__ zi = idxs + ((int64_t)n * P1 + i) * K;
      float* __restrict__ zd = dists + ((int64_t)n * P1 + i) * K;
      for (int k = 0; k < K; ++k) {
        zi[k] = 0;
        zd[k] = 0.0f;
      }
      return;
    }
  }
  const int64_t row = (int64_t)n * P1 + i;
  float a[D];
#pragma unroll
  for (int d = 0; d < D; ++d) a[d] = p1[row * D + d];
  TopK<KC> top;
  top.init();
  if (S == 1) {
    scan_cloud<D, KC, NORM>(a, p2 + (int64_t)n * P2 * D, 0, len2, top);
    write_row<KC>(top, K, len2, idxs + row * K, dists + row * K);
    return;
  }
  const int jbeg = (int)((int64_t)len2 * split / S), jend = (int)((int64_t)len2 * (split + 1) / S);
  scan_cloud<D, KC, NORM>(a, p2 + (int64_t)n * P2 * D, jbeg, jend, top);
  // partial list of this slice as (dist bits, idx) keys; empty slots order last
  unsigned long long* __restrict__ o = partial + (row * S + split) * K;
  const int have = min(K, jend - jbeg);
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (k < K) o[k] = k < have ? TopKLex<KC>::make(top.dk[k], top.ik[k]) : TopKLex<KC>::kEmpty;
}

// merge of the S partial lists of a query (each ascending, slices in index order): one lane per
// query, 64-bit lexicographic keys, early exit per list at the first key that cannot enter
template <int KC>
__global__ __launch_bounds__(256) void knn_merge_kernel(
    const unsigned long long* __restrict__ partial, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, int P1, int P2, int K, int S, int64_t total, int64_t* __restrict__ idxs,
    float* __restrict__ dists) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= total) return;
  const int n = (int)(row / P1), i = (int)(row - (int64_t)n * P1);
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  if (len2 < 0) len2 = 0;
  const bool live = i < (int)lengths1[n];
  TopKLex<KC> top;
  top.init();
  if (live) {
    for (int s = 0; s < S; ++s) {
      const unsigned long long* __restrict__ src = partial + (row * S + s) * K;
      for (int k = 0; k < K; ++k) {
        const unsigned long long key = src[k];
        if (!(key < top.key[KC - 1])) break;
        top.insert(key);
      }
    }
  }
  write_row<KC>(top, K, live ? len2 : 0, idxs + row * K, dists + row * K);
}

// ---------------------------------------------------------------------------
// Generic kernel: any D, any K.  The sorted list lives in the output rows
// themselves (private to the thread), the query is re-read from global memory
// (L1-resident).  Correctness fallback for D > 8 or K > 32.
// ---------------------------------------------------------------------------
template <int NORM>
__global__ __launch_bounds__(kKnnBlock) void knn_generic_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2, int P1, int P2,
    int D, int K, int tiles_per_cloud, int64_t* __restrict__ idxs, float* __restrict__ dists) {
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int i = tile * kKnnBlock + threadIdx.x;
  if (i >= P1) return;
  const int len1 = (int)lengths1[n];
  int len2 = (int)lengths2[n];
  if (len2 > P2) len2 = P2;
  if (len2 < 0) len2 = 0;
  const int64_t row = (int64_t)n * P1 + i;
  int64_t* orow_i = idxs + row * K;
  float* orow_d = dists + row * K;
  int cnt = 0;
  if (i < len1) {
    const float* a = p1 + row * D;
    const float* q = p2 + (int64_t)n * P2 * D;
    float worst = __builtin_inff();
    for (int j = 0; j < len2; ++j) {
      const float* b = q + (int64_t)j * D;
      float acc = 0.0f;
      for (int d = 0; d < D; ++d) {
        const float diff = a[d] - b[d];
        acc = (NORM == 1) ? (acc + __builtin_fabsf(diff)) : (acc + diff * diff);
      }
      if (cnt < K || acc < worst) {
        int pos = cnt < K ? cnt : K - 1;  // slot that is freed / appended
        while (pos > 0 && acc < orow_d[pos - 1]) {
          orow_d[pos] = orow_d[pos - 1];
          orow_i[pos] = orow_i[pos - 1];
          --pos;
        }
        orow_d[pos] = acc;
        orow_i[pos] = j;
        if (cnt < K) ++cnt;
        if (cnt == K) worst = orow_d[K - 1];
      }
    }
  }
  for (int k = cnt; k < K; ++k) {
    orow_i[k] = 0;
    orow_d[k] = 0.0f;
  }
}

// ---------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------
struct RegLaunch {
  int block, tiles, S;
  unsigned long long* partial;
};

template <int D, int KC, int NORM>
static void launch_reg(const KnnArgs& a, const RegLaunch& r) {
  const dim3 grid((unsigned)(a.N * r.tiles), (unsigned)r.S);
  hipLaunchKernelGGL((knn_reg_kernel<D, KC, NORM>), grid, dim3(r.block), 0, a.stream, a.p1, a.p2, a.l1, a.l2, a.P1,
                     a.P2, a.K, r.tiles, a.qlist, a.qcount, r.S, r.partial, a.idxs, a.dists);
}

template <int D, int NORM>
static void dispatch_k(const KnnArgs& a, const RegLaunch& r) {
  const int K = a.K;
  if (K <= 1) launch_reg<D, 1, NORM>(a, r);
  else if (K <= 2) launch_reg<D, 2, NORM>(a, r);
  else if (K <= 4) launch_reg<D, 4, NORM>(a, r);
  else if (K <= 8) launch_reg<D, 8, NORM>(a, r);
  else if (K <= 16) launch_reg<D, 16, NORM>(a, r);
  else if (K <= 24) launch_reg<D, 24, NORM>(a, r);
  else launch_reg<D, 32, NORM>(a, r);
}

template <int NORM>
static void dispatch_d(const KnnArgs& a, const RegLaunch& r) {
  switch (a.D) {
    case 1: dispatch_k<1, NORM>(a, r); break;
    case 2: dispatch_k<2, NORM>(a, r); break;
    case 3: dispatch_k<3, NORM>(a, r); break;
    case 4: dispatch_k<4, NORM>(a, r); break;
    case 5: dispatch_k<5, NORM>(a, r); break;
    case 6: dispatch_k<6, NORM>(a, r); break;
    case 7: dispatch_k<7, NORM>(a, r); break;
    case 8: dispatch_k<8, NORM>(a, r); break;
    default: break;
  }
}

int knn_split_count(int64_t N, int64_t P1, int64_t P2, int64_t K) {
  if (K > 32) return 1;
  const int64_t waves = N * ceil_div(P1, 64);
  if (waves >= 2048 || P2 < 512) return 1;
  int64_t s = ceil_div(4096, waves);
  s = std::min<int64_t>(s, 8);
  s = std::min<int64_t>(s, P2 / 128);  // a slice keeps >= 128 candidates (cfg1, B=2 N=M=1024 K=8, us per call by (max
                                       // slices, candidates): (8, 256) 48.6, (8, 128) 37.5, (16, 64) 41.4, (32, 32) 56.0)
  return (int)std::max<int64_t>(s, 1);
}

size_t knn_split_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K) {
  const int S = knn_split_count(N, P1, P2, K);
  return S > 1 ? sizeof(unsigned long long) * (size_t)(N * P1 * S * K) : 0;
}

void knn_merge_partials(const KnnArgs& a, int S, const void* workspace) {
  const int64_t total = a.N * a.P1;
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  const unsigned long long* ws = (const unsigned long long*)workspace;
#define PO_MERGE(KC)                                                                                              \
  hipLaunchKernelGGL(knn_merge_kernel<KC>, grid, block, 0, a.stream, ws, a.l1, a.l2, a.P1, a.P2, a.K, S, total, \
                     a.idxs, a.dists)
  const int K = a.K;
  if (K <= 1) PO_MERGE(1);
  else if (K <= 2) PO_MERGE(2);
  else if (K <= 4) PO_MERGE(4);
  else if (K <= 8) PO_MERGE(8);
  else if (K <= 16) PO_MERGE(16);
  else if (K <= 24) PO_MERGE(24);
  else PO_MERGE(32);
#undef PO_MERGE
}

void launch_knn_bruteforce(const KnnArgs& a, int norm, void* workspace) {
  // small batches: 64-lane workgroups (4x as many) and p2 slices; the query-list mode of the grid
  // fallback always scans whole clouds
  RegLaunch r{kKnnBlock, a.tiles, 1, nullptr};
  if (a.qlist == nullptr && a.N * a.tiles < 2048) {
    r.block = 64;
    r.tiles = (int)ceil_div(a.P1, 64);
    if (workspace != nullptr) {
      r.S = knn_split_count(a.N, a.P1, a.P2, a.K);
      r.partial = (unsigned long long*)workspace;
    }
  }
  if (norm == 1) dispatch_d<1>(a, r);
  else dispatch_d<2>(a, r);
  if (r.S > 1) knn_merge_partials(a, r.S, workspace);
}

}  // namespace pointops

using namespace pointops;

extern "C" {

int pointops_knn_check_version(int version, int64_t D, int64_t K) {
  // Kernel families of this library.  As in the reference (csrc/knn/knn.cu:292-312) the
  // highest valid version is the fastest and `version` never changes results:
  //   0 any D, any K: LDS-transposed queries with register (K <= 32) or LDS lists (knn_wide.hip);
  //     beyond the LDS budget the plain generic kernel (list kept in the output rows)
  //   1, 2 register top-K brute-force scan (D in [1,8], K in [1,32])
  //   3 exact grid-pruned search + brute-force fallback (D in [1,3], K in [1,128]: lane-private lists up to 64,
  //     wave-per-query sorting above)
  if (version == 0) return 1;
  if (version == 1 || version == 2) return (D >= 1 && D <= 8 && K >= 1 && K <= 32) ? 1 : 0;
  if (version == 3) return (D >= 1 && D <= 3 && K >= 1 && K <= 128) ? 1 : 0;
  return 0;
}

static int choose_version(int version, int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K) {
  // clouds of up to 2^24 - 16 points (24-bit record indices in the searches' run words); any batch size (slices)
  const bool grid_ok = pointops_knn_check_version(3, D, K) && P2 <= knn_grid_max_points();
  if (version == 3 && !grid_ok) version = -1;
  if (version >= 0 && version <= 3 && pointops_knn_check_version(version, D, K)) return version;
  // auto: the grid only pays once the all-pairs scan is longer than the grid's floor of ~12 launches (65 us at K=1,
  // 80 at K=8, 120-130 at K=32).  Round 3's wave-per-query scan (knn_small.hip) moved the crossover up for lists of
  // 8-16 slots (4096 x 4096, K=8: scan 25 us, grid 75; 4 x 4096 x 4096: 72 / 80; 2 x 8192 x 8192: 107 / 80) and made
  // few-query shapes cheap enough to need a rule of their own: with fewer than 4096 queries the scan runs on a
  // part-filled chip, so the pairs are counted as if there were 4096 (1024 x 65536: scan 80-200 us, grid 69-131;
  // 512 x 262144: 297-492 / 74-139).  profiles/r03_knn_small_sweep.jsonl, r02_knn_crossover.txt.
  const double pairs = (double)std::max<int64_t>(N * P1, 4096) * (double)P2;
  const double cross = K <= 2 ? 1.5 * (double)(1LL << 27)    // K=1: 2 x 8192^2 scan 52 / grid 68; 16384^2 100 / 67
                       : K <= 8 ? 1.5 * (double)(1LL << 26)  // K=8: 4 x 4096^2 69 / 80; 2 x 8192^2 99 / 79
                       : K <= 16 ? (double)(1LL << 25)
                       : K <= 32 ? 1.5 * (double)(1LL << 24)  // K=32: 4096^2 94 / 118; 4 x 4096^2 301 / 127
                                 : (double)(1LL << 24);
  if (grid_ok && P2 >= 4096 && pairs >= cross) return 3;
  if (pointops_knn_check_version(2, D, K)) return 2;
  return 0;
}

size_t pointops_knn_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K,
                                    int version) {
  if (N <= 0 || P1 <= 0 || D < 1 || K < 1) return 0;
  const int v = choose_version(version, N, P1, P2, D, K);
  if (v == 0 && !knn_wide_supported(D, K)) return 0;
  if ((v == 1 || v == 2) && knn_small_applies(N, P1, P2, D, K)) return 0;
  if (v != 3) return knn_split_workspace_bytes(N, P1, P2, K);
  return knn_grid_workspace_bytes(N, P1, P2, K);
}

int pointops_knn_points_idx(const float* p1, const float* p2, const int64_t* lengths1,
                            const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2, int64_t D,
                            int norm, int64_t K, int version, int64_t* idxs, float* dists,
                            void* workspace, size_t workspace_bytes, void* stream) {
  return pointops_knn_points_idx_reuse(p1, p2, lengths1, lengths2, N, P1, P2, D, norm, K, version, idxs, dists,
                                       workspace, workspace_bytes, 0, stream);
}

int pointops_knn_uses_grid(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K, int version) {
  if (N <= 0 || P1 <= 0 || D < 1 || K < 1) return 0;
  return choose_version(version, N, P1, P2, D, K) == 3 ? 1 : 0;
}

int pointops_knn_points_idx_reuse(const float* p1, const float* p2, const int64_t* lengths1,
                                  const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                  int norm, int64_t K, int version, int64_t* idxs, float* dists,
                                  void* workspace, size_t workspace_bytes, int reuse, void* stream) {
  POINTOPS_REQUIRE(reuse >= 0 && reuse <= 2, "knn_points_idx: reuse must be 0, 1 or 2");
  POINTOPS_REQUIRE(norm == 1 || norm == 2, "knn_points_idx: norm must be 1 or 2 (got %d)", norm);
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && K >= 1,
                   "knn_points_idx: bad sizes N=%lld P1=%lld P2=%lld D=%lld K=%lld", (long long)N,
                   (long long)P1, (long long)P2, (long long)D, (long long)K);
  POINTOPS_REQUIRE(P1 < (1LL << 31) && P2 < (1LL << 31) && K < (1LL << 20) && D < (1LL << 16),
                   "knn_points_idx: P1/P2 must fit int32");
  if (N == 0 || P1 == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(p1 && p2 && lengths1 && lengths2 && idxs && dists,
                   "knn_points_idx: null pointer");
  KnnArgs a;
  a.p1 = p1; a.p2 = p2; a.l1 = lengths1; a.l2 = lengths2;
  a.P1 = (int)P1; a.P2 = (int)P2; a.D = (int)D; a.K = (int)K; a.N = N;
  a.tiles = (int)ceil_div(P1, kKnnBlock);
  a.qlist = nullptr; a.qcount = nullptr;
  a.idxs = idxs; a.dists = dists; a.stream = (hipStream_t)stream;
  POINTOPS_REQUIRE(N * a.tiles < (1LL << 31), "knn_points_idx: grid too large");

  const int v = choose_version(version, N, P1, P2, D, K);
  if (v == 3) {
    const size_t need = knn_grid_workspace_bytes(N, P1, P2, K);
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("knn_points_idx: workspace of %zu bytes required (got %zu)", need, workspace_bytes);
      return POINTOPS_EWORKSPACE;
    }
    const int rc = knn_grid_run(a, norm, workspace, reuse);
    if (rc != POINTOPS_OK) return rc;
  } else if (v == 0 && knn_wide_supported(D, K) && debug_knob("knn_generic", 0) == 0) {
    // any D / long lists: LDS-transposed queries (POINTOPS_KNN_GENERIC=1 keeps the plain fallback, tests)
    const size_t need = knn_split_workspace_bytes(N, P1, P2, K);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
      set_error("knn_points_idx: workspace of %zu bytes required (got %zu)", need, workspace_bytes);
      return POINTOPS_EWORKSPACE;
    }
    const int rc = launch_knn_wide(a, norm, workspace);
    if (rc != POINTOPS_OK) return rc;
  } else if (v == 0) {
    const dim3 grid((unsigned)(N * a.tiles));
    if (norm == 1)
      hipLaunchKernelGGL(knn_generic_kernel<1>, grid, dim3(kKnnBlock), 0, a.stream, p1, p2, lengths1,
                         lengths2, a.P1, a.P2, a.D, a.K, a.tiles, idxs, dists);
    else
      hipLaunchKernelGGL(knn_generic_kernel<2>, grid, dim3(kKnnBlock), 0, a.stream, p1, p2, lengths1,
                         lengths2, a.P1, a.P2, a.D, a.K, a.tiles, idxs, dists);
  } else if (knn_small_applies(N, P1, P2, D, K)) {
    launch_knn_small(a, norm);  // few queries: one wave per query (knn_small.hip)
  } else {
    const size_t need = knn_split_workspace_bytes(N, P1, P2, K);
    launch_knn_bruteforce(a, norm, workspace != nullptr && workspace_bytes >= need && need > 0 ? workspace : nullptr);
  }
  return check_launch("knn_points_idx");
}

}  // extern "C"
