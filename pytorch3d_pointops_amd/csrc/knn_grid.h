// knn_grid.h -- host-side interface between knn.hip (dispatch, brute-force kernels)
// and knn_grid.hip (exact grid-pruned search).
#pragma once
#include "common.h"

namespace pointops {

struct KnnArgs {
  const float *p1, *p2;
  const int64_t *l1, *l2;
  int P1, P2, D, K, tiles;
  int64_t N;
  const int* qlist;   // optional per-cloud query lists (N x P1) for the fallback pass
  const int* qcount;  // (N,) entries used in each list
  int64_t* idxs;
  float* dists;
  hipStream_t stream;
};

// brute-force register-top-K scan (knn.hip); honours a.qlist / a.qcount.  With `workspace` (of
// knn_split_workspace_bytes) a small batch is scanned in p2 slices and merged (knn_split_count > 1).
void launch_knn_bruteforce(const KnnArgs& a, int norm, void* workspace = nullptr);

// p2 slices per query so that a small batch still fills the chip (1024 SIMDs): S partial lists per
// query (64-bit (dist, idx) keys) are merged by knn_merge_partials
int knn_split_count(int64_t N, int64_t P1, int64_t P2, int64_t K);
size_t knn_split_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K);
void knn_merge_partials(const KnnArgs& a, int S, const void* workspace);

// few queries (knn_small.hip): one wave per query, the cloud dealt over the lanes; D <= 8, K <= 32, no workspace
bool knn_small_applies(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K);
void launch_knn_small(const KnnArgs& a, int norm);

// brute-force scan for any D / long lists (knn_wide.hip): LDS-transposed queries, register or LDS lists
bool knn_wide_supported(int64_t D, int64_t K);
int launch_knn_wide(const KnnArgs& a, int norm, void* workspace);

// long lists, 64 < K <= 128: wave-per-query search of the 3x3x3 cube with one 2048-key sort (knn_grid_wsort.hip);
// uncertified queries are appended to ws.fb2_list
struct GridWs;
void grid_search_wsort(const KnnArgs& a, const GridWs& ws, int norm);

// exact grid search (knn_grid.hip); clouds of up to knn_grid_max_points() points, any batch size
int64_t knn_grid_max_points();
size_t knn_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K);
// reuse: 0 = build everything; 1 = the workspace still holds the point side (p2, lengths2) of the previous call;
// 2 = and the query side (p1, lengths1) too
int knn_grid_run(const KnnArgs& a, int norm, void* workspace, int reuse = 0);

// ball query for few queries (ball_small.hip): one wave per query, any D, any K, no workspace
bool ball_small_applies(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K);
void launch_ball_small(const float* p1, const float* p2, const int64_t* lengths1, const int64_t* lengths2, int64_t N,
                       int64_t P1, int64_t P2, int64_t D, int64_t K, float radius2, int64_t* idxs, float* dists,
                       hipStream_t stream);

// ball query through the same grid (knn_grid.hip); see ball_query.hip for the operator
size_t ball_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2);
int ball_grid_run(const KnnArgs& a, float radius, void* workspace, const int** flag, const int** qcount,
                  const int** qlist);

}  // namespace pointops
