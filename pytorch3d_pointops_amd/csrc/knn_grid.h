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

// brute-force register-top-K scan (knn.hip); honours a.qlist / a.qcount
void launch_knn_bruteforce(const KnnArgs& a, int norm);

// exact grid search (knn_grid.hip)
size_t knn_grid_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t K);
int knn_grid_run(const KnnArgs& a, int norm, void* workspace);

}  // namespace pointops
