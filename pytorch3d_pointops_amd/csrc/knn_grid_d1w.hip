// knn_grid_d1w.hip -- instantiates the grid search kernels (knn_grid_search.h) for D = 1, clouds of more than 2^21 - 16 points
// (8-bit run lengths: knn_grid_search.h, kRunBitsBig).
#include "knn_grid_search.h"
#include "knn_grid_box.h"

namespace pointops {

void grid_search_d1w(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad) {
  if (norm == 1) grid_search_dispatch<1, 1, kRunBitsBig>(a, ws, kc, quad);
  else grid_search_dispatch<1, 2, kRunBitsBig>(a, ws, kc, quad);
}

}  // namespace pointops
