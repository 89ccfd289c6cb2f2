// knn_grid_d3w.hip -- instantiates the grid search kernels (knn_grid_search.h) for D = 3, clouds of more than 2^21 - 16 points
// (8-bit run lengths: knn_grid_search.h, kRunBitsBig).
#include "knn_grid_search.h"
#include "knn_grid_box.h"

namespace pointops {

void grid_search_d3w(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad) {
  if (norm == 1) grid_search_dispatch<3, 1, kRunBitsBig>(a, ws, kc, quad);
  else grid_search_dispatch<3, 2, kRunBitsBig>(a, ws, kc, quad);
}

}  // namespace pointops
