// knn_grid_d3.hip -- instantiates the grid search kernels (knn_grid_search.h) for D = 3.
#include "knn_grid_search.h"
#include "knn_grid_box.h"

namespace pointops {

void grid_search_d3(const KnnArgs& a, const GridWs& ws, int norm, int kc, bool quad) {
  if (norm == 1) grid_search_dispatch<3, 1, kRunBitsStd>(a, ws, kc, quad);
  else grid_search_dispatch<3, 2, kRunBitsStd>(a, ws, kc, quad);
}

}  // namespace pointops
