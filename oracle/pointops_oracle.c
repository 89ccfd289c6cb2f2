/*
 * pointops_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's CPU kernels for
 * the batched point-cloud neighbour hot path.  It exists only as the parity
 * checker (tests/, __graft_entry__.smoke()) and as bench.py's `cpu_baseline`
 * leg; nothing under pytorch3d_pointops_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  the tests/golden npz files were generated in the build
 * container from the reference itself (its Python wrappers imported from
 * /root/reference plus its CPU kernels compiled by oracle/build_ref.py) by
 * tests/golden/make_golden.py; tests/test_oracle.py checks every function
 * below bit-for-bit against those vectors.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  The
 * distance arithmetic must stay an unfused fp32 multiply followed by an add,
 * accumulated in d order from 0.0f, because the reference build has no -mfma
 * (SURVEY.md section 3.1); -ffp-contract=off guarantees that here.
 *
 * All tensors are dense row-major; lengths / indices are int64, point data fp32.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* (dist, idx) max-heap with the ordering of std::tuple<float,int>     */
/* reference: csrc/knn/knn_cpu.cpp:40 (std::priority_queue of tuples)  */
/* ------------------------------------------------------------------ */
typedef struct {
  float d;
  int i;
} heap_item;

static inline int item_less(heap_item a, heap_item b) {
  /* std::less<std::tuple<float,int>>: lexicographic */
  if (a.d < b.d) return 1;
  if (b.d < a.d) return 0;
  return a.i < b.i;
}

static void heap_push(heap_item* h, int* n, heap_item x) {
  int c = (*n)++;
  h[c] = x;
  while (c > 0) {
    int p = (c - 1) / 2;
    if (!item_less(h[p], h[c])) break;
    heap_item t = h[p];
    h[p] = h[c];
    h[c] = t;
    c = p;
  }
}

static heap_item heap_pop(heap_item* h, int* n) {
  heap_item top = h[0];
  int m = --(*n);
  h[0] = h[m];
  int p = 0;
  for (;;) {
    int l = 2 * p + 1, r = l + 1, b = p;
    if (l < m && item_less(h[b], h[l])) b = l;
    if (r < m && item_less(h[b], h[r])) b = r;
    if (b == p) break;
    heap_item t = h[p];
    h[p] = h[b];
    h[b] = t;
    p = b;
  }
  return top;
}

/*
 * K nearest neighbours.  Follows KNearestNeighborIdxCpu,
 * csrc/knn/knn_cpu.cpp:13-69:
 *   outputs pre-filled with 0 (:25-26); for n / i1<length1 / i2<length2 (:35-41);
 *   dist accumulated over d from 0 as diff*diff (L2) or |diff| (L1) (:42-50);
 *   insert when size<K or dist < top.dist, strict (:52); pop when size>=K (:54-56);
 *   drain largest-first into slot k=q.size() => ascending (dist, idx) (:59-65).
 * idxs: (N,P1,K) int64, dists: (N,P1,K) fp32, both fully written here.
 */
void oracle_knn_points_idx(const float* p1, const float* p2,
                           const int64_t* lengths1, const int64_t* lengths2,
                           int64_t N, int64_t P1, int64_t P2, int64_t D,
                           int norm, int64_t K, int64_t* idxs, float* dists) {
  memset(idxs, 0, sizeof(int64_t) * (size_t)(N * P1 * K));
  memset(dists, 0, sizeof(float) * (size_t)(N * P1 * K));
  if (K <= 0) return;
  heap_item* h = (heap_item*)malloc(sizeof(heap_item) * (size_t)(K + 1));
  for (int64_t n = 0; n < N; ++n) {
    const int64_t length1 = lengths1[n], length2 = lengths2[n];
    for (int64_t i1 = 0; i1 < length1; ++i1) {
      const float* a = p1 + (n * P1 + i1) * D;
      int size = 0;
      for (int64_t i2 = 0; i2 < length2; ++i2) {
        const float* b = p2 + (n * P2 + i2) * D;
        float dist = 0;
        for (int64_t d = 0; d < D; ++d) {
          float diff = a[d] - b[d];
          if (norm == 1) {
            dist += fabsf(diff);
          } else {
            dist += diff * diff;
          }
        }
        if (size < K || dist < h[0].d) {
          heap_item x = {dist, (int)i2};
          int was = size;
          heap_push(h, &size, x);
          if (was >= K) (void)heap_pop(h, &size);
        }
      }
      while (size > 0) {
        heap_item t = heap_pop(h, &size);
        dists[(n * P1 + i1) * K + size] = t.d;
        idxs[(n * P1 + i1) * K + size] = t.i;
      }
    }
  }
  free(h);
}

/*
 * KNN backward.  Follows KNearestNeighborBackwardCpu,
 * csrc/knn/knn_cpu.cpp:75-128: k runs to min(length2, K) (:101-102), idx == -1
 * skipped (:109-111), L2: diff = 2*g*(p1-p2), L1: g*sign with sign = p1>p2 ? 1 : -1
 * (:114-121); grad_p1 += diff, grad_p2 += -diff in (n, i1, k, d) order (:122-123).
 */
void oracle_knn_points_backward(const float* p1, const float* p2,
                                const int64_t* lengths1, const int64_t* lengths2,
                                const int64_t* idxs, const float* grad_dists,
                                int64_t N, int64_t P1, int64_t P2, int64_t D,
                                int64_t K, int norm, float* grad_p1, float* grad_p2) {
  memset(grad_p1, 0, sizeof(float) * (size_t)(N * P1 * D));
  memset(grad_p2, 0, sizeof(float) * (size_t)(N * P2 * D));
  for (int64_t n = 0; n < N; ++n) {
    const int64_t length1 = lengths1[n];
    int64_t length2 = lengths2[n];
    length2 = (length2 < K) ? length2 : K;
    for (int64_t i1 = 0; i1 < length1; ++i1) {
      for (int64_t k = 0; k < length2; ++k) {
        const int64_t i2 = idxs[(n * P1 + i1) * K + k];
        if (i2 == -1) continue;
        const float g = grad_dists[(n * P1 + i1) * K + k];
        for (int64_t d = 0; d < D; ++d) {
          const float a = p1[(n * P1 + i1) * D + d];
          const float b = p2[(n * P2 + i2) * D + d];
          float diff;
          if (norm == 1) {
            float sign = (a > b) ? 1.0f : -1.0f;
            diff = g * sign;
          } else {
            diff = 2.0f * g * (a - b);
          }
          grad_p1[(n * P1 + i1) * D + d] += diff;
          grad_p2[(n * P2 + i2) * D + d] += -1.0f * diff;
        }
      }
    }
  }
}

/*
 * Ball query.  Follows BallQueryCpu, csrc/ball_query/ball_query_cpu.cpp:12-54:
 * idxs pre-filled -1, dists 0 (:24-25); radius2 = radius*radius in fp32 (:26);
 * scan j<length2 while count<K, accept dist2 < radius2 strictly (:39-50).
 */
void oracle_ball_query(const float* p1, const float* p2, const int64_t* lengths1,
                       const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2,
                       int64_t D, int64_t K, float radius, int64_t* idxs,
                       float* dists) {
  for (int64_t t = 0; t < N * P1 * K; ++t) idxs[t] = -1;
  memset(dists, 0, sizeof(float) * (size_t)(N * P1 * K));
  const float radius2 = radius * radius;
  for (int64_t n = 0; n < N; ++n) {
    const int64_t length1 = lengths1[n], length2 = lengths2[n];
    for (int64_t i = 0; i < length1; ++i) {
      const float* a = p1 + (n * P1 + i) * D;
      int64_t count = 0;
      for (int64_t j = 0; j < length2 && count < K; ++j) {
        const float* b = p2 + (n * P2 + j) * D;
        float dist2 = 0;
        for (int64_t d = 0; d < D; ++d) {
          float diff = a[d] - b[d];
          dist2 += diff * diff;
        }
        if (dist2 < radius2) {
          dists[(n * P1 + i) * K + count] = dist2;
          idxs[(n * P1 + i) * K + count] = j;
          ++count;
        }
      }
    }
  }
}

/*
 * Farthest point sampling.  Follows FarthestPointSamplingCpu,
 * csrc/sample_farthest_points/sample_farthest_points_cpu.cpp:14-103:
 * out (N, max_K) pre-filled -1 (:28); per cloud dists=FLT_MAX, mask=false (:45-50);
 * idx[n][0]=start (:53-57); batch_k=min(length,K[n]) (:62); each iteration: selected
 * points get dist 0, others min(dist, dist2-to-last) with strict < (:67-87); next =
 * FIRST maximum (std::max_element, :91-92).
 * lengths[n]==0 is undefined behaviour in the reference (:57); here such clouds are
 * skipped (row stays -1) -- not pinned.
 */
void oracle_sample_farthest_points(const float* points, const int64_t* lengths,
                                   const int64_t* K, const int64_t* start_idxs,
                                   int64_t N, int64_t P, int64_t D, int64_t max_K,
                                   int64_t* out) {
  for (int64_t t = 0; t < N * max_K; ++t) out[t] = -1;
  float* dists = (float*)malloc(sizeof(float) * (size_t)(P > 0 ? P : 1));
  unsigned char* mask = (unsigned char*)malloc((size_t)(P > 0 ? P : 1));
  for (int64_t n = 0; n < N; ++n) {
    const int64_t len = lengths[n];
    if (len <= 0 || max_K <= 0) continue;
    for (int64_t p = 0; p < len; ++p) {
      dists[p] = FLT_MAX;
      mask[p] = 0;
    }
    int64_t last = start_idxs[n];
    out[n * max_K + 0] = last;
    mask[last] = 1;
    const int64_t batch_k = len < K[n] ? len : K[n];
    for (int64_t k = 1; k < batch_k; ++k) {
      const float* a = points + (n * P + last) * D;
      for (int64_t p = 0; p < len; ++p) {
        if (mask[p]) {
          dists[p] = 0.0f;
          continue;
        }
        const float* b = points + (n * P + p) * D;
        float dist2 = 0.0f;
        for (int64_t d = 0; d < D; ++d) {
          float diff = a[d] - b[d];
          dist2 += diff * diff;
        }
        if (dist2 < dists[p]) dists[p] = dist2;
      }
      int64_t best = 0;
      for (int64_t p = 1; p < len; ++p) {
        if (dists[best] < dists[p]) best = p; /* first maximum wins */
      }
      last = best;
      out[n * max_K + k] = last;
      mask[last] = 1;
    }
  }
  free(dists);
  free(mask);
}

/*
 * packed -> padded.  Follows PackedToPaddedCpu,
 * csrc/packed_to_padded_tensor/packed_to_padded_tensor_cpu.cpp:11-40: output zeroed
 * (:21-22); rows [first_idxs[b], first_idxs[b+1]) (last cloud: num_inputs) copied to
 * padded[b, 0:num] (:29-38).  No clamp of num to max_size in the reference; callers
 * must respect it (here rows beyond max_size are dropped to stay memory safe).
 */
void oracle_packed_to_padded(const float* packed, const int64_t* first_idxs,
                             int64_t num_inputs, int64_t batch, int64_t max_size,
                             int64_t D, float* padded) {
  memset(padded, 0, sizeof(float) * (size_t)(batch * max_size * D));
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t start = first_idxs[b];
    const int64_t end = b + 1 < batch ? first_idxs[b + 1] : num_inputs;
    int64_t num = end - start;
    if (num > max_size) num = max_size;
    for (int64_t i = 0; i < num; ++i)
      for (int64_t j = 0; j < D; ++j)
        padded[(b * max_size + i) * D + j] = packed[(start + i) * D + j];
  }
}

/* padded -> packed.  Follows PaddedToPackedCpu, same file :42-70. */
void oracle_padded_to_packed(const float* padded, const int64_t* first_idxs,
                             int64_t num_inputs, int64_t batch, int64_t max_size,
                             int64_t D, float* packed) {
  memset(packed, 0, sizeof(float) * (size_t)(num_inputs * D));
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t start = first_idxs[b];
    const int64_t end = b + 1 < batch ? first_idxs[b + 1] : num_inputs;
    int64_t num = end - start;
    if (num > max_size) num = max_size;
    for (int64_t i = 0; i < num; ++i)
      for (int64_t j = 0; j < D; ++j)
        packed[(start + i) * D + j] = padded[(b * max_size + i) * D + j];
  }
}

/*
 * Inverse-CDF sampling.  Follows SamplePdfCpu_worker, csrc/sample_pdf/sample_pdf_cpu.cpp:23-99,
 * in the build the reference ships (`#define USE_BINARY_SEARCH`, :19): inclusive fp32 partial
 * sums of the bin weights in bin order (:52-58), total += eps (:59); per sample
 * uniform = total * u (:68); i_bin = std::lower_bound over the first n_bins-1 partial sums
 * (first partial sum that is not < uniform, else n_bins-1) (:70-72); uniform -= partial_sums[i_bin-1]
 * when i_bin > 0 (:73-75); linear interpolation inside the bin with the `uniform > w` and
 * `w > eps` guards (:87-95).  In place on `outputs`.
 */
void oracle_sample_pdf(const float* bins, const float* weights, float* outputs, int64_t batch,
                       int64_t n_bins, int64_t n_samples, float eps) {
  float* partial = (float*)malloc(sizeof(float) * (size_t)(n_bins > 0 ? n_bins : 1));
  for (int64_t b = 0; b < batch; ++b) {
    const float* bin = bins + b * (n_bins + 1);
    const float* w = weights + b * n_bins;
    float total = 0;
    for (int64_t i = 0; i < n_bins; ++i) {
      total += w[i];
      partial[i] = total;
    }
    total += eps;
    for (int64_t s = 0; s < n_samples; ++s) {
      float* o = outputs + b * n_samples + s;
      float uniform = total * *o;
      int64_t lo = 0, hi = n_bins - 1; /* lower_bound on [0, n_bins-1) */
      while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (partial[mid] < uniform) lo = mid + 1;
        else hi = mid;
      }
      const int64_t i = lo;
      if (i > 0) uniform -= partial[i - 1];
      const float bin_start = bin[i], bin_end = bin[i + 1], bin_weight = w[i];
      float v = bin_start;
      if (uniform > bin_weight) {
        v = bin_end;
      } else if (bin_weight > eps) {
        v += (uniform / bin_weight) * (bin_end - bin_start);
      }
      *o = v;
    }
  }
  free(partial);
}
