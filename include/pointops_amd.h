/*
 * pointops_amd.h -- C ABI of libpointops_amd.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the batched point-cloud neighbour hot path of
 * pytorch3d_pointops.  Each entry point replaces one operator of the reference's
 * pybind11 module `pytorch3d_pointops._C` (reference: csrc/ext.cpp:15-27); the
 * argument meaning, padding conventions and output layout are the reference's.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch CUDA-HIP tensor storage),
 *     dense row-major, fp32 for point data and int64 for lengths / indices;
 *   - inputs are borrowed and never written; outputs are FULLY written by the call
 *     (padding included), so callers may pass uninitialised buffers;
 *   - `stream` is a hipStream_t (NULL = default stream); calls enqueue work and
 *     return without synchronising (the reference does the same on its current
 *     stream: csrc/knn/knn.cu:331);
 *   - the calling thread's current HIP device must own the buffers;
 *   - return value: 0 on success, a negative POINTOPS_E* code otherwise;
 *     pointops_last_error() gives a thread-local message.
 *   - no torch types, no C++ types, no global state besides the error string.
 */
#ifndef POINTOPS_AMD_H_
#define POINTOPS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POINTOPS_OK 0
#define POINTOPS_EINVAL (-1)      /* bad argument (shape, norm, K ...)           */
#define POINTOPS_ELAUNCH (-2)     /* HIP launch / runtime error                  */
#define POINTOPS_EWORKSPACE (-3)  /* workspace too small                         */

#define POINTOPS_ABI_VERSION 1

/* Library identification: ABI version and the ISA the kernels were built for. */
int pointops_abi_version(void);
const char* pointops_target_arch(void); /* "gfx950" */
const char* pointops_last_error(void);

/*
 * K nearest neighbours -- replaces `_C.knn_points_idx`
 * (reference: csrc/knn/knn.h:59-80, CPU semantics csrc/knn/knn_cpu.cpp:13-69).
 *   p1 (N,P1,D), p2 (N,P2,D), lengths1 (N,), lengths2 (N,), norm in {1,2}, K >= 1.
 *   idxs (N,P1,K) int64 and dists (N,P1,K) fp32: for each query the K smallest
 *   (dist, idx) pairs in ascending lexicographic order; rows >= lengths1[n] and
 *   slots >= lengths2[n] are 0 / 0.0f.  Distances are unfused fp32
 *   ((dx*dx + dy*dy) + dz*dz), i.e. bit-equal to the reference CPU path.
 *   `version` is accepted for signature compatibility (reference: knn.h:45-57):
 *   -1 = auto; 0..3 select a kernel family when valid, exactly like the reference
 *   they never change results.
 *   workspace: pointops_knn_workspace_bytes() bytes of device scratch (may be NULL
 *   when that returns 0).
 */
size_t pointops_knn_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D,
                                    int64_t K, int version);
int pointops_knn_points_idx(const float* p1, const float* p2, const int64_t* lengths1,
                            const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2,
                            int64_t D, int norm, int64_t K, int version, int64_t* idxs,
                            float* dists, void* workspace, size_t workspace_bytes,
                            void* stream);

/*
 * The same operator REUSING the cell grid a previous call left in `workspace` (the grid-pruned family only:
 * pointops_knn_uses_grid(...) != 0 for the shape; other families ignore `reuse`).  The reference has no counterpart --
 * its operator (csrc/knn/knn.h:59-66) is stateless -- so this is an extension for callers that query the same target
 * cloud repeatedly (chamfer against a fixed ground truth, knn_points followed by more queries of the same cloud):
 *   reuse = 0  build everything (= pointops_knn_points_idx);
 *   reuse = 1  p2, lengths2, N, P1, P2, D, K and version are those of the previous call into this workspace and the
 *              bytes of p2 / lengths2 are unchanged: the bounding boxes, cell tables, the sort of p2 and its refined
 *              cells are reused, only the queries are sorted;
 *   reuse = 2  p1 and lengths1 are unchanged too: no build pass runs at all.
 * The caller vouches for these conditions; results equal reuse = 0 bit for bit when they hold.
 */
int pointops_knn_uses_grid(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K, int version);
int pointops_knn_points_idx_reuse(const float* p1, const float* p2, const int64_t* lengths1,
                                  const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                  int norm, int64_t K, int version, int64_t* idxs, float* dists,
                                  void* workspace, size_t workspace_bytes, int reuse, void* stream);

/*
 * Diagnostics for the grid-pruned KNN family (version 3): after a pointops_knn_points_idx
 * call that used `workspace`, copies the per-cloud number of queries that the pruning
 * bound could not certify (and that were answered by the whole-cloud scan instead) into
 * counts (2,N) int32 on the device: row 0 = queries re-searched wave-per-query on a growing
 * cell cube, row 1 = queries that ended in the whole-cloud scan.  Stream-ordered, no sync.
 */
int pointops_knn_grid_fallback_counts(const void* workspace, int64_t N, int64_t P1, int64_t P2,
                                      int64_t K, int32_t* counts, void* stream);

/*
 * Diagnostics, same contract: stats (N,14) int32 on the device = cells per dimension G[0..2], cell count,
 * 1 if the cloud was searched through a grid, queries uncertified after the lane pass / after the quad and box passes /
 * sent to the whole-cloud scan, queries deferred to the box search, refined cells, bins of the point sort / of the
 * query sort, crowded bins of the point sort / of the query sort.  (No reference counterpart; used by tools/ and the
 * distribution benchmarks.)
 */
int pointops_knn_grid_stats(const void* workspace, int64_t N, int64_t P1, int64_t P2, int64_t K,
                            int32_t* stats, void* stream);

/* Replaces `_C.knn_check_version` (reference: csrc/knn/knn.h:161, knn.cu:292-303). */
int pointops_knn_check_version(int version, int64_t D, int64_t K);

/*
 * KNN backward -- replaces `_C.knn_points_backward`
 * (reference: csrc/knn/knn.h:127-149, csrc/knn/knn_cpu.cpp:75-128).
 *   grad_p1 (N,P1,D) is a per-query sum over k in k order (deterministic, equal to
 *   the CPU order); grad_p2 (N,P2,D) is a scatter-add (fp32 atomics, order-dependent
 *   last bits, like the reference CUDA path csrc/knn/knn.cu:514-515,538).
 *   Skips k >= min(lengths2[n], K), i >= lengths1[n] and idx == -1.
 */
int pointops_knn_points_backward(const float* p1, const float* p2, const int64_t* lengths1,
                                 const int64_t* lengths2, const int64_t* idxs,
                                 const float* grad_dists, int64_t N, int64_t P1, int64_t P2,
                                 int64_t D, int64_t K, int norm, float* grad_p1,
                                 float* grad_p2, void* stream);

/*
 * Ball query -- replaces `_C.ball_query`
 * (reference: csrc/ball_query/ball_query.h:62-93, ball_query_cpu.cpp:12-54).
 *   First K points of p2 (index order) with dist2 < radius*radius (fp32 product,
 *   strict); idxs padded with -1, dists with 0.
 *   With a `workspace` of pointops_ball_query_workspace_bytes() (0 = not useful for the shape;
 *   workspace may then be NULL) clouds whose balls are SPARSE are answered through a cell grid
 *   instead of the index-order scan; the choice is made per cloud on the device and never changes
 *   results.  Without workspace every cloud is scanned.
 */
size_t pointops_ball_query_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D, int64_t K);
int pointops_ball_query(const float* p1, const float* p2, const int64_t* lengths1,
                        const int64_t* lengths2, int64_t N, int64_t P1, int64_t P2, int64_t D,
                        int64_t K, float radius, int64_t* idxs, float* dists, void* workspace,
                        size_t workspace_bytes, void* stream);

/*
 * Farthest point sampling -- replaces `_C.sample_farthest_points`
 * (reference: csrc/sample_farthest_points/sample_farthest_points.h:55-76,
 *  sample_farthest_points_cpu.cpp:14-103).
 *   points (N,P,D), lengths (N,), K (N,), start_idxs (N,), out idxs (N,max_K) int64
 *   padded with -1 beyond min(lengths[n], K[n]).  max_K = max(K) is passed by the
 *   host (the reference reads it with a host sync: sample_farthest_points.cu:132).
 *   workspace: pointops_fps_workspace_bytes(N, P, max_K) bytes of device scratch (running
 *   min-distance array of the single-workgroup kernel + the per-iteration exchange slots of
 *   the multi-workgroup cluster kernel).
 */
size_t pointops_fps_workspace_bytes(int64_t N, int64_t P, int64_t max_K);
int pointops_sample_farthest_points(const float* points, const int64_t* lengths,
                                    const int64_t* K, const int64_t* start_idxs, int64_t N,
                                    int64_t P, int64_t D, int64_t max_K, int64_t* idxs,
                                    void* workspace, size_t workspace_bytes, void* stream);

/*
 * packed <-> padded -- replace `_C.packed_to_padded` / `_C.padded_to_packed`
 * (reference: csrc/packed_to_padded_tensor/packed_to_padded_tensor.h:78-113,
 *  packed_to_padded_tensor_cpu.cpp:11-70).
 *   packed (F,D), first_idxs (B,), padded (B,max_size,D); cloud b owns packed rows
 *   [first_idxs[b], first_idxs[b+1]) (last cloud: F).  Padding is zero-filled; packed
 *   rows owned by no cloud are zero.
 */
int pointops_packed_to_padded(const float* packed, const int64_t* first_idxs, int64_t F,
                              int64_t B, int64_t max_size, int64_t D, float* padded,
                              void* stream);
int pointops_padded_to_packed(const float* padded, const int64_t* first_idxs, int64_t F,
                              int64_t B, int64_t max_size, int64_t D, float* packed,
                              void* stream);

/*
 * Neighbour gather -- the device half of `knn_gather` / `masked_gather`
 * (reference: functions/knn.py:200-250, functions/utils.py:20-65).
 *   x (N,M,U), idx (N,L,K) -> out (N,L,K,U) with out[n,l,k,:] = x[n, idx[n,l,k], :];
 *   zero where k >= lengths[n] (lengths may be NULL) or idx < 0.
 * Backward: grad_x (N,M,U) += scatter of grad_out with the same masks (fp32 atomics);
 *   grad_x is zero-filled by the call.
 */
int pointops_gather_neighbors(const float* x, const int64_t* idx, const int64_t* lengths,
                              int64_t N, int64_t M, int64_t U, int64_t L, int64_t K,
                              float* out, void* stream);
int pointops_gather_neighbors_backward(const float* grad_out, const int64_t* idx,
                                       const int64_t* lengths, int64_t N, int64_t M,
                                       int64_t U, int64_t L, int64_t K, float* grad_x,
                                       void* stream);

/*
 * DETERMINISTIC forms of the two scatter backward passes (csrc/backward_det.hip): the neighbour table is inverted
 * with a stable radix sort and every target row sums its addends in table order -- for knn_points_backward the order
 * of the reference's CPU loop (csrc/knn/knn_cpu.cpp:100-125), so grad_p2 is bit-equal to it.  Same arguments and
 * masks as the calls above plus `workspace` of pointops_backward_det_workspace_bytes(N, L, K, M) bytes
 * (knn: L = P1, M = P2; gather: L, M as given).  The neighbour table must have fewer than 2^31 entries.
 * The host wrappers select them under torch.use_deterministic_algorithms(True).
 */
size_t pointops_backward_det_workspace_bytes(int64_t N, int64_t L, int64_t K, int64_t M);
int pointops_knn_points_backward_det(const float* p1, const float* p2, const int64_t* lengths1,
                                     const int64_t* lengths2, const int64_t* idxs,
                                     const float* grad_dists, int64_t N, int64_t P1, int64_t P2,
                                     int64_t D, int64_t K, int norm, float* grad_p1, float* grad_p2,
                                     void* workspace, size_t workspace_bytes, void* stream);
int pointops_gather_neighbors_backward_det(const float* grad_out, const int64_t* idx,
                                           const int64_t* lengths, int64_t N, int64_t M, int64_t U,
                                           int64_t L, int64_t K, float* grad_x, void* workspace,
                                           size_t workspace_bytes, void* stream);

/*
 * Chamfer point-loss reduction -- the fused tail of
 * `_chamfer_distance_single_direction` (reference: functions/chamfer.py:134-185)
 * for point_reduction in {"sum","mean"}:
 *   out[n] = (sum_{i < lengths[n]} dists[n,i]) * (weights ? weights[n] : 1)
 *            / (mean ? max(lengths[n],1) : 1)
 *   dists (N,P) are the K=1 KNN distances.  Summation order is a fixed tree per
 *   cloud (deterministic), compared with the reference at 1e-5 relative.
 */
int pointops_chamfer_reduce(const float* dists, const int64_t* lengths, const float* weights,
                            int64_t N, int64_t P, int mean, float* out, void* stream);

/*
 * Fused single-direction chamfer terms for K = 1 and point_reduction in {"sum","mean"} --
 * the tail of `_chamfer_distance_single_direction` after its knn_points call
 * (reference: functions/chamfer.py:135-185) including the per-feature cosine loss:
 *   out[0][n]   = w_n / den_n * sum_{i < xlen_n} dists[n,i]
 *   out[1+f][n] = w_n / den_n * sum_{i < xlen_n} (1 - |cos(xf_f[n,i], yf_f[n, idx[n,i]])|)   (|.| if abs_cosine)
 * with den_n = max(xlen_n, 1) for "mean", 1 for "sum"; cos uses F.cosine_similarity's eps = 1e-6.
 * x_feats / y_feats / C are HOST arrays of F (<= 4) device pointers / channel counts (<= 16).
 * The backward is closed-form: grad_x, grad_x_feats written densely, grad_y / grad_y_feats are
 * zero-filled then scatter-added with fp32 atomics.  grad_out is (1+F, N).
 */
size_t pointops_chamfer_workspace_bytes(int64_t N, int64_t P1);
int pointops_chamfer_forward(const float* dists, const int64_t* idx, const int64_t* x_lengths,
                             const int64_t* y_lengths, const float* weights, int64_t N, int64_t P1,
                             int64_t P2, int F, const float* const* x_feats, const float* const* y_feats,
                             const int64_t* C, int abs_cosine, int mean, float* out, void* workspace,
                             size_t workspace_bytes, void* stream);
int pointops_chamfer_backward(const float* x, const float* y, const int64_t* idx, const int64_t* x_lengths,
                              const int64_t* y_lengths, const float* weights, const float* grad_out,
                              int64_t N, int64_t P1, int64_t P2, int64_t D, int norm, int F,
                              const float* const* x_feats, const float* const* y_feats, const int64_t* C,
                              int abs_cosine, int mean, float* grad_x, float* grad_y,
                              float* const* grad_x_feats, float* const* grad_y_feats, void* stream);
/*
 * The same gradients ADDED to buffers that already hold gradients (no reference counterpart: the reference sums the
 * two directions of a bidirectional chamfer distance through autograd).  After pointops_chamfer_backward for the
 * direction x -> y, the direction y -> x calls this with the roles swapped -- (x, y) := (y, x), grad_x := the first
 * call's grad_y, grad_y := its grad_x, likewise the feature gradients: the dense query-side terms are added element
 * by element, the target-side terms by the same fp32 atomics, nothing is zero-filled.
 */
int pointops_chamfer_backward_accumulate(const float* x, const float* y, const int64_t* idx,
                                         const int64_t* x_lengths, const int64_t* y_lengths, const float* weights,
                                         const float* grad_out, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                         int norm, int F, const float* const* x_feats, const float* const* y_feats,
                                         const int64_t* C, int abs_cosine, int mean, float* grad_x, float* grad_y,
                                         float* const* grad_x_feats, float* const* grad_y_feats, void* stream);

/*
 * BOTH directions of an unweighted chamfer distance with point_reduction in {"sum","mean"} behind one entry (no
 * reference counterpart; the composition of `chamfer_distance`, reference: functions/chamfer.py:188-312, for
 * weights=None): the K=1 searches x -> y and y -> x (pointops_knn_points_idx), pointops_chamfer_forward on each, the
 * sum of the two directions and the batch reduction.
 *   batch_reduction: 0 = None -> each of the 1+F outputs is (N,);  1 = "mean", 2 = "sum" -> each is one float
 *   outs: HOST array of 1+F device pointers (the point term, then one per feature pair)
 *   idx_xy (N,P1), idx_yx (N,P2): the neighbour indices, outputs, needed by the backward.
 * The backward takes 1+F gradient pointers (HOST array; a null entry is a zero gradient; each points to one float
 * after a batch reduction, to N floats without one) and writes grad_x, grad_y, grad_x_feats, grad_y_feats
 * (every element written: no pre-zeroing needed); workspace: (1+F)*N floats.
 */
size_t pointops_chamfer_pair_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D, int F);
int pointops_chamfer_pair_forward(const float* x, const float* y, const int64_t* x_lengths,
                                  const int64_t* y_lengths, int64_t N, int64_t P1, int64_t P2, int64_t D, int norm,
                                  int F, const float* const* x_feats, const float* const* y_feats, const int64_t* C,
                                  int abs_cosine, int mean, int batch_reduction, int64_t* idx_xy, int64_t* idx_yx,
                                  float* const* outs, void* workspace, size_t workspace_bytes, void* stream);
int pointops_chamfer_pair_backward(const float* x, const float* y, const int64_t* idx_xy, const int64_t* idx_yx,
                                   const int64_t* x_lengths, const int64_t* y_lengths, const float* const* grads,
                                   int64_t N, int64_t P1, int64_t P2, int64_t D, int norm, int F,
                                   const float* const* x_feats, const float* const* y_feats, const int64_t* C,
                                   int abs_cosine, int mean, int batch_reduction, float* grad_x, float* grad_y,
                                   float* const* grad_x_feats, float* const* grad_y_feats, void* workspace,
                                   size_t workspace_bytes, void* stream);

/*
 * Inverse-CDF sampling -- replaces `_C.sample_pdf` (reference: csrc/sample_pdf/sample_pdf.h:58-78,
 * CPU semantics sample_pdf_cpu.cpp:19-99, the USE_BINARY_SEARCH build).
 *   bins (batch, n_bins+1), weights (batch, n_bins), outputs (batch, n_samples) holding the quantiles
 *   u in [0,1] on entry and the samples on return (in place, like the reference).
 */
int pointops_sample_pdf(const float* bins, const float* weights, float* outputs, int64_t batch,
                        int64_t n_bins, int64_t n_samples, float eps, void* stream);

/*
 * Per-point covariance of a gathered K-neighbourhood, fused -- device half of get_point_covariances
 * (reference: functions/utils.py:111-153, which materialises a (N,P,K,D,D) tensor).
 *   knn (N,P,K,D) fp32 -> cov (N,P,D,D): cov[a][b] = mean_k (x_k[a]-m[a])(x_k[b]-m[b]), m = mean_k x_k.
 *   backward: grad_knn (N,P,K,D) = (G + G^T)(x_k - m) / K for G = grad_cov (N,P,D,D).  1 <= D <= 8.
 */
int pointops_point_covariances(const float* knn, int64_t N, int64_t P, int64_t K, int64_t D, float* cov,
                               void* stream);
int pointops_point_covariances_backward(const float* knn, const float* grad_cov, int64_t N, int64_t P, int64_t K,
                                        int64_t D, float* grad_knn, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTOPS_AMD_H_ */
