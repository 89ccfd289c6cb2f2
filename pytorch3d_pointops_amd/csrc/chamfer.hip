// chamfer.hip -- fused per-cloud reduction of the K=1 nearest-neighbour distances.
//
// Fuses the tail of _chamfer_distance_single_direction (reference:
// functions/chamfer.py:135-185) for point_reduction in {"sum","mean"}:
//   mask rows i >= lengths[n]  ->  sum over points  ->  * weights[n]
//   ->  / max(lengths[n], 1) ("mean")
// into one launch with one workgroup per cloud (the reference runs ~6 elementwise /
// reduction torch kernels and one host sync here).  Fixed-shape tree reduction, so
// the result is deterministic run to run.
#include "common.h"
#include "debug.h"

namespace pointops {

constexpr int kChBlock = 1024;

__global__ __launch_bounds__(kChBlock) void chamfer_reduce_kernel(
    const float* __restrict__ dists, const int64_t* __restrict__ lengths,
    const float* __restrict__ weights, int64_t P, int mean, float* __restrict__ out) {
  const int n = blockIdx.x;
  int64_t len = lengths[n];
  if (len > P) len = P;
  const float* __restrict__ row = dists + (int64_t)n * P;
  float acc = 0.0f;
  for (int64_t i = threadIdx.x; i < len; i += kChBlock) acc += row[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  __shared__ float s[kChBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) s[wave] = acc;
  __syncthreads();
  if (wave == 0) {
    float v = lane < kChBlock / kWave ? s[lane] : 0.0f;
#pragma unroll
    for (int off = kChBlock / kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) {
      if (weights != nullptr) v *= weights[n];
      if (mean) v /= (float)(len < 1 ? 1 : len);
      out[n] = v;
    }
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_chamfer_reduce(const float* dists, const int64_t* lengths,
                                       const float* weights, int64_t N, int64_t P, int mean,
                                       float* out, void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && N < (1LL << 31), "chamfer_reduce: bad sizes");
  if (N == 0) return POINTOPS_OK;
  hipLaunchKernelGGL(chamfer_reduce_kernel, dim3((unsigned)N), dim3(kChBlock), 0, (hipStream_t)stream_,
                     dists, lengths, weights, P, mean, out);
  return check_launch("chamfer_reduce");
}

// ===========================================================================
// Fused single-direction chamfer terms (K = 1) with a closed-form backward.
//
// Replaces, for point_reduction in {"sum","mean"}, everything the reference does after its
// K=1 knn_points call in _chamfer_distance_single_direction (functions/chamfer.py:135-185):
// masking, weights, knn_gather of the neighbour's features, F.cosine_similarity(eps=1e-6),
// |.|, 1 - cos, the per-cloud sums and the division by clamp(lengths,1) -- ~25 torch kernels
// forward and as many backward per direction -- by ONE forward kernel (+ a tiny finalize) and
// ONE backward kernel.  Forward sums are a fixed two-level tree (deterministic); the backward
// writes grad_x / grad_x_feature rows directly and scatter-adds grad_y / grad_y_feature with
// fp32 atomics (zero contributions skipped), like knn_points_backward.
// cosine(x1, x2) = sum_i (x1_i / max(|x1|, eps)) * (x2_i / max(|x2|, eps))   (ATen's formulation)
// ===========================================================================
namespace pointops {

constexpr int kCfBlock = 256;
constexpr int kCfPerThread = 2;  // (4 / 2 / 1 points per thread: forward + finalize 33.2 / 27.5 / 27.2 us per direction at cfg4)
constexpr int kCfMaxFeat = 4;
constexpr int kCfMaxC = 16;  // feature channels held in registers

struct ChamferFeat {
  const float* x[kCfMaxFeat];
  const float* y[kCfMaxFeat];
  float* gx[kCfMaxFeat];
  float* gy[kCfMaxFeat];
  int C[kCfMaxFeat];
  int F;
};

__device__ __forceinline__ float cf_block_sum(float v, float* s_red) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  if (threadIdx.x == 0) {
    for (int w = 0; w < kCfBlock / kWave; ++w) t += s_red[w];
  }
  __syncthreads();
  return t;  // valid in thread 0
}

// cos and the normalised vectors of one (x feature row, neighbour feature row) pair
__device__ __forceinline__ float cf_cosine(const float* __restrict__ xf, const float* __restrict__ yf, int C,
                                           bool y_valid, float eps, float (&xn)[kCfMaxC], float (&yn)[kCfMaxC],
                                           float& nx, float& ny) {
  float sx = 0.0f, sy = 0.0f;
#pragma unroll
  for (int c = 0; c < kCfMaxC; ++c) {  // static indices keep xn / yn in registers
    const float a = c < C ? xf[c] : 0.0f, b = (c < C && y_valid) ? yf[c] : 0.0f;
    xn[c] = a;
    yn[c] = b;
    sx += a * a;
    sy += b * b;
  }
  nx = sqrtf(sx);
  ny = sqrtf(sy);
  const float ix = 1.0f / fmaxf(nx, eps), iy = 1.0f / fmaxf(ny, eps);
  float cosv = 0.0f;
#pragma unroll
  for (int c = 0; c < kCfMaxC; ++c) {
    xn[c] *= ix;
    yn[c] *= iy;
    cosv += xn[c] * yn[c];
  }
  return cosv;
}

__global__ __launch_bounds__(kCfBlock) void chamfer_forward_kernel(
    const float* __restrict__ dists, const int64_t* __restrict__ idx, const int64_t* __restrict__ x_lengths,
    const int64_t* __restrict__ y_lengths, int64_t P1, int64_t P2, ChamferFeat ft, int abs_cosine, int chunks,
    float* __restrict__ partial) {
  __shared__ float s_red[kCfBlock / kWave];
  const int n = blockIdx.y, chunk = blockIdx.x;
  int64_t len = x_lengths[n];
  if (len > P1) len = P1;
  const bool y_valid = y_lengths[n] > 0;
  float acc[1 + kCfMaxFeat];
#pragma unroll
  for (int t = 0; t < 1 + kCfMaxFeat; ++t) acc[t] = 0.0f;
  for (int r = 0; r < kCfPerThread; ++r) {
    const int64_t i = (int64_t)chunk * (kCfBlock * kCfPerThread) + r * kCfBlock + threadIdx.x;
    if (i < len) {
      const int64_t row = (int64_t)n * P1 + i;
      acc[0] += dists[row];
      const int64_t j = idx[row];
#pragma unroll
      for (int f = 0; f < kCfMaxFeat; ++f) {
        if (f < ft.F) {
          const int C = ft.C[f];
          float xn[kCfMaxC], yn[kCfMaxC], nx, ny;
          const float cosv = cf_cosine(ft.x[f] + row * C, ft.y[f] + ((int64_t)n * P2 + j) * C, C, y_valid, 1e-6f,
                                       xn, yn, nx, ny);
          acc[1 + f] += 1.0f - (abs_cosine ? fabsf(cosv) : cosv);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 1 + kCfMaxFeat; ++t) {
    if (t < 1 + ft.F) {  // wave-uniform
      const float s = cf_block_sum(acc[t], s_red);
      if (threadIdx.x == 0) partial[((int64_t)n * chunks + chunk) * (1 + kCfMaxFeat) + t] = s;
    }
  }
}

__global__ __launch_bounds__(kWave) void chamfer_finalize_kernel(const float* __restrict__ partial,
                                                                const int64_t* __restrict__ x_lengths,
                                                                const float* __restrict__ weights, int N, int chunks,
                                                                int F, int mean, float* __restrict__ out) {
  const int n = blockIdx.x;
  for (int t = 0; t < 1 + F; ++t) {
    float v = 0.0f;
    for (int c = threadIdx.x; c < chunks; c += kWave) v += partial[((int64_t)n * chunks + c) * (1 + kCfMaxFeat) + t];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (threadIdx.x == 0) {
      if (weights != nullptr) v *= weights[n];
      if (mean) {
        const int64_t len = x_lengths[n];
        v /= (float)(len < 1 ? 1 : len);
      }
      out[(int64_t)t * N + n] = v;
    }
  }
}

// dense gradient of the query side: stored, or -- accumulating call (pointops_chamfer_backward_accumulate) -- added to
// what the other direction left there (the element has ONE owner thread per launch, launches are stream-ordered)
__device__ __forceinline__ void cf_put(float* __restrict__ p, float v, int acc) { *p = acc ? *p + v : v; }

template <int NORM>
__global__ __launch_bounds__(kCfBlock) void chamfer_backward_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ x_lengths, const int64_t* __restrict__ y_lengths,
    const float* __restrict__ weights, const float* __restrict__ grad_out, int N, int64_t P1, int64_t P2, int D,
    ChamferFeat ft, int abs_cosine, int mean, int acc, float* __restrict__ grad_x, float* __restrict__ grad_y) {
  const int n = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * kCfBlock + threadIdx.x;
  if (i >= P1) return;
  int64_t len = x_lengths[n];
  if (len > P1) len = P1;
  const int64_t row = (int64_t)n * P1 + i;
  const bool y_valid = y_lengths[n] > 0;
  if (i >= len) {  // masked rows: zero gradients (accumulating: nothing to add)
    if (acc) return;
    for (int d = 0; d < D; ++d) grad_x[row * D + d] = 0.0f;
#pragma unroll
    for (int f = 0; f < kCfMaxFeat; ++f)
      if (f < ft.F)
        for (int c = 0; c < ft.C[f]; ++c) ft.gx[f][row * ft.C[f] + c] = 0.0f;
    return;
  }
  float scale = weights != nullptr ? weights[n] : 1.0f;
  if (mean) scale /= (float)(len < 1 ? 1 : len);
  const int64_t j = idx[row];
  const int64_t yrow = (int64_t)n * P2 + j;
  // point term: d/dx of dist(x_i, y_j)   (same expressions as knn_points_backward)
  const float a = grad_out[n] * scale;
  for (int d = 0; d < D; ++d) {
    float diff = 0.0f;
    if (y_valid) {
      const float xv = x[row * D + d], yv = y[yrow * D + d];
      if (NORM == 1) diff = a * ((xv > yv) ? 1.0f : -1.0f);
      else diff = 2.0f * a * (xv - yv);
      if (diff != 0.0f) atomicAdd(grad_y + yrow * D + d, -1.0f * diff);
    }
    cf_put(grad_x + row * D + d, diff, acc);
  }
  // feature terms: d/d(features) of 1 - |cos| (or 1 - cos)
#pragma unroll
  for (int f = 0; f < kCfMaxFeat; ++f) {
    if (f >= ft.F) continue;
    const int C = ft.C[f];
    float xn[kCfMaxC], yn[kCfMaxC], nx, ny;
    const float eps = 1e-6f;
    const float cosv = cf_cosine(ft.x[f] + row * C, ft.y[f] + yrow * C, C, y_valid, eps, xn, yn, nx, ny);
    float s = -1.0f;  // d(1 - cos)/dcos
    if (abs_cosine) s = cosv > 0.0f ? -1.0f : (cosv < 0.0f ? 1.0f : 0.0f);
    const float b = grad_out[(int64_t)(1 + f) * N + n] * scale * s;
    const float ix = 1.0f / fmaxf(nx, eps), iy = 1.0f / fmaxf(ny, eps);
#pragma unroll
    for (int c = 0; c < kCfMaxC; ++c) {
      if (c < C) {
        // dcos/dx1 = (x2n - cos * x1n) / |x1| when |x1| > eps, x2n / eps otherwise (clamped norm)
        const float gx = b * ix * (nx > eps ? (yn[c] - cosv * xn[c]) : yn[c]);
        cf_put(ft.gx[f] + row * C + c, gx, acc);
        if (y_valid) {
          const float gy = b * iy * (ny > eps ? (xn[c] - cosv * yn[c]) : xn[c]);
          if (gy != 0.0f) atomicAdd(ft.gy[f] + yrow * C + c, gy);
        }
      }
    }
  }
}


// Channel-parallel backward for D <= 4 and C <= 4 (points + normals/colours): FOUR lanes per
// point, lane c owning coordinate / channel c.  The scatter into grad_y / grad_y_feat is bound by
// the memory-side fp32 atomic rate, which depends on how many distinct rows one wave instruction
// touches (64 different rows is ~17x slower than contiguous bytes: MI355X_MICROARCH.md, global
// float atomics).  With a lane per point the three coordinates of a target row leave in three
// instructions of 64 rows each; here they leave in ONE instruction covering 16 rows x 12-16
// contiguous bytes.  The per-point scalars (norms, cosine) are recomputed by the four lanes.
// Measured and dropped: grad_y / grad_y_feat through the LDS tiles of tiled_scatter.h (K = 1 table, 25
// tiles per 200k-point cloud): 1.50 vs 1.30 ms for the cfg4 step -- every tile workgroup re-streams the
// cloud's table in ~50 latency-bound steps, which costs more than 5 M device atomics.
template <int NORM>
__global__ __launch_bounds__(kCfBlock) void chamfer_backward4_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const int64_t* __restrict__ idx,
    const int64_t* __restrict__ x_lengths, const int64_t* __restrict__ y_lengths,
    const float* __restrict__ weights, const float* __restrict__ grad_out, int N, int64_t P1, int64_t P2, int D,
    ChamferFeat ft, int abs_cosine, int mean, int acc, float* __restrict__ grad_x, float* __restrict__ grad_y) {
  const int n = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * kCfBlock + threadIdx.x;
  const int64_t i = t >> 2;
  const int c = (int)(t & 3);
  if (i >= P1) return;
  int64_t len = x_lengths[n];
  if (len > P1) len = P1;
  const int64_t row = (int64_t)n * P1 + i;
  const bool y_valid = y_lengths[n] > 0;
  if (i >= len) {
    if (acc) return;
    if (c < D) grad_x[row * D + c] = 0.0f;
#pragma unroll
    for (int f = 0; f < kCfMaxFeat; ++f)
      if (f < ft.F && c < ft.C[f]) ft.gx[f][row * ft.C[f] + c] = 0.0f;
    return;
  }
  float scale = weights != nullptr ? weights[n] : 1.0f;
  if (mean) scale /= (float)(len < 1 ? 1 : len);
  const int64_t j = idx[row];
  const int64_t yrow = (int64_t)n * P2 + j;
  if (c < D) {
    const float a = grad_out[n] * scale;
    float diff = 0.0f;
    if (y_valid) {
      const float xv = x[row * D + c], yv = y[yrow * D + c];
      if (NORM == 1) diff = a * ((xv > yv) ? 1.0f : -1.0f);
      else diff = 2.0f * a * (xv - yv);
      if (diff != 0.0f) atomicAdd(grad_y + yrow * D + c, -1.0f * diff);
    }
    cf_put(grad_x + row * D + c, diff, acc);
  }
#pragma unroll
  for (int f = 0; f < kCfMaxFeat; ++f) {
    if (f >= ft.F) continue;
    const int C = ft.C[f];
    // norms and cosine of the (<= 4-channel) pair, same summation order as cf_cosine
    float xv[4], yv[4];
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xv[k] = k < C ? ft.x[f][row * C + k] : 0.0f;
      yv[k] = (k < C && y_valid) ? ft.y[f][yrow * C + k] : 0.0f;
      sx += xv[k] * xv[k];
      sy += yv[k] * yv[k];
    }
    const float eps = 1e-6f;
    const float nx = sqrtf(sx), ny = sqrtf(sy);
    const float ix = 1.0f / fmaxf(nx, eps), iy = 1.0f / fmaxf(ny, eps);
    float cosv = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xv[k] *= ix;
      yv[k] *= iy;
      cosv += xv[k] * yv[k];
    }
    if (c < C) {
      float sgn = -1.0f;
      if (abs_cosine) sgn = cosv > 0.0f ? -1.0f : (cosv < 0.0f ? 1.0f : 0.0f);
      const float b = grad_out[(int64_t)(1 + f) * N + n] * scale * sgn;
      float xc = 0.0f, yc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // static select keeps xv / yv in registers
        xc = (k == c) ? xv[k] : xc;
        yc = (k == c) ? yv[k] : yc;
      }
      cf_put(ft.gx[f] + row * C + c, b * ix * (nx > eps ? (yc - cosv * xc) : yc), acc);
      if (y_valid) {
        const float gy = b * iy * (ny > eps ? (xc - cosv * yc) : xc);
        if (gy != 0.0f) atomicAdd(ft.gy[f] + yrow * C + c, gy);
      }
    }
  }
}

}  // namespace pointops

extern "C" size_t pointops_chamfer_workspace_bytes(int64_t N, int64_t P1) {
  using namespace pointops;
  const int64_t chunks = ceil_div(P1 > 0 ? P1 : 1, (int64_t)kCfBlock * kCfPerThread);
  return sizeof(float) * (size_t)(N * chunks * (1 + kCfMaxFeat));
}

static int cf_fill(pointops::ChamferFeat* ft, int F, const float* const* xf, const float* const* yf,
                   float* const* gxf, float* const* gyf, const int64_t* C) {
  using namespace pointops;
  POINTOPS_REQUIRE(F >= 0 && F <= kCfMaxFeat, "chamfer: at most %d feature tensors in the fused path", kCfMaxFeat);
  ft->F = F;
  for (int f = 0; f < kCfMaxFeat; ++f) {
    ft->x[f] = f < F ? xf[f] : nullptr;
    ft->y[f] = f < F ? yf[f] : nullptr;
    ft->gx[f] = (f < F && gxf) ? gxf[f] : nullptr;
    ft->gy[f] = (f < F && gyf) ? gyf[f] : nullptr;
    ft->C[f] = f < F ? (int)C[f] : 0;
    POINTOPS_REQUIRE(ft->C[f] <= kCfMaxC, "chamfer: feature channels must be <= %d in the fused path", kCfMaxC);
  }
  return POINTOPS_OK;
}

extern "C" int pointops_chamfer_forward(const float* dists, const int64_t* idx, const int64_t* x_lengths,
                                        const int64_t* y_lengths, const float* weights, int64_t N, int64_t P1,
                                        int64_t P2, int F, const float* const* x_feats,
                                        const float* const* y_feats, const int64_t* C, int abs_cosine, int mean,
                                        float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  using namespace pointops;
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && N < 65536, "chamfer_forward: bad sizes");
  if (N == 0) return POINTOPS_OK;
  ChamferFeat ft;
  const int rc = cf_fill(&ft, F, x_feats, y_feats, nullptr, nullptr, C);
  if (rc != POINTOPS_OK) return rc;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= pointops_chamfer_workspace_bytes(N, P1),
                   "chamfer_forward: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int chunks = (int)ceil_div(P1 > 0 ? P1 : 1, (int64_t)kCfBlock * kCfPerThread);
  hipLaunchKernelGGL(chamfer_forward_kernel, dim3((unsigned)chunks, (unsigned)N), dim3(kCfBlock), 0, stream, dists,
                     idx, x_lengths, y_lengths, P1, P2, ft, abs_cosine, chunks, (float*)workspace);
  hipLaunchKernelGGL(chamfer_finalize_kernel, dim3((unsigned)N), dim3(kWave), 0, stream, (const float*)workspace,
                     x_lengths, weights, (int)N, chunks, F, mean, out);
  return check_launch("chamfer_forward");
}

static int chamfer_backward_impl(const float* x, const float* y, const int64_t* idx, const int64_t* x_lengths,
                                 const int64_t* y_lengths, const float* weights, const float* grad_out, int64_t N,
                                 int64_t P1, int64_t P2, int64_t D, int norm, int F, const float* const* x_feats,
                                 const float* const* y_feats, const int64_t* C, int abs_cosine, int mean,
                                 float* grad_x, float* grad_y, float* const* grad_x_feats,
                                 float* const* grad_y_feats, void* stream_, int acc) {
  using namespace pointops;
  POINTOPS_REQUIRE(norm == 1 || norm == 2, "chamfer_backward: norm must be 1 or 2");
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && N < 65536, "chamfer_backward: bad sizes");
  ChamferFeat ft;
  const int rc = cf_fill(&ft, F, x_feats, y_feats, grad_x_feats, grad_y_feats, C);
  if (rc != POINTOPS_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  if (!acc && N * P2 * D > 0 && hipMemsetAsync(grad_y, 0, sizeof(float) * (size_t)(N * P2 * D), stream) != hipSuccess)
    return check_launch("chamfer_backward(memset)");
  for (int f = 0; f < F; ++f)
    if (!acc && N * P2 * C[f] > 0 &&
        hipMemsetAsync(grad_y_feats[f], 0, sizeof(float) * (size_t)(N * P2 * C[f]), stream) != hipSuccess)
      return check_launch("chamfer_backward(memset)");
  if (N == 0 || P1 == 0) return POINTOPS_OK;
  bool four = D <= 4;
  for (int f = 0; f < F; ++f) four = four && C[f] <= 4;
  if (four) {
    const dim3 grid4((unsigned)ceil_div(P1 * 4, kCfBlock), (unsigned)N), block4(kCfBlock);
    if (norm == 1)
      hipLaunchKernelGGL(chamfer_backward4_kernel<1>, grid4, block4, 0, stream, x, y, idx, x_lengths, y_lengths,
                         weights, grad_out, (int)N, P1, P2, (int)D, ft, abs_cosine, mean, acc, grad_x, grad_y);
    else
      hipLaunchKernelGGL(chamfer_backward4_kernel<2>, grid4, block4, 0, stream, x, y, idx, x_lengths, y_lengths,
                         weights, grad_out, (int)N, P1, P2, (int)D, ft, abs_cosine, mean, acc, grad_x, grad_y);
    return check_launch("chamfer_backward");
  }
  const dim3 grid((unsigned)ceil_div(P1, kCfBlock), (unsigned)N), block(kCfBlock);
  if (norm == 1)
    hipLaunchKernelGGL(chamfer_backward_kernel<1>, grid, block, 0, stream, x, y, idx, x_lengths, y_lengths, weights,
                       grad_out, (int)N, P1, P2, (int)D, ft, abs_cosine, mean, acc, grad_x, grad_y);
  else
    hipLaunchKernelGGL(chamfer_backward_kernel<2>, grid, block, 0, stream, x, y, idx, x_lengths, y_lengths, weights,
                       grad_out, (int)N, P1, P2, (int)D, ft, abs_cosine, mean, acc, grad_x, grad_y);
  return check_launch("chamfer_backward");
}

extern "C" int pointops_chamfer_backward(const float* x, const float* y, const int64_t* idx,
                                         const int64_t* x_lengths, const int64_t* y_lengths, const float* weights,
                                         const float* grad_out, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                         int norm, int F, const float* const* x_feats, const float* const* y_feats,
                                         const int64_t* C, int abs_cosine, int mean, float* grad_x, float* grad_y,
                                         float* const* grad_x_feats, float* const* grad_y_feats, void* stream) {
  return chamfer_backward_impl(x, y, idx, x_lengths, y_lengths, weights, grad_out, N, P1, P2, D, norm, F, x_feats,
                               y_feats, C, abs_cosine, mean, grad_x, grad_y, grad_x_feats, grad_y_feats, stream, 0);
}

extern "C" int pointops_chamfer_backward_accumulate(const float* x, const float* y, const int64_t* idx,
                                                    const int64_t* x_lengths, const int64_t* y_lengths,
                                                    const float* weights, const float* grad_out, int64_t N,
                                                    int64_t P1, int64_t P2, int64_t D, int norm, int F,
                                                    const float* const* x_feats, const float* const* y_feats,
                                                    const int64_t* C, int abs_cosine, int mean, float* grad_x,
                                                    float* grad_y, float* const* grad_x_feats,
                                                    float* const* grad_y_feats, void* stream) {
  return chamfer_backward_impl(x, y, idx, x_lengths, y_lengths, weights, grad_out, N, P1, P2, D, norm, F, x_feats,
                               y_feats, C, abs_cosine, mean, grad_x, grad_y, grad_x_feats, grad_y_feats, stream, 1);
}

// ===========================================================================
// BOTH directions of an unweighted chamfer distance behind one entry: two K=1 searches, the two fused
// reductions, their sum and the batch reduction -- six launches from one host call instead of four host calls and
// five tensor ops (at the reference's example sizes the op is bound by the host: 95 us per forward for ~40 us of
// kernels).  The backward twin expands the (1+F) incoming gradients to (1+F, N) and runs the two closed-form
// backward passes into one set of buffers.
// ===========================================================================
namespace pointops {

__host__ __device__ inline size_t cp_align(size_t b) { return (b + 255) & ~(size_t)255; }

// rows = rows_a + rows_b, (1+F, N); reduce: 0 none -> out (1+F, N); 1 mean, 2 sum over the batch -> out (1+F)
struct PairOuts {
  float* o[1 + kCfMaxFeat];  // one tensor per output: (N,) without a batch reduction, () with one
};

__global__ __launch_bounds__(kWave) void chamfer_pair_combine_kernel(const float* __restrict__ a,
                                                                      const float* __restrict__ b, int N, int reduce,
                                                                      PairOuts outs) {
  const int f = blockIdx.x, lane = threadIdx.x;
  float* __restrict__ out = outs.o[f];
  float acc = 0.0f;
  for (int n = lane; n < N; n += kWave) {
    const float v = a[(int64_t)f * N + n] + b[(int64_t)f * N + n];
    if (reduce == 0) out[n] = v;
    acc += v;
  }
  if (reduce == 0) return;
  for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  if (lane == 0) out[0] = reduce == 1 ? acc / (float)(N > 0 ? N : 1) : acc;
}

struct PairGrads {
  const float* g[1 + kCfMaxFeat];  // incoming gradients, null = zero; () after a batch reduction, (N,) without
};

// g (1+F, N) for the two backward passes: the batch reduction's own backward (a broadcast, / N for "mean")
__global__ __launch_bounds__(kWave) void chamfer_pair_grad_kernel(PairGrads in, int N, int reduce,
                                                                   float* __restrict__ g) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const float* __restrict__ src = in.g[f];
  for (int n = lane; n < N; n += kWave) {
    float v = 0.0f;
    if (src != nullptr) v = reduce == 0 ? src[n] : (reduce == 1 ? src[0] / (float)(N > 0 ? N : 1) : src[0]);
    g[(int64_t)f * N + n] = v;
  }
}

}  // namespace pointops

// The two searches are independent until their results meet: when both go through the cell grid (big clouds: a dozen
// latency-bound build launches each) the reverse direction runs on a side stream of its own, forked from and joined to
// the caller's stream by events, with its own search workspace and distance buffer.  POINTOPS_DEBUG chamfer_overlap=0
// keeps one stream.
static bool pair_overlap(int64_t N, int64_t P1, int64_t P2, int64_t D) {
  return pointops::debug_knob("chamfer_overlap", 1) != 0 && pointops_knn_uses_grid(N, P1, P2, D, 1, -1) &&
         pointops_knn_uses_grid(N, P2, P1, D, 1, -1);
}

struct PairSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static PairSide* pair_side() {  // per host thread and device; created on first use, never destroyed
  static thread_local PairSide sides[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  PairSide* s = &sides[dev];
  if (s->stream == nullptr) {
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      s->stream = nullptr;
      return nullptr;
    }
  }
  return s;
}

extern "C" size_t pointops_chamfer_pair_workspace_bytes(int64_t N, int64_t P1, int64_t P2, int64_t D, int F) {
  using namespace pointops;
  if (N <= 0) return 0;
  const size_t knn_a = pointops_knn_workspace_bytes(N, P1, P2, D, 1, -1), knn_b = pointops_knn_workspace_bytes(N, P2, P1, D, 1, -1);
  const size_t ch_a = pointops_chamfer_workspace_bytes(N, P1), ch_b = pointops_chamfer_workspace_bytes(N, P2);
  // (laid out for two concurrent searches whether or not they overlap)
  return cp_align(knn_a) + cp_align(knn_b) + cp_align(sizeof(float) * (size_t)(N * P1)) +
         cp_align(sizeof(float) * (size_t)(N * P2)) + cp_align(ch_a > ch_b ? ch_a : ch_b) +
         2 * cp_align(sizeof(float) * (size_t)((1 + F) * N));
}

extern "C" int pointops_chamfer_pair_forward(const float* x, const float* y, const int64_t* x_lengths,
                                             const int64_t* y_lengths, int64_t N, int64_t P1, int64_t P2, int64_t D,
                                             int norm, int F, const float* const* x_feats,
                                             const float* const* y_feats, const int64_t* C, int abs_cosine, int mean,
                                             int batch_reduction, int64_t* idx_xy, int64_t* idx_yx,
                                             float* const* outs, void* workspace, size_t workspace_bytes,
                                             void* stream_) {
  using namespace pointops;
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && N < 65536 && F >= 0 && F <= kCfMaxFeat,
                   "chamfer_pair_forward: bad sizes");
  POINTOPS_REQUIRE(batch_reduction >= 0 && batch_reduction <= 2, "chamfer_pair_forward: batch_reduction must be 0, 1 or 2");
  if (N == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= pointops_chamfer_pair_workspace_bytes(N, P1, P2, D, F),
                   "chamfer_pair_forward: workspace too small");
  const size_t knn_a = pointops_knn_workspace_bytes(N, P1, P2, D, 1, -1), knn_b = pointops_knn_workspace_bytes(N, P2, P1, D, 1, -1);
  const size_t ch_a = pointops_chamfer_workspace_bytes(N, P1), ch_b = pointops_chamfer_workspace_bytes(N, P2);
  char* w = (char*)workspace;
  void* knn_ws_a = w;
  w += cp_align(knn_a);
  void* knn_ws_b = w;
  w += cp_align(knn_b);
  float* dists_a = (float*)w;
  w += cp_align(sizeof(float) * (size_t)(N * P1));
  float* dists_b = (float*)w;
  w += cp_align(sizeof(float) * (size_t)(N * P2));
  void* ch_ws = w;
  w += cp_align(ch_a > ch_b ? ch_a : ch_b);
  float* rows_a = (float*)w;
  w += cp_align(sizeof(float) * (size_t)((1 + F) * N));
  float* rows_b = (float*)w;
  hipStream_t main_stream = (hipStream_t)stream_;
  PairSide* side = pair_overlap(N, P1, P2, D) ? pair_side() : nullptr;
  void* stream_b = stream_;
  if (side != nullptr) {
    if (hipEventRecord(side->fork, main_stream) != hipSuccess || hipStreamWaitEvent(side->stream, side->fork, 0) != hipSuccess)
      return check_launch("chamfer_pair_forward(fork)");
    stream_b = (void*)side->stream;
  }
  // the reverse search first: on its own stream it overlaps everything the forward direction does
  int rc = pointops_knn_points_idx(y, x, y_lengths, x_lengths, N, P2, P1, D, norm, 1, -1, idx_yx, dists_b, knn_ws_b, knn_b,
                                   stream_b);
  if (side != nullptr && hipEventRecord(side->join, side->stream) != hipSuccess) rc = check_launch("chamfer_pair_forward(join)");
  const int rc_b = rc;
  rc = pointops_knn_points_idx(x, y, x_lengths, y_lengths, N, P1, P2, D, norm, 1, -1, idx_xy, dists_a, knn_ws_a, knn_a,
                               stream_);
  if (rc == POINTOPS_OK)
    rc = pointops_chamfer_forward(dists_a, idx_xy, x_lengths, y_lengths, nullptr, N, P1, P2, F, x_feats, y_feats, C,
                                  abs_cosine, mean, rows_a, ch_ws, ch_a, stream_);
  // join before anything else can fail out: the caller's stream must not run ahead of the side stream's use of the workspace
  if (side != nullptr && hipStreamWaitEvent(main_stream, side->join, 0) != hipSuccess)
    return check_launch("chamfer_pair_forward(join)");
  if (rc_b != POINTOPS_OK) return rc_b;
  if (rc != POINTOPS_OK) return rc;
  rc = pointops_chamfer_forward(dists_b, idx_yx, y_lengths, x_lengths, nullptr, N, P2, P1, F, y_feats, x_feats, C,
                                abs_cosine, mean, rows_b, ch_ws, ch_b, stream_);
  if (rc != POINTOPS_OK) return rc;
  PairOuts po;
  for (int f = 0; f < 1 + kCfMaxFeat; ++f) po.o[f] = f <= F ? outs[f] : nullptr;
  hipLaunchKernelGGL(chamfer_pair_combine_kernel, dim3((unsigned)(1 + F)), dim3(kWave), 0, (hipStream_t)stream_, rows_a,
                     rows_b, (int)N, batch_reduction, po);
  return check_launch("chamfer_pair_forward");
}

extern "C" int pointops_chamfer_pair_backward(const float* x, const float* y, const int64_t* idx_xy,
                                              const int64_t* idx_yx, const int64_t* x_lengths,
                                              const int64_t* y_lengths, const float* const* grads, int64_t N,
                                              int64_t P1, int64_t P2, int64_t D, int norm, int F,
                                              const float* const* x_feats, const float* const* y_feats,
                                              const int64_t* C, int abs_cosine, int mean, int batch_reduction,
                                              float* grad_x, float* grad_y, float* const* grad_x_feats,
                                              float* const* grad_y_feats, void* workspace, size_t workspace_bytes,
                                              void* stream_) {
  using namespace pointops;
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && N < 65536 && F >= 0 && F <= kCfMaxFeat,
                   "chamfer_pair_backward: bad sizes");
  POINTOPS_REQUIRE(batch_reduction >= 0 && batch_reduction <= 2, "chamfer_pair_backward: batch_reduction must be 0, 1 or 2");
  if (N == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= sizeof(float) * (size_t)((1 + F) * N),
                   "chamfer_pair_backward: workspace of (1 + F) * N floats required");
  PairGrads in;
  for (int f = 0; f < 1 + kCfMaxFeat; ++f) in.g[f] = f <= F ? grads[f] : nullptr;
  float* g = (float*)workspace;
  hipLaunchKernelGGL(chamfer_pair_grad_kernel, dim3((unsigned)(1 + F)), dim3(kWave), 0, (hipStream_t)stream_, in, (int)N,
                     batch_reduction, g);
  int rc = pointops_chamfer_backward(x, y, idx_xy, x_lengths, y_lengths, nullptr, g, N, P1, P2, D, norm, F, x_feats,
                                     y_feats, C, abs_cosine, mean, grad_x, grad_y, grad_x_feats, grad_y_feats, stream_);
  if (rc != POINTOPS_OK) return rc;
  // the reverse direction ADDS into the same buffers: its dense terms into grad_y, its atomics into grad_x
  return pointops_chamfer_backward_accumulate(y, x, idx_yx, y_lengths, x_lengths, nullptr, g, N, P2, P1, D, norm, F,
                                              y_feats, x_feats, C, abs_cosine, mean, grad_y, grad_x, grad_y_feats,
                                              grad_x_feats, stream_);
}
