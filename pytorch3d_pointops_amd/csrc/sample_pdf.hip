// sample_pdf.hip -- inverse-CDF sampling of piecewise-constant PDFs (NeRF hierarchical sampling).
//
// Replaces SamplePdf (reference: csrc/sample_pdf/sample_pdf.h:58-78) with the semantics of the CPU
// build the reference ships (sample_pdf_cpu.cpp:19 `#define USE_BINARY_SEARCH`, :23-99): inclusive
// fp32 partial sums of the bin weights IN BIN ORDER, total + eps, uniform = total * u,
// std::lower_bound over the first n_bins-1 partial sums, one subtraction of the previous partial
// sum, linear interpolation with the `uniform > w` / `w > eps` guards.  In place on `outputs`.
// (The reference's CUDA kernel scans the bins linearly and subtracts weight by weight, which rounds
// differently; parity is pinned to the CPU path like everywhere else.)
//
// One workgroup per batch row.  The partial sums must be accumulated sequentially to be bit-equal,
// so lane 0 builds them in LDS (n_bins is ~64-128 in practice); the samples of the row are then
// independent: one lane per sample, binary search in LDS, coalesced in-place update.
#include "common.h"

namespace pointops {

constexpr int kPdfBlock = 256;

__global__ __launch_bounds__(kPdfBlock) void sample_pdf_kernel(const float* __restrict__ bins,
                                                             const float* __restrict__ weights,
                                                             float* __restrict__ outputs, int n_bins,
                                                             int64_t n_samples, float eps) {
  extern __shared__ float s_partial[];  // n_bins
  __shared__ float s_total;
  const int64_t b = blockIdx.x;
  const float* __restrict__ bin = bins + b * (n_bins + 1);
  const float* __restrict__ w = weights + b * n_bins;
  if (threadIdx.x == 0) {
    float total = 0.0f;
    for (int i = 0; i < n_bins; ++i) {
      total += w[i];
      s_partial[i] = total;
    }
    s_total = total + eps;
  }
  __syncthreads();
  const float total = s_total;
  float* __restrict__ out = outputs + b * n_samples;
  for (int64_t s = threadIdx.x; s < n_samples; s += kPdfBlock) {
    float uniform = total * out[s];
    int lo = 0, hi = n_bins - 1;  // lower_bound on [0, n_bins-1)
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (s_partial[mid] < uniform) lo = mid + 1;
      else hi = mid;
    }
    const int i = lo;
    if (i > 0) uniform -= s_partial[i - 1];
    const float bin_start = bin[i], bin_end = bin[i + 1], bin_weight = w[i];
    float v = bin_start;
    if (uniform > bin_weight) {
      v = bin_end;
    } else if (bin_weight > eps) {
      v += (uniform / bin_weight) * (bin_end - bin_start);
    }
    out[s] = v;
  }
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_sample_pdf(const float* bins, const float* weights, float* outputs, int64_t batch,
                                   int64_t n_bins, int64_t n_samples, float eps, void* stream_) {
  POINTOPS_REQUIRE(batch >= 0 && n_bins >= 1 && n_samples >= 0, "sample_pdf: bad sizes");
  POINTOPS_REQUIRE(batch < (1LL << 31) && n_bins <= 16384, "sample_pdf: batch < 2^31 and n_bins <= 16384");
  if (batch == 0 || n_samples == 0) return POINTOPS_OK;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)batch), dim3(kPdfBlock), sizeof(float) * (size_t)n_bins,
                     (hipStream_t)stream_, bins, weights, outputs, (int)n_bins, n_samples, eps);
  return check_launch("sample_pdf");
}
