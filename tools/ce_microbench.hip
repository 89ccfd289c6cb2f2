// ce_microbench.hip -- cost and exactness of 64-bit compare-exchange forms on gfx950 (wave64).
// Keys are (fp32 distance bits << 32 | index) with distance >= +0, i.e. bit patterns of non-negative,
// non-NaN doubles: their unsigned order IS their order as doubles, so v_min_f64 / v_max_f64 sort them.
//   form 0: v_cmp_lt_u64 + mask + 4 v_bfi_b32      (round-1 key_ce)
//   form 1: v_min_f64 + v_max_f64
//   form 2: v_cmp_lt_u64 + 4 v_cndmask_b32 (what the compiler emits for a ternary swap)
// Each wave sorts 16 keys with the 60-comparator network + one 16-merge, `iters` times.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ce_microbench.hip -o gpurun_out/ce_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned long long u64;

template <int FORM>
__device__ __forceinline__ void ce(u64& a, u64& b) {
  if (FORM == 0) {
    unsigned m = (b < a) ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    const unsigned alo = (unsigned)a, ahi = (unsigned)(a >> 32), blo = (unsigned)b, bhi = (unsigned)(b >> 32);
    const unsigned lo_lo = (m & blo) | (~m & alo), lo_hi = (m & bhi) | (~m & ahi);
    const unsigned hi_lo = (m & alo) | (~m & blo), hi_hi = (m & ahi) | (~m & bhi);
    a = ((u64)lo_hi << 32) | lo_lo;
    b = ((u64)hi_hi << 32) | hi_lo;
  } else if (FORM == 1) {
    double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b), lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    a = (u64)__double_as_longlong(lo);
    b = (u64)__double_as_longlong(hi);
  } else {
    const bool sw = b < a;
    const u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
  }
}

__device__ constexpr unsigned char kA[60] = {0, 1, 2,  3,  4, 5, 7,  9,  0, 1, 2, 3, 6,  8,  10, 11, 0, 2, 4, 6,
                                             7, 10, 12, 14, 0, 1, 4,  5,  6, 8, 12, 13, 1, 3,  4,  5,  8, 9, 13, 1,
                                             2, 5,  7,  9,  11, 2, 3, 9,  11, 3, 6, 7,  10, 3, 5,  7,  9,  11, 6, 8};
__device__ constexpr unsigned char kB[60] = {13, 12, 15, 14, 8,  6,  11, 10, 5,  7,  9,  4,  13, 14, 15, 12, 1, 3, 5, 8,
                                             9,  11, 13, 15, 2,  3,  10, 11, 7,  9,  14, 15, 2,  12, 6,  7,  10, 11, 14, 4,
                                             6,  8,  10, 13, 14, 4,  6,  12, 13, 5,  8,  9,  12, 4,  6,  8,  10, 12, 7, 9};

template <int FORM>
__global__ void bench(const u64* __restrict__ in, u64* __restrict__ out, int iters) {
  u64 k[16];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i] = in[(size_t)t * 16 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 60; ++i) ce<FORM>(k[kA[i]], k[kB[i]]);
    if (it + 1 < iters) {  // perturb so the next round is not a no-op: reverse halves (still a permutation)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u64 tmp = k[i];
        k[i] = k[15 - i];
        k[15 - i] = tmp;
      }
      asm volatile("" : "+v"(k[0]));
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) out[(size_t)t * 16 + i] = k[i];
}

template <int FORM>
int run(const char* name, int waves_per_simd, const u64* in, u64* out, const std::vector<u64>& h_in, bool verify) {
  const int iters = 2000, threads = 256, blocks = 256 * waves_per_simd;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(bench<FORM>, dim3(blocks), dim3(threads), 0, 0, in, out, 3);
  CHECK(hipDeviceSynchronize());
  if (verify) {
    std::vector<u64> h((size_t)blocks * threads * 16);
    CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t t = 0; t < (size_t)blocks * threads; ++t) {
      u64 ref[16];
      for (int i = 0; i < 16; ++i) ref[i] = h_in[t * 16 + i];
      std::sort(ref, ref + 16);
      for (int i = 0; i < 16; ++i) bad += ref[i] != h[t * 16 + i];
    }
    printf("%-34s verify: %zu mismatching keys of %zu\n", name, bad, h.size());
  }
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<FORM>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  const double ce_per_simd = (double)iters * 60 * waves_per_simd;
  printf("%-34s waves/SIMD=%d  %.3f ms  cycles per wave compare-exchange per SIMD (2.4 GHz) = %.2f\n", name,
         waves_per_simd, ms, ms * 1e-3 * 2.4e9 / ce_per_simd);
  return 0;
}

int main() {
  const size_t n = (size_t)256 * 8 * 256 * 16;
  std::vector<u64> h(n);
  unsigned long long s = 0x9E3779B97F4A7C15ULL;
  auto next = [&]() {
    s += 0x9E3779B97F4A7C15ULL;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  };
  for (size_t i = 0; i < n; ++i) {
    const u64 r = next();
    unsigned dist;
    switch (r & 7) {  // distance bit patterns: normal range, tiny (double-denormal keys), zero, +inf, ties
      case 0: dist = (unsigned)(r >> 40) & 0x000fffffu; break;             // top 12 bits zero -> denormal double
      case 1: dist = 0u; break;
      case 2: dist = 0x7f800000u; break;
      case 3: dist = 0x3f000000u + ((unsigned)(r >> 50) & 3u); break;      // heavy ties
      default: dist = (unsigned)(r >> 33) % 0x7f800001u; break;
    }
    h[i] = ((u64)dist << 32) | (unsigned)((r >> 8) & 0x7fffffffu);
  }
  u64 *in, *out;
  CHECK(hipMalloc(&in, n * 8));
  CHECK(hipMalloc(&out, n * 8));
  CHECK(hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice));
  for (int w : {1, 2, 4, 8}) {
    run<0>("cmp_u64 + 4 bfi (round 1)", w, in, out, h, w == 8);
    run<1>("v_min_f64 + v_max_f64", w, in, out, h, w == 8);
    run<2>("ternary swap (compiler)", w, in, out, h, w == 8);
  }
  return 0;
}
