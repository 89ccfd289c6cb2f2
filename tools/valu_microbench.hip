// valu_microbench.hip -- measures fp32 VALU issue rates on gfx950 (wave64) to size the KNN scan:
// plain vs packed (v_pk_*_f32) add / mul / fma, v_cmp, v_min3, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o gpurun_out/valu_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void bench(float* out, int iters, float c) {
  float a[8];
  float2_ p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = float2_{a[i], a[i] + 0.5f}; }
  float2_ c2 = {c, c * 1.5f};
  unsigned long long m = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 1) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));
        if (MODE == 2) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "v"(c2));
        if (MODE == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));
        if (MODE == 6) asm volatile("v_min3_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
        if (MODE == 8) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(c));  // SGPR operand
        if (MODE == 9) asm volatile("v_med3_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 10) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(c) : );
        if (MODE == 11) asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[i]) : "s"(c2));  // SGPR pair
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[threadIdx.x] = s + (float)m;
}

template <int MODE>
int run(const char* name, int waves_per_simd, float* out) {
  const int iters = 20000;
  const int threads = 256;                       // 4 waves: one per SIMD
  const int blocks = 256 * waves_per_simd;       // per CU: waves_per_simd blocks
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0001f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double instr_per_wave = (double)iters * 32;
  const double wave_instr_per_simd = instr_per_wave * waves_per_simd;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s waves/SIMD=%d  %.3f ms  cycles/wave-instr/SIMD (at 2.4GHz) = %.2f\n", name, waves_per_simd, ms,
         cycles / wave_instr_per_simd);
  return 0;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 4096));
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_add_f32", w, out);
    run<1>("v_pk_add_f32", w, out);
    run<2>("v_fma_f32", w, out);
    run<3>("v_pk_fma_f32", w, out);
    run<4>("v_mul_f32", w, out);
    run<5>("v_pk_mul_f32", w, out);
    run<6>("v_min3_f32", w, out);
    run<7>("v_cmp_lt_f32", w, out);
    run<8>("v_sub_f32 (sgpr src)", w, out);
    run<9>("v_med3_f32", w, out);
    run<10>("v_cndmask_b32", w, out);
    run<11>("v_pk_add_f32 (sgpr pair)", w, out);
  }
  return 0;
}
