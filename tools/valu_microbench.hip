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
  const unsigned long long msk = 0x5555555555555555ULL & (unsigned long long)iters * 0x100000001ULL;
  const float vm = c * 3.0f;
  if (MODE == 23) asm volatile("s_mov_b64 vcc, 0" ::: "vcc");
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
        // round 2: select / integer / 64-bit forms used by the grid walk and the sorting networks
        if (MODE == 12) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(c), "s"(msk));
        if (MODE == 13) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(vm), "v"(c));
        if (MODE == 14) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 15) asm volatile("v_min_f64 %0, %1, %0" : "+v"(p[i]) : "v"(c2));
        if (MODE == 16) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(p[i]), "v"(c2) : "vcc");
        if (MODE == 17) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");
        if (MODE == 18) asm volatile("v_mov_b64 %0, %1" : "+v"(p[i]) : "v"(c2));
        if (MODE == 19) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 20) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(p[i]) : "v"(c2));
        if (MODE == 21) asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(c), "v"(vm) : "vcc");
        if (MODE == 22) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(c));
        if (MODE == 23) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(c) : "vcc");  // vcc = 0 set before the loop
        if (MODE == 24) asm volatile("v_max3_u32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 25) asm volatile("v_lshl_or_b32 %0, %1, 9, %0" : "+v"(a[i]) : "v"(c));
        if (MODE == 27) asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(c), "v"(vm) : "vcc");
        if (MODE == 28) asm volatile("v_cmp_gt_u32 vcc, %2, %3\n\tv_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %1, %2, %1, vcc" : "+v"(a[i]), "+v"(p[i].x) : "v"(c), "v"(vm) : "vcc");
        if (MODE == 29) asm volatile("v_cmp_gt_u32 s[10:11], %1, %2\n\tv_cndmask_b32_e64 %0, %1, %0, s[10:11]" : "+v"(a[i]) : "v"(c), "v"(vm) : "s10", "s11");
        if (MODE == 30) asm volatile("v_cmp_gt_u32 vcc, %2, %3\n\tv_add_u32 %1, %2, %1\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(a[i]), "+v"(p[i].x) : "v"(c), "v"(vm) : "vcc");
        if (MODE == 31) asm volatile("v_cmp_gt_u32 s[10:11], %2, %3\n\tv_cndmask_b32_e64 %0, %2, %0, s[10:11]\n\tv_cndmask_b32_e64 %1, %2, %1, s[10:11]" : "+v"(a[i]), "+v"(p[i].x) : "v"(c), "v"(vm) : "s10", "s11");
        if (MODE == 32) asm volatile("v_cmp_gt_u32 s[10:11], %2, %3\n\tv_add_u32 %1, %2, %1\n\tv_cndmask_b32_e64 %0, %2, %0, s[10:11]" : "+v"(a[i]), "+v"(p[i].x) : "v"(c), "v"(vm) : "s10", "s11");
        if (MODE == 33) asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\ts_and_saveexec_b64 s[10:11], vcc\n\tv_add_u32 %0, %1, %0\n\ts_mov_b64 exec, s[10:11]" : "+v"(a[i]) : "v"(c), "v"(vm) : "vcc", "s10", "s11");
        if (MODE == 34) asm volatile("v_cmpx_gt_u32 %1, %2\n\tv_add_u32 %0, %1, %0\n\ts_mov_b64 exec, -1" : "+v"(a[i]) : "v"(c), "v"(vm) : "vcc");
        if (MODE == 26) asm volatile("v_sub_f32 %0, %1, %0\n\tv_mul_f32 %0, %0, %0" : "+v"(a[i]) : "v"(c));
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
  for (int w : {3}) {
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
    run<12>("v_cndmask_b32_e64 (sgpr mask)", w, out);
    run<13>("v_bfi_b32", w, out);
    run<14>("v_min_u32", w, out);
    run<15>("v_min_f64", w, out);
    run<16>("v_cmp_lt_u64", w, out);
    run<17>("v_cmp_gt_u32", w, out);
    run<18>("v_mov_b64", w, out);
    run<19>("v_add_u32", w, out);
    run<20>("v_lshl_add_u64", w, out);
    run<21>("v_cmp_gt_u32 + v_cndmask (pair)", w, out);
    run<22>("v_mov_b32", w, out);
    run<23>("v_cndmask_b32 (vcc = 0)", w, out);
    run<24>("v_max3_u32", w, out);
    run<25>("v_lshl_or_b32", w, out);
    run<26>("v_sub_f32 + v_mul_f32 (pair)", w, out);
    run<27>("cmp; s_nop 1; cndmask vcc (2 VALU)", w, out);
    run<28>("cmp; 2 x cndmask vcc (3 VALU)", w, out);
    run<29>("cmp_e64 s; cndmask_e64 s (2 VALU)", w, out);
    run<30>("cmp; add; cndmask vcc (3 VALU)", w, out);
    run<31>("cmp_e64 s; 2 x cndmask_e64 s (3 VALU)", w, out);
    run<32>("cmp_e64 s; add; cndmask_e64 (3 VALU)", w, out);
    run<33>("cmp; saveexec; add; restore (2 VALU+2 SALU)", w, out);
    run<34>("cmpx; add; s_mov exec (2 VALU+1 SALU)", w, out);
  }
  return 0;
}
