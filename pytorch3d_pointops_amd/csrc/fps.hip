// fps.hip -- iterative farthest point sampling for gfx950.
//
// Replaces FarthestPointSampling (reference:
// csrc/sample_farthest_points/sample_farthest_points.h:55-76) with the CPU path's
// semantics (sample_farthest_points_cpu.cpp:14-103): idx[n][0] = start_idxs[n];
// every further sample is the FIRST index with the largest running min-distance
// to the selected set (std::max_element), unfused fp32 distances, -1 padding
// beyond min(lengths[n], K[n]).
//
// v1 layout: one workgroup of 1024 lanes (16 wave64) per cloud; the running
// min-distance array lives in HBM/L2 (`min_dist_ws`, N*P floats), each lane owns
// the points p = tid, tid+1024, ... so no lane ever reads another lane's entry.
// Per iteration: update + per-lane argmax (strict > keeps the lowest index),
// wave argmax by xor-butterfly on (value, index) keys preferring the LOWER index on
// ties, one LDS exchange across the 16 waves, broadcast of the winner.
#include <float.h>

#include "common.h"

namespace pointops {

constexpr int kFpsBlock = 1024;
constexpr int kFpsWaves = kFpsBlock / kWave;

__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
  // larger value wins; on equal values the lower index wins (first maximum)
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

template <int DT>
__global__ __launch_bounds__(kFpsBlock) void fps_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ lengths,
    const int64_t* __restrict__ Ks, const int64_t* __restrict__ start_idxs, int P, int Drt,
    int max_K, int64_t* __restrict__ idxs, float* __restrict__ min_dist) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  int len = (int)lengths[n];
  if (len > P) len = P;
  int64_t kn64 = Ks[n];
  int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
  if (kn > max_K) kn = max_K;
  if (kn < 0) kn = 0;
  int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
  // -1 padding beyond the samples this cloud produces
  for (int k = (kn > 0 ? kn : 0) + tid; k < max_K; k += kFpsBlock) out[k] = -1;
  if (len <= 0 || kn <= 0) return;

  const float* __restrict__ pts = points + (int64_t)n * P * D;
  float* __restrict__ md = min_dist + (int64_t)n * P;
  for (int p = tid; p < len; p += kFpsBlock) md[p] = FLT_MAX;

  __shared__ float s_val[kFpsWaves];
  __shared__ int s_idx[kFpsWaves];
  __shared__ int s_last;

  int last = (int)start_idxs[n];
  if (last < 0 || last >= len) last = 0;  // guard (reference: undefined behaviour)
  if (tid == 0) out[0] = last;

  for (int k = 1; k < kn; ++k) {
    float best = -1.0f;
    int besti = 0x7fffffff;
    if constexpr (DT > 0) {
      float c[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) c[d] = pts[(int64_t)last * DT + d];  // wave-uniform
      for (int p = tid; p < len; p += kFpsBlock) {
        float acc;
        {
          const float diff = c[0] - pts[(int64_t)p * DT];
          acc = diff * diff;
        }
#pragma unroll
        for (int d = 1; d < DT; ++d) {
          const float diff = c[d] - pts[(int64_t)p * DT + d];
          acc = acc + diff * diff;
        }
        float m = md[p];
        if (acc < m) {
          m = acc;
          md[p] = m;
        }
        if (m > best) {
          best = m;
          besti = p;
        }
      }
    } else {
      const float* __restrict__ c = pts + (int64_t)last * D;
      for (int p = tid; p < len; p += kFpsBlock) {
        const float* __restrict__ b = pts + (int64_t)p * D;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) {
          const float diff = c[d] - b[d];
          acc = acc + diff * diff;
        }
        float m = md[p];
        if (acc < m) {
          m = acc;
          md[p] = m;
        }
        if (m > best) {
          best = m;
          besti = p;
        }
      }
    }
    // wave64 argmax butterfly
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, kWave);
      const int oi = __shfl_xor(besti, off, kWave);
      argmax_combine(best, besti, ov, oi);
    }
    if (lane == 0) {
      s_val[wave] = best;
      s_idx[wave] = besti;
    }
    __syncthreads();
    if (wave == 0) {
      float v = lane < kFpsWaves ? s_val[lane] : -2.0f;
      int ix = lane < kFpsWaves ? s_idx[lane] : 0x7fffffff;
#pragma unroll
      for (int off = kFpsWaves / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, kWave);
        const int oi = __shfl_xor(ix, off, kWave);
        argmax_combine(v, ix, ov, oi);
      }
      if (lane == 0) {
        s_last = ix;
        out[k] = ix;
      }
    }
    __syncthreads();
    last = s_last;
  }
}


// ---------------------------------------------------------------------------
// v2: register-resident clusters.  A cloud is split over G workgroups (G = ceil(P /
// (PPT*1024)), up to one workgroup per CU); every lane keeps its PPT points AND their
// running min-distances in VGPRs, so an iteration touches no memory except the
// exchange: each workgroup reduces its local argmax and publishes it with ONE relaxed
// agent-scope 8-byte store into its own slot of the (cloud, iteration) row -- key =
// distance bits << 32 | ~index, never 0; the key IS the message, so there is no flag, no
// counter and no fence (cdna guide G16, R2 "the data is the flag") -- then wave 0 polls the
// G slots of the row (relaxed agent-scope loads + s_sleep) until all are non-zero and
// takes their maximum: the largest distance and, on ties, the LOWEST index, i.e.
// std::max_element's first maximum.  Rows are per iteration, zeroed by a memset node before
// the launch; only agent-scope atomics ever touch them, the points are read-only input.  All workgroups of the grid are resident by
// construction (grid <= number of CUs, one 1024-lane workgroup per CU), and every spin
// is bounded.
// Measured and dropped: four self-validating 8-byte units per slot (key + three tagged coordinates)
// so that the winner's coordinates arrive with its key instead of through the dependent load of
// pts[last] at the top of the iteration -- 3.07 -> 3.71 ms at 16 x 131072 -> 1024: the extra
// stores, the index-tracking reduction and the coordinate shuffles sit on the critical path and
// cost more than the (L2-resident) load they remove.
// ---------------------------------------------------------------------------
constexpr unsigned kFpsSpinLimit = 1u << 24;

template <int DT, int PPT>
__global__ __launch_bounds__(kFpsBlock) void fps_cluster_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ lengths,
    const int64_t* __restrict__ Ks, const int64_t* __restrict__ start_idxs, int N, int P, int max_K,
    int G, int n_clusters, unsigned long long* __restrict__ slots,
    unsigned* __restrict__ timeout_flag, int64_t* __restrict__ idxs) {
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const int cluster = blockIdx.x / G;
  const int member = blockIdx.x - cluster * G;
  __shared__ float s_val[kFpsWaves];
  __shared__ int s_idx[kFpsWaves];
  __shared__ int s_last;

  for (int n = cluster; n < N; n += n_clusters) {
    int len = (int)lengths[n];
    if (len > P) len = P;
    int64_t kn64 = Ks[n];
    int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
    if (kn > max_K) kn = max_K;
    if (kn < 0) kn = 0;
    int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
    if (member == 0) {
      for (int k = (kn > 0 ? kn : 0) + tid; k < max_K; k += kFpsBlock) out[k] = -1;
    }
    if (len <= 0 || kn <= 0) continue;

    const float* __restrict__ pts = points + (int64_t)n * P * DT;
    // this lane's points: p = member*PPT*1024 + i*1024 + tid  (ascending in i)
    float px[PPT][DT];
    float md[PPT];
    const int base = member * (PPT * kFpsBlock) + tid;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = base + i * kFpsBlock;
      const bool valid = p < len;
#pragma unroll
      for (int d = 0; d < DT; ++d) px[i][d] = valid ? pts[(int64_t)p * DT + d] : 0.0f;
      md[i] = valid ? FLT_MAX : -1.0f;  // -1: never the maximum
    }
    int last = (int)start_idxs[n];
    if (last < 0 || last >= len) last = 0;
    if (member == 0 && tid == 0) out[0] = last;
    unsigned long long* __restrict__ cslots = slots + (int64_t)n * max_K * G;

    for (int k = 1; k < kn; ++k) {
      float c[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) c[d] = pts[(int64_t)last * DT + d];  // wave-uniform, read-only input
      float best = -1.0f;
      int besti = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        float acc;
        {
          const float diff = c[0] - px[i][0];
          acc = diff * diff;
        }
#pragma unroll
        for (int d = 1; d < DT; ++d) {
          const float diff = c[d] - px[i][d];
          acc = acc + diff * diff;
        }
        float m = md[i];
        if (acc < m) m = acc;  // padded lanes keep -1 (acc >= 0 is never < -1)
        md[i] = m;
        if (m > best) {
          best = m;
          besti = base + i * kFpsBlock;
        }
      }
#pragma unroll
      for (int off = kWave / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, kWave);
        const int oi = __shfl_xor(besti, off, kWave);
        argmax_combine(best, besti, ov, oi);
      }
      if (lane == 0) {
        s_val[wave] = best;
        s_idx[wave] = besti;
      }
      __syncthreads();
      if (wave == 0) {
        float v = lane < kFpsWaves ? s_val[lane] : -2.0f;
        int ix = lane < kFpsWaves ? s_idx[lane] : 0x7fffffff;
#pragma unroll
        for (int off = kFpsWaves / 2; off > 0; off >>= 1) {
          const float ov = __shfl_xor(v, off, kWave);
          const int oi = __shfl_xor(ix, off, kWave);
          argmax_combine(v, ix, ov, oi);
        }
        int win = ix;  // valid in lane 0
        if (G > 1) {
          // Exchange, whole wave 0: lane 0 publishes this member's key with ONE relaxed agent-scope
          // store into its own slot of the (cloud, iteration) row -- the 8-byte key IS the message
          // (distance bits << 32 | ~index, never 0), so no flag and no fence are needed; then the
          // lanes poll the G slots of the row (one 128-byte line for G = 16) until all are non-zero
          // and take the maximum: largest distance, lowest index on ties.
          unsigned long long* __restrict__ rowk = cslots + (int64_t)k * G;
          if (lane == 0) {
            const unsigned long long key =
                (v >= 0.0f) ? (((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(0xffffffffu - (unsigned)ix))
                            : 1ull;  // this member holds no valid point (below every real key)
            __hip_atomic_store(rowk + member, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          unsigned long long best = 0ull;
          for (int base = 0; base < G; base += kWave) {
            const int m = base + lane;
            unsigned long long mine = 1ull;
            unsigned spins = 0;
            for (;;) {
              mine = m < G ? __hip_atomic_load(rowk + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull;
              if (__all(mine != 0ull)) break;
              __builtin_amdgcn_s_sleep(1);
              if (++spins > kFpsSpinLimit) {  // exit condition every wave reaches
                if (lane == 0) __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            best = mine > best ? mine : best;
          }
#pragma unroll
          for (int off = kWave / 2; off > 0; off >>= 1) {
            const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), off, kWave);
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, off, kWave);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o > best ? o : best;
          }
          win = (int)(0xffffffffu - (unsigned)(best & 0xffffffffull));
          if (win < 0 || win >= len) win = 0;  // only reachable after a timeout
        }
        if (lane == 0) {
          s_last = win;
          if (member == 0) out[k] = win;
        }
      }
      __syncthreads();
      last = s_last;
    }
    __syncthreads();  // s_last / s_val reuse by the next cloud of this cluster
  }
}

}  // namespace pointops

using namespace pointops;

static int fps_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 64;
  }
  return cus;
}

// cluster geometry of the register-resident kernel: points per lane and workgroups per cloud
static void fps_plan(int64_t P, int* ppt, int* G) {
  const int cus = fps_num_cus();
  int p = P <= 8 * (int64_t)kFpsBlock * cus ? 8 : 16;
  if (P <= 4 * (int64_t)kFpsBlock) p = 4;
  *ppt = p;
  *G = (int)ceil_div(P > 0 ? P : 1, (int64_t)p * kFpsBlock);
}

extern "C" size_t pointops_fps_workspace_bytes(int64_t N, int64_t P, int64_t max_K) {
  // v1 running min-distance array (N*P floats) + v2 exchange rows: one u64 slot per
  // (cloud, iteration, cluster member) + one timeout word
  int ppt, G;
  fps_plan(P, &ppt, &G);
  const size_t md = sizeof(float) * (size_t)(N * P);
  const size_t ex = sizeof(unsigned long long) * (size_t)(N * max_K) * (size_t)(G > 1 ? G : 1) + 64;
  return ((md + 255) & ~(size_t)255) + ex;
}

extern "C" int pointops_sample_farthest_points(const float* points, const int64_t* lengths,
                                               const int64_t* K, const int64_t* start_idxs,
                                               int64_t N, int64_t P, int64_t D, int64_t max_K,
                                               int64_t* idxs, void* workspace, size_t workspace_bytes,
                                               void* stream_) {
  POINTOPS_REQUIRE(N >= 0 && P >= 0 && D >= 1 && max_K >= 0, "sample_farthest_points: bad sizes");
  POINTOPS_REQUIRE(P < (1LL << 31) && max_K < (1LL << 31) && D < (1LL << 16) && N < (1LL << 31),
                   "sample_farthest_points: sizes must fit int32");
  if (N == 0 || max_K == 0) return POINTOPS_OK;
  POINTOPS_REQUIRE(workspace != nullptr && workspace_bytes >= pointops_fps_workspace_bytes(N, P, max_K),
                   "sample_farthest_points: workspace of %zu bytes required",
                   pointops_fps_workspace_bytes(N, P, max_K));
  hipStream_t stream = (hipStream_t)stream_;
  float* min_dist_ws = (float*)workspace;
  char* ex = (char*)workspace + ((sizeof(float) * (size_t)(N * P) + 255) & ~(size_t)255);
  int ppt, G;
  fps_plan(P, &ppt, &G);
  const size_t slot_bytes = sizeof(unsigned long long) * (size_t)(N * max_K) * (size_t)(G > 1 ? G : 1);
  unsigned long long* slots = (unsigned long long*)ex;
  unsigned* timeout_flag = (unsigned*)(ex + slot_bytes);

  // v2 (register-resident clusters) for D in {2,3}: up to PPT*1024 points per workgroup
  const int cus = fps_num_cus();
  if ((D == 3 || D == 2) && P >= 1) {
    if (G <= cus) {
      int n_clusters = cus / G;
      if (n_clusters > N) n_clusters = (int)N;
      if (n_clusters < 1) n_clusters = 1;
      if (G == 1) n_clusters = (int)N;  // no exchange: one independent workgroup per cloud
      if (G > 1) {
        if (hipMemsetAsync(ex, 0, slot_bytes + 64, stream) != hipSuccess) return check_launch("fps(memset)");
      }
      const dim3 grid((unsigned)(n_clusters * G)), block(kFpsBlock);
#define PO_LAUNCH_C(DT, PPT)                                                                         \
  hipLaunchKernelGGL((fps_cluster_kernel<DT, PPT>), grid, block, 0, stream, points, lengths, K, start_idxs, \
                     (int)N, (int)P, (int)max_K, G, n_clusters, slots, timeout_flag, idxs)
      if (D == 3) {
        if (ppt == 4) PO_LAUNCH_C(3, 4);
        else if (ppt == 8) PO_LAUNCH_C(3, 8);
        else PO_LAUNCH_C(3, 16);
      } else {
        if (ppt == 4) PO_LAUNCH_C(2, 4);
        else if (ppt == 8) PO_LAUNCH_C(2, 8);
        else PO_LAUNCH_C(2, 16);
      }
#undef PO_LAUNCH_C
      return check_launch("sample_farthest_points");
    }
  }
  const dim3 grid((unsigned)N), block(kFpsBlock);
#define PO_LAUNCH(DT)                                                                             \
  hipLaunchKernelGGL((fps_kernel<DT>), grid, block, 0, stream, points, lengths, K, start_idxs,     \
                     (int)P, (int)D, (int)max_K, idxs, min_dist_ws)
  switch (D) {
    case 2: PO_LAUNCH(2); break;
    case 3: PO_LAUNCH(3); break;
    default: PO_LAUNCH(0); break;
  }
#undef PO_LAUNCH
  return check_launch("sample_farthest_points");
}
