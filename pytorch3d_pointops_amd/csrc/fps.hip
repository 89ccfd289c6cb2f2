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
//
// MEMORY-MODEL NOTE on the cluster kernel's fastest exchange (fps_mode 2, the default where it verifies): members of
// a cloud publish their candidate with WORKGROUP-scope stores that OTHER workgroups poll with L1-bypassing loads.
// That hand-off is not promised by the HIP memory model; it works because gfx950's vector L1 is write-through and
// all members of a verified cloud share one XCD's L2.  It is therefore never trusted for correctness: a cloud only
// uses it after its members have proven (through the agent-scope protocol, row 0) that they run on one XCD, every
// spin is bounded, and a cloud whose exchange times out is flagged and recomputed by the single-workgroup kernel
// (tests force that with fps_spin_limit=0).  On another chip the guard would route every cloud to the agent-scope
// exchange or the repair pass -- slower, still exact.
#include <float.h>

#include "common.h"
#include "debug.h"

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
    int max_K, int64_t* __restrict__ idxs, float* __restrict__ min_dist, const unsigned* __restrict__ only_flagged) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  // repair pass behind the cluster kernel: only the clouds whose exchange timed out are redone here
  if (only_flagged != nullptr && only_flagged[n] == 0u) return;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  int len = (int)lengths[n];
  if (len > P) len = P;
  int64_t kn64 = Ks[n];
  int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
  if (kn > max_K) kn = max_K;
  if (kn < 0) kn = 0;
  int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
  // -1 padding beyond the samples this cloud produces; a non-empty cloud always reports its start index, even with
  // K[n] = 0 (sample_farthest_points_cpu.cpp:53-57 writes it before looking at K)
  const int keep = len > 0 ? (kn > 1 ? kn : 1) : 0;
  for (int k = keep + tid; k < max_K; k += kFpsBlock) out[k] = -1;
  if (len <= 0) return;
  if (kn <= 1) {
    int first = (int)start_idxs[n];
    if (first < 0 || first >= len) first = 0;
    if (tid == 0) out[0] = first;
    return;
  }

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
// exchange: each workgroup reduces its local argmax and publishes it with ONE 8-byte store into its
// own slot of the (cloud, iteration) row -- key = distance bits << 32 | ~index, never 0; the key IS
// the message, so there is no flag, no counter and no fence (cdna guide G16, R2 "the data is the
// flag") -- then wave 0 polls the G slots of the row (L1-bypassing loads + s_sleep) until all are
// non-zero and takes their maximum: the largest distance and, on ties, the LOWEST index, i.e.
// std::max_element's first maximum.  Rows are per iteration, zeroed by a memset node before
// the launch; the points are read-only input.
//
// Placement (MODE, debug knob fps_mode).  Workgroups are dealt round-robin over the 8 XCDs, so blocks
// b and b + 8 share an XCD and its L2.  MODE 0 (round 1): member = blockIdx % G -- the G members of a
// cloud sit on all 8 XCDs, every exchange crosses the fabric and the dependent load of the winner's
// coordinates misses this XCD's L2 7 times out of 8.  MODE 1: the members of a cloud are blocks with
// equal blockIdx % 8 (one XCD when G <= CUs/8), same agent-scope exchange.  MODE 2: as 1, plus each
// cloud VERIFIES its placement once -- every member publishes its HW_REG_XCC_ID through the agent-scope
// protocol (row 0 of the cloud's slots, unused otherwise) -- and when all members report one XCD the
// per-iteration stores become workgroup-scope (written through the L1 into that XCD's L2, where the
// pollers' L1-bypassing loads find them) instead of agent-scope write-through to memory.  Placement is
// never assumed for correctness: an unverified cloud keeps the agent-scope protocol.
//
// Liveness.  All workgroups of the grid are resident by construction (grid <= CUs x occupancy of this
// kernel, queried from the runtime) unless another kernel or a CU mask takes CUs away; so every spin is
// bounded, a member whose wait times out flags ITS CLOUD in `timeout_flags` and abandons it, and the
// host enqueues the single-workgroup kernel behind this one for exactly the flagged clouds.  Results are
// therefore always the reference's; a timeout only costs time.
// Measured and dropped: four self-validating 8-byte units per slot (key + three tagged coordinates)
// so that the winner's coordinates arrive with its key instead of through the dependent load of
// pts[last] at the top of the iteration -- 3.07 -> 3.71 ms at 16 x 131072 -> 1024.
// ---------------------------------------------------------------------------
constexpr int kXcds = 8;

// Argmax keys: (running min-distance bits << 32) | ~index.  Distances are >= +0 (padded slots carry -1.0f), so
// such a key, read as a double, is a non-NaN double whose order is the order we want: larger distance first,
// then the LOWER index (std::max_element's first maximum); a padded slot is a negative double, below every real
// one.  One v_max_f64 therefore replaces "compare distances, compare indices, two selects" -- and v_cndmask_b32
// pairs on a shared VCC cost ~22 cycles each on gfx950 (tools/valu_microbench.hip).
__device__ __forceinline__ double fps_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fps_min(float a, float b) {  // IEEE minNum: a NaN distance never replaces the minimum
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// k = max(k, k of the lane that DPP control CTRL names); lanes outside the rows of ROWS keep k.  Two v_mov_b32_dpp and
// one v_max_f64: ~25 cycles a step, where the ds_bpermute pair of __shfl_xor costs ~120 (a trip through the LDS pipe)
template <int CTRL, int ROWS>
__device__ __forceinline__ double fps_dpp_max(double k) {
  const int hi = __double2hiint(k), lo = __double2loint(k);
  const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false);
  const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWS, 0xf, false);
  return fps_max(k, __hiloint2double(ohi, olo));
}
// maximum over each row of 16 lanes, in every lane of the row
__device__ __forceinline__ double fps_row_max(double k) {
  k = fps_dpp_max<0xB1, 0xf>(k);   // quad_perm [1,0,3,2]
  k = fps_dpp_max<0x4E, 0xf>(k);   // quad_perm [2,3,0,1]
  k = fps_dpp_max<0x124, 0xf>(k);  // row_ror 4
  k = fps_dpp_max<0x128, 0xf>(k);  // row_ror 8
  return k;
}
// maximum over the wave, wave-uniform (row maxima carried up the rows into lane 63, read back through SGPRs)
__device__ __forceinline__ double fps_wave_max(double k) {
  k = fps_row_max(k);
  k = fps_dpp_max<0x142, 0xa>(k);  // row_bcast15 into rows 1, 3
  k = fps_dpp_max<0x143, 0xc>(k);  // row_bcast31 into rows 2, 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(k), 63), __builtin_amdgcn_readlane(__double2loint(k), 63));
}

__device__ __forceinline__ unsigned xcc_id() {
  return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;  // HW_REG_XCC_ID[3:0]
}

// wave 0 of a member: wait until all G slots of `row` are non-zero, return their maximum (every lane);
// 0 after `spin_limit` polls without success.  Slot values are argmax keys (or the tiny placeholders 1 /
// 0x100 | xcc of the placement round): non-negative doubles, so the maximum is again v_max_f64.
__device__ __forceinline__ unsigned long long fps_gather_row(const unsigned long long* __restrict__ row, int G,
                                                             int lane, unsigned spin_limit) {
  double best = 0.0;
  bool ok = true;
  for (int base = 0; base < G && ok; base += kWave) {
    const int m = base + lane;
    unsigned long long mine = 1ull;
    unsigned spins = 0;
    for (;;) {
      mine = m < G ? __hip_atomic_load(row + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull;
      if (__all(mine != 0ull)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {  // exit condition every wave reaches
        ok = false;
        break;
      }
    }
    best = fps_max(best, __longlong_as_double((long long)mine));
  }
  best = fps_wave_max(best);
  return ok ? (unsigned long long)__double_as_longlong(best) : 0ull;
}

template <int DT, int PPT>
__global__ __launch_bounds__(kFpsBlock) void fps_cluster_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ lengths,
    const int64_t* __restrict__ Ks, const int64_t* __restrict__ start_idxs, int N, int P, int max_K,
    int G, int n_clusters, int mode, unsigned spin_limit, unsigned long long* __restrict__ slots,
    unsigned* __restrict__ timeout_flags, int64_t* __restrict__ idxs) {
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  int cluster, member;
  if (mode == 0) {
    cluster = blockIdx.x / G;
    member = blockIdx.x - cluster * G;
  } else {
    // blocks with equal blockIdx % 8 share an XCD: deal each XCD's blocks to whole clusters
    const int x = blockIdx.x % kXcds, j = blockIdx.x / kXcds;  // j-th block of XCD group x
    cluster = (j / G) * kXcds + x;
    member = j % G;
  }
  if (cluster >= n_clusters) return;  // (grid rounded up to a multiple of 8 * G)
  __shared__ double s_key[kFpsWaves];
  __shared__ int s_last;
  __shared__ int s_flag;

  for (int n = cluster; n < N; n += n_clusters) {
    int len = (int)lengths[n];
    if (len > P) len = P;
    int64_t kn64 = Ks[n];
    int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
    if (kn > max_K) kn = max_K;
    if (kn < 0) kn = 0;
    int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
    const int keep = len > 0 ? (kn > 1 ? kn : 1) : 0;  // (the start index is reported even with K[n] = 0: cpu.cpp:53-57)
    if (member == 0) {
      for (int k = keep + tid; k < max_K; k += kFpsBlock) out[k] = -1;
    }
    if (len <= 0) continue;
    if (kn <= 1) {  // (uniform over the cloud's members)
      int first = (int)start_idxs[n];
      if (first < 0 || first >= len) first = 0;
      if (member == 0 && tid == 0) out[0] = first;
      continue;
    }

    const float* __restrict__ pts = points + (int64_t)n * P * DT;
    // this lane's points: p = member*PPT*1024 + i*1024 + tid  (ascending in i); mk[i] = argmax key of point i:
    // hi word = its running min-distance (the only part an iteration rewrites), lo word = ~index
    float px[PPT][DT];
    double mk[PPT];
    const int base = member * (PPT * kFpsBlock) + tid;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = base + i * kFpsBlock;
      const bool valid = p < len;
#pragma unroll
      for (int d = 0; d < DT; ++d) px[i][d] = valid ? pts[(int64_t)p * DT + d] : 0.0f;
      mk[i] = __hiloint2double(__float_as_int(valid ? FLT_MAX : -1.0f), (int)(0xffffffffu - (unsigned)p));  // -1: never the maximum
    }
    int last = (int)start_idxs[n];
    if (last < 0 || last >= len) last = 0;
    if (member == 0 && tid == 0) out[0] = last;
    unsigned long long* __restrict__ cslots = slots + (int64_t)n * max_K * G;

    // MODE 2: one verified-placement round per cloud (row 0): all members on one XCD?
    bool same_xcd = false;
    bool dead = false;  // this member's wait timed out: abandon the cloud (the repair pass redoes it)
    if (G > 1 && mode == 2) {
      if (wave == 0) {
        const unsigned long long mine = 0x100ull | xcc_id();
        if (lane == 0) __hip_atomic_store(cslots + member, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long mx = fps_gather_row(cslots, G, lane, spin_limit);
        // all equal <=> max == every slot; re-read (they are final) and compare
        bool eq = mx != 0ull;
        for (int b2 = 0; b2 < G; b2 += kWave) {
          const int m = b2 + lane;
          const unsigned long long v = m < G ? __hip_atomic_load(cslots + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mx;
          eq = eq && __all(v == mx);
        }
        if (lane == 0) s_flag = mx == 0ull ? -1 : (eq ? 1 : 0);
      }
      __syncthreads();
      same_xcd = s_flag == 1;
      dead = s_flag < 0;
    }

    if (G == 1) {
      // One workgroup owns the cloud: no exchange.  ONE barrier per iteration -- every wave reduces the 16 wave keys
      // itself (double-buffered by iteration parity: a wave can be at most one barrier ahead of the slowest reader) --
      // and, for clouds of up to 4096 points, the winner's coordinates come from an LDS copy of the cloud instead of a
      // dependent load from global memory (~0.5 us of the 1.0 us iteration).
      __shared__ double s_key2[2][kFpsWaves];
      __shared__ float s_pts[PPT == 4 ? PPT * kFpsBlock * DT : 1];
      if constexpr (PPT == 4) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
          const int p = tid + i * kFpsBlock;
#pragma unroll
          for (int d = 0; d < DT; ++d) s_pts[p * DT + d] = px[i][d];
        }
        __syncthreads();
      }
      for (int k = 1; k < kn; ++k) {
        float c[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) c[d] = PPT == 4 ? s_pts[last * DT + d] : pts[(int64_t)last * DT + d];
        double best = __hiloint2double(__float_as_int(-1.0f), 0);
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
          const float m = fps_min(acc, __int_as_float(__double2hiint(mk[i])));
          mk[i] = __hiloint2double(__float_as_int(m), __double2loint(mk[i]));
          best = fps_max(best, mk[i]);
        }
        best = fps_wave_max(best);
        if (lane == 0) s_key2[k & 1][wave] = best;
        __syncthreads();
        const double v = fps_row_max(s_key2[k & 1][lane & (kFpsWaves - 1)]);  // every lane: the workgroup's maximum
        last = __builtin_amdgcn_readfirstlane((int)(0xffffffffu - (unsigned)__double2loint(v)));
        if (tid == 0) out[k] = last;
      }
      __syncthreads();  // s_pts / s_key2 reuse by the next cloud of this cluster
      continue;
    }

    for (int k = 1; k < kn && !dead; ++k) {
      float c[DT];
#pragma unroll
      for (int d = 0; d < DT; ++d) c[d] = pts[(int64_t)last * DT + d];  // wave-uniform, read-only input
      double best = __hiloint2double(__float_as_int(-1.0f), 0);
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
        // padded slots keep -1 (acc >= 0 is never below it)
        const float m = fps_min(acc, __int_as_float(__double2hiint(mk[i])));
        mk[i] = __hiloint2double(__float_as_int(m), __double2loint(mk[i]));
        best = fps_max(best, mk[i]);
      }
      best = fps_wave_max(best);
      if (lane == 0) s_key[wave] = best;
      __syncthreads();
      if (wave == 0) {
        static_assert(kFpsWaves == 16, "one DPP row holds the waves' keys");
        double v = lane < kFpsWaves ? s_key[lane] : __hiloint2double(__float_as_int(-2.0f), 0);
        v = fps_row_max(v);  // (lanes 0..15: the workgroup's maximum)
        // this member's key: distance bits << 32 | ~index (never 0); 1 when it holds no valid point
        const unsigned long long key = __double2hiint(v) >= 0 ? (unsigned long long)__double_as_longlong(v) : 1ull;
        int win = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));  // valid in lane 0 (G == 1: the result)
        if (G > 1) {
          // Exchange, whole wave 0: lane 0 publishes this member's key with ONE store into its own slot
          // of the (cloud, iteration) row -- the 8-byte key IS the message -- then the lanes poll the G slots of
          // the row (one 128-byte line for G = 16) until all are non-zero and take the maximum: largest
          // distance, lowest index on ties.
          unsigned long long* __restrict__ rowk = cslots + (int64_t)k * G;
          if (lane == 0) {
            if (same_xcd) __hip_atomic_store(rowk + member, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(rowk + member, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          const unsigned long long bestk = fps_gather_row(rowk, G, lane, spin_limit);
          win = (int)(0xffffffffu - (unsigned)(bestk & 0xffffffffull));
          if (bestk == 0ull) {
            win = -1;
            if (lane == 0) __hip_atomic_store(timeout_flags + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (lane == 0) {
          s_last = win;
          if (member == 0 && win >= 0) out[k] = win;
        }
      }
      __syncthreads();
      last = s_last;
      if (last < 0) dead = true;  // (workgroup-uniform)
    }
    if (dead && tid == 0) __hip_atomic_store(timeout_flags + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // s_last / s_val reuse by the next cloud of this cluster
  }
}

// ---------------------------------------------------------------------------
// Small clouds (<= 4096 points, D in {2,3}): ONE workgroup of FOUR waves per cloud -- a wave per SIMD.  Same
// register-resident keys and LDS copy of the cloud as the single-workgroup path above, but a barrier of 4 waves
// instead of 16 and, up to 1024 / 2048 points, a quarter / half of the arithmetic per iteration: 2 x 1024 points 0.84 ->
// 0.44 us per iteration, 2 x 4096 0.85 -> 0.77 (the reference's example sizes; tools/fps_small.py).
// ---------------------------------------------------------------------------
constexpr int kFpsSmallBlock = 256;
constexpr int kFpsSmallWaves = kFpsSmallBlock / kWave;

template <int DT, int PPT>
__global__ __launch_bounds__(kFpsSmallBlock) void fps_small_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ lengths, const int64_t* __restrict__ Ks,
    const int64_t* __restrict__ start_idxs, int P, int max_K, int64_t* __restrict__ idxs) {
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  __shared__ double s_key2[2][kFpsSmallWaves];
  __shared__ float s_pts[PPT * kFpsSmallBlock * DT];
  int len = (int)lengths[n];
  if (len > P) len = P;
  int64_t kn64 = Ks[n];
  int kn = (int)(kn64 < (int64_t)len ? kn64 : (int64_t)len);
  if (kn > max_K) kn = max_K;
  if (kn < 0) kn = 0;
  int64_t* __restrict__ out = idxs + (int64_t)n * max_K;
  const int keep = len > 0 ? (kn > 1 ? kn : 1) : 0;  // (the start index is reported even with K[n] = 0: cpu.cpp:53-57)
  for (int k = keep + tid; k < max_K; k += kFpsSmallBlock) out[k] = -1;
  if (len <= 0) return;
  if (kn <= 1) {
    int first = (int)start_idxs[n];
    if (first < 0 || first >= len) first = 0;
    if (tid == 0) out[0] = first;
    return;
  }
  const float* __restrict__ pts = points + (int64_t)n * P * DT;
  float px[PPT][DT];
  double mk[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid + i * kFpsSmallBlock;
    const bool valid = p < len;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      px[i][d] = valid ? pts[(int64_t)p * DT + d] : 0.0f;
      s_pts[p * DT + d] = px[i][d];
    }
    mk[i] = __hiloint2double(__float_as_int(valid ? FLT_MAX : -1.0f), (int)(0xffffffffu - (unsigned)p));
  }
  int last = (int)start_idxs[n];
  if (last < 0 || last >= len) last = 0;
  if (tid == 0) out[0] = last;
  __syncthreads();
  for (int k = 1; k < kn; ++k) {
    float c[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) c[d] = s_pts[last * DT + d];
    double best = __hiloint2double(__float_as_int(-1.0f), 0);
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
      const float m = fps_min(acc, __int_as_float(__double2hiint(mk[i])));
      mk[i] = __hiloint2double(__float_as_int(m), __double2loint(mk[i]));
      best = fps_max(best, mk[i]);
    }
    best = fps_wave_max(best);
    if (lane == 0) s_key2[k & 1][wave] = best;
    __syncthreads();
    const double v = fps_row_max(s_key2[k & 1][lane & (kFpsSmallWaves - 1)]);  // every lane: the workgroup's maximum
    last = __builtin_amdgcn_readfirstlane((int)(0xffffffffu - (unsigned)__double2loint(v)));
    if (tid == 0) out[k] = last;
  }
}

}  // namespace pointops

using namespace pointops;

static int fps_num_cus() {
  // per DEVICE (a process may drive several GPUs; round 2 remembered the first device's count for all of them)
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    int c = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
    cus[dev] = c > 0 ? c : 64;
  }
  return cus[dev];
}

// cluster geometry of the register-resident kernel: points per lane and workgroups per cloud.
// Round 3: a cloud that needs an exchange anyway (more than 8192 points) takes FOUR points per lane -- twice the
// workgroups, half the arithmetic per iteration and workgroup -- while the batch still gets a CU per workgroup and a cloud
// still fits one XCD: 1 x 131072 -> 1024: 1.50 -> 1.35 ms, 4 x 32768 -> 512: 0.73 -> 0.64, 32 x 16384 -> 256: 0.38 -> 0.34.
// Not beyond that: two workgroups per CU (the <3, 4> kernel held to 64 VGPRs) do not overlap each other's latency
// (16 x 131072: 1.51 -> 1.73 ms), and a cloud that fits one workgroup is fastest without an exchange (64 x 8192 -> 256:
// 0.26 against 0.33 ms).  tools/fps_plan_sweep.py, profiles/r03_fps_plan_sweep.txt.
// D <= 0: the plan with the most workgroups per cloud any D could take (workspace sizing).
static void fps_plan(int64_t N, int64_t P, int64_t D, int* ppt, int* G) {
  const int cus = fps_num_cus();
  int p = P <= 8 * (int64_t)kFpsBlock * cus ? 8 : 16;
  if (P <= 4 * (int64_t)kFpsBlock) p = 4;
  if (p == 8 && (D == 3 || D <= 0) && debug_knob("fps_small_ppt", 1) != 0) {
    const int64_t g4 = ceil_div(P, (int64_t)4 * kFpsBlock);
    if (P > 8 * (int64_t)kFpsBlock && N * g4 <= (int64_t)cus && g4 <= (int64_t)cus / 8) p = 4;
  }
  const long forced = debug_knob("fps_ppt", 0);  // experiments: 4 | 8 | 16
  if (forced == 4 || forced == 8 || forced == 16) p = (int)forced;
  *ppt = p;
  *G = (int)ceil_div(P > 0 ? P : 1, (int64_t)p * kFpsBlock);
}

// workgroups of `kernel` the device can hold at once: CUs x the runtime's occupancy answer (the cluster
// exchange needs every workgroup of the grid resident)
template <class Kernel>
static int fps_resident_blocks(Kernel kernel) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kFpsBlock, 0) != hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    return 0;
  }
  return fps_num_cus() * (per_cu > 1 ? 1 : per_cu);  // one 1024-lane workgroup per CU is what the plan uses
}

static size_t fps_slot_bytes(int64_t N, int64_t max_K, int G) {
  return sizeof(unsigned long long) * (size_t)(N * max_K) * (size_t)(G > 1 ? G : 1);
}

extern "C" size_t pointops_fps_workspace_bytes(int64_t N, int64_t P, int64_t max_K) {
  // v1 running min-distance array (N*P floats) + v2 exchange rows: one u64 slot per
  // (cloud, iteration, cluster member) + one timeout word per cloud
  int ppt, G;
  fps_plan(N, P, 0, &ppt, &G);
  const size_t md = sizeof(float) * (size_t)(N * P);
  const size_t ex = fps_slot_bytes(N, max_K, G) + sizeof(unsigned) * (size_t)N + 64;
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
  fps_plan(N, P, D, &ppt, &G);
  const size_t slot_bytes = fps_slot_bytes(N, max_K, G);
  unsigned long long* slots = (unsigned long long*)ex;
  unsigned* timeout_flags = (unsigned*)(ex + slot_bytes);

#define PO_LAUNCH(DT, FLAGS)                                                                      \
  hipLaunchKernelGGL((fps_kernel<DT>), dim3((unsigned)N), dim3(kFpsBlock), 0, stream, points, lengths, K, \
                     start_idxs, (int)P, (int)D, (int)max_K, idxs, min_dist_ws, FLAGS)
  // small clouds: one four-wave workgroup per cloud
  if ((D == 3 || D == 2) && P >= 1 && P <= 16 * kFpsSmallBlock && debug_knob("fps_small", 1) != 0) {
#define PO_SMALL(DT, PPT)                                                                                          \
  hipLaunchKernelGGL((fps_small_kernel<DT, PPT>), dim3((unsigned)N), dim3(kFpsSmallBlock), 0, stream, points, lengths, K, \
                     start_idxs, (int)P, (int)max_K, idxs)
    if (D == 3) {
      if (P <= 4 * kFpsSmallBlock) PO_SMALL(3, 4);
      else if (P <= 8 * kFpsSmallBlock) PO_SMALL(3, 8);
      else PO_SMALL(3, 16);
    } else {
      if (P <= 4 * kFpsSmallBlock) PO_SMALL(2, 4);
      else if (P <= 8 * kFpsSmallBlock) PO_SMALL(2, 8);
      else PO_SMALL(2, 16);
    }
#undef PO_SMALL
    return check_launch("sample_farthest_points(small)");
  }
  // v2 (register-resident clusters) for D in {2,3}: up to PPT*1024 points per workgroup
  if ((D == 3 || D == 2) && P >= 1) {
    int resident = 0;
#define PO_RES(DT, PPT) resident = fps_resident_blocks(fps_cluster_kernel<DT, PPT>)
    if (D == 3) {
      if (ppt == 4) PO_RES(3, 4);
      else if (ppt == 8) PO_RES(3, 8);
      else PO_RES(3, 16);
    } else {
      if (ppt == 4) PO_RES(2, 4);
      else if (ppt == 8) PO_RES(2, 8);
      else PO_RES(2, 16);
    }
#undef PO_RES
    if (G <= resident) {
      int mode = (int)debug_knob("fps_mode", 2);
      const int per_xcd = resident / kXcds;  // blocks that share an XCD under round-robin dispatch
      if (G > per_xcd || per_xcd < 1) mode = 0;  // a cloud does not fit one XCD: spread it (agent-scope exchange)
      int n_clusters = mode == 0 ? resident / G : (per_xcd / G) * kXcds;
      if (n_clusters > N) n_clusters = (int)N;
      if (n_clusters < 1) n_clusters = 1;
      if (G == 1) {  // no exchange: one independent workgroup per cloud
        n_clusters = (int)N;
        mode = 0;
      }
      if (G > 1) {
        if (hipMemsetAsync(ex, 0, slot_bytes + sizeof(unsigned) * (size_t)N, stream) != hipSuccess)
          return check_launch("fps(memset)");
      }
      // XCD-local numbering: cluster c = (j / G) * 8 + x for the j-th block of XCD group x
      const int blocks = mode == 0 ? n_clusters * G : (int)ceil_div(n_clusters, kXcds) * G * kXcds;
      const unsigned spin_limit = (unsigned)debug_knob("fps_spin_limit", 1 << 20);
      const dim3 grid((unsigned)blocks), block(kFpsBlock);
#define PO_LAUNCH_C(DT, PPT)                                                                         \
  hipLaunchKernelGGL((fps_cluster_kernel<DT, PPT>), grid, block, 0, stream, points, lengths, K, start_idxs, \
                     (int)N, (int)P, (int)max_K, G, n_clusters, mode, spin_limit, slots, timeout_flags, idxs)
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
      int rc = check_launch("sample_farthest_points");
      if (rc != POINTOPS_OK || G == 1) return rc;
      // repair pass: clouds whose exchange timed out (CUs taken away by another kernel or a CU mask) are
      // redone by the single-workgroup kernel; a cloud that was not flagged returns at once
      if (D == 3) PO_LAUNCH(3, (const unsigned*)timeout_flags);
      else PO_LAUNCH(2, (const unsigned*)timeout_flags);
      return check_launch("sample_farthest_points(repair)");
    }
  }
  switch (D) {
    case 2: PO_LAUNCH(2, nullptr); break;
    case 3: PO_LAUNCH(3, nullptr); break;
    default: PO_LAUNCH(0, nullptr); break;
  }
#undef PO_LAUNCH
  return check_launch("sample_farthest_points");
}
