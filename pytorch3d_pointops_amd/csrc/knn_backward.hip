// knn_backward.hip -- gradient of the KNN / ball-query distances w.r.t. both clouds.
//
// Replaces KNearestNeighborBackward (reference: csrc/knn/knn.h:127-149) with the
// CPU path's semantics (csrc/knn/knn_cpu.cpp:75-128).  One lane per query point:
//   grad_p1[n,i,:] = sum_k c(n,i,k,:)   accumulated in registers in k order -- no
//       atomics, deterministic and in the same order as the CPU loop, so grad_p1 is
//       bit-equal to the reference CPU result;
//   grad_p2[n,idx,:] -= c               scatter-add with global_atomic_add_f32
//       (fp32, agent scope): the only cross-lane accumulation; order-dependent in
//       the last bits like the reference CUDA path (csrc/knn/knn.cu:514-515,538).
//   c = 2*g*(p1-p2) for L2, g*sign(p1>p2) for L1.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace pointops {

constexpr int kBwdBlock = 256;

template <int DT, int NORM>  // DT = compile-time D, 0 = runtime D
__global__ __launch_bounds__(kBwdBlock) void knn_backward_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2,
    const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists, int P1, int P2, int Drt,
    int K, int tiles_per_cloud, float* __restrict__ grad_p1, float* __restrict__ grad_p2) {
  const int D = DT > 0 ? DT : Drt;
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int i = tile * kBwdBlock + threadIdx.x;
  if (i >= P1) return;
  const int len1 = (int)lengths1[n];
  int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  float* __restrict__ g1 = grad_p1 + row * D;
  if (i >= len1) {
    for (int d = 0; d < D; ++d) g1[d] = 0.0f;
    return;
  }
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float* __restrict__ a = p1 + row * D;
  if constexpr (DT > 0) {
    float av[DT], acc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      av[d] = a[d];
      acc[d] = 0.0f;
    }
    for (int k = 0; k < kmax; ++k) {
      const int64_t i2 = irow[k];
      if (i2 < 0 || i2 >= P2) continue;  // -1 padding (ball query); also guards bad input
      const float g = grow[k];
      const float* __restrict__ b = p2 + ((int64_t)n * P2 + i2) * DT;
      float* __restrict__ g2 = grad_p2 + ((int64_t)n * P2 + i2) * DT;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const float bv = b[d];
        float diff;
        if (NORM == 1) diff = g * ((av[d] > bv) ? 1.0f : -1.0f);
        else diff = 2.0f * g * (av[d] - bv);
        acc[d] = acc[d] + diff;
        if (diff != 0.0f) atomicAdd(g2 + d, -1.0f * diff);  // +-0 never changes a sum started at +0
      }
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) g1[d] = acc[d];
  } else {
    for (int d = 0; d < D; ++d) {
      const float av = a[d];
      float acc = 0.0f;
      for (int k = 0; k < kmax; ++k) {
        const int64_t i2 = irow[k];
        if (i2 < 0 || i2 >= P2) continue;
        const float g = grow[k];
        const float bv = p2[((int64_t)n * P2 + i2) * D + d];
        float diff;
        if (NORM == 1) diff = g * ((av > bv) ? 1.0f : -1.0f);
        else diff = 2.0f * g * (av - bv);
        acc = acc + diff;
        if (diff != 0.0f) atomicAdd(grad_p2 + ((int64_t)n * P2 + i2) * D + d, -1.0f * diff);
      }
      g1[d] = acc;
    }
  }
}


// D <= 4: FOUR lanes per query, lane c owning coordinate c.  grad_p1[n,i,c] is still one lane's
// register sum in k order (bit-equal to the CPU loop); the grad_p2 atomics of one neighbour row
// now leave as ONE wave instruction over 16 rows x 12-16 contiguous bytes instead of three
// instructions over 64 different rows each (the memory-side fp32 atomic rate depends on the rows
// touched per instruction: MI355X_MICROARCH.md, global float atomics).
template <int NORM, bool SCATTER>
__global__ __launch_bounds__(kBwdBlock) void knn_backward4_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2,
    const int64_t* __restrict__ lengths1, const int64_t* __restrict__ lengths2,
    const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists, int P1, int P2, int D, int K,
    int tiles_per_cloud, float* __restrict__ grad_p1, float* __restrict__ grad_p2) {
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int t = tile * kBwdBlock + threadIdx.x;
  const int i = t >> 2, c = t & 3;
  if (i >= P1 || c >= D) return;
  const int len1 = (int)lengths1[n];
  const int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  if (i >= len1) {
    grad_p1[row * D + c] = 0.0f;
    return;
  }
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float av = p1[row * D + c];
  float acc = 0.0f;
  for (int k = 0; k < kmax; ++k) {
    const int64_t i2 = irow[k];
    if (i2 < 0 || i2 >= P2) continue;  // -1 padding (ball query); also guards bad input
    const float g = grow[k];
    const int64_t o = ((int64_t)n * P2 + i2) * D + c;
    const float bv = p2[o];
    float diff;
    if (NORM == 1) diff = g * ((av > bv) ? 1.0f : -1.0f);
    else diff = 2.0f * g * (av - bv);
    acc = acc + diff;
    if (SCATTER && diff != 0.0f) atomicAdd(grad_p2 + o, -1.0f * diff);
  }
  grad_p1[row * D + c] = acc;
}


// grad_p1 only (the LDS-tile path below owns grad_p2): ONE lane per query row, the row's idx and
// grad entries fetched as 16-byte vectors eight neighbours at a time, all issued before use, so a
// row's 128-byte line is consumed while it is in flight instead of being re-fetched per k (the
// 4-lanes-per-row kernel above exists for contiguous atomics, which this path does not issue).
// The sum stays one register chain in k order: bit-equal to the CPU loop.
struct alignas(8) I64x2 { int64_t v[2]; };
struct alignas(4) F32x4 { float v[4]; };
struct alignas(4) F32x3 { float v[3]; };

template <int DT, int NORM>  // DT = compile-time D (1..4)
__global__ __launch_bounds__(kBwdBlock) void knn_backward_rows_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists,
    int P1, int P2, int K, int tiles_per_cloud, float* __restrict__ grad_p1) {
  const int n = blockIdx.x / tiles_per_cloud;
  const int tile = blockIdx.x - n * tiles_per_cloud;
  const int i = tile * kBwdBlock + threadIdx.x;
  if (i >= P1) return;
  const int len1 = (int)lengths1[n];
  const int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  const int64_t row = (int64_t)n * P1 + i;
  float* __restrict__ g1 = grad_p1 + row * DT;
  float acc[DT], av[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) acc[d] = 0.0f;
  if (i >= len1) {
#pragma unroll
    for (int d = 0; d < DT; ++d) g1[d] = 0.0f;
    return;
  }
#pragma unroll
  for (int d = 0; d < DT; ++d) av[d] = p1[row * DT + d];
  const int64_t* __restrict__ irow = idxs + row * K;
  const float* __restrict__ grow = grad_dists + row * K;
  const float* __restrict__ p2n = p2 + (int64_t)n * P2 * DT;
  auto one = [&](int64_t i2, float g) {
    // -1 padding (ball query) and bad input contribute an exact +0 (branch-free: the eight
    // gathers of a chunk can all be in flight)
    const bool valid = i2 >= 0 && i2 < P2;
    i2 = valid ? i2 : 0;
    float bv[DT];
    if constexpr (DT == 3) {
      const F32x3 b = *reinterpret_cast<const F32x3*>(p2n + i2 * 3);
      bv[0] = b.v[0], bv[1] = b.v[1], bv[2] = b.v[2];
    } else {
#pragma unroll
      for (int d = 0; d < DT; ++d) bv[d] = p2n[i2 * DT + d];
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      float diff;
      if (NORM == 1) diff = g * ((av[d] > bv[d]) ? 1.0f : -1.0f);
      else diff = 2.0f * g * (av[d] - bv[d]);
      acc[d] = acc[d] + (valid ? diff : 0.0f);
    }
  };
  int k = 0;
  for (; k + 8 <= kmax; k += 8) {
    I64x2 iv[4];
    F32x4 gv[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) iv[q] = *reinterpret_cast<const I64x2*>(irow + k + 2 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) gv[q] = *reinterpret_cast<const F32x4*>(grow + k + 4 * q);
#pragma unroll
    for (int q = 0; q < 8; ++q) one(iv[q >> 1].v[q & 1], gv[q >> 2].v[q & 3]);
  }
  for (; k < kmax; ++k) one(irow[k], grow[k]);
#pragma unroll
  for (int d = 0; d < DT; ++d) g1[d] = acc[d];
}

// ---------------------------------------------------------------------------
// grad_p2 without device-scope atomics (D <= 4, P2 <= 16 tiles): the memory-side fp32 atomic
// rate (~55 G/s measured) bounds the kernels above at ~1.8 ms for 32 x 65536 x 16 neighbours,
// 30x the time their bytes need.  Here a 1024-thread workgroup owns one TILE of a cloud's p2
// rows as fp32 accumulators in LDS (<= 96 KB), streams the cloud's WHOLE (idx, grad) table
// (coalesced 8-byte loads), compacts the entries that point into its tile through a per-wave
// LDS stage (ballot + mbcnt) so that the gather/accumulate body runs on full waves, adds with
// ds_add_f32, and finally stores its tile with plain coalesced writes: every grad_p2 element is
// written exactly once, so there is no memset either.  The idx table is read once per tile
// (P2 / TILE times); the workgroups that share a table are placed on the SAME XCD
// (blockIdx % 8 = XCD under round-robin dispatch) and start together, so the re-reads are L2
// hits and HBM sees the table about once.  With few clouds the rows are split over S workgroups
// per tile, whose partial tiles meet with atomics (S * P2 * D of them instead of P1 * K * D).
// Measured (32 x 65536 x 16, D = 3): 0.85 ms against 1.85 ms for the device-atomic kernel
// (rows kernel 0.23 ms + this kernel 0.61 ms).  The LDS pipeline is busy for the whole kernel
// (SQ_ACTIVE_INST_LDS = kernel duration, SQ_WAIT_INST_LDS 37 % of wave cycles): ds_add_f32 retires
// ~1 lane every 3 cycles, i.e. ~200 G scatter-adds/s for the chip against ~55 G/s at the L2, so
// deeper prefetch of the table or of the gathers does not move it (tried: no change).
// ---------------------------------------------------------------------------
constexpr int kTiledBlock = 1024;
constexpr int kTiledAccFloats = 24576;  // 96 KB
constexpr int kTiledUnroll = 4;         // 64-entry groups scanned per step
constexpr int kTiledBatch = 128;        // staged entries accumulated per drain (two per lane)
constexpr int kTiledStage = kTiledBatch + 64 * kTiledUnroll;  // < one batch carried over + one scan step
constexpr int kXcds = 8;

__host__ __device__ inline int tiled_tile_points(int D) { return (kTiledAccFloats / D) & ~63; }

template <int DT, int NORM>  // DT = compile-time D (1..4)
__global__ __launch_bounds__(kTiledBlock) void knn_backward_tiled_kernel(
    const float* __restrict__ p1, const float* __restrict__ p2, const int64_t* __restrict__ lengths1,
    const int64_t* __restrict__ lengths2, const int64_t* __restrict__ idxs, const float* __restrict__ grad_dists,
    int N, int P1, int P2, int K, unsigned kmagic, int kshift, int parts, int S, float* __restrict__ grad_p2) {
  __shared__ float s_acc[kTiledAccFloats];
  __shared__ int2 s_stage[kTiledBlock / 64][kTiledStage];  // (entry, row in tile)

  // XCD-aware placement: the `parts` workgroups of one (cloud, row split) share an XCD
  const int x = blockIdx.x % kXcds, y = blockIdx.x / kXcds;
  const int cs = (y / parts) * kXcds + x, part = y % parts;
  if (cs >= N * S) return;
  const int n = cs / S, split = cs - n * S;
  const int tile = tiled_tile_points(DT);
  const int j0 = part * tile;
  const int jn = min(tile, P2 - j0);  // rows of this tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int t = tid; t < jn * DT; t += kTiledBlock) s_acc[t] = 0.0f;
  __syncthreads();

  const int len1 = (int)min((int64_t)P1, lengths1[n]);
  const int64_t len2 = lengths2[n];
  const int kmax = (int)(len2 < K ? len2 : K);
  // rows [r0, r1) of this split; entries are the flat (row, k) table
  const int rows_per = (len1 + S - 1) / S;
  const int r0 = min(len1, split * rows_per), r1 = min(len1, r0 + rows_per);
  const int64_t ebase = (int64_t)n * P1 * K;
  const int64_t* __restrict__ itab = idxs + ebase;
  const float* __restrict__ gtab = grad_dists + ebase;
  const float* __restrict__ p1n = p1 + (int64_t)n * P1 * DT;
  const float* __restrict__ p2t = p2 + ((int64_t)n * P2 + j0) * DT;
  const int e0 = r0 * K, e1 = r1 * K;  // P1 * K < 2^31 - 2^20 checked by the host
  int2* __restrict__ stage = s_stage[wave];
  int staged = 0;  // wave-uniform

  // A drain pops one batch of staged entries.  It is split in two so that the gathers of a batch
  // (its grad, p1 row and p2 row per entry, two entries per lane) fly while the wave goes on
  // scanning: drain_issue() starts them, drain_finish() -- one scan step later -- turns them into
  // LDS atomics.
  constexpr int kPer = kTiledBatch / 64;
  float dg[kPer], dav[kPer][DT], dbv[kPer][DT];
  int djl[kPer];
  bool don[kPer];
  bool pending = false;  // wave-uniform
  auto drain_issue = [&](int first, int count) {
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      don[q] = q * 64 + lane < count;
      const int2 se = don[q] ? stage[first + q * 64 + lane] : make_int2(e0, 0);
      const int e = se.x;
      djl[q] = se.y;
      const int i = kshift < 0 ? e : (int)(__umulhi((unsigned)e, kmagic) >> kshift);  // e / K
      const int k = e - i * K;
      don[q] = don[q] && k < kmax;
      dg[q] = gtab[e];
      if constexpr (DT == 3) {
        const F32x3 a = *reinterpret_cast<const F32x3*>(p1n + (int64_t)i * 3);
        const F32x3 b = *reinterpret_cast<const F32x3*>(p2t + (int64_t)djl[q] * 3);
#pragma unroll
        for (int c = 0; c < 3; ++c) dav[q][c] = a.v[c], dbv[q][c] = b.v[c];
      } else {
#pragma unroll
        for (int c = 0; c < DT; ++c) {
          dav[q][c] = p1n[(int64_t)i * DT + c];
          dbv[q][c] = p2t[(int64_t)djl[q] * DT + c];
        }
      }
    }
    pending = true;
  };
  auto drain_finish = [&]() {
    if (!pending) return;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      if (don[q]) {
#pragma unroll
        for (int c = 0; c < DT; ++c) {
          float diff;
          if (NORM == 1) diff = dg[q] * ((dav[q][c] > dbv[q][c]) ? 1.0f : -1.0f);
          else diff = 2.0f * dg[q] * (dav[q][c] - dbv[q][c]);
          if (diff != 0.0f) atomicAdd(&s_acc[djl[q] * DT + c], -1.0f * diff);
        }
      }
    }
    pending = false;
  };

  constexpr int kStep = 64 * kTiledUnroll;
  const int wave_stride = (kTiledBlock / 64) * kStep;
  auto load_step = [&](int eb, int64_t (&j)[kTiledUnroll]) {
#pragma unroll
    for (int u = 0; u < kTiledUnroll; ++u) {
      const int e = eb + u * 64 + lane;
      j[u] = e < e1 ? itab[e] : -1;
    }
  };
  // one scan step: 4 x 64 table entries (loaded two steps earlier) -> stage; then pop full batches
  auto scan_step = [&](int eb, int64_t (&jbuf)[kTiledUnroll]) {
    int64_t j[kTiledUnroll];
#pragma unroll
    for (int u = 0; u < kTiledUnroll; ++u) j[u] = jbuf[u];
    if (eb + 2 * wave_stride < e1) load_step(eb + 2 * wave_stride, jbuf);  // two steps ahead
#pragma unroll
    for (int u = 0; u < kTiledUnroll; ++u) {
      const uint64_t rel = (uint64_t)(j[u] - (int64_t)j0);
      const bool hit = rel < (uint64_t)jn;
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int pos =
            staged + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        stage[pos] = make_int2(eb + u * 64 + lane, (int)rel);
      }
      staged += __popcll(m);
    }
    drain_finish();  // the batch issued one step ago
    // < kTiledBatch + 4 * 64 staged here; full batches pop from the BACK of the stage
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    while (staged >= kTiledBatch) {
      drain_finish();
      staged -= kTiledBatch;
      drain_issue(staged, kTiledBatch);
    }
    __builtin_amdgcn_wave_barrier();
  };
  int eb = e0 + wave * kStep;
  int64_t ja[kTiledUnroll], jb[kTiledUnroll];
  if (eb < e1) load_step(eb, ja);
  if (eb + wave_stride < e1) load_step(eb + wave_stride, jb);
  for (; eb < e1; eb += 2 * wave_stride) {
    scan_step(eb, ja);
    if (eb + wave_stride < e1) scan_step(eb + wave_stride, jb);
  }
  drain_finish();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  drain_issue(0, staged);
  drain_finish();
  __syncthreads();

  float* __restrict__ out = grad_p2 + ((int64_t)n * P2 + j0) * DT;
  if (S == 1) {
    for (int t = tid; t < jn * DT; t += kTiledBlock) out[t] = s_acc[t];
  } else {
    for (int t = tid; t < jn * DT; t += kTiledBlock) {
      const float v = s_acc[t];
      if (v != 0.0f) atomicAdd(out + t, v);
    }
  }
}

// e / K for 0 <= e < 2^31 as umulhi(e, magic) >> shift: with l = ceil(log2 K) and
// magic = ceil(2^(31+l) / K) < 2^32 the quotient is exact for every 31-bit dividend
// (round-up method); K = 1 is flagged by shift = -1.
static void division_magic(unsigned K, unsigned* magic, int* shift) {
  int l = 0;
  while ((1ull << l) < K) ++l;
  if (l == 0) {
    *magic = 0;
    *shift = -1;
    return;
  }
  *magic = (unsigned)(((1ull << (31 + l)) + K - 1) / K);
  *shift = l - 1;
}

}  // namespace pointops

using namespace pointops;

extern "C" int pointops_knn_points_backward(const float* p1, const float* p2,
                                            const int64_t* lengths1, const int64_t* lengths2,
                                            const int64_t* idxs, const float* grad_dists, int64_t N,
                                            int64_t P1, int64_t P2, int64_t D, int64_t K, int norm,
                                            float* grad_p1, float* grad_p2, void* stream_) {
  POINTOPS_REQUIRE(norm == 1 || norm == 2, "knn_points_backward: norm must be 1 or 2");
  POINTOPS_REQUIRE(N >= 0 && P1 >= 0 && P2 >= 0 && D >= 1 && K >= 0, "knn_points_backward: bad sizes");
  POINTOPS_REQUIRE(P1 < (1LL << 31) && P2 < (1LL << 31) && K < (1LL << 31) && D < (1LL << 16),
                   "knn_points_backward: sizes must fit int32");
  hipStream_t stream = (hipStream_t)stream_;
  // grad_p2 mode: LDS tiles when the neighbour table is large and p2 splits into few tiles
  int parts = 0, S = 1;
  bool tiled = false;
  if (D <= 4 && N > 0 && P1 > 0 && P2 > 0 && K > 0) {
    parts = (int)ceil_div(P2, tiled_tile_points((int)D));
    tiled = parts <= 16 && N * P1 * K >= (1 << 21) && P1 * K < (1LL << 31) - (1 << 20);
    if (const char* e = getenv("POINTOPS_KNN_BWD_MODE")) {  // A/B measurements and tests
      if (e[0] == 'a') tiled = false;
      if (e[0] == 't') tiled = parts <= 64 && P1 * K < (1LL << 31) - (1 << 20);
    }
    if (tiled) {
      // few clouds: split the rows so that ~256 workgroups exist; partial tiles meet with atomics
      const int64_t wgs = N * parts;
      if (wgs < 192) S = (int)std::min<int64_t>(ceil_div(256, wgs), std::max<int64_t>(1, P1 / 4096));
      if (const char* e = getenv("POINTOPS_KNN_BWD_SPLIT")) S = std::max(1, atoi(e));
    }
  }
  if (N * P2 * D > 0 && !(tiled && S == 1)) {
    if (hipMemsetAsync(grad_p2, 0, sizeof(float) * (size_t)(N * P2 * D), stream) != hipSuccess)
      return check_launch("knn_points_backward(memset)");
  }
  if (N == 0 || P1 == 0) return POINTOPS_OK;
  if (D <= 4) {
    const int tiles4 = (int)ceil_div(P1 * 4, kBwdBlock);
    POINTOPS_REQUIRE(N * tiles4 < (1LL << 31), "knn_points_backward: grid too large");
    const dim3 grid4((unsigned)(N * tiles4)), block4(kBwdBlock);
#define PO_LAUNCH4(NORM, SCATTER)                                                                               \
  hipLaunchKernelGGL((knn_backward4_kernel<NORM, SCATTER>), grid4, block4, 0, stream, p1, p2, lengths1, lengths2, \
                     idxs, grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles4, grad_p1, grad_p2)
    if (!tiled) {
      if (norm == 1) PO_LAUNCH4(1, true);
      else PO_LAUNCH4(2, true);
      return check_launch("knn_points_backward");
    }
#undef PO_LAUNCH4
    const int tiles = (int)ceil_div(P1, kBwdBlock);
    const dim3 gridr((unsigned)(N * tiles)), blockr(kBwdBlock);
    const int64_t groups = ceil_div(N * S, kXcds);
    POINTOPS_REQUIRE(groups * kXcds * parts < (1LL << 31), "knn_points_backward: grid too large");
    const dim3 gridt((unsigned)(groups * kXcds * parts)), blockt(kTiledBlock);
    unsigned magic;
    int shift;
    division_magic((unsigned)K, &magic, &shift);
#define PO_TILED(DT, NORM)                                                                                        \
  do {                                                                                                            \
    hipLaunchKernelGGL((knn_backward_rows_kernel<DT, NORM>), gridr, blockr, 0, stream, p1, p2, lengths1,          \
                       lengths2, idxs, grad_dists, (int)P1, (int)P2, (int)K, tiles, grad_p1);                     \
    hipLaunchKernelGGL((knn_backward_tiled_kernel<DT, NORM>), gridt, blockt, 0, stream, p1, p2, lengths1,         \
                       lengths2, idxs, grad_dists, (int)N, (int)P1, (int)P2, (int)K, magic, shift, parts, S,      \
                       grad_p2);                                                                                  \
  } while (0)
    if (norm == 1) {
      if (D == 1) PO_TILED(1, 1);
      else if (D == 2) PO_TILED(2, 1);
      else if (D == 3) PO_TILED(3, 1);
      else PO_TILED(4, 1);
    } else {
      if (D == 1) PO_TILED(1, 2);
      else if (D == 2) PO_TILED(2, 2);
      else if (D == 3) PO_TILED(3, 2);
      else PO_TILED(4, 2);
    }
#undef PO_TILED
    return check_launch("knn_points_backward(tiled)");
  }
  const int tiles = (int)ceil_div(P1, kBwdBlock);
  POINTOPS_REQUIRE(N * tiles < (1LL << 31), "knn_points_backward: grid too large");
  const dim3 grid((unsigned)(N * tiles)), block(kBwdBlock);
#define PO_LAUNCH(DT, NORM)                                                                       \
  hipLaunchKernelGGL((knn_backward_kernel<DT, NORM>), grid, block, 0, stream, p1, p2, lengths1,   \
                     lengths2, idxs, grad_dists, (int)P1, (int)P2, (int)D, (int)K, tiles, grad_p1, \
                     grad_p2)
  if (norm == 1) {
    if (D == 3) PO_LAUNCH(3, 1);
    else if (D == 2) PO_LAUNCH(2, 1);
    else PO_LAUNCH(0, 1);
  } else {
    if (D == 3) PO_LAUNCH(3, 2);
    else if (D == 2) PO_LAUNCH(2, 2);
    else PO_LAUNCH(0, 2);
  }
#undef PO_LAUNCH
  return check_launch("knn_points_backward");
}
