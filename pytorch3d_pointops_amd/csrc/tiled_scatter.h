// tiled_scatter.h -- scatter-add of a per-cloud (rows x K) neighbour table into the rows it
// points at, WITHOUT device-scope atomics.  Shared by knn_points_backward (grad_p2) and
// gather_neighbors_backward (grad_x).
//
// The memory-side fp32 atomic rate (~55 G/s measured on MI355X) bounds the plain scatter kernels
// at ~1.8 ms for 32 x 65536 x 16 neighbours x 3 channels, 30x the time their bytes need.  Here a
// 1024-thread workgroup owns one TILE of a cloud's target rows as fp32 accumulators in LDS
// (<= 96 KB), streams the cloud's WHOLE idx table (coalesced 8-byte loads, two steps ahead),
// compacts the entries that point into its tile through a per-wave LDS stage (ballot + mbcnt) so
// that the fetch/accumulate body runs on full waves, adds with ds_add_f32, and finally stores its
// tile with plain coalesced writes: every target element is written exactly once, so there is no
// memset either.  The idx table is read once per tile (M / TILE times); the workgroups that share
// a table are placed on the SAME XCD (blockIdx % 8 = XCD under round-robin dispatch) and start
// together, so the re-reads are L2 hits and HBM sees the table about once.  With few clouds the
// rows are split over S workgroups per tile, whose partial tiles meet with atomics (S * M * C of
// them instead of rows * K * C).
//
// Measured (knn backward, 32 x 65536 x 16, C = 3): this kernel 0.61 ms + 0.23 ms for grad_p1
// against 1.85 ms for the device-atomic kernel.  The LDS pipeline is busy for the whole kernel
// (SQ_ACTIVE_INST_LDS = kernel duration, SQ_WAIT_INST_LDS 37 % of wave cycles): ds_add_f32 retires
// ~1 lane every 3 cycles, i.e. ~200 G scatter-adds/s for the chip against ~55 G/s at the L2, so
// deeper prefetch of the table or of the operands does not move it (tried: no change).
//
// A SOURCE functor supplies the values (C = Src::kChannels <= 4, compile time):
//   struct Src { static constexpr int kChannels; struct Regs;
//     __device__ int rows(int n) const;            // valid table rows of cloud n
//     __device__ int kmax(int n) const;            // valid entries per row of cloud n
//     __device__ void issue(int n, int e, int i, int k, int j, Regs&) const;  // start the loads
//     __device__ float value(const Regs&, int c) const; };                    // addend of channel c
#pragma once
#include "common.h"
#include "debug.h"

#include <algorithm>
#include <cstdlib>

namespace pointops {

constexpr int kTiledBlock = 1024;
constexpr int kTiledAccFloats = 24576;  // 96 KB
constexpr int kTiledUnroll = 4;         // 64-entry groups scanned per step
constexpr int kTiledBatch = 128;        // staged entries accumulated per drain (two per lane)
constexpr int kTiledStage = kTiledBatch + 64 * kTiledUnroll;  // < one batch carried over + one scan step
constexpr int kXcds = 8;

__host__ __device__ inline int tiled_tile_rows(int C) { return (kTiledAccFloats / C) & ~63; }

// e / K for 0 <= e < 2^31 as umulhi(e, magic) >> shift: with l = ceil(log2 K) and
// magic = ceil(2^(31+l) / K) < 2^32 the quotient is exact for every 31-bit dividend
// (round-up method); K = 1 is flagged by shift = -1.
struct DivMagic {
  unsigned magic;
  int shift;
};
static inline DivMagic division_magic(unsigned K) {
  int l = 0;
  while ((1ull << l) < K) ++l;
  if (l == 0) return DivMagic{0u, -1};
  return DivMagic{(unsigned)(((1ull << (31 + l)) + K - 1) / K), l - 1};
}

struct TiledPlan {
  bool tiled;
  int parts, S;
  dim3 grid;
};

// tiles when the table is large and the target splits into few tiles; the debug knobs `env_mode` = a(tomic) |
// t(iled) and `env_split` override the choice (A/B measurements and tests; debug.h)
static inline TiledPlan tiled_plan(int64_t N, int64_t rows, int64_t K, int64_t M, int C, const char* env_mode,
                                   const char* env_split) {
  TiledPlan p{false, 0, 1, dim3(1)};
  if (C < 1 || C > 4 || N <= 0 || rows <= 0 || M <= 0 || K <= 0) return p;
  p.parts = (int)ceil_div(M, tiled_tile_rows(C));
  const bool fits = rows * K < (1LL << 31) - (1 << 20);
  p.tiled = fits && p.parts <= 16 && N * rows * K >= (1 << 21);
  const char mode = debug_knob_c(env_mode);
  if (mode == 'a') p.tiled = false;
  if (mode == 't') p.tiled = fits && p.parts <= 64;
  if (!p.tiled) return p;
  // few clouds: split the rows so that ~256 workgroups exist; partial tiles meet with atomics
  const int64_t wgs = N * p.parts;
  if (wgs < 192) p.S = (int)std::min<int64_t>(ceil_div(256, wgs), std::max<int64_t>(1, rows / 4096));
  p.S = std::max<int>(1, (int)debug_knob(env_split, p.S));
  const int64_t groups = ceil_div(N * p.S, kXcds);
  if (groups * kXcds * p.parts >= (1LL << 31)) {
    p.tiled = false;
    p.S = 1;
    return p;
  }
  p.grid = dim3((unsigned)(groups * kXcds * p.parts));
  return p;
}

template <class Src>
__global__ __launch_bounds__(kTiledBlock) void tiled_scatter_kernel(
    const Src src, const int64_t* __restrict__ idxs, int N, int R, int M, int K, DivMagic dm, int parts, int S,
    float* __restrict__ target) {
  constexpr int C = Src::kChannels;
  __shared__ float s_acc[kTiledAccFloats];
  __shared__ int2 s_stage[kTiledBlock / 64][kTiledStage];  // (entry, row in tile)

  // XCD-aware placement: the `parts` workgroups of one (cloud, row split) share an XCD
  const int x = blockIdx.x % kXcds, y = blockIdx.x / kXcds;
  const int cs = (y / parts) * kXcds + x, part = y % parts;
  if (cs >= N * S) return;
  const int n = cs / S, split = cs - n * S;
  const int tile = tiled_tile_rows(C);
  const int j0 = part * tile;
  const int jn = min(tile, M - j0);  // rows of this tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int t = tid; t < jn * C; t += kTiledBlock) s_acc[t] = 0.0f;
  __syncthreads();

  const int len = min(R, src.rows(n));
  const int kmax = min(K, src.kmax(n));
  // rows [r0, r1) of this split; entries are the flat (row, k) table
  const int rows_per = (len + S - 1) / S;
  const int r0 = min(len, split * rows_per), r1 = min(len, r0 + rows_per);
  const int64_t* __restrict__ itab = idxs + (int64_t)n * R * K;
  const int e0 = r0 * K, e1 = r1 * K;  // R * K < 2^31 - 2^20 checked by the host
  int2* __restrict__ stage = s_stage[wave];
  int staged = 0;  // wave-uniform

  // A drain pops one batch of staged entries.  It is split in two so that the loads of a batch
  // (two entries per lane) fly while the wave goes on scanning: drain_issue() starts them,
  // drain_finish() -- one scan step later -- turns them into LDS atomics.
  constexpr int kPer = kTiledBatch / 64;
  typename Src::Regs regs[kPer];
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
      const int i = dm.shift < 0 ? e : (int)(__umulhi((unsigned)e, dm.magic) >> dm.shift);  // e / K
      const int k = e - i * K;
      don[q] = don[q] && k < kmax;
      src.issue(n, e, i, k, j0 + djl[q], regs[q]);
    }
    pending = true;
  };
  auto drain_finish = [&]() {
    if (!pending) return;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      if (don[q]) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float v = src.value(regs[q], c);
          if (v != 0.0f) atomicAdd(&s_acc[djl[q] * C + c], v);  // +-0 never changes a sum started at +0
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
      const uint64_t rel = (uint64_t)(j[u] - (int64_t)j0);  // -1 padding and bad rows fall outside
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

  float* __restrict__ out = target + ((int64_t)n * M + j0) * C;
  if (S == 1) {
    for (int t = tid; t < jn * C; t += kTiledBlock) out[t] = s_acc[t];
  } else {
    for (int t = tid; t < jn * C; t += kTiledBlock) {
      const float v = s_acc[t];
      if (v != 0.0f) atomicAdd(out + t, v);
    }
  }
}

}  // namespace pointops
