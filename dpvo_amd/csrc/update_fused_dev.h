// update_fused_dev.h -- device building blocks shared by the row-tile-resident update kernels (update_fused.hip, update_fused_k7.hip).
// See update_fused.hip for the design notes (geometry, P order, register image, weight fragment image).
//
// Every block is a template over NT = the number of 32-feature output tiles ONE WAVE owns: a workgroup is NW = 12 / NT waves.
//   NT = 3: 4 waves, one per SIMD, 96 features each (rounds 2-5; the chain kernels at two workgroups per CU);
//   NT = 1: 12 waves, three per SIMD, 32 features each (round 6: K1 and K7 -- a third of the accumulators / state / weight ring per wave
//           fits the 168 registers a wave has at three per SIMD, so another wave's MFMAs run while one converts, normalises or waits).
// Feature tile ft = NT w + t of wave w is the same object in both geometries: same packed weight fragments (the image is
// [ft / 3][k-step][ft % 3][lane][8 halves]: a fragment is 1 KB whichever wave loads it), same P order position, same slot of the f32
// register image, same MFMA chain -- so the geometries are bit-identical wherever no cross-wave reduction is involved, and the
// LayerNorm statistics are reduced per feature tile in a fixed order so that they are, too.
#pragma once
#include "common.h"
#include <atomic>

#ifdef FU_TRACE
// per-workgroup timeline (100 MHz wall clock) for tools/fu_trace.py: [kernel id 8][block 1024][wave 4][stamp 16]
static __device__ unsigned long long* g_fu_trace = nullptr;      // (one per translation unit: the setters below fill each)
#define FU_T(k, i)                                                                                                       \
  do {                                                                                                                   \
    if (g_fu_trace && (threadIdx.x & 63) == 0 && threadIdx.x < 256 && blockIdx.x < 1024)                                 \
      g_fu_trace[(((size_t)(k) * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64();              \
  } while (0)
#else
#define FU_T(k, i) do {} while (0)
#endif

// ---- shared between the translation units of the update operator (update_fused.hip, update_fused_k7.hip): external linkage, hidden
namespace dpvo_fu {
struct Lin { const void* w; const _Float16* b; };     // packed image + f16 bias
struct P7 {
  Lin h;
  Lin gate[2], res0[2], res2[2];
  const float *ln_g[2], *ln_b[2];
  const _Float16 *d_w, *d_b, *w_w, *w_b;            // heads: [2,384], [2] f16, feature order
  const _Float16* y; const int32_t* rows;           // y[pu]
  const float* img;
  const float* coords; int pp;                      // optional: target = coords[..., P/2, P/2] + delta
  float *net_out, *delta, *weight, *target;
  int64_t E;
  int skew;                                          // soft start: workgroup b waits (b & 3) * skew / 4 microseconds (0: off)
};
// K7 lives in its own translation unit: it parks 144 state registers in AGPRs and therefore needs its MFMAs in the VGPR form
// (-amdgpu-mfma-vgpr-form is a per-compilation switch; K1 spills under it).  tiles x 256 threads, dynamic LDS set inside.
// waves12 != 0: the 12-wave geometry (NT = 1)
__attribute__((visibility("hidden"))) int launch_k7(int64_t tiles, const P7& p, int waves12, void* stream);
__attribute__((visibility("hidden"))) int k7_set_trace(unsigned long long* buf);      // trace builds only
}  // namespace dpvo_fu

namespace {
namespace fu {
using ::dpvo_fu::Lin;
using ::dpvo_fu::P7;

// Soft start (dpvo_update_fused_params_t.start_skew): the FIRST ROUND of workgroups of a launch (one per CU: blockIdx < 256 on MI355X)
// begins in four groups, skew / 4 microseconds apart, instead of all 256 CUs entering the same phase in the same microsecond -- every
// workgroup runs the same chain of memory and compute phases, so in lock step the HBM is saturated during the gathers / image loads /
// stores and idle during the GEMMs.  The groups interleave inside an XCD (block b runs on XCD b % 8: (b >> 3) & 3 staggers neighbours on
// one L2).  Later rounds start whenever a CU falls free, i.e. already out of step: delaying them would be pure loss.
__device__ __forceinline__ void soft_start(int skew_us) {
  const unsigned g = (blockIdx.x >> 3) & 3;
  if (skew_us > 0 && g && blockIdx.x < 256) {
    const unsigned long long t0 = wall_clock64(), d = (unsigned long long)g * (unsigned)skew_us * 25ull;   // 100 MHz ticks
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
}

constexpr int D = 384;
constexpr int PITCH = 784;             // bytes per LDS activation row (768 + 16: ds_read_b128 / ds_write_b128 conflict free)
constexpr int CPITCH = 272;            // bytes per LDS row of one K chunk of the correlation GEMM (256 + 16)
constexpr int KCH = 128;               // halves per K chunk
constexpr int KS384 = 24;              // k-steps (of 16) of a 384-wide layer

template <int RT, int NT = 3> struct Geo {
  static constexpr int R = 32 * RT;
  static constexpr int NW = 12 / NT;                            // waves per workgroup
  static constexpr int NTHR = 64 * NW;
  static constexpr int ACT_BYTES = R * PITCH;
  static constexpr int RED_BYTES = 2 * R * 12 * 4;              // LayerNorm: 2 x [R][12 feature tiles] floats
  static constexpr int LDS_BYTES = ACT_BYTES + RED_BYTES;
  static_assert(12 % NT == 0, "NT = 1, 2, 3");
  static_assert(2 * R * CPITCH <= ACT_BYTES, "the two K-chunk stages of the correlation GEMM alias the activation tile");
  static_assert(R * 12 * 16 <= ACT_BYTES, "the heads' partial sums alias the activation tile");
};

struct Lane { int tid, lane, w, n, h; };                        // w: wave of the workgroup, 0 .. NW - 1
__device__ __forceinline__ Lane lane_of() {
  Lane l;
  l.tid = threadIdx.x;
  l.lane = l.tid & 63;
  l.w = __builtin_amdgcn_readfirstlane(l.tid >> 6);
  l.n = l.lane & 31;
  l.h = l.lane >> 5;
  return l;
}

// sigmoid in f32 (its result is rounded to f16 by every caller): v_exp + v_rcp, 1 ulp -- an IEEE division costs ten more VALU
// instructions per value, a third of K7's VALU work
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32); }

// ------------------------------------------------------------------------------------------------ weights
// packed image of a [384, K] layer: [ft / 3][k-step K/16][ft % 3][lane 64][8 halves], ft = 32-feature tile 0..11
template <int DW, int NT>
__device__ __forceinline__ void w_preload(h8 (&wf)[DW][NT], const h8* __restrict__ wp) {
#pragma unroll
  for (int d = 0; d < DW; ++d)
#pragma unroll
    for (int t = 0; t < NT; ++t) wf[d][t] = wp[(d * 3 + t) * 64];
}
// this lane's base into the packed image of a layer with KS k-steps: fragment (k-step s, own tile t) is at base[(3 s + t) * 64]
template <int NT>
__device__ __forceinline__ const h8* w_base(const void* img, int KS, const Lane& l) {
  const int ft0 = NT * l.w;
  return reinterpret_cast<const h8*>(img) + ((size_t)(ft0 / 3) * KS * 3 + ft0 % 3) * 64 + l.lane;
}

// accumulators start at the bias (f16 [384], feature order): feature 96 w + 32 t + 8 j + 4 h + q.  The loads are issued
// EARLY (next to the weight preload, before the epilogue of the previous layer): with one wave per SIMD nothing else hides
// a dependent L2 round trip (~1 us under load) in front of the first MFMA.
template <int NT> struct Bias { h4 v[NT][4]; };
template <int NT>
__device__ __forceinline__ void bias_load(Bias<NT>& b, const _Float16* __restrict__ bias, const Lane& l) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[t][j] = *reinterpret_cast<const h4*>(bias + 32 * (NT * l.w + t) + 8 * j + 4 * l.h);
}
// (several workgroups per CU: no early bias load, the accumulators start straight from memory)
template <int RT, int NT>
__device__ __forceinline__ void acc_init_mem(f16v (&acc)[RT][NT], const _Float16* __restrict__ bias, const Lane& l) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f16v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const h4 b = *reinterpret_cast<const h4*>(bias + 32 * (NT * l.w + t) + 8 * j + 4 * l.h);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[4 * j + q] = (float)b[q];
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r][t] = v;
  }
}
template <int RT, int NT>
__device__ __forceinline__ void acc_init(f16v (&acc)[RT][NT], const Bias<NT>& b) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f16v v;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[4 * j + q] = (float)b.v[t][j][q];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r][t] = v;
  }
}

// ------------------------------------------------------------------------------------------------ GEMM over the LDS tile
// acc[r][t] += W(wave's features, tile t) . act(rows of row tile r)^T over KS k-steps; `bl` = this lane's B-fragment
// address of k-step 0, row tile 0 (tile base + n * pitch + 16 h); the ring holds k-steps 0 .. DW-1 on entry.
template <int RT, int KS, int DW, int PITCH_B, int NT>
__device__ __forceinline__ void gemm_lds(f16v (&acc)[RT][NT], h8 (&wf)[DW][NT], const h8* __restrict__ wp, const char* bl) {
  h8 bf[2][RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl + r * 32 * PITCH_B);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) {
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[(s + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * PITCH_B + (s + 1) * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[s & 1][r], acc[r][t], 0, 0, 0);
    if (s + DW < KS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);          // (keeps the scheduler from hoisting later k-steps' loads: register pressure)
  }
}

// ------------------------------------------------------------------------------------------------ epilogue pieces
// one rounding to f16 (what nn.Linear returns under autocast), kept in f32 registers
template <int RT, int NT>
__device__ __forceinline__ void round_f16(f16v (&v)[RT][NT]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) v[r][t][k] = (float)(_Float16)v[r][t][k];
}

// v -> f16 -> the LDS tile in P order.  ACT: 0 none, 1 relu, 2 sigmoid.  `al` = tile base + n * PITCH + 16 h (this lane's row 0)
template <int RT, int ACT, int NT>
__device__ __forceinline__ void to_lds(const f16v (&v)[RT][NT], char* al, const Lane& l) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          _Float16 x = (_Float16)v[r][t][8 * c + i];
          if (ACT == 1) x = x > (_Float16)0 ? x : (_Float16)0;
          if (ACT == 2) x = (_Float16)sigm((float)x);
          o[i] = x;
        }
        *reinterpret_cast<h8*>(al + r * 32 * PITCH + ((NT * l.w + t) * 2 + c) * 32) = o;
      }
}

// v -> f16 -> global rows in P order (dst = row 0 of the tile, ld halves per row): 16-byte pieces, two lanes per 32 B
template <int RT, int NT>
__device__ __forceinline__ void to_rows(const f16v (&v)[RT][NT], _Float16* dst, int64_t ld, int64_t row0, int64_t E, const Lane& l) {
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int64_t g = row0 + r * 32 + l.n;
    if (g < E) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          h8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[r][t][8 * c + i];
          *reinterpret_cast<h8*>(dst + g * ld + ((NT * l.w + t) * 2 + c) * 16 + 8 * l.h) = o;
        }
    }
  }
}

// LayerNorm over the 384 features of every row of the tile (two-pass, f32), in place.  Two barriers; the first one also
// orders every wave's LDS reads of the preceding GEMM before whatever is written to the tile afterwards.
// Statistics: one partial per (row, 32-feature tile) -- the 16 values of a lane in k order plus the other half-wave's -- stored at
// red[row][ft]; every lane then adds the twelve partials of its row in ONE fixed order, so the 4-wave and the 12-wave geometry (and
// any two runs) agree bit for bit.
__device__ __forceinline__ float sum12(const float* p) {
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4), c = *reinterpret_cast<const f4*>(p + 8);
  return (((a[0] + a[1]) + a[2]) + ((a[3] + b[0]) + b[1])) + (((b[2] + b[3]) + c[0]) + ((c[1] + c[2]) + c[3]));
}
template <int RT, int NT>
__device__ __forceinline__ void ln_stats(const f16v (&v)[RT][NT], float* red, float (&mean)[RT], float (&rstd)[RT], const Lane& l) {
  constexpr int R = 32 * RT;
  float* red1 = red;
  float* red2 = red + R * 12;
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[r][t][k];
      s += xhalf(s);
      if (l.h == 0) red1[(r * 32 + l.n) * 12 + NT * l.w + t] = s;
    }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    mean[r] = sum12(red1 + (r * 32 + l.n) * 12) * (1.0f / D);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) { const float d = v[r][t][k] - mean[r]; q += d * d; }
      q += xhalf(q);
      if (l.h == 0) red2[(r * 32 + l.n) * 12 + NT * l.w + t] = q;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) rstd[r] = rsqrtf(sum12(red2 + (r * 32 + l.n) * 12) * (1.0f / D) + 1e-3f);
}
// affine parameters requested BEFORE the statistics (their round trip hides behind the two reductions): one wave per SIMD
template <int RT, int NT>
__device__ __forceinline__ void layernorm_tile(f16v (&v)[RT][NT], float* red, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, const Lane& l) {
  float mean[RT], rstd[RT];
  f4 gm[NT][4], bt[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 32 * (NT * l.w + t) + 8 * j + 4 * l.h;
      gm[t][j] = *reinterpret_cast<const f4*>(gamma + f);
      bt[t][j] = *reinterpret_cast<const f4*>(beta + f);
    }
  ln_stats<RT, NT>(v, red, mean, rstd, l);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[r][t][4 * j + q] = (v[r][t][4 * j + q] - mean[r]) * rstd[r] * gm[t][j][q] + bt[t][j][q];
}
// same, parameters read AFTER the statistics -- from LDS (gp = [gamma 384 | beta 384] f32 copied once per workgroup, bp = gp + 384: no
// 96-register parameter block while K7's 144-register state is live) or from global memory (several waves per SIMD cover the round trip)
template <int RT, int NT>
__device__ __forceinline__ void layernorm_tile_late(f16v (&v)[RT][NT], float* red, const float* gp, const float* bp, const Lane& l) {
  float mean[RT], rstd[RT];
  ln_stats<RT, NT>(v, red, mean, rstd, l);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 32 * (NT * l.w + t) + 8 * j + 4 * l.h;
      const f4 g = *reinterpret_cast<const f4*>(gp + f), b = *reinterpret_cast<const f4*>(bp + f);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[r][t][4 * j + q] = (v[r][t][4 * j + q] - mean[r]) * rstd[r] * g[q] + b[q];
    }
}
template <int RT, int NT>
__device__ __forceinline__ void layernorm_tile_lds(f16v (&v)[RT][NT], float* red, const float* gb, const Lane& l) {
  layernorm_tile_late<RT, NT>(v, red, gb, gb + D, l);
}

// register image of the f32 hidden state: [32-row tile][feature tile 12][j 4] x 1 KB (independent of RT and NT: kernels may tile differently)
constexpr int IMG_RT_STRIDE = 12 * 4 * 256;          // floats per 32-row tile
template <int RT, int NT>
__device__ __forceinline__ float* img_ptr(float* img, int64_t tile, const Lane& l) {
  return img + (size_t)(tile * RT) * IMG_RT_STRIDE + (NT * l.w) * (4 * 256) + l.lane * 4;
}
// the image is requested one GEMM ahead of the epilogue that adds it (144 registers at RT = 3: this is what the one wave
// per SIMD configuration has them for) ...
template <int RT, int NT> struct Img { f4 v[RT][NT][4]; };
template <int RT, int NT>
__device__ __forceinline__ void img_load(Img<RT, NT>& m, const float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) m.v[r][t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
}
template <int RT, int NT>
__device__ __forceinline__ void img_add(f16v (&v)[RT][NT], const Img<RT, NT>& m) {        // ... v += image
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[r][t][4 * j + q] += m.v[r][t][j][q];
}
template <int RT, int NT>
__device__ __forceinline__ void img_store(const f16v (&v)[RT][NT], float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = v[r][t][4 * j + q];
        *reinterpret_cast<f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256) = x;
      }
}

// two workgroups per CU: no registers for an image in flight under a GEMM (and the other workgroup covers the latency):
// v += image, image = v, one 32-row tile at a time
template <int RT, int NT>
__device__ __forceinline__ void img_add_store_stream(f16v (&v)[RT][NT], float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    Img<1, NT> m;
    img_load<1, NT>(m, ip + r * IMG_RT_STRIDE);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f4 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[r][t][4 * j + q] += m.v[0][t][j][q]; x[q] = v[r][t][4 * j + q]; }
        *reinterpret_cast<f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256) = x;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows of a P-order f16 matrix [., 384] -> the LDS tile.  rows == nullptr: row0 + i; an index < 0 or a row >= E: zeros.
// The row indices are read ONCE per row (one coalesced load, through `sidx`: 32 RT ints of LDS that are free at this point)
// instead of once per 16-byte piece: 18 dependent 4-byte loads per thread in front of the data loads were a third of the
// gather's 9-11 us.  Contains one barrier.
template <int RT, int NTHR = 256>
__device__ __forceinline__ void gather_rows(char* act, const _Float16* __restrict__ src, const int32_t* __restrict__ rows,
                                            int64_t row0, int64_t E, int tid, int32_t* sidx) {
  constexpr int N = RT * 6 * 256 / NTHR;         // 32 RT rows x 48 pieces of 16 B over NTHR threads
  if (tid < 32 * RT) {
    const int64_t g = row0 + tid;
    sidx[tid] = g < E ? (rows ? rows[g] : (int32_t)g) : -1;
  }
  __syncthreads();
  h8 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + NTHR * i, row = idx / 48, ch = idx - 48 * row;
    const int32_t sr = sidx[row];
    v[i] = sr >= 0 ? *reinterpret_cast<const h8*>(src + (int64_t)sr * D + ch * 8) : (h8)(_Float16)0;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + NTHR * i, row = idx / 48, ch = idx - 48 * row;
    *reinterpret_cast<h8*>(act + row * PITCH + ch * 16) = v[i];
  }
}
// the LDS tile -> rows [row0, row0 + R) of a P-order f16 matrix [E, 384]
template <int RT, int NTHR = 256>
__device__ __forceinline__ void scatter_rows(const char* act, _Float16* __restrict__ dst, int64_t row0, int64_t E, int tid) {
  constexpr int N = RT * 6 * 256 / NTHR;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + NTHR * i, row = idx / 48, ch = idx - 48 * row;
    const h8 v = *reinterpret_cast<const h8*>(act + row * PITCH + ch * 16);
    if (row0 + row < E) *reinterpret_cast<h8*>(dst + (row0 + row) * D + ch * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------ parameter blocks
// The kernel is a non-type template parameter, so every kernel instantiation owns its flag word: bit d = the dynamic-LDS
// attribute has been set on device d (relaxed atomics: setting it twice is harmless, it only must not be skipped).
template <auto KERN, typename P, int NTHR = 256>
int launch(int64_t tiles, int lds, const P& p, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DPVO_E_INVALID;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
    if (hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPVO_E_UNSUPPORTED;
    attr_done.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(KERN, dim3((unsigned)tiles), dim3(NTHR), (size_t)lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace fu
}  // namespace
