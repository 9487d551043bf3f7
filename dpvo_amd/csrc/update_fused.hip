// update_fused.hip -- the update operator (reference dpvo/net.py:74-92, dpvo/blocks.py:15-48, under autocast
// dpvo/dpvo.py:332) as SEVEN launches of row-tile-resident MFMA kernels for gfx950.
//
// Why: launch by launch (update.hip) every one of the 21 Linear layers streams an E x 384 activation through memory,
// which makes 0.26 TFLOP of GEMM work memory bound.  Here a workgroup owns a tile of 32 RT edge rows for a whole CHAIN of
// layers; only what another tile must see (the rows gathered by the neighbour MLPs and by SoftAgg) and one f32 image
// of the hidden state per chain cross a kernel boundary:
//
//   K1  corr[E,896] -> Linear ReLU Linear LN ReLU Linear, + net + inp, LayerNorm            (net.py:77-78)
//   K2  net[ix] -> c1 (Linear ReLU Linear) -> net +=                                        (net.py:80-82)
//   K3  net[jx] -> c2 -> net += ; f | g = agg_kk.f(net), agg_kk.g(net)                       (net.py:83-85,87)
//   SA  segmented softmax-sum of f by g over the kk groups (softagg_kernel of update.hip)   (blocks.py:41-43)
//   K5  y[ku] -> agg_kk.h -> net += ; f | g of agg_ij                                        (net.py:87-88)
//   SA  ... over the (ii, jj) groups
//   K7  y[pu] -> agg_ij.h -> net += ; 2 x (LayerNorm, x + gate(x) * res(x)); heads           (net.py:88-92)
//
// (h is applied to the gathered group row of every edge instead of once per group: same bits per row, 2 x 14 GFLOP of
// extra MFMA work, two launches and two dependent small GEMMs less.)
//
// Geometry of every kernel: 256 threads = 4 waves, wave w owns output features [96 w, 96 w + 96) of ALL rows of the tile.
// The product is formed transposed, D[feature][row] = W . act^T, with v_mfma_f32_32x32x16_f16:
//   A operand = a 32-feature x 16-k fragment of W, fetched straight from global memory / L2 into VGPRs.  The host-side
//     pack kernel stores W fragment by fragment in lane order, so a fragment is ONE contiguous 1 KB load per wave, and a
//     wave streams exactly its own quarter of W per layer: no LDS traffic and no synchronisation on the weight stream;
//   B operand = 16 k x 32 rows of the activation tile, ds_read_b128 from LDS (row pitch 784 B: conflict free);
//   D: lane (n = lane & 31, h = lane >> 5) holds, for row n of a row tile, features 8 (k / 4) + 4 h + (k % 4), k = 0..15,
//     of a 32-feature tile.  Everything row-wise (bias, activation, residual, gate, LayerNorm statistics, the heads'
//     dot products) is therefore lane-local; LayerNorm and the heads need one exchange with lane ^ 32 and one 4-way
//     cross-wave sum through LDS.
// Chaining without a transpose: the 16 values a lane holds after a 32-feature tile are exactly two B-operand fragments
// (8 halves each) of the NEXT layer if that layer's k index is permuted: position p = ((3 w + t) 2 + c) 16 + 8 h + i
// holds feature 96 w + 32 t + 16 c + 8 (i / 4) + 4 h + (i % 4) ("P order").  The pack kernel permutes the K dimension
// of every chained layer's weights accordingly (a dot product does not care about the order of its terms), all
// f16 activation rows inside the operator (LDS tile, gathered rows, f | g, y) are kept in P order, and only the
// external tensors (corr, net, inp, the outputs) are in feature order.
// The f32 hidden state between kernels is stored as a REGISTER IMAGE ([tile][wave][row tile][feature tile][j][lane] x 16 B):
// the lane that wrote a value is the lane that reads it in the next kernel, every access is a fully coalesced 1 KB.
//
// Occupancy: ONE workgroup per CU, one wave per SIMD with the whole 512-entry register file, because what limits a wave
// is how far ahead of the MFMAs its weight fragments are requested: an L2 hit is ~2000 clocks away under load, a k-step
// of 3 RT MFMAs lasts 96 RT clocks, so the ring of fragments in flight must be >= 6 k-steps deep (72 registers) -- with two
// workgroups per CU (256 registers each, ring of 3) the same kernels ran at 15 % of the MFMA peak.
//
// Precision contract: unchanged from update.hip: Linear = f16 operands, f32 accumulate, one rounding to
// f16; LayerNorm -> f32; residual adds in f32; gate * res is a half product.
#include "common.h"
#include <atomic>

#ifdef FU_TRACE
// per-workgroup timeline (100 MHz wall clock) for tools/fu_trace.py: [kernel id 8][block 1024][wave 4][stamp 16]
__device__ unsigned long long* g_fu_trace = nullptr;
#define FU_T(k, i)                                                                                                       \
  do {                                                                                                                   \
    if (g_fu_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                      \
      g_fu_trace[(((size_t)(k) * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64();              \
  } while (0)
#else
#define FU_T(k, i) do {} while (0)
#endif

namespace {
namespace fu {

// Soft start (dpvo_update_fused_start_skew): the workgroups of a launch begin in four groups, skew / 4 microseconds apart,
// instead of all 256 CUs entering the same phase in the same microsecond.  A candidate of the autotune only: it costs a few
// microseconds per kernel on a normal box (measured: 0 / +10 / +60 us at 4 / 10 / 20 us) and exists for the boxes on which the
// FIRST, synchronous round of workgroups of every such kernel runs 2x slower than the second, staggered one.
__device__ __forceinline__ void soft_start(int skew_us) {
  if (skew_us > 0 && (blockIdx.x & 3)) {
    const unsigned long long t0 = wall_clock64(), d = (unsigned long long)(blockIdx.x & 3) * (unsigned)skew_us * 25ull;   // 100 MHz ticks
    while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
  }
}

constexpr int D = 384;
constexpr int PITCH = 784;             // bytes per LDS activation row (768 + 16: ds_read_b128 / ds_write_b128 conflict free)
constexpr int CPITCH = 272;            // bytes per LDS row of one K chunk of the correlation GEMM (256 + 16)
constexpr int KCH = 128;               // halves per K chunk
constexpr int KS384 = 24;              // k-steps (of 16) of a 384-wide layer

template <int RT> struct Geo {
  static constexpr int R = 32 * RT;
  static constexpr int ACT_BYTES = R * PITCH;
  static constexpr int RED_BYTES = R * 64;                      // LN: 2 x [R][4] floats; heads: [R][4 waves][4] floats
  static constexpr int LDS_BYTES = ACT_BYTES + RED_BYTES;
  static_assert(2 * R * CPITCH <= ACT_BYTES, "the two K-chunk stages of the correlation GEMM alias the activation tile");
};

struct Lane { int tid, lane, w, n, h; };
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
// packed image of a [384, K] layer: [wave 4][k-step K/16][tile 3][lane 64][8 halves]
template <int DW>
__device__ __forceinline__ void w_preload(h8 (&wf)[DW][3], const h8* __restrict__ wp) {
#pragma unroll
  for (int d = 0; d < DW; ++d)
#pragma unroll
    for (int t = 0; t < 3; ++t) wf[d][t] = wp[(d * 3 + t) * 64];
}
// this lane's base into the packed image of a layer with KS k-steps
__device__ __forceinline__ const h8* w_base(const void* img, int KS, const Lane& l) {
  return reinterpret_cast<const h8*>(img) + (size_t)l.w * KS * 3 * 64 + l.lane;
}

// accumulators start at the bias (f16 [384], feature order): feature 96 w + 32 t + 8 j + 4 h + q.  The loads are issued
// EARLY (next to the weight preload, before the epilogue of the previous layer): with one wave per SIMD nothing else hides
// a dependent L2 round trip (~1 us under load) in front of the first MFMA.
struct Bias { h4 v[3][4]; };
__device__ __forceinline__ void bias_load(Bias& b, const _Float16* __restrict__ bias, const Lane& l) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[t][j] = *reinterpret_cast<const h4*>(bias + 96 * l.w + 32 * t + 8 * j + 4 * l.h);
}
// (several workgroups per CU: no early bias load, the accumulators start straight from memory)
template <int RT>
__device__ __forceinline__ void acc_init_mem(f16v (&acc)[RT][3], const _Float16* __restrict__ bias, const Lane& l) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    f16v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const h4 b = *reinterpret_cast<const h4*>(bias + 96 * l.w + 32 * t + 8 * j + 4 * l.h);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[4 * j + q] = (float)b[q];
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r][t] = v;
  }
}
template <int RT>
__device__ __forceinline__ void acc_init(f16v (&acc)[RT][3], const Bias& b) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
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
template <int RT, int KS, int DW, int PITCH_B>
__device__ __forceinline__ void gemm_lds(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* __restrict__ wp, const char* bl) {
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
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[s & 1][r], acc[r][t], 0, 0, 0);
    if (s + DW < KS) {
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);          // (keeps the scheduler from hoisting later k-steps' loads: register pressure)
  }
}

// ------------------------------------------------------------------------------------------------ epilogue pieces
// one rounding to f16 (what nn.Linear returns under autocast), kept in f32 registers
template <int RT>
__device__ __forceinline__ void round_f16(f16v (&v)[RT][3]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) v[r][t][k] = (float)(_Float16)v[r][t][k];
}

// v -> f16 -> the LDS tile in P order.  ACT: 0 none, 1 relu, 2 sigmoid.  `al` = tile base + n * PITCH + 16 h (this lane's row 0)
template <int RT, int ACT>
__device__ __forceinline__ void to_lds(const f16v (&v)[RT][3], char* al, const Lane& l) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
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
        *reinterpret_cast<h8*>(al + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32) = o;
      }
}

// v -> f16 -> global rows in P order (dst = row 0 of the tile, ld halves per row): 16-byte pieces, two lanes per 32 B
template <int RT>
__device__ __forceinline__ void to_rows(const f16v (&v)[RT][3], _Float16* dst, int64_t ld, int64_t row0, int64_t E, const Lane& l) {
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int64_t g = row0 + r * 32 + l.n;
    if (g < E) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          h8 o;
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[r][t][8 * c + i];
          *reinterpret_cast<h8*>(dst + g * ld + ((3 * l.w + t) * 2 + c) * 16 + 8 * l.h) = o;
        }
    }
  }
}

// LayerNorm over the 384 features of every row of the tile (two-pass, f32), in place.  Two barriers; the first one also
// orders every wave's LDS reads of the preceding GEMM before whatever is written to the tile afterwards.
template <int RT>
__device__ __forceinline__ void layernorm_tile(f16v (&v)[RT][3], float* red, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, const Lane& l) {
  constexpr int R = 32 * RT;
  float* red1 = red;
  float* red2 = red + R * 4;
  float mean[RT], rstd[RT];
  // the affine parameters of this lane's 48 features, requested before the statistics (their round trip hides behind the
  // two reductions)
  f4 gm[3][4], bt[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
      gm[t][j] = *reinterpret_cast<const f4*>(gamma + f);
      bt[t][j] = *reinterpret_cast<const f4*>(beta + f);
    }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[r][t][k];
    s += xhalf(s);
    if (l.h == 0) red1[(r * 32 + l.n) * 4 + l.w] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const f4 p = *reinterpret_cast<const f4*>(red1 + (r * 32 + l.n) * 4);
    mean[r] = ((p[0] + p[1]) + (p[2] + p[3])) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) { const float d = v[r][t][k] - mean[r]; q += d * d; }
    q += xhalf(q);
    if (l.h == 0) red2[(r * 32 + l.n) * 4 + l.w] = q;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const f4 p = *reinterpret_cast<const f4*>(red2 + (r * 32 + l.n) * 4);
    rstd[r] = rsqrtf(((p[0] + p[1]) + (p[2] + p[3])) * (1.0f / D) + 1e-3f);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[r][t][4 * j + q] = (v[r][t][4 * j + q] - mean[r]) * rstd[r] * gm[t][j][q] + bt[t][j][q];
}

// same, affine parameters in LDS (gb: [gamma 384 | beta 384] f32, feature order; copied once per persistent workgroup):
// no 96-register parameter block while the 144-register state is live
template <int RT>
__device__ __forceinline__ void layernorm_tile_late(f16v (&v)[RT][3], float* red, const float* gp, const float* bp, const Lane& l);
template <int RT>
__device__ __forceinline__ void layernorm_tile_lds(f16v (&v)[RT][3], float* red, const float* gb, const Lane& l) {
  layernorm_tile_late<RT>(v, red, gb, gb + D, l);
}
// (also with the parameters in global memory when several workgroups share a CU and cover each other's round trips)
template <int RT>
__device__ __forceinline__ void layernorm_tile_late(f16v (&v)[RT][3], float* red, const float* gp, const float* bp, const Lane& l) {
  constexpr int R = 32 * RT;
  float* red1 = red;
  float* red2 = red + R * 4;
  float mean[RT], rstd[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) s += v[r][t][k];
    s += xhalf(s);
    if (l.h == 0) red1[(r * 32 + l.n) * 4 + l.w] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const f4 p = *reinterpret_cast<const f4*>(red1 + (r * 32 + l.n) * 4);
    mean[r] = ((p[0] + p[1]) + (p[2] + p[3])) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) { const float d = v[r][t][k] - mean[r]; q += d * d; }
    q += xhalf(q);
    if (l.h == 0) red2[(r * 32 + l.n) * 4 + l.w] = q;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const f4 p = *reinterpret_cast<const f4*>(red2 + (r * 32 + l.n) * 4);
    rstd[r] = rsqrtf(((p[0] + p[1]) + (p[2] + p[3])) * (1.0f / D) + 1e-3f);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
      const f4 g = *reinterpret_cast<const f4*>(gp + f), b = *reinterpret_cast<const f4*>(bp + f);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[r][t][4 * j + q] = (v[r][t][4 * j + q] - mean[r]) * rstd[r] * g[q] + b[q];
    }
}

// register image of the f32 hidden state: [32-row tile][wave][t][j] x 1 KB (independent of RT: kernels may tile differently)
constexpr int IMG_RT_STRIDE = 4 * 3 * 4 * 256;       // floats per 32-row tile
template <int RT>
__device__ __forceinline__ float* img_ptr(float* img, int64_t tile, const Lane& l) {
  return img + (size_t)(tile * RT) * IMG_RT_STRIDE + l.w * (3 * 4 * 256) + l.lane * 4;
}
// the image is requested one GEMM ahead of the epilogue that adds it (144 registers at RT = 3: this is what the one wave
// per SIMD configuration has them for) ...
template <int RT> struct Img { f4 v[RT][3][4]; };
template <int RT>
__device__ __forceinline__ void img_load(Img<RT>& m, const float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) m.v[r][t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
}
template <int RT>
__device__ __forceinline__ void img_add(f16v (&v)[RT][3], const Img<RT>& m) {        // ... v += image
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[r][t][4 * j + q] += m.v[r][t][j][q];
}
template <int RT>
__device__ __forceinline__ void img_store(const f16v (&v)[RT][3], float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
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
template <int RT>
__device__ __forceinline__ void img_add_store_stream(f16v (&v)[RT][3], float* ip) {
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    Img<1> m;
    img_load<1>(m, ip + r * IMG_RT_STRIDE);
#pragma unroll
    for (int t = 0; t < 3; ++t)
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
template <int RT>
__device__ __forceinline__ void gather_rows(char* act, const _Float16* __restrict__ src, const int32_t* __restrict__ rows,
                                            int64_t row0, int64_t E, int tid, int32_t* sidx) {
  constexpr int N = RT * 6;                      // 32 RT rows x 48 pieces of 16 B over 256 threads
  if (tid < 32 * RT) {
    const int64_t g = row0 + tid;
    sidx[tid] = g < E ? (rows ? rows[g] : (int32_t)g) : -1;
  }
  __syncthreads();
  h8 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    const int32_t sr = sidx[row];
    v[i] = sr >= 0 ? *reinterpret_cast<const h8*>(src + (int64_t)sr * D + ch * 8) : (h8)(_Float16)0;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    *reinterpret_cast<h8*>(act + row * PITCH + ch * 16) = v[i];
  }
}
// the LDS tile -> rows [row0, row0 + R) of a P-order f16 matrix [E, 384]
template <int RT>
__device__ __forceinline__ void scatter_rows(const char* act, _Float16* __restrict__ dst, int64_t row0, int64_t E, int tid) {
  constexpr int N = RT * 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    const h8 v = *reinterpret_cast<const h8*>(act + row * PITCH + ch * 16);
    if (row0 + row < E) *reinterpret_cast<h8*>(dst + (row0 + row) * D + ch * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------ parameter blocks
struct Lin { const void* w; const _Float16* b; };     // packed image + f16 bias

struct P1 {
  Lin c0, c2, c5;
  const float *cln_g, *cln_b, *norm_g, *norm_b;
  const _Float16* corr; int64_t ld_corr;
  const float* net;                                   // [E,384] f32, feature order
  const int64_t* net_rows; int64_t n_kept;            // optional: row g of the state is net[net_rows[g]] for g < n_kept, zero after
  const _Float16* inp; const int64_t* inp_rows; int64_t inp_mod;
  float* img; _Float16* rows16;                       // out
  int64_t E;
  int skew;                                          // soft start: workgroup b waits (b & 3) * skew / 4 microseconds (0: off)
};

// K1 -----------------------------------------------------------------------------------------------
template <int RT, int DW, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void k1_corr_norm(const P1 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = Geo<RT>::R;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  float* red = reinterpret_cast<float*>(smem + Geo<RT>::ACT_BYTES);
  char* al = act + l.n * PITCH + 16 * l.h;

  soft_start(p.skew);
  FU_T(0, 0);
  f16v acc[RT][3];
  h8 wf[DW][3];
  Bias bias;
  // ---- Linear(882 -> 384) + ReLU over K = 896 streamed from global memory in 7 chunks of 128, two LDS stages
  {
    const h8* wp = w_base(p.c0.w, 56, l);
    w_preload<DW>(wf, wp);
    if constexpr (OCC == 1) bias_load(bias, p.c0.b, l);
    constexpr int NS = RT * 2;                         // 16-byte pieces per thread per chunk
    // chunk kc + 2 is requested (into one of two register sets) while chunk kc is multiplied and chunk kc + 1 waits in the
    // other set for its LDS stage: a chunk lasts ~1 us of MFMAs, a miss to HBM under load about twice that.
    h8 st[2][NS];
    auto load = [&](h8 (&d)[NS], int kc) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
        int64_t g = row0 + row;
        g = g < p.E ? g : p.E - 1;
        d[i] = *reinterpret_cast<const h8*>(p.corr + g * p.ld_corr + kc * KCH + ch * 8);
      }
    };
    auto store = [&](const h8 (&d)[NS], int b) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
        *reinterpret_cast<h8*>(smem + b * R * CPITCH + row * CPITCH + ch * 16) = d[i];
      }
    };
    load(st[0], 0);
    load(st[1], 1);
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.c0.b, l);
    store(st[0], 0);
    __syncthreads();
    FU_T(0, 1);
#pragma unroll
    for (int kc = 0; kc < 7; ++kc) {
      if (kc + 2 < 7) load(st[kc & 1], kc + 2);
      const char* bl = smem + (kc & 1) * R * CPITCH + l.n * CPITCH + 16 * l.h;
      h8 bf[2][RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int s = kc * 8 + ks;
        if (ks + 1 < 8) {
#pragma unroll
          for (int r = 0; r < RT; ++r) bf[(ks + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH + (ks + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[ks & 1][r], acc[r][t], 0, 0, 0);
        if (s + DW < 56) {
#pragma unroll
          for (int t = 0; t < 3; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kc + 1 < 7) store(st[(kc + 1) & 1], (kc + 1) & 1);
      __syncthreads();
    }
  }
  FU_T(0, 2);
  const h8* wp2 = w_base(p.c2.w, KS384, l);
  w_preload<DW>(wf, wp2);
  if constexpr (OCC == 1) bias_load(bias, p.c2.b, l);
  to_lds<RT, 1>(acc, al, l);                            // (the last barrier of the chunk loop freed the stages)
  __syncthreads();
  // ---- Linear, LayerNorm, ReLU
  if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.c2.b, l);
  gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp2, al);
  FU_T(0, 3);
  const h8* wp3 = w_base(p.c5.w, KS384, l);
  w_preload<DW>(wf, wp3);
  if constexpr (OCC == 1) bias_load(bias, p.c5.b, l);
  round_f16<RT>(acc);
  if constexpr (OCC == 1) layernorm_tile<RT>(acc, red, p.cln_g, p.cln_b, l); else layernorm_tile_late<RT>(acc, red, p.cln_g, p.cln_b, l);
  to_lds<RT, 1>(acc, al, l);
  __syncthreads();
  FU_T(0, 4);
  // ---- Linear; net = LayerNorm(net + inp + .).
  if constexpr (OCC == 1) {
    // one workgroup per CU: the rows of net (feature order) are requested before the GEMM
    Img<RT> nv;
    int64_t ir[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t g = row0 + r * 32 + l.n;
      g = g < p.E ? g : p.E - 1;
      ir[r] = g;
      if (p.inp_rows) { ir[r] = p.inp_rows[g]; if (p.inp_mod > 0) ir[r] %= p.inp_mod; }
      const int64_t gs = !p.net_rows ? g : (g < p.n_kept ? p.net_rows[g] : -1);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          nv.v[r][t][j] = gs >= 0 ? *reinterpret_cast<const f4*>(p.net + gs * D + 96 * l.w + 32 * t + 8 * j + 4 * l.h) : (f4)0.f;
    }
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp3, al);
    FU_T(0, 5);
    h4 iv[RT][3][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) iv[r][t][j] = *reinterpret_cast<const h4*>(p.inp + ir[r] * D + 96 * l.w + 32 * t + 8 * j + 4 * l.h);
    round_f16<RT>(acc);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[r][t][4 * j + q] = (nv.v[r][t][j][q] + (float)iv[r][t][j][q]) + acc[r][t][4 * j + q];
  } else {
    // several workgroups per CU: no registers for rows in flight under the GEMM; one 32-row tile at a time afterwards
    acc_init_mem<RT>(acc, p.c5.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp3, al);
    FU_T(0, 5);
    round_f16<RT>(acc);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t g = row0 + r * 32 + l.n;
      g = g < p.E ? g : p.E - 1;
      int64_t ir = g;
      if (p.inp_rows) { ir = p.inp_rows[g]; if (p.inp_mod > 0) ir %= p.inp_mod; }
      const int64_t gs = !p.net_rows ? g : (g < p.n_kept ? p.net_rows[g] : -1);
      f4 nv[3][4]; h4 iv[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nv[t][j] = gs >= 0 ? *reinterpret_cast<const f4*>(p.net + gs * D + 96 * l.w + 32 * t + 8 * j + 4 * l.h) : (f4)0.f;
          iv[t][j] = *reinterpret_cast<const h4*>(p.inp + ir * D + 96 * l.w + 32 * t + 8 * j + 4 * l.h);
        }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[r][t][4 * j + q] = (nv[t][j][q] + (float)iv[t][j][q]) + acc[r][t][4 * j + q];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (OCC == 1) layernorm_tile<RT>(acc, red, p.norm_g, p.norm_b, l); else layernorm_tile_late<RT>(acc, red, p.norm_g, p.norm_b, l);
  FU_T(0, 6);
  img_store<RT>(acc, img_ptr<RT>(p.img, tile, l));
  to_lds<RT, 0>(acc, al, l);
  __syncthreads();
  scatter_rows<RT>(act, p.rows16, row0, p.E, l.tid);
  FU_T(0, 7);
}

// K2 / K3 / K5 ---------------------------------------------------------------------------------------
struct P2 {
  Lin a, b;                     // MLP: Linear a, ReLU, Linear b (MODE_H: only b)
  Lin f, g;                     // SoftAgg f, g (MODE_C1: unused)
  const _Float16* src; const int32_t* rows;         // gathered input rows (P order)
  float* img;                                       // in / out
  _Float16* rows16;                                 // MODE_C1: out
  _Float16* fg;                                     // else: out [E, 768]
  int64_t E;
  int skew;                                          // soft start: workgroup b waits (b & 3) * skew / 4 microseconds (0: off)
};
enum { MODE_C1 = 0, MODE_C2 = 1, MODE_H = 2 };

template <int RT, int DW, int MODE, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void k_chain(const P2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = Geo<RT>::R;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  char* al = act + l.n * PITCH + 16 * l.h;

  soft_start(p.skew);
  FU_T(1 + MODE, 0);
  f16v acc[RT][3];
  h8 wf[DW][3];
  Bias bias;
  Img<RT> im;
  float* ip = img_ptr<RT>(p.img, tile, l);
  const h8* wpa = w_base(MODE == MODE_H ? p.b.w : p.a.w, KS384, l);
  w_preload<DW>(wf, wpa);
  if constexpr (OCC == 1) bias_load(bias, MODE == MODE_H ? p.b.b : p.a.b, l);
  gather_rows<RT>(act, p.src, p.rows, row0, p.E, l.tid, reinterpret_cast<int32_t*>(smem + Geo<RT>::ACT_BYTES));
  if constexpr (MODE == MODE_H && OCC == 1) img_load<RT>(im, ip);
  FU_T(1 + MODE, 1);
  __syncthreads();
  FU_T(1 + MODE, 2);
  if constexpr (MODE != MODE_H) {
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.a.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpa, al);
    const h8* wpb = w_base(p.b.w, KS384, l);
    FU_T(1 + MODE, 3);
    w_preload<DW>(wf, wpb);
    if constexpr (OCC == 1) bias_load(bias, p.b.b, l);
    if constexpr (OCC == 1) img_load<RT>(im, ip);   // lands under the second GEMM
    __syncthreads();
    to_lds<RT, 1>(acc, al, l);
    __syncthreads();
    FU_T(1 + MODE, 4);
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.b.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpb, al);
  } else {
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.b.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpa, al);
  }
  FU_T(1 + MODE, 5);
  const h8* wpf = nullptr;
  if constexpr (MODE != MODE_C1) { wpf = w_base(p.f.w, KS384, l); w_preload<DW>(wf, wpf); if constexpr (OCC == 1) bias_load(bias, p.f.b, l); }
  round_f16<RT>(acc);
  if constexpr (OCC == 1) {
    img_add<RT>(acc, im);
    img_store<RT>(acc, ip);
  } else {
    img_add_store_stream<RT>(acc, ip);
  }
  FU_T(1 + MODE, 6);
  __syncthreads();
  to_lds<RT, 0>(acc, al, l);
  __syncthreads();
  FU_T(1 + MODE, 7);
  if constexpr (MODE == MODE_C1) {
    scatter_rows<RT>(act, p.rows16, row0, p.E, l.tid);
    FU_T(1 + MODE, 8);
  } else {
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.f.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpf, al);
    FU_T(1 + MODE, 8);
    const h8* wpg = w_base(p.g.w, KS384, l);
    w_preload<DW>(wf, wpg);
    if constexpr (OCC == 1) bias_load(bias, p.g.b, l);
    to_rows<RT>(acc, p.fg, 768, row0, p.E, l);
    FU_T(1 + MODE, 9);
    if constexpr (OCC == 1) acc_init<RT>(acc, bias); else acc_init_mem<RT>(acc, p.g.b, l);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpg, al);
    FU_T(1 + MODE, 10);
    to_rows<RT>(acc, p.fg + D, 768, row0, p.E, l);
    FU_T(1 + MODE, 11);
  }
}

// K7 -----------------------------------------------------------------------------------------------
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

template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void k7_gru_heads(const P7 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = Geo<RT>::R;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  float* red = reinterpret_cast<float*>(smem + 2 * Geo<RT>::ACT_BYTES);
  char* al = act + l.n * PITCH + 16 * l.h;
  char* gl = al + Geo<RT>::ACT_BYTES;               // second tile: where every lane parks its own gate values (no barriers)

  soft_start(p.skew);
  FU_T(4, 0);
  f16v x[RT][3];
  h8 wf[DW][3];
  Bias bias;
  const h8* wp = w_base(p.h.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.h.b, l);
  float* lnp = reinterpret_cast<float*>(smem + 2 * Geo<RT>::ACT_BYTES + Geo<RT>::RED_BYTES);      // [gamma | beta] x 2, f32
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.ln_g[0][i]; lnp[D + i] = p.ln_b[0][i]; lnp[2 * D + i] = p.ln_g[1][i]; lnp[3 * D + i] = p.ln_b[1][i];
  }
  gather_rows<RT>(act, p.y, p.rows, row0, p.E, l.tid, reinterpret_cast<int32_t*>(red));
  float* ip = img_ptr<RT>(const_cast<float*>(p.img), tile, l);
  {
    Img<RT> im;
    img_load<RT>(im, ip);       // lands under the first GEMM
    __syncthreads();
    FU_T(4, 1);
    acc_init<RT>(x, bias);
    gemm_lds<RT, KS384, DW, PITCH>(x, wf, wp, al);
    FU_T(4, 2);
    round_f16<RT>(x);
    img_add<RT>(x, im);
  }
  // From here on the f32 state is in registers only between the last GEMM of a gated residual and the next LayerNorm: it is
  // written back to its (lane-private) image slot after each LayerNorm and re-read, one row tile at a time, for the residual add.
  // Holding it across the three GEMMs (144 + 144 accumulator registers > the 256-entry accumulation file) made the compiler spill
  // ~120 registers, and every scratch reload is an exposed memory round trip with one wave per SIMD.
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    layernorm_tile_lds<RT>(x, red, lnp + 2 * D * G, l);
    FU_T(4, 3 + 5 * G);
    wp = w_base(p.gate[G].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gate[G].b, l);
    img_store<RT>(x, ip);
    to_lds<RT, 0>(x, al, l);
    __syncthreads();
    FU_T(4, 4 + 5 * G);
    f16v acc[RT][3];
    // gate = sigmoid(Linear(x))
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp, al);
    FU_T(4, 5 + 5 * G);
    wp = w_base(p.res0[G].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.res0[G].b, l);
    to_lds<RT, 2>(acc, gl, l);
    // res = Linear(relu(Linear(x)))
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp, al);
    FU_T(4, 6 + 5 * G);
    wp = w_base(p.res2[G].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.res2[G].b, l);
    __syncthreads();
    to_lds<RT, 1>(acc, al, l);
    __syncthreads();
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp, al);
    FU_T(4, 7 + 5 * G);
    // x = x(image) + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 m[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * c + i;
            const _Float16 rv = (_Float16)acc[r][t][k];
            x[r][t][k] = m[t][k >> 2][k & 3] + (float)(_Float16)(gt[i] * rv);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  FU_T(4, 13);
  // ---- hidden state out (feature order) and the heads
  // d and w (two Linear(384, 2) on relu(net), net.py:92) as ONE MFMA chain per wave over its own 96 features: the B fragment of
  // k-step (3w + t) * 2 + c is exactly what to_lds would write for this lane -- relu(x[r][t][8c .. 8c + 7]) in f16 -- so it is
  // built in registers; the A fragment carries the four head rows (rows 4..31 zero) in the same P order.  18 MFMAs instead of
  // ~1 000 VALU instructions (relu, two conversions and four FMAs per value); f16 operands, f32 accumulate as before.
  f16v hacc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) hacc[r][k] = 0.f;
  {
    const int m = l.n;                              // A row = output row of the 32-row MFMA tile: d0, d1, w0, w1, then zeros
    const _Float16* wrow = m == 0 ? p.d_w : m == 1 ? p.d_w + D : m == 2 ? p.w_w : p.w_w + D;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int base = 96 * l.w + 32 * t + 16 * c + 4 * l.h;      // features base + {0..3} and base + 8 + {0..3}
        h8 af = (h8)(_Float16)0;
        if (m < 4) {
          const h4 lo = *reinterpret_cast<const h4*>(wrow + base), hi = *reinterpret_cast<const h4*>(wrow + base + 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) { af[i] = lo[i]; af[4 + i] = hi[i]; }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          h8 bfr;
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float v = x[r][t][8 * c + i]; bfr[i] = (_Float16)(v > 0.f ? v : 0.f); }
          hacc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfr, hacc[r], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int64_t g = row0 + r * 32 + l.n;
        f4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = x[r][t][4 * j + q];
        if (g < p.E) *reinterpret_cast<f4*>(p.net_out + g * D + f) = o4;
      }
    }
  __syncthreads();                                  // (the LayerNorm partials in `red` are dead)
  // D[row m][col n]: lane (n, h) holds rows (j & 3) + 8 (j >> 2) + 4 h in register j -> the four head sums of tile row n sit
  // in registers 0..3 of the lanes with h == 0, both K halves already added
#pragma unroll
  for (int r = 0; r < RT; ++r)
    if (l.h == 0) *reinterpret_cast<f4*>(red + ((r * 32 + l.n) * 4 + l.w) * 4) = (f4){hacc[r][0], hacc[r][1], hacc[r][2], hacc[r][3]};
  __syncthreads();
  if (l.tid < R) {
    const int64_t g = row0 + l.tid;
    if (g < p.E) {
      f4 s = *reinterpret_cast<const f4*>(red + (l.tid * 4 + 0) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f4 q = *reinterpret_cast<const f4*>(red + (l.tid * 4 + w) * 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) s[o] += q[o];
      }
      const float d0 = (float)(_Float16)(s[0] + (float)p.d_b[0]), d1 = (float)(_Float16)(s[1] + (float)p.d_b[1]);
      const _Float16 h0 = (_Float16)(s[2] + (float)p.w_b[0]), h1 = (_Float16)(s[3] + (float)p.w_b[1]);
      p.delta[2 * g + 0] = d0;
      p.delta[2 * g + 1] = d1;
      p.weight[2 * g + 0] = (float)(_Float16)sigm((float)h0);
      p.weight[2 * g + 1] = (float)(_Float16)sigm((float)h1);
      if (p.target) {
        p.target[2 * g + 0] = p.coords[(g * 2 + 0) * p.pp + p.pp / 2] + d0;
        p.target[2 * g + 1] = p.coords[(g * 2 + 1) * p.pp + p.pp / 2] + d1;
      }
    }
  }
  FU_T(4, 14);
}


// ================================================================================================================
// Patch-major path: FOUR launches.
//
// With the edges taken in the plan's per-patch order (perm_k: sorted by patch, then target frame) two of the three
// exchanges of the operator stop crossing tiles when a tile holds whole patches:
//   * the neighbour rows of c1 / c2 (fastba.neighbors, ba.cpp:59-97: previous / next edge of the same patch in jj order) are
//     the rows q - 1 / q + 1 of the sorted order: the GEMM just reads the LDS tile one row up or down (a zero row at a patch
//     boundary), no gather at all;
//   * agg_kk (SoftAgg over the edges of a patch, net.py:87) is a segmented softmax over contiguous rows of the tile.
// Only agg_ij (groups by frame pair) still needs every tile.  So:
//     pm_prepare   tiles of <= 96 rows made of whole patches (greedy, 16 waves), inverse of perm_k
//     KA           corr MLP + norm, c1, c2, agg_kk (f, g, softmax-sum, h), f | g of agg_ij       -- f32 state in registers
//     SA           agg_ij softmax-sum over the frame-pair groups (rows addressed through the inverse permutation)
//     KB           agg_ij.h, 2 x (LayerNorm, gated residual), heads
// HBM traffic per edge: corr 1792 + net 1536 in, image 1536 + f | g 1536 out (KA); 1536 in (SA); image 1536 in, net 1536 out
// (KB) = ~11 KB instead of ~26 KB for the seven-launch path, and the workgroups (persistent, tiles handed out by an atomic
// counter) drift apart after their first tile, so the memory phases of one CU overlap the MFMA phases of the others.
// ================================================================================================================
struct Tile { int32_t row0, nrows, p0, np; };
enum { HDR_NTILES = 0, HDR_CTR_A = 1, HDR_CTR_B = 2, HDR_ERR = 3, HDR_INTS = 8 };
constexpr int PM_MAX_TILES = 2048;

// block 0: greedy packing (wave w packs the patches [w C, (w + 1) C)); every block: invk[perm_k[q]] = q
__global__ __launch_bounds__(1024) void pm_prepare_kernel(const int32_t* __restrict__ perm_k, const int32_t* __restrict__ patch_off,
                                                          const int32_t* __restrict__ counts, int64_t E, int R,
                                                          int32_t* __restrict__ invk, Tile* __restrict__ tiles,
                                                          int32_t* __restrict__ hdr, int max_tiles) {
  for (int64_t q = blockIdx.x * 1024ll + threadIdx.x; q < E; q += gridDim.x * 1024ll) invk[perm_k[q]] = (int32_t)q;
  if (blockIdx.x != 0) return;
  __shared__ Tile s_tiles[16][128];
  __shared__ int s_cnt[16], s_err;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int np = counts[0];
  const int C = (np + 15) / 16;
  const int pbeg = w * C, pend = (pbeg + C < np) ? pbeg + C : np;
  if (threadIdx.x == 0) s_err = 0;
  __syncthreads();
  int cnt = 0, err = 0;
  for (int p = pbeg; p < pend;) {
    const int row0 = patch_off[p];
    const int pl = p + lane;
    const int endl = (pl < pend) ? patch_off[pl + 1] : 0x7fffffff;
    const bool ok = (pl < pend) && (endl - row0 <= R) && (lane < 32);
    const unsigned long long m = __ballot(ok);
    int n = (~m == 0ull) ? 64 : __builtin_ctzll(~m);            // leading run of patches that fit (sizes are monotone)
    if (n < 1) { n = 1; err = 1; }                                // a patch with more than R edges: not representable here
    int nrows = patch_off[p + n] - row0;
    if (nrows > R) nrows = R;
    if (lane == 0) {
      if (cnt < 128) s_tiles[w][cnt] = Tile{row0, nrows, p, n};
      else err = 1;
    }
    ++cnt;
    p += n;
  }
  if (cnt > 128) cnt = 128;
  if (lane == 0) { s_cnt[w] = cnt; if (err) s_err = 1; }
  __syncthreads();
  int base = 0, total = 0;
  for (int i = 0; i < 16; ++i) { if (i < w) base += s_cnt[i]; total += s_cnt[i]; }
  for (int i = lane; i < cnt; i += 64) tiles[base + i] = s_tiles[w][i];
  // (more tiles than the launch grids of KA / KB cover: only possible if the caller's bound on the edges of a patch is wrong)
  if (threadIdx.x == 0) { hdr[HDR_NTILES] = total < max_tiles ? total : max_tiles; hdr[HDR_ERR] = (s_err || total > max_tiles) ? 1 : 0; }
}

// SoftAgg over groups whose member rows are addressed through an inverse permutation (row = invk[perm[p]])
__global__ __launch_bounds__(384) void softagg_inv_kernel(const _Float16* __restrict__ fg, int64_t ldfg,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ invk,
                                                          const int32_t* __restrict__ off, const int32_t* __restrict__ n_groups,
                                                          _Float16* __restrict__ y) {
  __shared__ float part[4][3][384];
  const int ng = *n_groups;
  const int q = threadIdx.x / 96, cq = threadIdx.x - 96 * q;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int b = off[g], e = off[g + 1];
    float m[4], s[4], a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; s[r] = 0.f; a[r] = 0.f; }
    for (int p = b + q; p < e; p += 4) {
      const _Float16* rowp = fg + (int64_t)invk[perm[p]] * ldfg + 4 * cq;
      const h4 fx = *reinterpret_cast<const h4*>(rowp), gx = *reinterpret_cast<const h4*>(rowp + D);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g_ = (float)gx[r];
        const float mn = fmaxf(m[r], g_);
        const float sc = __expf(m[r] - mn), wv = __expf(g_ - mn);
        s[r] = s[r] * sc + wv;
        a[r] = a[r] * sc + wv * (float)fx[r];
        m[r] = mn;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[q][0][4 * cq + r] = m[r]; part[q][1][4 * cq + r] = s[r]; part[q][2][4 * cq + r] = a[r]; }
    __syncthreads();
    {
      const int c = threadIdx.x;
      float M = part[0][0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k) M = fmaxf(M, part[k][0][c]);
      float S = 0.f, A = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mk = part[k][0][c];
        const float sc = (mk == -INFINITY) ? 0.f : __expf(mk - M);
        S += part[k][1][c] * sc;
        A += part[k][2][c] * sc;
      }
      y[(int64_t)g * D + c] = (_Float16)(A / S);
    }
    __syncthreads();
  }
}

// k-loop of a 384-wide layer as a ROLLED loop (DW k-steps per trip, DW even and a divisor of 24): ~1 KB of code per call
// site instead of 4 KB -- KA chains twelve GEMMs and must stay friendly to the 64 KB instruction cache.  Per-row-tile
// B-fragment addresses (bl[r]) so that a lane can read a shifted row or the zero row.  The prefetch index is clamped
// (the last trips re-request k-step 23, harmless).
template <int RT, int DW>
__device__ __forceinline__ void gemm384(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* __restrict__ wp, const char* const (&bl)[RT]) {
  static_assert(24 % DW == 0 && (DW & 1) == 0, "ring");
  h8 bf[2][RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl[r]);
#pragma unroll 1
  for (int s0 = 0; s0 < 24; s0 += DW) {
#pragma unroll
    for (int d = 0; d < DW; ++d) {
      const int s = s0 + d;
      const int sn = s + 1 < 24 ? s + 1 : 23;
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[(d + 1) & 1][r] = *reinterpret_cast<const h8*>(bl[r] + sn * 32);
      __builtin_amdgcn_sched_barrier(0);        // (else the scheduler sinks these reads below the MFMAs to save 4 RT registers,
                                                //  and every k-step then waits a full LDS round trip)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r)
          acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[d][t], bf[d & 1][r], acc[r][t], 0, 0, 0);
      const int sw = s + DW < 24 ? s + DW : 23;
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[d][t] = wp[(sw * 3 + t) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// v = f32(f16(v)) and x += v
template <int RT>
__device__ __forceinline__ void residual_add(f16v (&x)[RT][3], const f16v (&v)[RT][3]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) x[r][t][k] += (float)(_Float16)v[r][t][k];
}

// the LDS tile -> rows [row0, row0 + nrows) of a P-order f16 matrix with leading dimension ld (halves)
template <int RT>
__device__ __forceinline__ void tile_to_rows(const char* act, _Float16* __restrict__ dst, int64_t ld, int64_t row0, int nrows, int tid) {
  constexpr int N = RT * 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    const h8 v = *reinterpret_cast<const h8*>(act + row * PITCH + ch * 16);
    if (row < nrows) *reinterpret_cast<h8*>(dst + (row0 + row) * ld + ch * 8) = v;
  }
}

struct PA {
  Lin c0, c2, c5, c1a, c1b, c2a, c2b, fk, gk, hk, fi, gi;
  const float *cln_g, *cln_b, *norm_g, *norm_b;
  const _Float16* corr; int64_t ld_corr;
  const float* net;
  const _Float16* inp; const int64_t* inp_rows; int64_t inp_mod;
  const int32_t *perm_k, *ix, *jx, *ku;
  const Tile* tiles; int32_t* hdr;
  float* img; _Float16* fg;
  int64_t E;
};

template <int RT> struct GeoA {
  static constexpr int R = 32 * RT;
  static constexpr int T1 = 0, T2 = R * PITCH, RED = 2 * R * PITCH, ZERO = RED + R * 32, META = ZERO + PITCH;
  static constexpr int LNP = META + R * 8 + 16;                  // two LayerNorms: 2 x [gamma | beta] f32
  static constexpr int LDS_BYTES = LNP + 2 * 2 * D * 4;
  static_assert(LDS_BYTES <= 163840, "LDS");
};

template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void ka_kernel(const PA p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA<RT>;
  constexpr int R = G::R;
  const Lane l = lane_of();
  char* t1 = smem + G::T1;
  char* t2 = smem + G::T2;
  float* red = reinterpret_cast<float*>(smem + G::RED);
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);          // edge id of row i (-1: no row)
  int32_t* meta_f = meta_e + R;                                            // patch index in the tile | first << 8 | last << 9
  int32_t* slot = meta_f + R;                                              // the tile this workgroup drew
  if ((int)blockIdx.x >= p.hdr[HDR_NTILES]) return;
  for (int i = l.tid; i < PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(smem + G::ZERO)[i] = 0u;
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.cln_g[i]; lnp[D + i] = p.cln_b[i]; lnp[2 * D + i] = p.norm_g[i]; lnp[3 * D + i] = p.norm_b[i];
  }
  // one tile per workgroup, the grid is the host's upper bound on the number of tiles (surplus workgroups leave at once): the
  // hardware dispatcher hands out tiles as CUs become free.  (A persistent loop around this body made the compiler hoist
  // ~400 loop-invariant addresses out of it and spill them: 4x slower.)
  (void)slot;
  const int n_tiles = p.hdr[HDR_NTILES];
  {
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    FU_T(5, 0);
    const Tile T = p.tiles[tile];
    const int64_t row0 = T.row0;
    if (l.tid < R) {
      const int i = l.tid;
      int e = -1, f = (1 << 8) | (1 << 9);
      if (i < T.nrows) {
        e = p.perm_k[row0 + i];
        f = (p.ku[e] - T.p0) | ((p.ix[e] < 0) << 8) | ((p.jx[e] < 0) << 9);
      }
      meta_e[i] = e;
      meta_f[i] = f;
    }
    __syncthreads();

    f16v acc[RT][3], x[RT][3];
    h8 wf[DW][3];
    Bias bias;
    const char* al = t1 + l.n * PITCH + 16 * l.h;          // this lane's row of row tile 0 (write side / unshifted read side)
    const char* bl0[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) bl0[r] = al + r * 32 * PITCH;
    const char* bl2[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) bl2[r] = bl0[r] + G::T2;
    int eg[RT];                                              // edge id of this lane's rows (clamped to a valid one for loads)
#pragma unroll
    for (int r = 0; r < RT; ++r) { const int e = meta_e[r * 32 + l.n]; eg[r] = e < 0 ? 0 : e; }

    // ---- Linear(882 -> 384) + ReLU: K = 896 streamed from the rows corr[e] in 7 chunks of 128 through two LDS stages
    {
      const h8* wp = w_base(p.c0.w, 56, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.c0.b, l);
      constexpr int NS = RT * 2;
      const _Float16* src[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
        const int e = meta_e[row];
        src[i] = p.corr + (int64_t)(e < 0 ? 0 : e) * p.ld_corr + ch * 8;
      }
      h8 st[2][NS];
      auto load = [&](h8 (&d)[NS], int kc) {
#pragma unroll
        for (int i = 0; i < NS; ++i) d[i] = *reinterpret_cast<const h8*>(src[i] + kc * KCH);
      };
      auto store = [&](const h8 (&d)[NS], int b) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
          *reinterpret_cast<h8*>(smem + b * R * CPITCH + row * CPITCH + ch * 16) = d[i];
        }
      };
      load(st[0], 0);
      load(st[1], 1);
      acc_init<RT>(acc, bias);
      store(st[0], 0);
      __syncthreads();
#pragma unroll
      for (int kc = 0; kc < 7; ++kc) {
        if (kc + 2 < 7) load(st[kc & 1], kc + 2);
        const char* bl = smem + (kc & 1) * R * CPITCH + l.n * CPITCH + 16 * l.h;
        h8 bf[2][RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int s = kc * 8 + ks;
          if (ks + 1 < 8) {
#pragma unroll
            for (int r = 0; r < RT; ++r) bf[(ks + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH + (ks + 1) * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < RT; ++r)
              acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[ks & 1][r], acc[r][t], 0, 0, 0);
          if (s + DW < 56) {
#pragma unroll
            for (int t = 0; t < 3; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kc + 1 < 7) store(st[(kc + 1) & 1], (kc + 1) & 1);
        __syncthreads();
      }
    }
    FU_T(5, 1);
    const h8* wp = w_base(p.c2.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.c2.b, l);
    to_lds<RT, 1>(acc, const_cast<char*>(al), l);
    __syncthreads();
    // ---- Linear, LayerNorm, ReLU
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.c5.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.c5.b, l);
    round_f16<RT>(acc);
    layernorm_tile_lds<RT>(acc, red, lnp, l);
    to_lds<RT, 1>(acc, const_cast<char*>(al), l);
    __syncthreads();
    FU_T(5, 2);
    // ---- Linear; x = LayerNorm(net + inp + .)      (net.py:77-78)
    {
      acc_init<RT>(x, bias);
      gemm384<RT, DW>(x, wf, wp, bl0);
      wp = w_base(p.c1a.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.c1a.b, l);
      round_f16<RT>(x);
      // rows net[e] (f32) and inp[kk[e] % mod] (f16) in feature order, one row tile at a time (72 registers in flight)
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        int64_t ir = eg[r];
        if (p.inp_rows) { ir = p.inp_rows[eg[r]]; if (p.inp_mod > 0) ir %= p.inp_mod; }
        f4 nv[3][4];
        h4 iv[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
            nv[t][j] = *reinterpret_cast<const f4*>(p.net + (int64_t)eg[r] * D + f);
            iv[t][j] = *reinterpret_cast<const h4*>(p.inp + ir * D + f);
          }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) x[r][t][4 * j + q] = (nv[t][j][q] + (float)iv[t][j][q]) + x[r][t][4 * j + q];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    layernorm_tile_lds<RT>(x, red, lnp + 2 * D, l);
    to_lds<RT, 0>(x, const_cast<char*>(al), l);
    __syncthreads();
    FU_T(5, 3);
    // ---- x += c1(previous edge of the patch); x += c2(next edge of the patch)              (net.py:80-85)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* bls[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int i = r * 32 + l.n;
        const int f = meta_f[i];
        const bool none = k == 0 ? ((f >> 8) & 1) : ((f >> 9) & 1);
        bls[r] = none ? smem + G::ZERO + 16 * l.h : bl0[r] + (k == 0 ? -PITCH : PITCH);
      }
      acc_init<RT>(acc, bias);
      gemm384<RT, DW>(acc, wf, wp, bls);
      wp = w_base(k == 0 ? p.c1b.w : p.c2b.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, k == 0 ? p.c1b.b : p.c2b.b, l);
      to_lds<RT, 1>(acc, const_cast<char*>(al) + G::T2, l);
      __syncthreads();
      acc_init<RT>(acc, bias);
      gemm384<RT, DW>(acc, wf, wp, bl2);
      wp = w_base(k == 0 ? p.c2a.w : p.fk.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, k == 0 ? p.c2a.b : p.fk.b, l);
      residual_add<RT>(x, acc);
      to_lds<RT, 0>(x, const_cast<char*>(al), l);          // (T1 was last read before the barrier above)
      __syncthreads();
    }
    FU_T(5, 4);
    // ---- agg_kk: f -> T2, g -> T1, segmented softmax-sum over the rows of every patch, y -> T1 rows [0, np)   (blocks.py:40-43)
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.gk.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gk.b, l);
    to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.hk.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.hk.b, l);
    __syncthreads();
    to_lds<RT, 0>(acc, const_cast<char*>(al), l);
    __syncthreads();
    FU_T(5, 5);
    if (l.tid < 192) {
      // channels 2 tid, 2 tid + 1 (P order; f and g of a channel sit at the same position of T2 / T1)
      const char* gp = t1 + 4 * l.tid;
      const char* fp = t2 + 4 * l.tid;
      float m0 = -INFINITY, m1 = -INFINITY, s0 = 0.f, s1 = 0.f, a0 = 0.f, a1 = 0.f;
      int cur = 0;
      for (int i = 0; i < T.nrows; ++i) {
        const int pl = meta_f[i] & 0xff;
        if (pl != cur) {
          h2 o; o[0] = (_Float16)(a0 / s0); o[1] = (_Float16)(a1 / s1);
          *reinterpret_cast<h2*>(t1 + cur * PITCH + 4 * l.tid) = o;      // row `cur` of g is already consumed (cur <= first row of the patch)
          cur = pl; m0 = m1 = -INFINITY; s0 = s1 = a0 = a1 = 0.f;
        }
        const h2 gv = *reinterpret_cast<const h2*>(gp + i * PITCH), fv = *reinterpret_cast<const h2*>(fp + i * PITCH);
        {
          const float g_ = (float)gv[0], mn = fmaxf(m0, g_), sc = __expf(m0 - mn), wv = __expf(g_ - mn);
          s0 = s0 * sc + wv; a0 = a0 * sc + wv * (float)fv[0]; m0 = mn;
        }
        {
          const float g_ = (float)gv[1], mn = fmaxf(m1, g_), sc = __expf(m1 - mn), wv = __expf(g_ - mn);
          s1 = s1 * sc + wv; a1 = a1 * sc + wv * (float)fv[1]; m1 = mn;
        }
      }
      h2 o; o[0] = (_Float16)(a0 / s0); o[1] = (_Float16)(a1 / s1);
      *reinterpret_cast<h2*>(t1 + cur * PITCH + 4 * l.tid) = o;
    }
    __syncthreads();
    FU_T(5, 6);
    // ---- h on the (<= 32) patch rows, expanded back to the edges: x += h(y)[patch of the row]       (blocks.py:45-48)
    {
      f16v hy[1][3];
      const char* blh[1] = {al};
      acc_init<1>(hy, bias);
      gemm384<1, DW>(hy, wf, wp, blh);
      wp = w_base(p.fi.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.fi.b, l);
      to_lds<1, 0>(hy, const_cast<char*>(al) + G::T2, l);               // (T2 = f: consumed before the barrier above)
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int pl = meta_f[r * 32 + l.n] & 0xff;
      const char* hp = t2 + pl * PITCH + 16 * l.h;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 v = *reinterpret_cast<const h8*>(hp + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) x[r][t][8 * c + i] += (float)v[i];
        }
    }
    to_lds<RT, 0>(x, const_cast<char*>(al), l);            // (T1 = y: consumed by the h GEMM before the barrier above)
    img_store<RT>(x, img_ptr<RT>(p.img, tile, l));
    __syncthreads();
    FU_T(5, 7);
    // ---- f | g of agg_ij for every edge row, stored at the sorted position
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.gi.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gi.b, l);
    to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);               // (T2 = h(y): consumed before the barrier above)
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    __syncthreads();
    to_lds<RT, 0>(acc, const_cast<char*>(al), l);
    __syncthreads();
    tile_to_rows<RT>(t2, p.fg, 768, row0, T.nrows, l.tid);
    tile_to_rows<RT>(t1, p.fg + D, 768, row0, T.nrows, l.tid);
    FU_T(5, 8);
  }
}

struct PB {
  Lin h;
  Lin gate[2], res0[2], res2[2];
  const float *ln_g[2], *ln_b[2];
  const _Float16 *d_w, *d_b, *w_w, *w_b;
  const _Float16* y; const int32_t *pu, *perm_k;
  const Tile* tiles; int32_t* hdr;
  const float* img;
  const float* coords; int pp;
  float *net_out, *delta, *weight, *target;
  int64_t E;
};

template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void kb_kernel(const PB p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA<RT>;
  constexpr int R = G::R;
  const Lane l = lane_of();
  char* t1 = smem + G::T1;
  float* red = reinterpret_cast<float*>(smem + G::RED);          // (the heads need R x 64 B: RED + ZERO + META are contiguous, R x 32 + 784 + ...)
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);
  int32_t* slot = meta_e + 2 * R;
  const int n_tiles = p.hdr[HDR_NTILES];
  if ((int)blockIdx.x >= n_tiles) return;
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.ln_g[0][i]; lnp[D + i] = p.ln_b[0][i]; lnp[2 * D + i] = p.ln_g[1][i]; lnp[3 * D + i] = p.ln_b[1][i];
  }
  char* al = t1 + l.n * PITCH + 16 * l.h;
  char* gl = al + G::T2;
  const char* bl0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bl0[r] = al + r * 32 * PITCH;

  (void)slot;
  {
    const int tile = blockIdx.x;
    FU_T(6, 0);
#ifdef FU_TRACE
    if (g_fu_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)
      g_fu_trace[(((size_t)6 * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + 12] = clock64();
#endif
    const Tile T = p.tiles[tile];
    const int64_t row0 = T.row0;
    if (l.tid < R) meta_e[l.tid] = l.tid < T.nrows ? p.perm_k[row0 + l.tid] : -1;
    __syncthreads();

    f16v x[RT][3];
    h8 wf[DW][3];
    Bias bias;
    const h8* wp = w_base(p.h.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.h.b, l);
    float* ip = img_ptr<RT>(const_cast<float*>(p.img), tile, l);
    {
      // rows y[pu[e]] of the group table -> T1
      constexpr int N = RT * 6;
      h8 v[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int idx = l.tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
        const int e = meta_e[row];
        v[i] = e >= 0 ? *reinterpret_cast<const h8*>(p.y + (int64_t)p.pu[e] * D + ch * 8) : (h8)(_Float16)0;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int idx = l.tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
        *reinterpret_cast<h8*>(t1 + row * PITCH + ch * 16) = v[i];
      }
    }
    __syncthreads();
    acc_init<RT>(x, bias);
    gemm384<RT, DW>(x, wf, wp, bl0);
    round_f16<RT>(x);
    // x = image + h(y); from here on the f32 state is in registers only BETWEEN the GEMMs of a gated residual: it is written
    // back to its (lane-private) image slot after each LayerNorm and re-read, one row tile at a time, for the residual add.
    // Holding it across the three GEMMs (144 + 144 accumulator registers > the 256-entry accumulation file) made the compiler
    // spill ~125 registers, and a scratch reload costs a full memory round trip with one wave per SIMD.
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 m[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) x[r][t][4 * j + q] += m[t][j][q];
      __builtin_amdgcn_sched_barrier(0);
    }
    FU_T(6, 1);
#pragma unroll
    for (int Gi = 0; Gi < 2; ++Gi) {
      if (Gi == 0) FU_T(7, 0);
      layernorm_tile_lds<RT>(x, red, lnp + 2 * D * Gi, l);
      if (Gi == 0) FU_T(7, 1);
      wp = w_base(p.gate[Gi].w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.gate[Gi].b, l);
      img_store<RT>(x, ip);
      if (Gi == 0) FU_T(7, 2);
      to_lds<RT, 0>(x, al, l);
      __syncthreads();
      if (Gi == 0) FU_T(7, 3);
      {
        f16v acc[RT][3];
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 4);
        wp = w_base(p.res0[Gi].w, KS384, l);
        w_preload<DW>(wf, wp);
        bias_load(bias, p.res0[Gi].b, l);
        to_lds<RT, 2>(acc, gl, l);
        if (Gi == 0) FU_T(7, 5);
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 6);
        wp = w_base(p.res2[Gi].w, KS384, l);
        w_preload<DW>(wf, wp);
        bias_load(bias, p.res2[Gi].b, l);
        __syncthreads();
        to_lds<RT, 1>(acc, al, l);
        __syncthreads();
        if (Gi == 0) FU_T(7, 7);
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 8);
        // x = x(image) + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          f4 m[3][4];
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int k = 8 * c + i;
                const _Float16 rv = (_Float16)acc[r][t][k];
                x[r][t][k] = m[t][k >> 2][k & 3] + (float)(_Float16)(gt[i] * rv);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FU_T(6, 2 + Gi);
    }
    // ---- hidden state out (rows net_out[e], feature order) and the heads
    int eg[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) eg[r] = meta_e[r * 32 + l.n];
    float dsum[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int o = 0; o < 4; ++o) dsum[r][o] = 0.f;
    h4 wv[3][4][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
        wv[t][j][0] = *reinterpret_cast<const h4*>(p.d_w + f);
        wv[t][j][1] = *reinterpret_cast<const h4*>(p.d_w + D + f);
        wv[t][j][2] = *reinterpret_cast<const h4*>(p.w_w + f);
        wv[t][j][3] = *reinterpret_cast<const h4*>(p.w_w + D + f);
      }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          f4 o4;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = x[r][t][4 * j + q];
            o4[q] = v;
            const float a = (float)(_Float16)(v > 0.f ? v : 0.f);
#pragma unroll
            for (int o = 0; o < 4; ++o) dsum[r][o] += a * (float)wv[t][j][o][q];
          }
          if (eg[r] >= 0) *reinterpret_cast<f4*>(p.net_out + (int64_t)eg[r] * D + f) = o4;
        }
      }
    __syncthreads();                                  // (the LayerNorm partials in `red` are dead; T1 reads retired)
    float* hred = reinterpret_cast<float*>(t1);       // [R][4 waves][4]: the tile is free now
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 s;
#pragma unroll
      for (int o = 0; o < 4; ++o) { s[o] = dsum[r][o]; s[o] += xhalf(s[o]); }
      if (l.h == 0) *reinterpret_cast<f4*>(hred + ((r * 32 + l.n) * 4 + l.w) * 4) = s;
    }
    __syncthreads();
    if (l.tid < R) {
      const int e = meta_e[l.tid];
      if (e >= 0) {
        f4 s = *reinterpret_cast<const f4*>(hred + (l.tid * 4 + 0) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const f4 q = *reinterpret_cast<const f4*>(hred + (l.tid * 4 + w) * 4);
#pragma unroll
          for (int o = 0; o < 4; ++o) s[o] += q[o];
        }
        const float d0 = (float)(_Float16)(s[0] + (float)p.d_b[0]), d1 = (float)(_Float16)(s[1] + (float)p.d_b[1]);
        const _Float16 h0 = (_Float16)(s[2] + (float)p.w_b[0]), h1 = (_Float16)(s[3] + (float)p.w_b[1]);
        p.delta[2 * (int64_t)e + 0] = d0;
        p.delta[2 * (int64_t)e + 1] = d1;
        p.weight[2 * (int64_t)e + 0] = (float)(_Float16)sigm((float)h0);
        p.weight[2 * (int64_t)e + 1] = (float)(_Float16)sigm((float)h1);
        if (p.target) {
          p.target[2 * (int64_t)e + 0] = p.coords[((int64_t)e * 2 + 0) * p.pp + p.pp / 2] + d0;
          p.target[2 * (int64_t)e + 1] = p.coords[((int64_t)e * 2 + 1) * p.pp + p.pp / 2] + d1;
        }
      }
    }
    FU_T(6, 4);
#ifdef FU_TRACE
    if (g_fu_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)
      g_fu_trace[(((size_t)6 * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + 13] = clock64();
#endif
  }
}

// ------------------------------------------------------------------------------------------------ weight packing
// W [384, ldw] f16 row-major (torch Linear layout), K columns used (zero beyond k_valid) -> fragment image.
// chained = 0: k-slot (s, h, i) holds input column 16 s + 8 h + i (inputs in feature order: corr);
// chained = 1: P order (inputs produced by another layer of the operator, or gathered P-order rows).
__global__ void pack_kernel(const _Float16* __restrict__ W, int64_t ldw, int K, int k_valid, int chained,
                            _Float16* __restrict__ out) {
  const int KS = K / 16;
  const int64_t total = (int64_t)4 * KS * 3 * 64 * 8;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(o & 7), lane = (int)((o >> 3) & 63);
    int64_t q = o >> 9;
    const int t = (int)(q % 3); q /= 3;
    const int s = (int)(q % KS);
    const int w = (int)(q / KS);
    const int m = lane & 31, h = lane >> 5;
    int k;
    if (chained) {
      const int w2 = s / 6, t2 = (s >> 1) % 3, c = s & 1;
      k = 96 * w2 + 32 * t2 + 16 * c + 8 * (i >> 2) + 4 * h + (i & 3);
    } else {
      k = 16 * s + 8 * h + i;
    }
    const int f = 96 * w + 32 * t + m;
    out[o] = k < k_valid ? W[(int64_t)f * ldw + k] : (_Float16)0;
  }
}

// The kernel is a non-type template parameter, so every kernel instantiation owns its flag word: bit d = the dynamic-LDS
// attribute has been set on device d (relaxed atomics: setting it twice is harmless, it only must not be skipped).
template <auto KERN, typename P>
int launch(int64_t tiles, int lds, const P& p, hipStream_t st) {
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DPVO_E_INVALID;
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_relaxed) & bit)) {
    if (hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPVO_E_UNSUPPORTED;
    attr_done.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(KERN, dim3((unsigned)tiles), dim3(256), (size_t)lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct Ws { size_t img, r16a, r16b, fg, y, total; };
template <int RT>
void ws_layout(int64_t E, int64_t maxg, Ws* w) {
  const size_t tiles = (size_t)cdiv64(E > 0 ? E : 1, 32 * RT), g = (size_t)(maxg > 0 ? maxg : 1), e = (size_t)(E > 0 ? E : 1);
  size_t o = 0;
  w->img = o; o += al256((size_t)cdiv64((int64_t)e, 192) * 192 * D * 4);       // (covers the 64-row and the 96-row tilings)
  w->r16a = o; o += al256(e * D * 2);
  w->r16b = o; o += al256(e * D * 2);
  w->fg = o; o += al256(e * 2 * D * 2);
  w->y = o; o += al256(g * D * 2);
  w->total = o;
}

}  // namespace fu
}  // namespace

#define FU_RT 3
#define FU_DW 6
#define FU_DW7 6
#ifndef FU_DW1
#define FU_DW1 6
#endif
#ifndef FU_DW2
#define FU_DW2 3
#endif
#ifndef FU_OCC2
#define FU_OCC2 2
#endif
#ifndef FU_RT2
#define FU_RT2 2
#endif
#ifndef FU_RT2C
#define FU_RT2C FU_RT2
#endif
#ifndef FU_DW2C
#define FU_DW2C 6            // the chain kernels fit a 6-deep ring in 250 registers at two workgroups per CU (K1 spills beyond 3)
#endif
#define FU_CFG_DEFAULT 1            // chains: 64-row tiles x 2 workgroups per CU; K1 and K7: 96-row tiles x 1 (A/B in the frame, round 3: cfg 1 / 3 / 0 / 2 = 828 / 817 / 814 / 800 frames/sec on one box)
#define FU_DWPM 4
#define FU_DWKB 6

extern "C" size_t dpvo_update_fused_pack_bytes(int K) { return K > 0 && (K % 16) == 0 ? (size_t)384 * K * 2 : 0; }

extern "C" int dpvo_update_fused_pack(const void* W, int64_t ldw, int K, int k_valid, int chained, void* out, void* stream) {
  if (!W || !out || K <= 0 || (K % 16) != 0 || k_valid < 0 || k_valid > K || ldw < k_valid) return DPVO_E_INVALID;
  if (chained && K != 384) return DPVO_E_UNSUPPORTED;
  hipLaunchKernelGGL(fu::pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)W, ldw, K, k_valid,
                     chained ? 1 : 0, (_Float16*)out);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

// Tiling of the seven-launch path (dpvo_update_fused_params_t.tiling).  bit 0: the three chain kernels (c1, c2 + f|g, h + f|g),
// bit 1: the correlation kernel (K1) run 64-row tiles with TWO workgroups per CU (256 registers per wave, nothing requested early:
// the neighbour workgroup covers the round trips) instead of 96-row tiles with one (512 registers, everything prefetched).  K7
// always runs 96 x 1: its phases between the GEMMs are VALU work, which a second workgroup on the same SIMDs does not hide
// (measured: +40 us).  Results are bit-identical across tilings.  The library keeps NO state: the tiling and the soft start
// travel with the parameter block of each call.
extern "C" int dpvo_update_fused_default_tiling(void) { return FU_CFG_DEFAULT; }

extern "C" size_t dpvo_update_fused_workspace_bytes(int64_t E, int64_t max_groups) {
  if (E < 0 || max_groups < 0) return 0;
  fu::Ws w;
  fu::ws_layout<FU_RT>(E, max_groups, &w);
  return w.total;
}

extern "C" int dpvo_update_forward_fused(const dpvo_update_fused_params_t* p, const float* net, const void* inp,
                                         const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr,
                                         const int32_t* plan, int64_t n_patches_ub, int64_t n_pairs_ub, const float* coords,
                                         int P, float* net_out, float* delta, float* weight, float* target, int64_t E,
                                         void* ws, size_t ws_bytes, void* stream) {
  return dpvo_update_forward_fused_rows(p, net, nullptr, 0, inp, inp_rows, inp_mod, corr, ld_corr, plan, n_patches_ub, n_pairs_ub,
                                        coords, P, net_out, delta, weight, target, E, ws, ws_bytes, stream);
}

// The same with the edge compaction of remove_factors (dpvo.py:223-238) folded into the first kernel: the hidden state of edge g
// is net[net_rows[g]] for g < n_kept (the `keep` list of the removal, ascending) and zero for the edges appended after it.
// net_out may alias net: the first kernel has read every row before the last one writes any.
extern "C" int dpvo_update_forward_fused_rows(const dpvo_update_fused_params_t* p, const float* net, const int64_t* net_rows,
                                              int64_t n_kept, const void* inp, const int64_t* inp_rows, int64_t inp_mod,
                                              const void* corr, int64_t ld_corr, const int32_t* plan, int64_t n_patches_ub,
                                              int64_t n_pairs_ub, const float* coords, int P, float* net_out, float* delta,
                                              float* weight, float* target, int64_t E, void* ws, size_t ws_bytes, void* stream) {
  using namespace fu;
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (net_rows && (n_kept < 0 || n_kept > E)) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8) || E >= (1ll << 31)) return DPVO_E_UNSUPPORTED;
  for (int i = 0; i < DPVO_UF_NLIN; ++i)
    if (!p->w[i] || !p->b[i]) return DPVO_E_INVALID;
  constexpr int RT = FU_RT, DW = FU_DW;
  constexpr int RT2 = FU_RT2, DW2 = FU_DW2, OCC2 = FU_OCC2;
  constexpr int RT2C = FU_RT2C, DW2C = FU_DW2C;       // chain kernels' own tile height / ring at several workgroups per CU  // several workgroups per CU, 64-row tiles
  const int cfg = (p->tiling < 0 ? FU_CFG_DEFAULT : p->tiling) & 3;
  const int skew = p->start_skew < 0 ? 0 : (p->start_skew > 1000 ? 1000 : p->start_skew);
  const int64_t maxg = n_patches_ub > n_pairs_ub ? n_patches_ub : n_pairs_ub;
  Ws L;
  ws_layout<RT>(E, maxg, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* wsb = (char*)ws;
  float* img = (float*)(wsb + L.img);
  _Float16 *r16a = (_Float16*)(wsb + L.r16a), *r16b = (_Float16*)(wsb + L.r16b), *fg = (_Float16*)(wsb + L.fg),
           *y = (_Float16*)(wsb + L.y);
  hipStream_t st = (hipStream_t)stream;
  const int64_t tiles = cdiv64(E, 32 * RT), tiles2 = cdiv64(E, 32 * RT2), tiles2c = cdiv64(E, 32 * RT2C);
  auto lin = [&](int i) { return Lin{p->w[i], (const _Float16*)p->b[i]}; };
  int rc;
#define FU(...) do { rc = (__VA_ARGS__); if (rc) return rc; } while (0)
  {
    P1 a{lin(DPVO_UF_C0), lin(DPVO_UF_C2), lin(DPVO_UF_C5), p->ln_g[0], p->ln_b[0], p->ln_g[1], p->ln_b[1],
         (const _Float16*)corr, ld_corr, net, net_rows, n_kept, (const _Float16*)inp, inp_rows, inp_mod, img, r16a, E, skew};
    if (cfg & 2) FU(launch<k1_corr_norm<RT2, DW2, OCC2>>(tiles2, Geo<RT2>::LDS_BYTES, a, st));
    else FU(launch<k1_corr_norm<RT, FU_DW1>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C1_0), lin(DPVO_UF_C1_2), Lin{nullptr, nullptr}, Lin{nullptr, nullptr}, r16a, plan + PL.ix, img, r16b,
         nullptr, E, skew};
    if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_C1, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
    else FU(launch<k_chain<RT, DW, MODE_C1>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C2N_0), lin(DPVO_UF_C2N_2), lin(DPVO_UF_AKK_F), lin(DPVO_UF_AKK_G), r16b, plan + PL.jx, img, nullptr, fg,
         E, skew};
    if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_C2, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
    else FU(launch<k_chain<RT, DW, MODE_C2>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  int64_t ngk = n_patches_ub < 1 ? 1 : (n_patches_ub > E ? E : n_patches_ub);
  int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
  FU(dpvo_softagg(fg, 768, plan + PL.perm_k, plan + PL.patch_off, plan + PL.counts + 0, ngk, y, 384, stream));
  {
    P2 a{Lin{nullptr, nullptr}, lin(DPVO_UF_AKK_H), lin(DPVO_UF_AIJ_F), lin(DPVO_UF_AIJ_G), y, plan + PL.ku, img, nullptr, fg, E, skew};
    if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_H, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
    else FU(launch<k_chain<RT, DW, MODE_H>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  FU(dpvo_softagg(fg, 768, plan + PL.perm_p, plan + PL.pair_off, plan + PL.counts + 1, ngp, y, 384, stream));
  {
    P7 a;
    a.h = lin(DPVO_UF_AIJ_H);
    a.gate[0] = lin(DPVO_UF_G0_GATE); a.res0[0] = lin(DPVO_UF_G0_RES0); a.res2[0] = lin(DPVO_UF_G0_RES2);
    a.gate[1] = lin(DPVO_UF_G1_GATE); a.res0[1] = lin(DPVO_UF_G1_RES0); a.res2[1] = lin(DPVO_UF_G1_RES2);
    a.ln_g[0] = p->ln_g[2]; a.ln_b[0] = p->ln_b[2]; a.ln_g[1] = p->ln_g[3]; a.ln_b[1] = p->ln_b[3];
    a.d_w = (const _Float16*)p->d_w; a.d_b = (const _Float16*)p->d_b; a.w_w = (const _Float16*)p->w_w; a.w_b = (const _Float16*)p->w_b;
    a.y = y; a.rows = plan + PL.pu; a.img = img; a.coords = coords; a.pp = P * P;
    a.net_out = net_out; a.delta = delta; a.weight = weight; a.target = target; a.E = E; a.skew = skew;
    FU(launch<k7_gru_heads<RT, FU_DW7>>(tiles, Geo<RT>::LDS_BYTES + Geo<RT>::ACT_BYTES + 4 * D * 4, a, st));
  }
#undef FU
  return DPVO_OK;
}


// ------------------------------------------------------------------------------------------------ patch-major entry
namespace {
namespace fu {
struct WsPm { size_t img, fg, y, invk, tiles, hdr, total; int64_t max_tiles; };
template <int RT>
void ws_layout_pm(int64_t E, int64_t maxg, WsPm* w) {
  const size_t e = (size_t)(E > 0 ? E : 1), g = (size_t)(maxg > 0 ? maxg : 1);
  int64_t mt = 2 * cdiv64((int64_t)e, 32 * RT) + 48;          // (greedy tiles of whole patches: two consecutive ones hold > R rows, + 16 wave tails)
  if (mt > PM_MAX_TILES) mt = PM_MAX_TILES;
  w->max_tiles = mt;
  size_t o = 0;
  w->img = o; o += al256((size_t)mt * 32 * RT * D * 4);
  w->fg = o; o += al256(e * 2 * D * 2);
  w->y = o; o += al256(g * D * 2);
  w->invk = o; o += al256(e * 4);
  w->tiles = o; o += al256((size_t)PM_MAX_TILES * sizeof(Tile));
  w->hdr = o; o += al256(HDR_INTS * 4);
  w->total = o;
}
}  // namespace fu
}  // namespace

extern "C" size_t dpvo_update_pm_workspace_bytes(int64_t E, int64_t max_groups) {
  if (E < 0 || max_groups < 0) return 0;
  fu::WsPm w;
  fu::ws_layout_pm<FU_RT>(E, max_groups, &w);
  return w.total;
}

extern "C" int dpvo_update_forward_pm(const dpvo_update_fused_params_t* p, const float* net, const void* inp,
                                      const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr,
                                      const int32_t* plan, int64_t n_patches_ub, int64_t n_pairs_ub, int64_t patch_edges_ub,
                                      const float* coords, int P, float* net_out, float* delta, float* weight, float* target,
                                      int64_t E, void* ws, size_t ws_bytes, int32_t* status, void* stream) {
  using namespace fu;
  constexpr int RT = FU_RT, DW = FU_DW;
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8) || E >= (1ll << 31)) return DPVO_E_UNSUPPORTED;
  // tiles hold whole patches: every patch must fit (32 RT rows), and the tile table has PM_MAX_TILES entries
  if (patch_edges_ub <= 0 || patch_edges_ub > 32 * RT) return DPVO_E_UNSUPPORTED;
  if (2 * cdiv64(E, 32 * RT) + 48 > PM_MAX_TILES) return DPVO_E_UNSUPPORTED;
  for (int i = 0; i < DPVO_UF_NLIN; ++i)
    if (!p->w[i] || !p->b[i]) return DPVO_E_INVALID;
  const int64_t maxg = n_patches_ub > n_pairs_ub ? n_patches_ub : n_pairs_ub;
  WsPm L;
  ws_layout_pm<RT>(E, maxg, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* wsb = (char*)ws;
  float* img = (float*)(wsb + L.img);
  _Float16 *fg = (_Float16*)(wsb + L.fg), *y = (_Float16*)(wsb + L.y);
  int32_t* invk = (int32_t*)(wsb + L.invk);
  Tile* tiles = (Tile*)(wsb + L.tiles);
  int32_t* hdr = (int32_t*)(wsb + L.hdr);
  hipStream_t st = (hipStream_t)stream;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  auto lin = [&](int i) { return Lin{p->w[i], (const _Float16*)p->b[i]}; };
  // upper bound on the number of tiles: a closed tile plus the next patch exceed R rows, so every tile but the last of each of
  // the 16 packing waves holds more than R - patch_edges_ub rows
  int64_t grid_tiles = E / (32 * RT - patch_edges_ub + 1) + 17;
  if (grid_tiles > L.max_tiles) grid_tiles = L.max_tiles;
  (void)n_cu;
  int rc;
#define FU(...) do { rc = (__VA_ARGS__); if (rc) return rc; } while (0)
  {
    int64_t g = cdiv64(E, 1024);
    if (g > 64) g = 64;
    hipLaunchKernelGGL(pm_prepare_kernel, dim3((unsigned)g), dim3(1024), 0, st, plan + PL.perm_k, plan + PL.patch_off,
                       plan + PL.counts, E, 32 * RT, invk, tiles, hdr, (int)grid_tiles);
    DPVO_LAUNCH_CHECK();
  }
  {
    PA a;
    a.c0 = lin(DPVO_UF_C0); a.c2 = lin(DPVO_UF_C2); a.c5 = lin(DPVO_UF_C5);
    a.c1a = lin(DPVO_UF_C1_0); a.c1b = lin(DPVO_UF_C1_2); a.c2a = lin(DPVO_UF_C2N_0); a.c2b = lin(DPVO_UF_C2N_2);
    a.fk = lin(DPVO_UF_AKK_F); a.gk = lin(DPVO_UF_AKK_G); a.hk = lin(DPVO_UF_AKK_H);
    a.fi = lin(DPVO_UF_AIJ_F); a.gi = lin(DPVO_UF_AIJ_G);
    a.cln_g = p->ln_g[0]; a.cln_b = p->ln_b[0]; a.norm_g = p->ln_g[1]; a.norm_b = p->ln_b[1];
    a.corr = (const _Float16*)corr; a.ld_corr = ld_corr; a.net = net; a.inp = (const _Float16*)inp; a.inp_rows = inp_rows;
    a.inp_mod = inp_mod;
    a.perm_k = plan + PL.perm_k; a.ix = plan + PL.ix; a.jx = plan + PL.jx; a.ku = plan + PL.ku;
    a.tiles = tiles; a.hdr = hdr; a.img = img; a.fg = fg; a.E = E;
    FU(launch<ka_kernel<RT, FU_DWPM>>(grid_tiles, GeoA<RT>::LDS_BYTES, a, st));
  }
  {
    int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
    const unsigned grid = (unsigned)(ngp < 8192 ? ngp : 8192);
    hipLaunchKernelGGL(softagg_inv_kernel, dim3(grid), dim3(384), 0, st, (const _Float16*)fg, (int64_t)768, plan + PL.perm_p,
                       (const int32_t*)invk, plan + PL.pair_off, plan + PL.counts + 1, y);
    DPVO_LAUNCH_CHECK();
  }
  {
    PB a;
    a.h = lin(DPVO_UF_AIJ_H);
    a.gate[0] = lin(DPVO_UF_G0_GATE); a.res0[0] = lin(DPVO_UF_G0_RES0); a.res2[0] = lin(DPVO_UF_G0_RES2);
    a.gate[1] = lin(DPVO_UF_G1_GATE); a.res0[1] = lin(DPVO_UF_G1_RES0); a.res2[1] = lin(DPVO_UF_G1_RES2);
    a.ln_g[0] = p->ln_g[2]; a.ln_b[0] = p->ln_b[2]; a.ln_g[1] = p->ln_g[3]; a.ln_b[1] = p->ln_b[3];
    a.d_w = (const _Float16*)p->d_w; a.d_b = (const _Float16*)p->d_b; a.w_w = (const _Float16*)p->w_w; a.w_b = (const _Float16*)p->w_b;
    a.y = y; a.pu = plan + PL.pu; a.perm_k = plan + PL.perm_k; a.tiles = tiles; a.hdr = hdr; a.img = img;
    a.coords = coords; a.pp = P * P;
    a.net_out = net_out; a.delta = delta; a.weight = weight; a.target = target; a.E = E;
    FU(launch<kb_kernel<RT, FU_DWKB>>(grid_tiles, GeoA<RT>::LDS_BYTES, a, st));
  }
  if (status) {      // device int32: 1 if a patch had more than 32 RT edges (results then undefined but memory safe)
    if (hipMemcpyAsync(status, hdr + HDR_ERR, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return DPVO_E_INVALID;
  }
#undef FU
  return DPVO_OK;
}

#ifdef FU_TRACE
extern "C" int dpvo_debug_fu_trace_buffer(void* buf) {      // trace builds only (make TRACE=1): device buffer of 8*1024*4*16 u64, or NULL
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fu_trace), &p, sizeof(p));
}
#endif
