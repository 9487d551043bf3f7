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

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
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

// rows of a P-order f16 matrix [., 384] -> the LDS tile.  rows == nullptr: row0 + i; an index < 0 or a row >= E: zeros
template <int RT>
__device__ __forceinline__ void gather_rows(char* act, const _Float16* __restrict__ src, const int32_t* __restrict__ rows,
                                            int64_t row0, int64_t E, int tid) {
  constexpr int N = RT * 6;                      // 32 RT rows x 48 pieces of 16 B over 256 threads
  h8 v[N];
  int32_t sr[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48;
    const int64_t g = row0 + row;
    sr[i] = g < E ? (rows ? rows[g] : (int32_t)g) : -1;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    v[i] = sr[i] >= 0 ? *reinterpret_cast<const h8*>(src + (int64_t)sr[i] * D + ch * 8) : (h8)(_Float16)0;
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
  const _Float16* inp; const int64_t* inp_rows; int64_t inp_mod;
  float* img; _Float16* rows16;                       // out
  int64_t E;
};

// K1 -----------------------------------------------------------------------------------------------
template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void k1_corr_norm(const P1 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = Geo<RT>::R;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  float* red = reinterpret_cast<float*>(smem + Geo<RT>::ACT_BYTES);
  char* al = act + l.n * PITCH + 16 * l.h;

  FU_T(0, 0);
  f16v acc[RT][3];
  h8 wf[DW][3];
  Bias bias;
  // ---- Linear(882 -> 384) + ReLU over K = 896 streamed from global memory in 7 chunks of 128, two LDS stages
  {
    const h8* wp = w_base(p.c0.w, 56, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.c0.b, l);
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
    acc_init<RT>(acc, bias);
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
  bias_load(bias, p.c2.b, l);
  to_lds<RT, 1>(acc, al, l);                            // (the last barrier of the chunk loop freed the stages)
  __syncthreads();
  // ---- Linear, LayerNorm, ReLU
  acc_init<RT>(acc, bias);
  gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp2, al);
  FU_T(0, 3);
  const h8* wp3 = w_base(p.c5.w, KS384, l);
  w_preload<DW>(wf, wp3);
  bias_load(bias, p.c5.b, l);
  round_f16<RT>(acc);
  layernorm_tile<RT>(acc, red, p.cln_g, p.cln_b, l);
  to_lds<RT, 1>(acc, al, l);
  __syncthreads();
  FU_T(0, 4);
  // ---- Linear; net = LayerNorm(net + inp + .).  The rows of net (feature order) are requested before the GEMM.
  Img<RT> nv;
  int64_t ir[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    int64_t g = row0 + r * 32 + l.n;
    g = g < p.E ? g : p.E - 1;
    ir[r] = g;
    if (p.inp_rows) { ir[r] = p.inp_rows[g]; if (p.inp_mod > 0) ir[r] %= p.inp_mod; }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) nv.v[r][t][j] = *reinterpret_cast<const f4*>(p.net + g * D + 96 * l.w + 32 * t + 8 * j + 4 * l.h);
  }
  acc_init<RT>(acc, bias);
  gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wp3, al);
  FU_T(0, 5);
  {
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
  }
  layernorm_tile<RT>(acc, red, p.norm_g, p.norm_b, l);
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
};
enum { MODE_C1 = 0, MODE_C2 = 1, MODE_H = 2 };

template <int RT, int DW, int MODE>
__global__ __launch_bounds__(256, 1) void k_chain(const P2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = Geo<RT>::R;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  char* al = act + l.n * PITCH + 16 * l.h;

  FU_T(1 + MODE, 0);
  f16v acc[RT][3];
  h8 wf[DW][3];
  Bias bias;
  Img<RT> im;
  float* ip = img_ptr<RT>(p.img, tile, l);
  const h8* wpa = w_base(MODE == MODE_H ? p.b.w : p.a.w, KS384, l);
  w_preload<DW>(wf, wpa);
  bias_load(bias, MODE == MODE_H ? p.b.b : p.a.b, l);
  gather_rows<RT>(act, p.src, p.rows, row0, p.E, l.tid);
  if constexpr (MODE == MODE_H) img_load<RT>(im, ip);
  FU_T(1 + MODE, 1);
  __syncthreads();
  FU_T(1 + MODE, 2);
  if constexpr (MODE != MODE_H) {
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpa, al);
    const h8* wpb = w_base(p.b.w, KS384, l);
    FU_T(1 + MODE, 3);
    w_preload<DW>(wf, wpb);
    bias_load(bias, p.b.b, l);
    img_load<RT>(im, ip);                           // lands under the second GEMM
    __syncthreads();
    to_lds<RT, 1>(acc, al, l);
    __syncthreads();
    FU_T(1 + MODE, 4);
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpb, al);
  } else {
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpa, al);
  }
  FU_T(1 + MODE, 5);
  const h8* wpf = nullptr;
  if constexpr (MODE != MODE_C1) { wpf = w_base(p.f.w, KS384, l); w_preload<DW>(wf, wpf); bias_load(bias, p.f.b, l); }
  round_f16<RT>(acc);
  img_add<RT>(acc, im);
  img_store<RT>(acc, ip);
  FU_T(1 + MODE, 6);
  __syncthreads();
  to_lds<RT, 0>(acc, al, l);
  __syncthreads();
  FU_T(1 + MODE, 7);
  if constexpr (MODE == MODE_C1) {
    scatter_rows<RT>(act, p.rows16, row0, p.E, l.tid);
    FU_T(1 + MODE, 8);
  } else {
    acc_init<RT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH>(acc, wf, wpf, al);
    FU_T(1 + MODE, 8);
    const h8* wpg = w_base(p.g.w, KS384, l);
    w_preload<DW>(wf, wpg);
    bias_load(bias, p.g.b, l);
    to_rows<RT>(acc, p.fg, 768, row0, p.E, l);
    FU_T(1 + MODE, 9);
    acc_init<RT>(acc, bias);
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

  FU_T(4, 0);
  f16v acc[RT][3], x[RT][3];
  h8 wf[DW][3];
  Bias bias;
  const h8* wp = w_base(p.h.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.h.b, l);
  gather_rows<RT>(act, p.y, p.rows, row0, p.E, l.tid);
  {
    Img<RT> im;
    img_load<RT>(im, img_ptr<RT>(const_cast<float*>(p.img), tile, l));       // lands under the first GEMM
    __syncthreads();
    FU_T(4, 1);
    acc_init<RT>(x, bias);
    gemm_lds<RT, KS384, DW, PITCH>(x, wf, wp, al);
    FU_T(4, 2);
    round_f16<RT>(x);
    img_add<RT>(x, im);
  }
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    layernorm_tile<RT>(x, red, p.ln_g[G], p.ln_b[G], l);
    FU_T(4, 3 + 5 * G);
    wp = w_base(p.gate[G].w, KS384, l);           // (after the LayerNorm: its registers are needed there)
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gate[G].b, l);
    to_lds<RT, 0>(x, al, l);
    __syncthreads();
    FU_T(4, 4 + 5 * G);
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
    // x = x + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const _Float16 rv = (_Float16)acc[r][t][8 * c + i];
            x[r][t][8 * c + i] += (float)(_Float16)(gt[i] * rv);
          }
        }
  }
  FU_T(4, 13);
  // ---- hidden state out (feature order) and the heads
  float dsum[RT][4];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int o = 0; o < 4; ++o) dsum[r][o] = 0.f;
  h4 wv[3][4][4];                                  // the four head rows at this lane's 48 features (one round trip)
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
        const int64_t g = row0 + r * 32 + l.n;
        f4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = x[r][t][4 * j + q];
          o4[q] = v;
          const float a = (float)(_Float16)(v > 0.f ? v : 0.f);
#pragma unroll
          for (int o = 0; o < 4; ++o) dsum[r][o] += a * (float)wv[t][j][o][q];
        }
        if (g < p.E) *reinterpret_cast<f4*>(p.net_out + g * D + f) = o4;
      }
    }
  __syncthreads();                                  // (the LayerNorm partials in `red` are dead)
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    f4 s;
#pragma unroll
    for (int o = 0; o < 4; ++o) { s[o] = dsum[r][o]; s[o] += xhalf(s[o]); }
    if (l.h == 0) *reinterpret_cast<f4*>(red + ((r * 32 + l.n) * 4 + l.w) * 4) = s;
  }
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

template <typename K, typename P>
int launch(K kern, int64_t tiles, int lds, const P& p, hipStream_t st) {
  static bool attr_done = false;          // (one flag per kernel: the template is instantiated per kernel type)
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return DPVO_E_UNSUPPORTED;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), (size_t)lds, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct Ws { size_t img, r16a, r16b, fg, y, total; };
template <int RT>
void ws_layout(int64_t E, int64_t maxg, Ws* w) {
  const size_t tiles = (size_t)cdiv64(E > 0 ? E : 1, 32 * RT), g = (size_t)(maxg > 0 ? maxg : 1), e = (size_t)(E > 0 ? E : 1);
  size_t o = 0;
  w->img = o; o += al256(tiles * 32 * RT * D * 4);
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

extern "C" size_t dpvo_update_fused_pack_bytes(int K) { return K > 0 && (K % 16) == 0 ? (size_t)384 * K * 2 : 0; }

extern "C" int dpvo_update_fused_pack(const void* W, int64_t ldw, int K, int k_valid, int chained, void* out, void* stream) {
  if (!W || !out || K <= 0 || (K % 16) != 0 || k_valid < 0 || k_valid > K || ldw < k_valid) return DPVO_E_INVALID;
  if (chained && K != 384) return DPVO_E_UNSUPPORTED;
  hipLaunchKernelGGL(fu::pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)W, ldw, K, k_valid,
                     chained ? 1 : 0, (_Float16*)out);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

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
  using namespace fu;
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8) || E >= (1ll << 31)) return DPVO_E_UNSUPPORTED;
  for (int i = 0; i < DPVO_UF_NLIN; ++i)
    if (!p->w[i] || !p->b[i]) return DPVO_E_INVALID;
  constexpr int RT = FU_RT, DW = FU_DW;
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
  const int64_t tiles = cdiv64(E, 32 * RT);
  auto lin = [&](int i) { return Lin{p->w[i], (const _Float16*)p->b[i]}; };
  int rc;
#define FU(call) do { rc = (call); if (rc) return rc; } while (0)
  {
    P1 a{lin(DPVO_UF_C0), lin(DPVO_UF_C2), lin(DPVO_UF_C5), p->ln_g[0], p->ln_b[0], p->ln_g[1], p->ln_b[1],
         (const _Float16*)corr, ld_corr, net, (const _Float16*)inp, inp_rows, inp_mod, img, r16a, E};
    FU(launch(k1_corr_norm<RT, DW>, tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C1_0), lin(DPVO_UF_C1_2), Lin{nullptr, nullptr}, Lin{nullptr, nullptr}, r16a, plan + PL.ix, img, r16b,
         nullptr, E};
    FU(launch(k_chain<RT, DW, MODE_C1>, tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C2N_0), lin(DPVO_UF_C2N_2), lin(DPVO_UF_AKK_F), lin(DPVO_UF_AKK_G), r16b, plan + PL.jx, img, nullptr, fg,
         E};
    FU(launch(k_chain<RT, DW, MODE_C2>, tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  int64_t ngk = n_patches_ub < 1 ? 1 : (n_patches_ub > E ? E : n_patches_ub);
  int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
  FU(dpvo_softagg(fg, 768, plan + PL.perm_k, plan + PL.patch_off, plan + PL.counts + 0, ngk, y, 384, stream));
  {
    P2 a{Lin{nullptr, nullptr}, lin(DPVO_UF_AKK_H), lin(DPVO_UF_AIJ_F), lin(DPVO_UF_AIJ_G), y, plan + PL.ku, img, nullptr, fg, E};
    FU(launch(k_chain<RT, DW, MODE_H>, tiles, Geo<RT>::LDS_BYTES, a, st));
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
    a.net_out = net_out; a.delta = delta; a.weight = weight; a.target = target; a.E = E;
    FU(launch(k7_gru_heads<RT, FU_DW7>, tiles, Geo<RT>::LDS_BYTES + Geo<RT>::ACT_BYTES, a, st));
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
