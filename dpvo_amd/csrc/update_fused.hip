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
//   K7  y[pu] -> agg_ij.h -> net += ; 2 x (LayerNorm, x + gate(x) * res(x)); heads           (net.py:88-92)   [update_fused_k7.hip]
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
#include "update_fused_dev.h"

namespace {
namespace fu {


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
// NT = 3: 4 waves; OCC = workgroups per CU (1: everything an epilogue needs is requested one GEMM early; 2: nothing is, the neighbour
// workgroup covers the round trips).  NT = 1: 12 waves, three per SIMD, one workgroup per CU; EARLY = what OCC == 1 means for NT = 3.
template <int RT, int DW, int OCC = 1, int NT = 3>
__global__ __launch_bounds__(64 * (12 / NT), OCC * 3 / NT) void k1_corr_norm(const P1 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G_ = Geo<RT, NT>;
  constexpr int R = G_::R, NTHR = G_::NTHR;
  constexpr bool EARLY = OCC == 1 && NT == 3;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  float* red = reinterpret_cast<float*>(smem + G_::ACT_BYTES);
  char* al = act + l.n * PITCH + 16 * l.h;

  soft_start(p.skew);
  FU_T(0, 0);
  f16v acc[RT][NT];
  h8 wf[DW][NT];
  Bias<NT> bias;
  // ---- Linear(882 -> 384) + ReLU over K = 896 streamed from global memory in 7 chunks of 128, two LDS stages
  {
    const h8* wp = w_base<NT>(p.c0.w, 56, l);
    w_preload<DW, NT>(wf, wp);
    if constexpr (EARLY) bias_load<NT>(bias, p.c0.b, l);
    constexpr int NS = RT * 2 * 256 / NTHR;            // 16-byte pieces per thread per chunk
    // chunk kc + 2 is requested (into one of two register sets) while chunk kc is multiplied and chunk kc + 1 waits in the
    // other set for its LDS stage: a chunk lasts ~1 us of MFMAs, a miss to HBM under load about twice that.
    h8 st[2][NS];
    auto load = [&](h8 (&d)[NS], int kc) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + NTHR * i, row = idx >> 4, ch = idx & 15;
        int64_t g = row0 + row;
        g = g < p.E ? g : p.E - 1;
        d[i] = *reinterpret_cast<const h8*>(p.corr + g * p.ld_corr + kc * KCH + ch * 8);
      }
    };
    auto store = [&](const h8 (&d)[NS], int b) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + NTHR * i, row = idx >> 4, ch = idx & 15;
        *reinterpret_cast<h8*>(smem + b * R * CPITCH + row * CPITCH + ch * 16) = d[i];
      }
    };
    load(st[0], 0);
    load(st[1], 1);
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.c0.b, l);
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
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[ks & 1][r], acc[r][t], 0, 0, 0);
        if (s + DW < 56) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kc + 1 < 7) store(st[(kc + 1) & 1], (kc + 1) & 1);
      __syncthreads();
    }
  }
  FU_T(0, 2);
  const h8* wp2 = w_base<NT>(p.c2.w, KS384, l);
  w_preload<DW, NT>(wf, wp2);
  if constexpr (EARLY) bias_load<NT>(bias, p.c2.b, l);
  to_lds<RT, 1, NT>(acc, al, l);                            // (the last barrier of the chunk loop freed the stages)
  __syncthreads();
  // ---- Linear, LayerNorm, ReLU
  if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.c2.b, l);
  gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp2, al);
  FU_T(0, 3);
  const h8* wp3 = w_base<NT>(p.c5.w, KS384, l);
  w_preload<DW, NT>(wf, wp3);
  if constexpr (EARLY) bias_load<NT>(bias, p.c5.b, l);
  round_f16<RT, NT>(acc);
  if constexpr (EARLY) layernorm_tile<RT, NT>(acc, red, p.cln_g, p.cln_b, l); else layernorm_tile_late<RT, NT>(acc, red, p.cln_g, p.cln_b, l);
  to_lds<RT, 1, NT>(acc, al, l);
  __syncthreads();
  FU_T(0, 4);
  // ---- Linear; net = LayerNorm(net + inp + .).
  if constexpr (OCC == 1) {
    // one workgroup per CU: the rows of net (feature order) are requested before the GEMM
    Img<RT, NT> nv;
    int64_t ir[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t g = row0 + r * 32 + l.n;
      g = g < p.E ? g : p.E - 1;
      ir[r] = g;
      if (p.inp_rows) { ir[r] = p.inp_rows[g]; if (p.inp_mod > 0) ir[r] %= p.inp_mod; }
      const int64_t gs = !p.net_rows ? g : (g < p.n_kept ? p.net_rows[g] : -1);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          nv.v[r][t][j] = gs >= 0 ? *reinterpret_cast<const f4*>(p.net + gs * D + 32 * (NT * l.w + t) + 8 * j + 4 * l.h) : (f4)0.f;
    }
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.c5.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp3, al);
    FU_T(0, 5);
    h4 iv[RT][NT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) iv[r][t][j] = *reinterpret_cast<const h4*>(p.inp + ir[r] * D + 32 * (NT * l.w + t) + 8 * j + 4 * l.h);
    round_f16<RT, NT>(acc);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[r][t][4 * j + q] = (nv.v[r][t][j][q] + (float)iv[r][t][j][q]) + acc[r][t][4 * j + q];
  } else {
    // several workgroups per CU: no registers for rows in flight under the GEMM; one 32-row tile at a time afterwards
    acc_init_mem<RT, NT>(acc, p.c5.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp3, al);
    FU_T(0, 5);
    round_f16<RT, NT>(acc);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t g = row0 + r * 32 + l.n;
      g = g < p.E ? g : p.E - 1;
      int64_t ir = g;
      if (p.inp_rows) { ir = p.inp_rows[g]; if (p.inp_mod > 0) ir %= p.inp_mod; }
      const int64_t gs = !p.net_rows ? g : (g < p.n_kept ? p.net_rows[g] : -1);
      f4 nv[NT][4]; h4 iv[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nv[t][j] = gs >= 0 ? *reinterpret_cast<const f4*>(p.net + gs * D + 32 * (NT * l.w + t) + 8 * j + 4 * l.h) : (f4)0.f;
          iv[t][j] = *reinterpret_cast<const h4*>(p.inp + ir * D + 32 * (NT * l.w + t) + 8 * j + 4 * l.h);
        }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[r][t][4 * j + q] = (nv[t][j][q] + (float)iv[t][j][q]) + acc[r][t][4 * j + q];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (EARLY) layernorm_tile<RT, NT>(acc, red, p.norm_g, p.norm_b, l); else layernorm_tile_late<RT, NT>(acc, red, p.norm_g, p.norm_b, l);
  FU_T(0, 6);
  img_store<RT, NT>(acc, img_ptr<RT, NT>(p.img, tile, l));
  to_lds<RT, 0, NT>(acc, al, l);
  __syncthreads();
  scatter_rows<RT, NTHR>(act, p.rows16, row0, p.E, l.tid);
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

template <int RT, int DW, int MODE, int OCC = 1, int NT = 3>
__global__ __launch_bounds__(64 * (12 / NT), OCC * 3 / NT) void k_chain(const P2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G_ = Geo<RT, NT>;
  constexpr int R = G_::R, NTHR = G_::NTHR;
  constexpr bool EARLY = OCC == 1 && NT == 3;       // one wave per SIMD: biases requested one GEMM early
  constexpr bool IMG_EARLY = OCC == 1;              // one workgroup per CU: the f32 image requested one GEMM early (12 waves: three
                                                    // dependent image round trips behind the last GEMM were 4 of a tile's 24 us)
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  char* al = act + l.n * PITCH + 16 * l.h;

  soft_start(p.skew);
  FU_T(1 + MODE, 0);
  f16v acc[RT][NT];
  h8 wf[DW][NT];
  Bias<NT> bias;
  Img<RT, NT> im;
  float* ip = img_ptr<RT, NT>(p.img, tile, l);
  const h8* wpa = w_base<NT>(MODE == MODE_H ? p.b.w : p.a.w, KS384, l);
  w_preload<DW, NT>(wf, wpa);
  if constexpr (EARLY) bias_load<NT>(bias, MODE == MODE_H ? p.b.b : p.a.b, l);
  gather_rows<RT, NTHR>(act, p.src, p.rows, row0, p.E, l.tid, reinterpret_cast<int32_t*>(smem + G_::ACT_BYTES));
  if constexpr (MODE == MODE_H && IMG_EARLY) img_load<RT, NT>(im, ip);
  FU_T(1 + MODE, 1);
  __syncthreads();
  FU_T(1 + MODE, 2);
  if constexpr (MODE != MODE_H) {
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.a.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wpa, al);
    const h8* wpb = w_base<NT>(p.b.w, KS384, l);
    FU_T(1 + MODE, 3);
    w_preload<DW, NT>(wf, wpb);
    if constexpr (EARLY) bias_load<NT>(bias, p.b.b, l);
    if constexpr (IMG_EARLY) img_load<RT, NT>(im, ip);   // lands under the second GEMM
    __syncthreads();
    to_lds<RT, 1, NT>(acc, al, l);
    __syncthreads();
    FU_T(1 + MODE, 4);
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.b.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wpb, al);
  } else {
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.b.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wpa, al);
  }
  FU_T(1 + MODE, 5);
  const h8* wpf = nullptr;
  if constexpr (MODE != MODE_C1) { wpf = w_base<NT>(p.f.w, KS384, l); w_preload<DW, NT>(wf, wpf); if constexpr (EARLY) bias_load<NT>(bias, p.f.b, l); }
  round_f16<RT, NT>(acc);
  if constexpr (IMG_EARLY) {
    img_add<RT, NT>(acc, im);
    img_store<RT, NT>(acc, ip);
  } else {
    img_add_store_stream<RT, NT>(acc, ip);
  }
  FU_T(1 + MODE, 6);
  __syncthreads();
  to_lds<RT, 0, NT>(acc, al, l);
  __syncthreads();
  FU_T(1 + MODE, 7);
  if constexpr (MODE == MODE_C1) {
    scatter_rows<RT, NTHR>(act, p.rows16, row0, p.E, l.tid);
    FU_T(1 + MODE, 8);
  } else {
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.f.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wpf, al);
    FU_T(1 + MODE, 8);
    const h8* wpg = w_base<NT>(p.g.w, KS384, l);
    w_preload<DW, NT>(wf, wpg);
    if constexpr (EARLY) bias_load<NT>(bias, p.g.b, l);
    to_rows<RT, NT>(acc, p.fg, 768, row0, p.E, l);
    FU_T(1 + MODE, 9);
    if constexpr (EARLY) acc_init<RT, NT>(acc, bias); else acc_init_mem<RT, NT>(acc, p.g.b, l);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wpg, al);
    FU_T(1 + MODE, 10);
    to_rows<RT, NT>(acc, p.fg + D, 768, row0, p.E, l);
    FU_T(1 + MODE, 11);
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

namespace {
// ---------------------------------------------------------------------------------------------------
// SoftAgg: one 384-thread block per group.  Thread (q, cq): member slice q = t / 96 (members b + q, b + q + 4, ...) and
// channels [4 cq, 4 cq + 4) -- 8-byte loads (with one channel per thread the kernel was bound by the NUMBER of 2-byte
// vector loads: 96 members x 12 wave-loads for a frame-pair group) and a dependent online-softmax chain of a quarter of
// the members; the four partial (max, sum, weighted sum) triples are merged through LDS in the fixed order q = 0..3.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void softagg_kernel(const _Float16* __restrict__ fg, int64_t ldfg,
                                                      const int32_t* __restrict__ perm, const int32_t* __restrict__ off,
                                                      const int32_t* __restrict__ n_groups, _Float16* __restrict__ y,
                                                      int D) {
  __shared__ float part[4][3][384];
  const int ng = *n_groups;
  const int q = threadIdx.x / 96, cq = threadIdx.x - 96 * q;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int b = off[g], e = off[g + 1];
    float m[4], s[4], a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; s[r] = 0.f; a[r] = 0.f; }
    // SA_U members of the slice in flight at a time: first their row ids, then their rows, then the (sequential) online-softmax
    // steps in member order -- the same operations in the same order as one member per trip, which was two DEPENDENT round trips per
    // member (id, then row): 24 trips for a frame-pair group at ~1 us each were the kernel's 24 us (round 6)
    constexpr int SA_U = 6;
    for (int p0 = b + q; p0 < e; p0 += 4 * SA_U) {
      int rid[SA_U];
      h4 fx[SA_U], gx[SA_U];
#pragma unroll
      for (int u = 0; u < SA_U; ++u) rid[u] = p0 + 4 * u < e ? perm[p0 + 4 * u] : -1;
#pragma unroll
      for (int u = 0; u < SA_U; ++u)
        if (rid[u] >= 0) {
          const _Float16* rowp = fg + (int64_t)rid[u] * ldfg + 4 * cq;
          fx[u] = *reinterpret_cast<const h4*>(rowp);
          gx[u] = *reinterpret_cast<const h4*>(rowp + D);
        }
#pragma unroll
      for (int u = 0; u < SA_U; ++u)
        if (rid[u] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g_ = (float)gx[u][r];
            const float mn = fmaxf(m[r], g_);
            const float sc = __expf(m[r] - mn), w = __expf(g_ - mn);
            s[r] = s[r] * sc + w;
            a[r] = a[r] * sc + w * (float)fx[u][r];
            m[r] = mn;
          }
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
        const float sc = (mk == -INFINITY) ? 0.f : __expf(mk - M);      // an empty slice (fewer than 4 members) contributes nothing
        S += part[k][1][c] * sc;
        A += part[k][2][c] * sc;
      }
      y[(int64_t)g * D + c] = (_Float16)(A / S);
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int dpvo_softagg(const void* fg, int64_t ldfg, const int32_t* perm, const int32_t* off,
                            const int32_t* n_groups, int64_t max_groups, void* y, int D, void* stream) {
  if (max_groups < 0) return DPVO_E_INVALID;
  if (max_groups == 0) return DPVO_OK;
  if (D != 384 || (ldfg % 4) || ((uintptr_t)fg & 7)) return DPVO_E_UNSUPPORTED;      // 8-byte row loads
  if (!fg || !perm || !off || !n_groups || !y) return DPVO_E_INVALID;
  const unsigned grid = (unsigned)(max_groups < 8192 ? max_groups : 8192);
  hipLaunchKernelGGL(softagg_kernel, dim3(grid), dim3(384), 0, (hipStream_t)stream, (const _Float16*)fg, ldfg, perm, off,
                     n_groups, (_Float16*)y, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}



// One fixed set of tile shapes / weight-ring depths (the A/B switches of rounds 2-5 are gone; their measurements: DESIGN.md 3.4):
#define FU_RT 3                // 96-row tiles (one workgroup per CU)
#define FU_DW 6                // weight ring of the 4-wave kernels at one workgroup per CU
#define FU_DW1 6               // ... of K1 there
#define FU_RT2 2               // 64-row tiles (two workgroups per CU, 4 waves each)
#define FU_DW2 3               // K1's ring there (spills beyond 3)
#define FU_DW2C 6              // the chain kernels' ring there (6 fits 250 registers)
#define FU_OCC2 2
// chains: 64-row tiles x 2 workgroups per CU, 4 waves each (A/B in the frame, round 3: cfg 1 / 3 / 0 / 2 = 828 / 817 / 814 / 800 frames/sec on one
// box); K1 and K7: 96-row tiles x 1 workgroup of TWELVE waves (round 6: 516 -> 483 us alone, 0.512 -> 0.483 ms in the frame, 956-966 -> 984-993
// frames/sec on one box, profiles/r06_b_*; the chains gain nothing from the 12-wave geometry: 481 us with all five kernels on it)
#define FU_CFG_DEFAULT 13
#define FU_DW1W 5              // K1's weight ring at three waves per SIMD (NT = 1: 6 spills one register)
#define FU_DWCW 5              // the chain kernels' ring at three waves per SIMD (6 spills six registers in the c1 kernel)

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
  constexpr int RT2 = FU_RT2, DW2 = FU_DW2, OCC2 = FU_OCC2;       // several workgroups per CU, 64-row tiles
  constexpr int RT2C = FU_RT2, DW2C = FU_DW2C;
  const int cfg = (p->tiling < 0 ? FU_CFG_DEFAULT : p->tiling) & 31;
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
    if (cfg & 8) FU(launch<k1_corr_norm<RT, FU_DW1W, 1, 1>, P1, Geo<RT, 1>::NTHR>(tiles, Geo<RT, 1>::LDS_BYTES, a, st));
    else if (cfg & 2) FU(launch<k1_corr_norm<RT2, DW2, OCC2>>(tiles2, Geo<RT2>::LDS_BYTES, a, st));
    else FU(launch<k1_corr_norm<RT, FU_DW1>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C1_0), lin(DPVO_UF_C1_2), Lin{nullptr, nullptr}, Lin{nullptr, nullptr}, r16a, plan + PL.ix, img, r16b,
         nullptr, E, skew};
    if (cfg & 16) FU(launch<k_chain<RT, FU_DWCW, MODE_C1, 1, 1>, P2, Geo<RT, 1>::NTHR>(tiles, Geo<RT, 1>::LDS_BYTES, a, st));
    else if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_C1, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
    else FU(launch<k_chain<RT, DW, MODE_C1>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  {
    P2 a{lin(DPVO_UF_C2N_0), lin(DPVO_UF_C2N_2), lin(DPVO_UF_AKK_F), lin(DPVO_UF_AKK_G), r16b, plan + PL.jx, img, nullptr, fg,
         E, skew};
    if (cfg & 16) FU(launch<k_chain<RT, FU_DWCW, MODE_C2, 1, 1>, P2, Geo<RT, 1>::NTHR>(tiles, Geo<RT, 1>::LDS_BYTES, a, st));
    else if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_C2, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
    else FU(launch<k_chain<RT, DW, MODE_C2>>(tiles, Geo<RT>::LDS_BYTES, a, st));
  }
  int64_t ngk = n_patches_ub < 1 ? 1 : (n_patches_ub > E ? E : n_patches_ub);
  int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
  FU(dpvo_softagg(fg, 768, plan + PL.perm_k, plan + PL.patch_off, plan + PL.counts + 0, ngk, y, 384, stream));
  {
    P2 a{Lin{nullptr, nullptr}, lin(DPVO_UF_AKK_H), lin(DPVO_UF_AIJ_F), lin(DPVO_UF_AIJ_G), y, plan + PL.ku, img, nullptr, fg, E, skew};
    if (cfg & 16) FU(launch<k_chain<RT, FU_DWCW, MODE_H, 1, 1>, P2, Geo<RT, 1>::NTHR>(tiles, Geo<RT, 1>::LDS_BYTES, a, st));
    else if (cfg & 1) FU(launch<k_chain<RT2C, DW2C, MODE_H, OCC2>>(tiles2c, Geo<RT2C>::LDS_BYTES, a, st));
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
    FU(dpvo_fu::launch_k7(tiles, a, cfg & 4, stream));          // (update_fused_k7.hip: 96-row tiles, one workgroup per CU)
  }
#undef FU
  return DPVO_OK;
}


#ifdef FU_TRACE
extern "C" int dpvo_debug_fu_trace_buffer(void* buf) {      // trace builds only (make TRACE=1): device buffer of 8*1024*4*16 u64, or NULL
  unsigned long long* p = (unsigned long long*)buf;
  const int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fu_trace), &p, sizeof(p));
  return rc ? rc : dpvo_fu::k7_set_trace(p);
}
#endif
