// update_fused_k7.hip -- the LAST of the update operator's seven launches (see update_fused.hip for the operator and its geometry):
//   K7  y[pu] -> agg_ij.h -> net += ; 2 x (LayerNorm, x + gate(x) * res(x)); heads; target          (net.py:88-92, blocks.py:15-29, dpvo.py:340)
// In its own translation unit since round 5 because it is compiled with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): the f32 state is
// parked in the wave's AGPRs across the three GEMMs of a gated residual, so the MFMA accumulators must live in VGPRs.
#include "update_fused_dev.h"

namespace {
namespace fu {

#define FU_RT 3
#ifndef FU_DW7
#define FU_DW7 6
#endif

// K7 -----------------------------------------------------------------------------------------------
// The f32 state across the three GEMMs of a gated residual, PARKED IN THE ACCUMULATION REGISTERS (round 5).  Rounds 2-4 wrote it to
// its image slot after each LayerNorm and re-read it for the residual add: 4 x 1.5 KB per edge of memory traffic that exists only
// because 144 state + 144 accumulator registers do not fit 256 VGPRs -- 588 of the 955 KB a 96-row tile moves, ~60 of K7's ~180 us
// at the rate these lock-step phases reach.  A wave at one per SIMD owns 256 AGPRs next to its 256 VGPRs; with the MFMAs in their VGPR
// form (csrc/Makefile: -amdgpu-mfma-vgpr-form for this unit) nothing else wants them, and the register allocator cannot be talked into
// leaving 144 values there by itself (it spills to scratch instead: round 2), so the moves are explicit: v_accvgpr_write / _read through
// "a"-constrained operands.  288 one-cycle moves per lane and gated residual instead of two exposed memory round trips.
template <int RT> struct Park { float a[RT][3][16]; };
template <int RT>
__device__ __forceinline__ void park(Park<RT>& P, const f16v (&v)[RT][3]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(P.a[r][t][k]) : "v"(v[r][t][k]));
}
__device__ __forceinline__ float unpark(const float& a) {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

#ifndef FU_K7_PARK
#define FU_K7_PARK 1
#endif

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
#if FU_K7_PARK
    Park<RT> px;
    park<RT>(px, x);
#else
    img_store<RT>(x, ip);
#endif
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
#if !FU_K7_PARK
      f4 m[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#endif
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * c + i;
            const _Float16 rv = (_Float16)acc[r][t][k];
#if FU_K7_PARK
            x[r][t][k] = unpark(px.a[r][t][k]) + (float)(_Float16)(gt[i] * rv);
#else
            x[r][t][k] = m[t][k >> 2][k & 3] + (float)(_Float16)(gt[i] * rv);
#endif
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



}  // namespace fu
}  // namespace

namespace dpvo_fu {
int launch_k7(int64_t tiles, const P7& p, void* stream) {
  using namespace fu;
  return launch<k7_gru_heads<FU_RT, FU_DW7>>(tiles, Geo<FU_RT>::LDS_BYTES + Geo<FU_RT>::ACT_BYTES + 4 * D * 4, p, (hipStream_t)stream);
}
int k7_set_trace(unsigned long long* buf) {
#ifdef FU_TRACE
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fu_trace), &buf, sizeof(buf));
#else
  (void)buf;
  return 0;
#endif
}
}  // namespace dpvo_fu
