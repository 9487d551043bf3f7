// update_fused_k7.hip -- the LAST of the update operator's seven launches (see update_fused.hip for the operator and its geometry):
//   K7  y[pu] -> agg_ij.h -> net += ; 2 x (LayerNorm, x + gate(x) * res(x)); heads; target          (net.py:88-92, blocks.py:15-29, dpvo.py:340)
// In its own translation unit since round 5 because it is compiled with -mllvm -amdgpu-mfma-vgpr-form (csrc/Makefile): the f32 state is
// parked in the wave's AGPRs across the three GEMMs of a gated residual, so the MFMA accumulators must live in VGPRs.
#include "update_fused_dev.h"

namespace {
namespace fu {

#define FU_RT 3
#define FU_DW7 6                // weight ring, one wave per SIMD (NT = 3)
#define FU_DW7W 6               // weight ring, three waves per SIMD (NT = 1)

// K7 -----------------------------------------------------------------------------------------------
// NT = 3 (4 waves, one per SIMD): the f32 state across the three GEMMs of a gated residual is PARKED IN THE ACCUMULATION REGISTERS
// (round 5).  Rounds 2-4 wrote it to its image slot after each LayerNorm and re-read it for the residual add: 4 x 1.5 KB per edge of
// memory traffic that exists only because 144 state + 144 accumulator registers do not fit 256 VGPRs.  A wave at one per SIMD owns
// 256 AGPRs next to its 256 VGPRs; with the MFMAs in their VGPR form (csrc/Makefile: -amdgpu-mfma-vgpr-form for this unit) nothing
// else wants them, and the register allocator cannot be talked into leaving 144 values there by itself (it spills to scratch instead:
// round 2), so the moves are explicit: v_accvgpr_write / _read through "a"-constrained operands.
// NT = 1 (12 waves, three per SIMD; round 6): 48 state + 48 accumulator registers per wave are plain VGPRs.
template <int RT, int NT> struct Park { float a[RT][NT][16]; };
template <int RT, int NT>
__device__ __forceinline__ void park(Park<RT, NT>& P, const f16v (&v)[RT][NT]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if constexpr (NT == 3) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(P.a[r][t][k]) : "v"(v[r][t][k]));
        else P.a[r][t][k] = v[r][t][k];
      }
}
template <int NT>
__device__ __forceinline__ float unpark(const float& a) {
  if constexpr (NT == 3) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
  } else {
    return a;
  }
}

template <int RT, int DW, int NT>
__global__ __launch_bounds__(64 * (12 / NT), 3 / NT) void k7_gru_heads(const P7 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G_ = Geo<RT, NT>;
  constexpr int R = G_::R, NW = G_::NW, NTHR = G_::NTHR;
  const Lane l = lane_of();
  const int64_t tile = blockIdx.x, row0 = tile * R;
  char* act = smem;
  float* red = reinterpret_cast<float*>(smem + 2 * G_::ACT_BYTES);
  char* al = act + l.n * PITCH + 16 * l.h;
  char* gl = al + G_::ACT_BYTES;                    // second tile: where every lane parks its own gate values (no barriers)
  // NT = 3: the CURRENT LayerNorm's [gamma | beta] in LDS (no 96-register parameter block beside the 144-register state);
  // NT = 1: read from global memory after the statistics (three waves per SIMD cover the round trip)
  float* lnp = reinterpret_cast<float*>(smem + 2 * G_::ACT_BYTES + G_::RED_BYTES);

  soft_start(p.skew);
  FU_T(4, 0);
  f16v x[RT][NT];
  h8 wf[DW][NT];
  Bias<NT> bias;
  const h8* wp = w_base<NT>(p.h.w, KS384, l);
  w_preload<DW, NT>(wf, wp);
  bias_load<NT>(bias, p.h.b, l);
  if constexpr (NT == 3)
    for (int i = l.tid; i < D; i += NTHR) { lnp[i] = p.ln_g[0][i]; lnp[D + i] = p.ln_b[0][i]; }
  gather_rows<RT, NTHR>(act, p.y, p.rows, row0, p.E, l.tid, reinterpret_cast<int32_t*>(red));
  float* ip = img_ptr<RT, NT>(const_cast<float*>(p.img), tile, l);
  {
    Img<RT, NT> im;
    img_load<RT, NT>(im, ip);       // lands under the first GEMM
    __syncthreads();
    FU_T(4, 1);
    acc_init<RT, NT>(x, bias);
    gemm_lds<RT, KS384, DW, PITCH, NT>(x, wf, wp, al);
    FU_T(4, 2);
    round_f16<RT, NT>(x);
    img_add<RT, NT>(x, im);
  }
#pragma unroll
  for (int G = 0; G < 2; ++G) {
    if constexpr (NT == 3) layernorm_tile_lds<RT, NT>(x, red, lnp, l);
    else layernorm_tile_late<RT, NT>(x, red, p.ln_g[G], p.ln_b[G], l);
    FU_T(4, 3 + 5 * G);
    wp = w_base<NT>(p.gate[G].w, KS384, l);
    w_preload<DW, NT>(wf, wp);
    bias_load<NT>(bias, p.gate[G].b, l);
    Park<RT, NT> px;
    park<RT, NT>(px, x);
    to_lds<RT, 0, NT>(x, al, l);
    __syncthreads();
    if constexpr (NT == 3)
      if (G == 0)                 // every wave is past the first LayerNorm: its parameters make room for the second one's
        for (int i = l.tid; i < D; i += NTHR) { lnp[i] = p.ln_g[1][i]; lnp[D + i] = p.ln_b[1][i]; }
    FU_T(4, 4 + 5 * G);
    f16v acc[RT][NT];
    // gate = sigmoid(Linear(x))
    acc_init<RT, NT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp, al);
    FU_T(4, 5 + 5 * G);
    wp = w_base<NT>(p.res0[G].w, KS384, l);
    w_preload<DW, NT>(wf, wp);
    bias_load<NT>(bias, p.res0[G].b, l);
    to_lds<RT, 2, NT>(acc, gl, l);
    // res = Linear(relu(Linear(x)))
    acc_init<RT, NT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp, al);
    FU_T(4, 6 + 5 * G);
    wp = w_base<NT>(p.res2[G].w, KS384, l);
    w_preload<DW, NT>(wf, wp);
    bias_load<NT>(bias, p.res2[G].b, l);
    __syncthreads();
    to_lds<RT, 1, NT>(acc, al, l);
    __syncthreads();
    acc_init<RT, NT>(acc, bias);
    gemm_lds<RT, KS384, DW, PITCH, NT>(acc, wf, wp, al);
    FU_T(4, 7 + 5 * G);
    // x = x + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((NT * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * c + i;
            const _Float16 rv = (_Float16)acc[r][t][k];
            x[r][t][k] = unpark<NT>(px.a[r][t][k]) + (float)(_Float16)(gt[i] * rv);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  FU_T(4, 13);
  // ---- hidden state out (feature order) and the heads
  // d and w (two Linear(384, 2) on relu(net), net.py:92) as ONE MFMA chain per wave over its own features: the B fragment of
  // k-step (NT w + t) * 2 + c is exactly what to_lds would write for this lane -- relu(x[r][t][8c .. 8c + 7]) in f16 -- so it is
  // built in registers; the A fragment carries the four head rows (rows 4..31 zero) in the same P order.  6 NT MFMAs instead of
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
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int base = 32 * (NT * l.w + t) + 16 * c + 4 * l.h;      // features base + {0..3} and base + 8 + {0..3}
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
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 32 * (NT * l.w + t) + 8 * j + 4 * l.h;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int64_t g = row0 + r * 32 + l.n;
        f4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = x[r][t][4 * j + q];
        if (g < p.E) *reinterpret_cast<f4*>(p.net_out + g * D + f) = o4;
      }
    }
  __syncthreads();                                  // (every wave is past its last GEMM: the activation tile is dead)
  // D[row m][col n]: lane (n, h) holds rows (j & 3) + 8 (j >> 2) + 4 h in register j -> the four head sums of tile row n sit
  // in registers 0..3 of the lanes with h == 0, both K halves already added.  Partials [row][wave] x f4 over the dead activation tile.
  float* hred = reinterpret_cast<float*>(act);
#pragma unroll
  for (int r = 0; r < RT; ++r)
    if (l.h == 0) *reinterpret_cast<f4*>(hred + ((r * 32 + l.n) * NW + l.w) * 4) = (f4){hacc[r][0], hacc[r][1], hacc[r][2], hacc[r][3]};
  __syncthreads();
  if (l.tid < R) {
    const int64_t g = row0 + l.tid;
    if (g < p.E) {
      f4 s = *reinterpret_cast<const f4*>(hred + (l.tid * NW + 0) * 4);
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        const f4 q = *reinterpret_cast<const f4*>(hred + (l.tid * NW + w) * 4);
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
int launch_k7(int64_t tiles, const P7& p, int waves12, void* stream) {
  using namespace fu;
  if (waves12)
    return launch<k7_gru_heads<FU_RT, FU_DW7W, 1>, P7, Geo<FU_RT, 1>::NTHR>(tiles, 2 * Geo<FU_RT, 1>::ACT_BYTES + Geo<FU_RT, 1>::RED_BYTES, p,
                                                                            (hipStream_t)stream);
  return launch<k7_gru_heads<FU_RT, FU_DW7, 3>, P7, 256>(tiles, 2 * Geo<FU_RT, 3>::ACT_BYTES + Geo<FU_RT, 3>::RED_BYTES + 2 * D * 4, p,
                                                         (hipStream_t)stream);
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
