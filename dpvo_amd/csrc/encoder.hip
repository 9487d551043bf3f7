// encoder.hip -- the two feature encoders of the Patchifier (reference dpvo/extractor.py:200-264 BasicEncoder4,
// ResidualBlock :6-55; called from dpvo/net.py:110-127) as hand-written MFMA implicit-GEMM convolutions for gfx950.
//
// SURVEY.md section 8(f) "next #1": the step right before the hot path, every frame, ~24 GFLOP.  MIOpen runs these
// small-channel (32/64) convolutions on dot2 VALU kernels (0.9 ms) plus ~56 elementwise launches for bias / norm /
// ReLU / residual; here
//   * activations live in NHWC f16, so one pixel's channels are 64 / 128 contiguous bytes = MFMA fragments;
//   * a workgroup computes an 8x32 output tile for 64 (or all) output channels: the input halo is staged ONCE in LDS
//     (with the producer's InstanceNorm + ReLU applied on the way in, so normalised tensors are never materialised),
//     then 9 taps x Cin/32 chunks of v_mfma_f32_16x16x32_f16 run from LDS; weights stream from L1/L2;
//   * the epilogue adds the bias, rounds to f16 exactly where autocast rounds, stores NHWC and emits per-workgroup
//     (sum, sum^2) partials per channel; the CONSUMER reduces them in a fixed order (deterministic InstanceNorm);
//   * fnet (InstanceNorm) and inet (no norm) share every launch (grid.z = encoder);
//   * a residual block's output relu(fx(x) + relu(fy(y))) (extractor.py:44-55) is not a launch of its own: the convolution that
//     consumes it forms it while staging its halo (from the two raw tensors and their statistics) and writes the interior of
//     its tile back for the later readers; the stride-2 1x1 "downsample" convolution of layer2.0 is the centre tap of that
//     block's first 3x3 stride-2 convolution and runs in the same launch: 10 launches per frame instead of 15.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;           // output tile (pixels)
constexpr int ENC_TH2 = 4, ENC_WN2 = 2, ENC_WN1 = 2, ENC_PD = 5;      // quarter-resolution tile height, wave tilings, weight prefetch depth (round 3)
constexpr float kInEps = 1e-5f;          // nn.InstanceNorm2d default eps
constexpr int kStatCopies = 16;          // accumulator copies per statistics set: spreads the same-address f64 atomics
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

struct EncPtrs {                         // one per encoder (z = 0 fnet, z = 1 inet)
  const _Float16* in;                    // input activation NHWC (raw conv output of the producer, or materialised)
  const double* in_part;                 // producer's per-channel (sum, sum^2) accumulators [2][64] or null
  const _Float16* w;                     // weights [K/32][Cout][32] (k-step major: fragment loads are 1 KB contiguous)
  const _Float16* bias;                  // [Cout]
  _Float16* out;                         // NHWC raw conv output (bias added, f16)
  double* out_part;                      // this conv's (sum, sum^2) accumulators [2][64] (zeroed by the host) or null
  int in_mode;                           // 0 identity, 1 relu, 2 instance-norm + relu, 3 residual-block output of (in, in2)
  int cout;                              // number of output channels of THIS encoder for this layer
  float out_scale;                       // multiplies the rounded f16 output (the "/ 4.0" of net.py:116-117)
  // in_mode 3: the input is relu(fx(in) + relu(fy(in2))) (extractor.py:44-55), formed while the halo is staged
  const _Float16* in2; const double* in2_part;
  int x_mode, y_mode;                    // x: 0 identity, 1 relu, 2 instance-norm, 3 instance-norm + relu; y: 1 relu, 3 instance-norm + relu
  _Float16* mat_out;                     // the interior of the tile of that tensor is written here (its later readers), or null
  // DS kernels: a second, 1x1 convolution of the same input at the same stride (the centre tap): layer2.0.downsample
  const _Float16* w2; const _Float16* bias2; _Float16* out2; double* out2_part;
};
__host__ __device__ inline EncPtrs enc_ptrs(const _Float16* in, const double* in_part, const _Float16* w, const _Float16* bias, _Float16* out,
                                            double* out_part, int in_mode, int cout, float out_scale) {
  EncPtrs p{};
  p.in = in; p.in_part = in_part; p.w = w; p.bias = bias; p.out = out; p.out_part = out_part; p.in_mode = in_mode; p.cout = cout;
  p.out_scale = out_scale;
  return p;
}
struct EncArgs { EncPtrs e[2]; };
#ifdef ENC_TRACE
__device__ unsigned long long g_enc_trace[2048][8];
#ifndef ENC_TRACE_CIN
#define ENC_TRACE_CIN 64
#define ENC_TRACE_KS 3
#endif
#ifndef ENC_TRACE_Z
#define ENC_TRACE_Z 0
#endif
#define ENC_T(i) do { if (CIN == ENC_TRACE_CIN && KS == ENC_TRACE_KS && threadIdx.x == 0 && blockIdx.z == ENC_TRACE_Z && blockIdx.y == 0) g_enc_trace[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define ENC_T(i) do {} while (0)
#endif

__device__ __forceinline__ int swz4(int px) { return (0x78 >> (2 * ((px >> 2) & 3))) & 3; }   // 64 B pixels (Cin 32)

// mean / rstd of the producer's channels from its f64 (sum, sum^2) accumulators [kStatCopies][2][64].  Producers add one
// f64 atomic per channel per workgroup (hardware global_atomic_add_f64) into copy blockIdx.x % kStatCopies -- a few
// hundred workgroups hitting the same 128 addresses serialise in L2 otherwise; summation order effects are ~1e-16
// relative, far below the f32 statistics derived here.
// (two halves: the loads are issued before the halo loads of the workgroup -- they come back first -- and the arithmetic runs
//  while the halo loads are in flight.  Two lanes per channel, 8 copies each: 32 registers instead of 64 across the halo loads.)
constexpr int kStatHalf = kStatCopies / 2;
struct StatRaw { double v1[kStatHalf], v2[kStatHalf]; };
// t: index inside the set's 2 * CIN threads (channel t >> 1, half t & 1)
__device__ __forceinline__ void stats_load(const double* acc, int t, StatRaw& r) {
  const int c = t >> 1, h = t & 1;
#pragma unroll
  for (int k = 0; k < kStatHalf; ++k) { r.v1[k] = acc[(h * kStatHalf + k) * 128 + c]; r.v2[k] = acc[(h * kStatHalf + k) * 128 + 64 + c]; }
}
__device__ __forceinline__ void stats_finish(const StatRaw& r, int t, float inv_n, float* s_mean, float* s_rstd) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < kStatHalf; ++k) { s1 += r.v1[k]; s2 += r.v2[k]; }
  // copies 0..7 + copies 8..15 (lane pair 2c, 2c + 1)
  const double o1 = __shfl_xor(s1, 1), o2 = __shfl_xor(s2, 1);
  if ((t & 1) == 0) {
    s1 += o1; s2 += o2;
    const double mean = s1 * (double)inv_n;
    const double var = s2 * (double)inv_n - mean * mean;
    s_mean[t >> 1] = (float)mean;
    s_rstd[t >> 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)kInEps));
  }
}

// ---------------------------------------------------------------------------------------------------
// conv KSxKS (KS = 3 pad 1, or KS = 1 pad 0), stride S, Cin in {32,64}, 64 output channels per workgroup (blockIdx.y)
// ---------------------------------------------------------------------------------------------------
// THT: rows of the output tile, 8 (wave w owns rows 2w, 2w + 1) or 4 (wave w owns row w).  The quarter-resolution layers have only
// 75 tiles of 8 x 32 per tower -- 150 workgroups on 256 CUs, each a ~15 us dependent chain; 4 x 32 tiles make it 300 shorter ones.
template <int CIN, int KS, int S, int NT, bool DS = false, int THT = TH, int WN = 1>
__global__ __launch_bounds__(256) void conv_kernel(EncArgs args, int Hin, int Win, int Hout, int Wout, int n_part_in) {
  constexpr int PAD = KS / 2;
  // the four waves tile the workgroup's [THT rows] x [16 NT channels] output as WM x WN: wave (wm, wc) owns rows [wm RPW, +RPW) and
  // N-tiles [wc NTW, +NTW).  WN = 2 halves the weight fragments every wave streams from L2 (each wave would otherwise load the whole
  // filter bank: the k-loop of the quarter-resolution layers was bound by that stream) for twice the LDS reads per MFMA.
  constexpr int WM = 4 / WN, NTW = NT / WN, RPW = THT / WM;
  constexpr int MT = RPW * 2;                               // 16-pixel M-tiles per wave
  static_assert(WN == 1 || WN == 2, "waves along N"); static_assert(THT % WM == 0 && NT % WN == 0, "tile split");
  constexpr int IH = (THT - 1) * S + KS, IW = (TW - 1) * S + KS;
  constexpr int CPP = CIN / 8;                              // 16-byte chunks per pixel
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* halo = reinterpret_cast<_Float16*>(smem_raw);   // [IH][IW][CIN] swizzled
  float* s_mean = reinterpret_cast<float*>(smem_raw + (size_t)IH * IW * CIN * 2);
  float* s_rstd = s_mean + CIN;
  float* s_mean2 = s_rstd + CIN;                            // (residual input: statistics of the second operand)
  float* s_rstd2 = s_mean2 + CIN;
  float* s_red = s_rstd2 + CIN;                             // [4 waves][2][64]

  const EncPtrs P = args.e[blockIdx.z];
  const int n0 = blockIdx.y * (16 * NT);
  if (n0 >= P.cout) return;
  const int tiles_x = (Wout + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * THT, ox0 = tx * TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wc = wave % WN;
  const int n0w = n0 + wc * NTW * 16;                       // first output channel of this wave

  ENC_T(0);
  // the epilogue's bias fragments are requested now: loaded where they are used they were an exposed round trip (~1.5 us of a 17 us
  // workgroup) in front of the output stores
  // (LEAN: the 32-channel stride-1 kernel runs 600 workgroups and must stay at three per CU, i.e. <= 112 VGPRs: it keeps the
  //  statistics in front of the halo loads and fetches its bias in the epilogue; early requests cost it 9-40 registers and its
  //  third workgroup: 16.6 -> 18.0 us)
  constexpr bool HALF = (CIN == 32 && KS == 3 && S == 1);       // the half-resolution 3x3 layers: 600 workgroups, three per CU wanted
  constexpr bool LEAN = HALF && WN == 1;
  constexpr bool EARLY_STATS = !HALF;
  h4 bias_pf[NTW];
  auto bias_load = [&]() {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = n0w + j * 16 + (lane >> 4) * 4;
      bias_pf[j] = (n < P.cout) ? *reinterpret_cast<const h4*>(P.bias + n) : (h4)(_Float16)0;
    }
  };
  if constexpr (!LEAN) bias_load();
  // the producers' statistics: threads [0, 2 CIN) request set 1, [2 CIN, 4 CIN) set 2 (whole waves either way), BEFORE the halo
  // loads; the arithmetic follows once the halo loads have been issued
  const bool st1 = P.in_mode == 2 || (P.in_mode == 3 && P.x_mode >= 2), st2 = P.in_mode == 3 && P.y_mode >= 2;
  const bool need_stats = st1 || st2;
  const bool my1 = st1 && tid < 2 * CIN, my2 = st2 && tid >= 2 * CIN && tid < 4 * CIN;
  StatRaw sraw;
  if (my1) stats_load(P.in_part, tid, sraw);
  if (my2) stats_load(P.in2_part, tid - 2 * CIN, sraw);
  auto stats = [&]() {
    if (my1) stats_finish(sraw, tid, 1.0f / (float)(Hin * Win), s_mean, s_rstd);
    if (my2) stats_finish(sraw, tid - 2 * CIN, 1.0f / (float)(Hin * Win), s_mean2, s_rstd2);
    if (need_stats) __syncthreads();
  };
  if constexpr (!EARLY_STATS) { stats(); ENC_T(1); }

  // ---- stage the input halo (producer's norm + relu applied here; zero padding applies to the transformed tensor)
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
  {
    // all global loads of this thread first (they are independent), then the transform + LDS writes: the naive
    // load -> normalise -> store loop is a chain of ~1 us round trips (measured: 4.7 us of a 23 us workgroup)
    constexpr int NCHUNK = IH * IW * CPP, NPT = (NCHUNK + 255) / 256;
    // (the stride-2 kernel's 17 x 65 halo is 18 chunks per thread: it only exists for residual inputs, which are staged in
    //  batches -- holding 18 chunks of a single input in registers would cost it its second workgroup per CU)
    constexpr bool DUAL_ONLY = (S == 2 && KS == 3);
    h8 pf[DUAL_ONLY ? 1 : NPT];
    unsigned inimg = 0;
    const bool dual = DUAL_ONLY || P.in_mode == 3;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = tid + 256 * k;
      const int ch = q % CPP, pix = q / CPP;
      const int ly = pix / IW, lx = pix - ly * IW;
      const int y = iy0 + ly, x = ix0 + lx;
      if constexpr (!DUAL_ONLY) pf[k] = (h8)(_Float16)0;
      if (q < NCHUNK && y >= 0 && y < Hin && x >= 0 && x < Win) {
        if constexpr (!DUAL_ONLY) { if (!dual) pf[k] = *reinterpret_cast<const h8*>(P.in + ((int64_t)y * Win + x) * CIN + ch * 8); }
        inimg |= 1u << k;
      }
    }
    // the producer's statistics are reduced AFTER this thread's halo loads have been issued: two independent round trips side
    // by side instead of one after the other
    if constexpr (EARLY_STATS) { if (!dual) { stats(); ENC_T(1); } }
    if (dual) {
      // relu(fx(x) + relu(fy(y))): the arithmetic and its f16 rounding points are those of the reference's f16 tensors
      // (extractor.py:44-55).  Both operands of a batch of KB chunks are requested before anything is consumed; batches keep
      // the staging registers at 2 x KB x 4 (the 17 x 65 halo of the stride-2 kernel is 18 chunks per thread)
      constexpr int KB = (NPT <= 8) ? NPT : 6;          // (the 1x1 layer's 8 chunks per thread go in one batch: one round trip)
#pragma unroll
      for (int k0 = 0; k0 < NPT; k0 += KB) {
        h8 px[KB], pg[KB];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
          const int k = k0 + kk;
          if (k >= NPT) break;
          const int q = tid + 256 * k;
          const int ch = q % CPP, pix = q / CPP;
          const int ly = pix / IW, lx = pix - ly * IW;
          px[kk] = pg[kk] = (h8)(_Float16)0;
          if ((inimg >> k) & 1) {
            const int64_t o = ((int64_t)(iy0 + ly) * Win + (ix0 + lx)) * CIN + ch * 8;
            px[kk] = *reinterpret_cast<const h8*>(P.in + o);
            pg[kk] = *reinterpret_cast<const h8*>(P.in2 + o);
          }
        }
        if constexpr (EARLY_STATS) { if (k0 == 0) { stats(); ENC_T(1); } }
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
          const int k = k0 + kk;
          if (k >= NPT) break;
          const int q = tid + 256 * k;
          if (q < NCHUNK) {
            const int ch = q % CPP, pix = q / CPP;
            const int ly = pix / IW, lx = pix - ly * IW;
            h8 v = (h8)(_Float16)0;
            if ((inimg >> k) & 1) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int c = ch * 8 + e;
                _Float16 a = px[kk][e], b = pg[kk][e];
                if (P.x_mode >= 2) a = (_Float16)(((float)a - s_mean[c]) * s_rstd[c]);
                if (P.x_mode & 1) a = a > (_Float16)0 ? a : (_Float16)0;
                if (P.y_mode >= 2) b = (_Float16)(((float)b - s_mean2[c]) * s_rstd2[c]);
                b = b > (_Float16)0 ? b : (_Float16)0;
                const _Float16 sum = a + b;                               // f16 add (x + y on f16 tensors)
                v[e] = sum > (_Float16)0 ? sum : (_Float16)0;
              }
              // the tile's own pixels of the block output go back to memory for its other readers (the skip connection of
              // the next block, the second convolution of a stride-2 block): every pixel is the interior of exactly one tile
              if (P.mat_out && blockIdx.y == 0 && ly >= PAD && ly < PAD + THT * S && lx >= PAD && lx < PAD + TW * S)
                *reinterpret_cast<h8*>(P.mat_out + ((int64_t)(iy0 + ly) * Win + (ix0 + lx)) * CIN + ch * 8) = v;
            }
            const int sch = (CIN == 32) ? (ch ^ swz4(lx)) : (ch ^ (lx & 7));
            *reinterpret_cast<h8*>(halo + ((int64_t)pix * CPP + sch) * 8) = v;
          }
        }
      }
    } else if constexpr (!DUAL_ONLY) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = tid + 256 * k;
      if (q < NCHUNK) {
        const int ch = q % CPP, pix = q / CPP;
        const int lx = pix % IW;
        h8 v = pf[k];
        if ((inimg >> k) & 1) {
          if (P.in_mode == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const _Float16 nv = (_Float16)(((float)v[e] - s_mean[ch * 8 + e]) * s_rstd[ch * 8 + e]);   // IN output is f16
              v[e] = nv > (_Float16)0 ? nv : (_Float16)0;
            }
          } else if (P.in_mode == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > (_Float16)0 ? v[e] : (_Float16)0;
          }
        }
        const int sch = (CIN == 32) ? (ch ^ swz4(lx)) : (ch ^ (lx & 7));
        *reinterpret_cast<h8*>(halo + ((int64_t)pix * CPP + sch) * 8) = v;
      }
    }
    }
  }
  __syncthreads();

  ENC_T(2);
  // ---- implicit GEMM: wave w owns output rows 2w, 2w+1 (4 M-tiles of 16 pixels), 4 N-tiles (64 channels)
  f4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f4)0.f;
  // M-tile i of this wave: output row / 16-pixel half of the tile
  auto row_of = [&](int i) { return wm * RPW + (i >> 1); };
  const int m = lane & 15, kg = lane >> 4;
  const int ncout = P.cout;
  static_assert(!DS || (KS == 3 && CIN == 32), "centre-tap second convolution: 3x3, one 32-channel k-step");
  constexpr int K = KS * KS * CIN;
  // flat k-steps t = (kh*KS + kw)*(CIN/32) + kc; the filter fragments of step t+2 are fetched (L2) while step t runs:
  // un-prefetched they were a dependent ~0.5 us round trip per step (9 us of a 23 us workgroup for 2 us of MFMA)
  constexpr int KC = CIN / 32, T = KS * KS * KC, PD = HALF ? 2 : ENC_PD;       // (the non-LEAN kernels run one workgroup per CU: registers to spare)
  constexpr int RING = (PD + 1 < T) ? PD + 1 : T;
  h8 fwr[RING][NTW];
  auto wload = [&](int t, h8 (&dst)[NTW]) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = n0w + j * 16 + m;
      dst[j] = (n < ncout) ? *reinterpret_cast<const h8*>(P.w + ((int64_t)t * ncout + n) * 32 + kg * 8) : (h8)(_Float16)0;
    }
  };
#pragma unroll
  for (int t = 0; t < PD && t < T; ++t) wload(t, fwr[t]);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + PD < T) wload(t + PD, fwr[(t + PD) % RING]);
    const int kc = t % KC, kw = (t / KC) % KS, kh = t / (KC * KS);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ly = row_of(i) * S + kh, lx = ((i & 1) * 16 + m) * S + kw;
      const int ch = kc * 4 + kg;
      const int sch = (CIN == 32) ? (ch ^ swz4(lx)) : (ch ^ (lx & 7));
      const h8 fa = *reinterpret_cast<const h8*>(halo + ((int64_t)(ly * IW + lx) * CPP + sch) * 8);
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fwr[t % RING][j], fa, acc[i][j], 0, 0, 0);
    }
  }

  ENC_T(3);
  // ---- epilogue: bias, f16 rounding, NHWC store, per-channel partial statistics of the ROUNDED values
  auto epilogue = [&](auto& A, const h4 (&bias4)[NTW], _Float16* out, double* out_part) {
  float ssum[NTW][4], ssq[NTW][4];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int oy = oy0 + row_of(i), ox = ox0 + (i & 1) * 16 + m;
    const bool inb = oy < Hout && ox < Wout;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = n0w + j * 16 + kg * 4;
      if (n >= ncout) continue;
      const h4 bv = bias4[j];
      h4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hv[r] = (_Float16)(A[i][j][r] + (float)bv[r]);
        if (inb) { const float f = (float)hv[r]; ssum[j][r] += f; ssq[j][r] += f * f; }
        hv[r] = (_Float16)((float)hv[r] * P.out_scale);
      }
      if (inb) *reinterpret_cast<h4*>(out + ((int64_t)oy * Wout + ox) * ncout + n) = hv;
    }
  }
  ENC_T(4);
  if (out_part) {
    // reduce over the 16 pixel lanes (xor 1,2,4,8 keeps kg), then over the 4 waves through LDS, fixed order
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = ssum[j][r], b = ssq[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        const int cr = (wc * NTW + j) * 16 + kg * 4 + r;              // channel relative to n0
        if (m == 0) { s_red[(wm * 2 + 0) * 64 + cr] = a; s_red[(wm * 2 + 1) * 64 + cr] = b; }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      if (c < 16 * NT && n0 + c < ncout) {
        float v = s_red[(0 * 2 + which) * 64 + c] + s_red[(1 * 2 + which) * 64 + c];
        if constexpr (WM == 4) v = v + (s_red[(2 * 2 + which) * 64 + c] + s_red[(3 * 2 + which) * 64 + c]);
        unsafeAtomicAdd(&out_part[(blockIdx.x % kStatCopies) * 128 + which * 64 + n0 + c], (double)v);
      }
    }
    __syncthreads();                        // (s_red is reused by a second epilogue)
  }
  };
  if constexpr (LEAN) bias_load();
  epilogue(acc, bias_pf, P.out, P.out_part);
  if constexpr (DS) {
    // the second, 1x1 convolution (layer2.0.downsample) = one k-step on the centre tap of the halo that is still in LDS; the
    // accumulators are reused (carrying both sets through the main loop cost the kernel its second workgroup per CU)
    h8 fw2[NTW];
    h4 bias2_pf[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int n = n0w + j * 16 + m, nb = n0w + j * 16 + kg * 4;
      fw2[j] = (n < ncout) ? *reinterpret_cast<const h8*>(P.w2 + (int64_t)n * 32 + kg * 8) : (h8)(_Float16)0;
      bias2_pf[j] = (nb < ncout) ? *reinterpret_cast<const h4*>(P.bias2 + nb) : (h4)(_Float16)0;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ly = row_of(i) * S + KS / 2, lx = ((i & 1) * 16 + m) * S + KS / 2;
      const int sch = kg ^ swz4(lx);
      const h8 fa = *reinterpret_cast<const h8*>(halo + ((int64_t)(ly * IW + lx) * CPP + sch) * 8);
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw2[j], fa, (f4)0.f, 0, 0, 0);
    }
    epilogue(acc, bias2_pf, P.out2, P.out2_part);
  }
  ENC_T(5);
}

// ---------------------------------------------------------------------------------------------------
// conv1: 7x7 stride 2 pad 3, 3 -> 32 channels, planar f16 input image [3][H][W] (shared by both encoders).
// K is ordered (kh, c, kw) with kw padded 7 -> 8 so that one k-group of 8 = one input row segment of 8 contiguous
// pixels (stride-1 in x for a fixed output pixel): K = 21 groups -> padded to 24 groups = 6 chunks of 32.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1_kernel(const _Float16* __restrict__ img, EncArgs args, int H, int W, int Hout,
                                                    int Wout) {
  constexpr int IH = (TH - 1) * 2 + 7, IW = (TW - 1) * 2 + 7 + 1, IWP = 72;     // 21 x 70 (+pad) per channel
  __shared__ __attribute__((aligned(16))) _Float16 halo[3 * IH * IWP];
  __shared__ float s_red[4 * 2 * 32];
  const EncPtrs P = args.e[blockIdx.z];
  const int tiles_x = (Wout + TW - 1) / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  {
    // independent loads first, LDS writes after (see conv_kernel)
    constexpr int NEL = 3 * IH * IWP, NPT = (NEL + 255) / 256;
    _Float16 pf[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = tid + 256 * k;
      const int lx = q % IWP, ly = (q / IWP) % IH, c = q / (IWP * IH);
      const int y = iy0 + ly, x = ix0 + lx;
      pf[k] = (_Float16)0;
      if (q < NEL && lx < IW && y >= 0 && y < H && x >= 0 && x < W) pf[k] = img[((int64_t)c * H + y) * W + x];
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int q = tid + 256 * k;
      if (q < NEL) halo[q] = pf[k];
    }
  }
  __syncthreads();
  f4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { acc[i][0] = (f4)0.f; acc[i][1] = (f4)0.f; }
  const int m = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int g = q * 4 + kg;                         // k-group: (kh, c) = (g / 3, g % 3), valid for g < 21
    const int kh = g / 3, c = g - 3 * kh;
    h8 fw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const h8*>(P.w + (int64_t)(j * 16 + m) * 192 + g * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h8 fa = (h8)(_Float16)0;
      if (g < 21) {
        const int ly = (2 * wave + (i >> 1)) * 2 + kh, lx = ((i & 1) * 16 + m) * 2;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(halo + (c * IH + ly) * IWP + lx);   // 4-byte aligned
        const u4v u = {p[0], p[1], p[2], p[3]};
        fa = __builtin_bit_cast(h8, u);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa, acc[i][j], 0, 0, 0);
    }
  }
  float ssum[2][4], ssq[2][4];
  h4 bias4[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bias4[j] = *reinterpret_cast<const h4*>(P.bias + j * 16 + kg * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[j][r] = 0.f; ssq[j][r] = 0.f; }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = oy0 + 2 * wave + (i >> 1), ox = ox0 + (i & 1) * 16 + m;
    const bool inb = oy < Hout && ox < Wout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = j * 16 + kg * 4;
      const h4 bv = bias4[j];
      h4 hv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hv[r] = (_Float16)(acc[i][j][r] + (float)bv[r]);
        if (inb) { const float f = (float)hv[r]; ssum[j][r] += f; ssq[j][r] += f * f; }
      }
      if (inb) *reinterpret_cast<h4*>(P.out + ((int64_t)oy * Wout + ox) * 32 + n) = hv;
    }
  }
  if (P.out_part) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = ssum[j][r], b = ssq[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (m == 0) { s_red[(wave * 2 + 0) * 32 + j * 16 + kg * 4 + r] = a; s_red[(wave * 2 + 1) * 32 + j * 16 + kg * 4 + r] = b; }
      }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      unsafeAtomicAdd(&P.out_part[(blockIdx.x % kStatCopies) * 128 + which * 64 + c],
                      (double)((s_red[(0 * 2 + which) * 32 + c] + s_red[(1 * 2 + which) * 32 + c]) +
                               (s_red[(2 * 2 + which) * 32 + c] + s_red[(3 * 2 + which) * 32 + c])));
    }
  }
}

template <int CIN, int KS, int S, int NT, bool DS = false, int THT = TH, int WN = 1>
int launch_conv(const EncArgs& a, int Hin, int Win, int Hout, int Wout, int n_part_in, int cout_max, hipStream_t st) {
  constexpr int IH = (THT - 1) * S + KS, IW = (TW - 1) * S + KS;
  const size_t sh = (size_t)IH * IW * CIN * 2 + (4 * CIN + 4 * 2 * 64) * 4;
  (void)hipFuncSetAttribute((const void*)conv_kernel<CIN, KS, S, NT, DS, THT, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  const int tiles = ((Hout + THT - 1) / THT) * ((Wout + TW - 1) / TW);
  hipLaunchKernelGGL((conv_kernel<CIN, KS, S, NT, DS, THT, WN>), dim3(tiles, (cout_max + 16 * NT - 1) / (16 * NT), 2), dim3(256), sh, st, a, Hin, Win,
                     Hout, Wout, n_part_in);
  return (int)hipGetLastError();
}

// 4x4 average pooling of an NHWC map (dpvo.py:438: fmap2_ = F.avg_pool2d(fmap, 4, 4)), f32 accumulate, one rounding
__global__ void pool4_nhwc_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int h, int w, int C) {
  pool4_nhwc_body(in, out, h, w, C, blockIdx.x, gridDim.x);
}

}  // namespace

// Weight / bias pointer table, per encoder (fnet then inet), in this order (all f16, repacked by the host):
//   0 conv1 w[32][192]  1 conv1 b     (K = (kh, c, kw padded to 8), zero padded to 192)
//   then for each 3x3 / 1x1 conv  w[K/32][Cout][32] with K = KS*KS*Cin ordered (kh, kw, cin) (k-step major), b[Cout]:
//   2,3   layer1.0.conv1 (32->32)     4,5   layer1.0.conv2      6,7   layer1.1.conv1     8,9   layer1.1.conv2
//   10,11 layer2.0.conv1 (32->64 s2)  12,13 layer2.0.conv2 (64) 14,15 layer2.0.downsample.0 (1x1 s2 32->64)
//   16,17 layer2.1.conv1 (64->64)     18,19 layer2.1.conv2      20,21 conv2 (1x1, 64 -> 128 | 384)
static inline size_t enc_al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t dpvo_encoders_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0 || (H % 16) || (W % 16)) return 0;
  const size_t h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4;
  const size_t a32 = enc_al(h2 * w2 * 32 * 2), a64 = enc_al(h4 * w4 * 64 * 2);
  const size_t t2 = ((h2 + TH - 1) / TH) * ((w2 + TW - 1) / TW);
  return 2 * (4 * a32 + 4 * a64) + enc_al((size_t)2 * 10 * kStatCopies * 128 * 8) + 4096;
}

// fmap_out [H/4][W/4][128], imap_out [H/4][W/4][384] f16 NHWC, both already divided by 4 (net.py:116-117).
extern "C" int dpvo_encoders_forward(const void* image_f16, const void* const* weights, void* fmap_out, void* imap_out,
                                     int H, int W, void* ws, size_t ws_bytes, void* stream) {
  return dpvo_encoders_forward_hold(image_f16, weights, fmap_out, imap_out, H, W, ws, ws_bytes, nullptr, 0, stream);
}

// The same, with the stream made to wait for `hold_event` (hipEvent_t, may be NULL) in front of launch number `hold_at`
// (0 = the first convolution ... 9 = the last): a tracker that runs the next frame's encoders on a side stream lets the first
// launches run beside the chip-filling kernels of the current frame and holds the rest back until those are through.
// (Measured alternatives to the event, round 3: hipStreamWaitValue32 on signal memory slows the OTHER queue's kernels by 15 %
//  while it polls; a one-lane kernel parked on a flag keeps one CU from hosting the update operator's 512-register workgroups --
//  K7's 256 persistent workgroups then need a second round.  The event's own cost: while the wait is pending, a thread of the HIP
//  runtime burns CPU, 0.7 ms per frame if the encoders are enqueued a whole update operator ahead.)
extern "C" int dpvo_encoders_forward_hold(const void* image_f16, const void* const* weights, void* fmap_out, void* imap_out,
                                          int H, int W, void* ws, size_t ws_bytes, void* hold_event, int hold_at, void* stream) {
  if (!image_f16 || !weights || !fmap_out || !imap_out || !ws) return DPVO_E_INVALID;
  if (H <= 0 || W <= 0 || (H % 16) || (W % 16)) return DPVO_E_UNSUPPORTED;
  if (ws_bytes < dpvo_encoders_workspace_bytes(H, W)) return DPVO_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4;
  const size_t a32 = enc_al((size_t)h2 * w2 * 32 * 2), a64 = enc_al((size_t)h4 * w4 * 64 * 2);
  const int t2 = ((h2 + TH - 1) / TH) * ((w2 + TW - 1) / TW), t4 = ((h4 + TH - 1) / TH) * ((w4 + TW - 1) / TW);
  char* base = (char*)ws;
  _Float16 *A[2][4], *B[2][4];
  for (int z = 0; z < 2; ++z) {
    for (int i = 0; i < 4; ++i) { A[z][i] = (_Float16*)base; base += a32; }
    for (int i = 0; i < 4; ++i) { B[z][i] = (_Float16*)base; base += a64; }
  }
  // ten (sum, sum^2) accumulator sets per tower, one per statistics-producing conv, zeroed once per forward
  double* Sacc = (double*)base;
  {
    hipError_t e = hipMemsetAsync(Sacc, 0, (size_t)2 * 10 * kStatCopies * 128 * 8, st);
    if (e != hipSuccess) return (int)e;
  }
  double* Pt[2][10];
  for (int z = 0; z < 2; ++z)
    for (int i = 0; i < 10; ++i) Pt[z][i] = Sacc + ((size_t)z * 10 + i) * kStatCopies * 128;
  auto Wp = [&](int z, int i) { return (const _Float16*)weights[z * 22 + i]; };
  const bool nm[2] = {true, false};                       // fnet: instance norm, inet: none (net.py:98-99)
  int rc;
  EncArgs a;
  const int cin_mode[2] = {2, 1};                         // consumer-side transform of a raw conv output: IN+relu | relu
  int launch_no = 0;
  auto hold = [&]() -> int {
    if (hold_event && launch_no++ == hold_at && hipStreamWaitEvent(st, (hipEvent_t)hold_event, 0) != hipSuccess) return DPVO_E_INVALID;
    return 0;
  };
#define HOLD() do { if ((rc = hold())) return rc; } while (0)
#define STATS(z, i) (nm[z] ? Pt[z][i] : nullptr)

  // Residual-block outputs are formed by their first consumer (in_mode 3) and written back by it (mat_out):
  //   X1 = relu(f(A0) + relu(f(A2)))  by layer1.1.conv1 -> A1      X2 = relu(X1 + relu(f(A0')))  by layer2.0.conv1 (+ downsample) -> A2
  //   X3 = relu(f(B2) + relu(f(B1)))  by layer2.1.conv1 -> B3      X4 = relu(X3 + relu(f(B1')))  by conv2 (not stored)
  auto res = [&](EncPtrs p, const _Float16* y, const double* y_part, int x_mode, int y_mode, _Float16* mat) {
    p.in_mode = 3; p.in2 = y; p.in2_part = y_part; p.x_mode = x_mode; p.y_mode = y_mode; p.mat_out = mat;
    return p;
  };
  // conv1 (7x7 s2) -> A0 raw, stats P0; x0 = relu(norm1(A0)) is applied on the fly by its consumers      :252-254
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(nullptr, nullptr, Wp(z, 0), Wp(z, 1), A[z][0], STATS(z, 0), 0, 32, 1.0f);
  HOLD();
  hipLaunchKernelGGL(conv1_kernel, dim3(t2, 1, 2), dim3(256), 0, st, (const _Float16*)image_f16, a, H, W, h2, w2);

  // ---- layer1.0 (ResidualBlock 32->32, :44-55): c1: x0 -> A1 (P1), c2 -> A2 (P2)
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(A[z][0], STATS(z, 0), Wp(z, 2), Wp(z, 3), A[z][1], STATS(z, 1), cin_mode[z], 32, 1.0f);
  HOLD();
  if ((rc = launch_conv<32, 3, 1, 2, false, TH, ENC_WN1>(a, h2, w2, h2, w2, t2, 32, st))) return rc;
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(A[z][1], STATS(z, 1), Wp(z, 4), Wp(z, 5), A[z][2], STATS(z, 2), cin_mode[z], 32, 1.0f);
  HOLD();
  if ((rc = launch_conv<32, 3, 1, 2, false, TH, ENC_WN1>(a, h2, w2, h2, w2, t2, 32, st))) return rc;
  // ---- layer1.1: input X1 = relu(x0 + relu(f(A2))) formed here and written to A1; c1 -> A3 (P3), c2 -> A0 (P4)
  for (int z = 0; z < 2; ++z)
    a.e[z] = res(enc_ptrs(A[z][0], STATS(z, 0), Wp(z, 6), Wp(z, 7), A[z][3], STATS(z, 3), 3, 32, 1.0f), A[z][2], STATS(z, 2),
                 nm[z] ? 3 : 1, nm[z] ? 3 : 1, A[z][1]);
  HOLD();
  if ((rc = launch_conv<32, 3, 1, 2, false, TH, ENC_WN1>(a, h2, w2, h2, w2, t2, 32, st))) return rc;
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(A[z][3], STATS(z, 3), Wp(z, 8), Wp(z, 9), A[z][0], STATS(z, 4), cin_mode[z], 32, 1.0f);
  HOLD();
  if ((rc = launch_conv<32, 3, 1, 2, false, TH, ENC_WN1>(a, h2, w2, h2, w2, t2, 32, st))) return rc;
  // ---- layer2.0 (32->64, stride 2): input X2 = relu(X1 + relu(f(A0))) formed here and written to A2;
  //      c1 -> B0 (P5) and, on the centre tap, downsample (1x1 stride 2) -> B2 (P7) in the same launch; c2 -> B1 (P6)
  for (int z = 0; z < 2; ++z) {
    a.e[z] = res(enc_ptrs(A[z][1], nullptr, Wp(z, 10), Wp(z, 11), B[z][0], STATS(z, 5), 3, 64, 1.0f), A[z][0], STATS(z, 4), 0,
                 nm[z] ? 3 : 1, A[z][2]);
    a.e[z].w2 = Wp(z, 14); a.e[z].bias2 = Wp(z, 15); a.e[z].out2 = B[z][2]; a.e[z].out2_part = STATS(z, 7);
  }
  HOLD();
  if ((rc = launch_conv<32, 3, 2, 4, true, ENC_TH2, ENC_WN2>(a, h2, w2, h4, w4, t2, 64, st))) return rc;
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(B[z][0], STATS(z, 5), Wp(z, 12), Wp(z, 13), B[z][1], STATS(z, 6), cin_mode[z], 64, 1.0f);
  HOLD();
  if ((rc = launch_conv<64, 3, 1, 4, false, ENC_TH2, ENC_WN2>(a, h4, w4, h4, w4, t4, 64, st))) return rc;
  // ---- layer2.1: input X3 = relu(norm(B2) + relu(f(B1))) formed here and written to B3; c1 -> B0 (P8), c2 -> B1 (P9)
  for (int z = 0; z < 2; ++z)
    a.e[z] = res(enc_ptrs(B[z][2], STATS(z, 7), Wp(z, 16), Wp(z, 17), B[z][0], STATS(z, 8), 3, 64, 1.0f), B[z][1], STATS(z, 6),
                 nm[z] ? 2 : 0, nm[z] ? 3 : 1, B[z][3]);
  HOLD();
  if ((rc = launch_conv<64, 3, 1, 4, false, ENC_TH2, ENC_WN2>(a, h4, w4, h4, w4, t4, 64, st))) return rc;
  for (int z = 0; z < 2; ++z) a.e[z] = enc_ptrs(B[z][0], STATS(z, 8), Wp(z, 18), Wp(z, 19), B[z][1], STATS(z, 9), cin_mode[z], 64, 1.0f);
  HOLD();
  if ((rc = launch_conv<64, 3, 1, 4, false, ENC_TH2, ENC_WN2>(a, h4, w4, h4, w4, t4, 64, st))) return rc;
  // ---- conv2 (1x1, 64 -> 128 | 384) on X4 = relu(X3 + relu(f(B1))), output / 4.0                        :259, net.py:116-117
  a.e[0] = res(enc_ptrs(B[0][3], nullptr, Wp(0, 20), Wp(0, 21), (_Float16*)fmap_out, nullptr, 3, 128, 0.25f), B[0][1], STATS(0, 9), 0,
               nm[0] ? 3 : 1, nullptr);
  a.e[1] = res(enc_ptrs(B[1][3], nullptr, Wp(1, 20), Wp(1, 21), (_Float16*)imap_out, nullptr, 3, 384, 0.25f), B[1][1], STATS(1, 9), 0,
               nm[1] ? 3 : 1, nullptr);
  HOLD();
  if ((rc = launch_conv<64, 1, 1, 4, false, TH, ENC_WN1>(a, h4, w4, h4, w4, t4, 384, st))) return rc;
#undef STATS
#undef HOLD
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_pool4_nhwc(const void* in, void* out, int h, int w, int C, void* stream) {
  if (!in || !out || h <= 0 || w <= 0 || (h % 4) || (w % 4) || (C % 8)) return DPVO_E_INVALID;
  const int64_t total = (int64_t)(h / 4) * (w / 4) * (C / 8);
  hipLaunchKernelGGL(pool4_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)in, (_Float16*)out, h, w, C);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
