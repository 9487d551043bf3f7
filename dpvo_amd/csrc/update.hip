// update.hip -- the recurrent update operator's device kernels for gfx950.
//
// Replaces the torch.nn / torch_scatter launches of Update.forward (reference dpvo/net.py:74-92,
// dpvo/blocks.py:15-48) under autocast (dpvo/dpvo.py:332): 21 nn.Linear layers as MFMA GEMMs with the
// bias / activation / residual / gate fused into the epilogue, LayerNorm (+ the net + inp + corr sum and
// the imap gather), the SoftAgg segmented softmax-sum, the group expand + residual, and the two heads.
//
// Precision contract (what autocast does in the reference): Linear = f16 inputs and weights,
// f32 accumulate, one rounding to f16; LayerNorm -> f32; residual adds in f32.  The scatter softmax runs in
// f32 here (the reference runs it in f16 through torch_scatter) and rounds once.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// linear_kernel: out[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)
//   128 x 128 x 32 tiles, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 f16 tiles.
//   The product is formed transposed (W rows as the MFMA A operand, activation rows as B) so that a lane
//   ends up with 4 consecutive output columns of one row: packed 8-byte f16 stores / 16-byte f32 RMW.
//   Operands are staged global -> registers -> LDS (rows padded to 80 B): this is what allows the f32 -> f16
//   conversion of `net` and the neighbour row gather (rows[] with -1 -> zero row) in the load path.
// ---------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32, LDT = 40;   // LDT: LDS row stride in halves (32 + 8 pad)

template <bool A_F32>
struct ARegs;
template <>
struct ARegs<false> { h8 v[2]; };
template <>
struct ARegs<true> { f4 v[4]; };

template <bool A_F32>
__device__ __forceinline__ void load_a(ARegs<A_F32>& r, const void* A, int64_t row, bool valid, int64_t lda, int k) {
  if constexpr (A_F32) {
    if (valid) {
      const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(A) + row * lda + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) r.v[i] = p[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) r.v[i] = (f4)0.f;
    }
  } else {
    if (valid) {
      const h8* p = reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(A) + row * lda + k);
      r.v[0] = p[0]; r.v[1] = p[1];
    } else {
      r.v[0] = (h8)(_Float16)0; r.v[1] = (h8)(_Float16)0;
    }
  }
}

template <bool A_F32>
__device__ __forceinline__ void store_a(const ARegs<A_F32>& r, _Float16* dst) {
  if constexpr (A_F32) {
    h8 o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o0[i] = (_Float16)r.v[0][i]; o0[4 + i] = (_Float16)r.v[1][i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { o1[i] = (_Float16)r.v[2][i]; o1[4 + i] = (_Float16)r.v[3][i]; }
    reinterpret_cast<h8*>(dst)[0] = o0; reinterpret_cast<h8*>(dst)[1] = o1;
  } else {
    reinterpret_cast<h8*>(dst)[0] = r.v[0]; reinterpret_cast<h8*>(dst)[1] = r.v[1];
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }


// Shared epilogue: acc[i][j][r] = C[m = m0 + wm*64 + i*16 + (lane&15)][n = n0 + wn*64 + j*16 + (lane>>4)*4 + r]
__device__ __forceinline__ void linear_epilogue(const f4 (&acc)[4][4], const _Float16* __restrict__ bias,
                                                void* __restrict__ out, int64_t ldo, const _Float16* __restrict__ gate,
                                                int64_t ldg, _Float16* __restrict__ out16, int64_t ld16, int epilogue,
                                                int n_split, int64_t M, int N, int64_t m0, int n0, int wm, int wn,
                                                int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= N) continue;
      h4 hv;
      const h4 bv = bias ? *reinterpret_cast<const h4*>(bias + n) : (h4)(_Float16)0;
#pragma unroll
      for (int r = 0; r < 4; ++r) hv[r] = (_Float16)(acc[i][j][r] + (float)bv[r]);
      if (epilogue == DPVO_EPI_RELU || (epilogue == DPVO_EPI_RELU_SIG && n < n_split)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = hv[r] > (_Float16)0 ? hv[r] : (_Float16)0;
      } else if (epilogue == DPVO_EPI_SIGMOID || epilogue == DPVO_EPI_RELU_SIG) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (_Float16)sigmoidf_((float)hv[r]);
      }
      if (epilogue == DPVO_EPI_RESADD || epilogue == DPVO_EPI_GATED) {
        if (epilogue == DPVO_EPI_GATED) {
          const h4 g = *reinterpret_cast<const h4*>(gate + m * ldg + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = g[r] * hv[r];          // half * half -> half (blocks.py:28-29)
        }
        f4* dst = reinterpret_cast<f4*>(reinterpret_cast<float*>(out) + m * ldo + n);
        f4 o = *dst;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += (float)hv[r];
        *dst = o;
        if (out16) {                                                   // f16 image of the updated row for the next GEMM
          h4 o16;
#pragma unroll
          for (int r = 0; r < 4; ++r) o16[r] = (_Float16)o[r];
          *reinterpret_cast<h4*>(out16 + m * ld16 + n) = o16;
        }
      } else {
        *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(out) + m * ldo + n) = hv;
      }
    }
  }
}

template <bool A_F32>
__global__ __launch_bounds__(256) void linear_kernel(const void* __restrict__ A, int64_t lda,
                                                     const int32_t* __restrict__ rows,
                                                     const _Float16* __restrict__ W, int64_t ldw,
                                                     const _Float16* __restrict__ bias, void* __restrict__ out,
                                                     int64_t ldo, const _Float16* __restrict__ gate, int64_t ldg,
                                                     int epilogue, int n_split, int64_t M, int N, int K) {
  __shared__ __attribute__((aligned(16))) _Float16 As[2][BM * LDT];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN * LDT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, a speed hint only), so the N-tiles
  // of one M-tile are given consecutive slots of ONE XCD and its private L2 serves the re-reads of the A rows.
  const int ntn = (N + BN - 1) / BN;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int64_t mt = (int64_t)(idx / ntn) * 8 + xcd;
  const int64_t m0 = mt * BM;
  const int n0 = (idx % ntn) * BN;
  if (m0 >= M) return;

  // this thread's staging slice: row tid>>1, 16 halves at k-offset (tid&1)*16
  const int lr = tid >> 1, lk = (tid & 1) * 16;
  const int64_t grow = m0 + lr;
  int64_t arow = grow;
  bool avalid = grow < M;
  if (avalid && rows) { const int32_t rr = rows[grow]; avalid = rr >= 0; arow = rr; }
  const int brow = n0 + lr;
  const bool bvalid = brow < N;

  // Register-staged software pipeline, prefetch distance 3 k-steps: the loads of tile kt+3 are issued while tile kt
  // is multiplied, and reach LDS two iterations later -- one k-step (256 MFMA cycles per wave) is far too short to
  // cover an L2 / Infinity-Cache round trip on its own.
  ARegs<A_F32> ra[3];
  h8 rb[3][2];
  const int nk = K / BK;
  auto load_tile = [&](ARegs<A_F32>& a, h8 (&bq)[2], int kt) {
    if (kt < nk) {
      load_a<A_F32>(a, A, arow, avalid, lda, kt * BK + lk);
      if (bvalid) {
        const h8* p = reinterpret_cast<const h8*>(W + (int64_t)brow * ldw + kt * BK + lk);
        bq[0] = p[0]; bq[1] = p[1];
      } else { bq[0] = (h8)(_Float16)0; bq[1] = (h8)(_Float16)0; }
    }
  };
  auto store_tile = [&](const ARegs<A_F32>& a, const h8 (&bq)[2], int buf) {
    store_a<A_F32>(a, &As[buf][lr * LDT + lk]);
    reinterpret_cast<h8*>(&Bs[buf][lr * LDT + lk])[0] = bq[0];
    reinterpret_cast<h8*>(&Bs[buf][lr * LDT + lk])[1] = bq[1];
  };

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4)0.f;

  load_tile(ra[0], rb[0], 0);
  load_tile(ra[1], rb[1], 1);
  load_tile(ra[2], rb[2], 2);
  store_tile(ra[0], rb[0], 0);
  __syncthreads();

  const int fr = lane & 15, fk = (lane >> 4) * 8;
  for (int kt0 = 0; kt0 < nk; kt0 += 3) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int kt = kt0 + p;
      if (kt < nk) {                                   // uniform across the workgroup
        const int cur = kt & 1;
        load_tile(ra[p], rb[p], kt + 3);               // slot p was drained into LDS one iteration ago
        h8 fa[4], fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[i] = *reinterpret_cast<const h8*>(&As[cur][(wm * 64 + i * 16 + fr) * LDT + fk]);
          fw[i] = *reinterpret_cast<const h8*>(&Bs[cur][(wn * 64 + i * 16 + fr) * LDT + fk]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_tile(ra[(p + 1) % 3], rb[(p + 1) % 3], cur ^ 1);
        __syncthreads();
      }
    }
  }

  linear_epilogue(acc, bias, out, ldo, gate, ldg, nullptr, 0, epilogue, n_split, M, N, m0, n0, wm, wn, lane);
}


// ---------------------------------------------------------------------------------------------------
// linear_dma_kernel: the hot variant for f16 activations (optionally row-gathered).
//   Both operands go global -> LDS by `global_load_lds_dwordx4` (LDS-DMA, no VGPR staging, no ds_write), into a
//   3-stage ring of 16 KB stages; two k-steps are in flight behind the one being multiplied, tracked with a counted
//   s_waitcnt vmcnt(4) and a raw s_barrier (one barrier per k-step).  A DMA lands lane-linear (wave base + lane*16 B),
//   so the bank-conflict swizzle is applied to the SOURCE chunk index and to the fragment read address:
//   LDS rows are 64 B (32 halves), slot = chunk ^ g[(row>>2)&3], g = {0,2,3,1}, which makes every hardware lane
//   group of ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte slots.
// ---------------------------------------------------------------------------------------------------
constexpr int NST = 3;
constexpr int STAGE_HALVES = (BM + BN) * BK;          // 8192 halves = 16 KB
__device__ unsigned g_zero_row[64];                    // 256 B of zeros: source of masked / gathered(-1) rows

__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__global__ __launch_bounds__(256) void linear_dma_kernel(const _Float16* __restrict__ A, int64_t lda,
                                                         const int32_t* __restrict__ rows,
                                                         const _Float16* __restrict__ W, int64_t ldw,
                                                         const _Float16* __restrict__ bias, void* __restrict__ out,
                                                         int64_t ldo, const _Float16* __restrict__ gate, int64_t ldg,
                                                         _Float16* __restrict__ out16, int64_t ld16, int epilogue,
                                                         int n_split, int64_t M, int N, int K) {
  __shared__ __attribute__((aligned(1024))) _Float16 smem[NST * STAGE_HALVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN;
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int64_t mt = (int64_t)(idx / ntn) * 8 + xcd;
  const int64_t m0 = mt * BM;
  const int n0 = (idx % ntn) * BN;
  if (m0 >= M) return;

  // per-lane DMA sources: wave w stages tile rows [32w, 32w+32) of A and of W, 16 rows (1 KB) per instruction
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_row);
  const _Float16* srcA[2]; const _Float16* srcW[2];
  int64_t stepA[2], stepW[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 32 + j * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz(r);
    const int64_t grow = m0 + r;
    bool va = grow < M;
    int64_t arow = grow;
    if (va && rows) { const int32_t rr = rows[grow]; va = rr >= 0; arow = rr; }
    srcA[j] = va ? A + arow * lda + c * 8 : zero;
    stepA[j] = va ? BK : 0;
    const int nrow = n0 + r;
    const bool vb = nrow < N;
    srcW[j] = vb ? W + (int64_t)nrow * ldw + c * 8 : zero;
    stepW[j] = vb ? BK : 0;
  }
  auto issue = [&](int stage, int kt) {
    _Float16* base = smem + stage * STAGE_HALVES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcA[j] + kt * stepA[j]),
                                       (lds_void_t*)(base + (wave * 32 + j * 16) * BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(srcW[j] + kt * stepW[j]),
                                       (lds_void_t*)(base + (BM + wave * 32 + j * 16) * BK), 16, 0, 0);
    }
  };

  f4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f4)0.f;

  const int nk = K / BK;
  issue(0, 0);
  if (nk > 1) issue(1, 1);

  const int fr = lane & 15;
  const int foff = fr * BK + (((lane >> 4) ^ swz(fr)) * 8);       // halves, inside a 16-row group
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue((kt + 2) % NST, kt + 2);
    const _Float16* sa = smem + (kt % NST) * STAGE_HALVES;
    const _Float16* sw = sa + BM * BK;
    h8 fa[4], fw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = *reinterpret_cast<const h8*>(sa + (wm * 64 + i * 16) * BK + foff);
      fw[i] = *reinterpret_cast<const h8*>(sw + (wn * 64 + i * 16) * BK + foff);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[j], fa[i], acc[i][j], 0, 0, 0);
  }
  linear_epilogue(acc, bias, out, ldo, gate, ldg, out16, ld16, epilogue, n_split, M, N, m0, n0, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over D = 384 with optional fused inputs: one wave per row, 6 elements per lane.
// ---------------------------------------------------------------------------------------------------
template <bool X_F32>
__global__ __launch_bounds__(256) void layernorm384_kernel(const void* __restrict__ x, const _Float16* __restrict__ add1,
                                                           const int64_t* __restrict__ add1_rows, int64_t add1_mod,
                                                           const _Float16* __restrict__ add2,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ y_f32,
                                                           _Float16* __restrict__ y_f16, int relu_f16, int64_t M) {
  constexpr int D = 384;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float v[6];
  if constexpr (X_F32) {
    const float* xr = reinterpret_cast<const float*>(x) + row * D;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = xr[lane + 64 * i];
  } else {
    const _Float16* xr = reinterpret_cast<const _Float16*>(x) + row * D;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = (float)xr[lane + 64 * i];
  }
  if (add1) {
    int64_t r1 = row;
    if (add1_rows) { r1 = add1_rows[row]; if (add1_mod > 0) r1 %= add1_mod; }
    const _Float16* a = add1 + r1 * D;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] += (float)a[lane + 64 * i];
  }
  if (add2) {
    const _Float16* a = add2 + row * D;
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] += (float)a[lane + 64 * i];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) s += v[i];
  const float mean = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) { const float d = v[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
    if (y_f32) y_f32[row * D + c] = o;
    if (y_f16) y_f16[row * D + c] = (_Float16)((relu_f16 && o < 0.f) ? 0.f : o);
  }
}

// ---------------------------------------------------------------------------------------------------
// SoftAgg: one 384-thread block per group, online softmax over the group's member rows.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void softagg_kernel(const _Float16* __restrict__ fg, int64_t ldfg,
                                                      const int32_t* __restrict__ perm, const int32_t* __restrict__ off,
                                                      const int32_t* __restrict__ n_groups, _Float16* __restrict__ y,
                                                      int D) {
  const int ng = *n_groups;
  const int c = threadIdx.x;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int b = off[g], e = off[g + 1];
    float m = -INFINITY, s = 0.f, a = 0.f;
    for (int p = b; p < e; ++p) {
      const _Float16* rowp = fg + (int64_t)perm[p] * ldfg;
      const float fx = (float)rowp[c], gx = (float)rowp[D + c];
      const float mn = fmaxf(m, gx);
      const float sc = __expf(m - mn), w = __expf(gx - mn);
      s = s * sc + w;
      a = a * sc + w * fx;
      m = mn;
    }
    y[(int64_t)g * D + c] = (_Float16)(a / s);
  }
}

__global__ void gather_add_kernel(float* __restrict__ net, const _Float16* __restrict__ hy,
                                  const int32_t* __restrict__ group, _Float16* __restrict__ net16, int64_t E, int D) {
  const int64_t total = E * (D / 4);
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = n / (D / 4);
    const int c = (int)(n - e * (D / 4)) * 4;
    const h4 h = *reinterpret_cast<const h4*>(hy + (int64_t)group[e] * D + c);
    f4* dst = reinterpret_cast<f4*>(net + e * D + c);
    f4 o = *dst;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += (float)h[r];
    *dst = o;
    if (net16) {
      h4 o16;
#pragma unroll
      for (int r = 0; r < 4; ++r) o16[r] = (_Float16)o[r];
      *reinterpret_cast<h4*>(net16 + e * D + c) = o16;
    }
  }
}

// heads: one wave per edge row; four 384-long dot products of relu(net) (rounded to f16 as autocast feeds
// the Linear) with f16 weights, f32 accumulate.
__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ net, const _Float16* __restrict__ Wd,
                                                    const _Float16* __restrict__ bd, const _Float16* __restrict__ Ww,
                                                    const _Float16* __restrict__ bw, float* __restrict__ delta,
                                                    float* __restrict__ weight, int64_t E) {
  constexpr int D = 384;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= E) return;
  float d0 = 0.f, d1 = 0.f, w0 = 0.f, w1 = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    float v = net[row * D + c];
    v = v > 0.f ? v : 0.f;
    v = (float)(_Float16)v;
    d0 += v * (float)Wd[c]; d1 += v * (float)Wd[D + c];
    w0 += v * (float)Ww[c]; w1 += v * (float)Ww[D + c];
  }
  d0 = wave_sum(d0); d1 = wave_sum(d1); w0 = wave_sum(w0); w1 = wave_sum(w1);
  if (lane == 0) {
    delta[2 * row + 0] = (float)(_Float16)(d0 + (float)bd[0]);
    delta[2 * row + 1] = (float)(_Float16)(d1 + (float)bd[1]);
    const _Float16 h0 = (_Float16)(w0 + (float)bw[0]), h1 = (_Float16)(w1 + (float)bw[1]);
    weight[2 * row + 0] = (float)(_Float16)sigmoidf_((float)h0);
    weight[2 * row + 1] = (float)(_Float16)sigmoidf_((float)h1);
  }
}

}  // namespace

extern "C" int dpvo_linear(const void* A, int a_dtype, int64_t lda, const int32_t* rows, const void* W, int64_t ldw,
                           const void* bias, void* out, int64_t ldo, const void* gate, int64_t ldg, void* out16,
                           int64_t ld16, int epilogue, int n_split, int64_t M, int N, int K, void* stream) {
  if (M < 0 || N <= 0 || K <= 0) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (!A || !W || !out) return DPVO_E_INVALID;
  if ((K % BK) != 0 || (N % 16) != 0) return DPVO_E_UNSUPPORTED;
  if ((lda % 8) != 0 || (ldw % 8) != 0 || (ldo % 4) != 0) return DPVO_E_UNSUPPORTED;
  if (epilogue < DPVO_EPI_NONE || epilogue > DPVO_EPI_RELU_SIG) return DPVO_E_INVALID;
  if (epilogue == DPVO_EPI_GATED && (!gate || (ldg % 4) != 0)) return DPVO_E_INVALID;
  const int64_t mtiles = cdiv64(M, BM), ntn = (N + BN - 1) / BN;
  const dim3 grid((unsigned)(cdiv64(mtiles, 8) * 8 * ntn));
  if (out16 && (ld16 % 4) != 0) return DPVO_E_UNSUPPORTED;
  if (a_dtype == DPVO_F32) {
    if (out16) return DPVO_E_UNSUPPORTED;                 // the f16 image output belongs to the f16 (DMA) variant
    hipLaunchKernelGGL(linear_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A, lda, rows, (const _Float16*)W,
                       ldw, (const _Float16*)bias, out, ldo, (const _Float16*)gate, ldg, epilogue, n_split, M, N, K);
  } else if (a_dtype == DPVO_F16) {
    hipLaunchKernelGGL(linear_dma_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)A, lda, rows,
                       (const _Float16*)W, ldw, (const _Float16*)bias, out, ldo, (const _Float16*)gate, ldg,
                       (_Float16*)out16, ld16, epilogue, n_split, M, N, K);
  } else {
    return DPVO_E_UNSUPPORTED;
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_layernorm(const void* x, int x_dtype, const void* add1, const int64_t* add1_rows, int64_t add1_mod,
                              const void* add2, const float* gamma, const float* beta, float eps, float* y_f32,
                              void* y_f16, int relu_f16, int64_t M, int D, void* stream) {
  if (M < 0) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (D != 384) return DPVO_E_UNSUPPORTED;
  if (!x || !gamma || !beta || (!y_f32 && !y_f16)) return DPVO_E_INVALID;
  const dim3 grid((unsigned)cdiv64(M, 4));
  if (x_dtype == DPVO_F32)
    hipLaunchKernelGGL(layernorm384_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)add1,
                       add1_rows, add1_mod, (const _Float16*)add2, gamma, beta, eps, y_f32, (_Float16*)y_f16, relu_f16, M);
  else if (x_dtype == DPVO_F16)
    hipLaunchKernelGGL(layernorm384_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)add1,
                       add1_rows, add1_mod, (const _Float16*)add2, gamma, beta, eps, y_f32, (_Float16*)y_f16, relu_f16, M);
  else
    return DPVO_E_UNSUPPORTED;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_softagg(const void* fg, int64_t ldfg, const int32_t* perm, const int32_t* off,
                            const int32_t* n_groups, int64_t max_groups, void* y, int D, void* stream) {
  if (max_groups < 0) return DPVO_E_INVALID;
  if (max_groups == 0) return DPVO_OK;
  if (D != 384) return DPVO_E_UNSUPPORTED;
  if (!fg || !perm || !off || !n_groups || !y) return DPVO_E_INVALID;
  const unsigned grid = (unsigned)(max_groups < 8192 ? max_groups : 8192);
  hipLaunchKernelGGL(softagg_kernel, dim3(grid), dim3(384), 0, (hipStream_t)stream, (const _Float16*)fg, ldfg, perm, off,
                     n_groups, (_Float16*)y, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_gather_add(float* net, const void* hy, const int32_t* group, void* net16, int64_t E, int D,
                               void* stream) {
  if (E < 0 || D <= 0 || (D % 4) != 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !hy || !group) return DPVO_E_INVALID;
  int64_t g = cdiv64(E * (D / 4), 256);
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(gather_add_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, net, (const _Float16*)hy,
                     group, (_Float16*)net16, E, D);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_heads(const float* net, const void* Wd, const void* bd, const void* Ww, const void* bw, float* delta,
                          float* weight, int64_t E, int D, void* stream) {
  if (E < 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (D != 384) return DPVO_E_UNSUPPORTED;
  if (!net || !Wd || !bd || !Ww || !bw || !delta || !weight) return DPVO_E_INVALID;
  hipLaunchKernelGGL(heads_kernel, dim3((unsigned)cdiv64(E, 4)), dim3(256), 0, (hipStream_t)stream, net,
                     (const _Float16*)Wd, (const _Float16*)bd, (const _Float16*)Ww, (const _Float16*)bw, delta, weight, E);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
