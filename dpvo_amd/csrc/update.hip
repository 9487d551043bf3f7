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
#include "../../include/dpvo_hip_cmp.h"

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

// (v_exp + v_rcp, 1 ulp; the result is rounded to f16 by every caller -- same function as update_fused.hip's, so that the two paths
// keep identical rounding points)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }


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
__device__ unsigned g_zero_row[256];                   // 1 KB of zeros: source of masked / gathered(-1) rows

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
// linear_ws_kernel: weights-stationary persistent GEMM for the 384-wide layers (K = 384, N a multiple of 384; each
// workgroup owns 384 output columns, N = 768 runs as two column groups).
//   The update operator multiplies E ~ 47 712 rows by 384x384 weights: the weights are tiny, the rows are many, and at
//   73 MB of activation traffic per 14 GFLOP the layer sits on the HBM side of the roofline.  So the kernel is built
//   around keeping the memory queue full, and everything else out of its way:
//     * each of the 4 waves of a workgroup (ONE wave per SIMD: the whole 512-entry VGPR+AGPR file is its own) keeps
//       its 96 output columns of W (72 MFMA fragments = 288 registers) resident for the whole kernel -- no weight
//       byte is re-fetched, and W itself arrives through LDS-DMA in full cache lines after one L2-warming touch
//       (all 256 workgroups miss on the same lines at the same time otherwise);
//     * activation rows stream through LDS exactly once: 32-row stages (rows padded to 1 KB = one LDS-DMA instruction
//       per row, source-side XOR swizzle of the 16-byte chunks with the row index), FOUR stage buffers, the DMA of
//       stage t+3 issued row by row between the MFMA k-steps of stage t, one barrier per stage;
//     * vmcnt is in-order and counts stores too, so waiting for "stage t has landed" is a COUNTED wait: every
//       vector-memory instruction of the stage body is unconditional (rows past M are clamped for loads and sent to a
//       sink for stores) so that the number of younger instructions is a compile-time constant;
//     * LDS reads are issued by hand (inline asm) two k-steps ahead of their MFMAs with counted lgkmcnt waits -- the
//       compiler would order every LDS read it knows about behind all outstanding LDS-DMA;
//     * the epilogue of a stage is split: E1 (activation, one rounding to f16, transpose through a wave-private LDS
//       scratch so that global accesses are 16-byte and row-contiguous) at the end of the stage, E2 (residual / gate
//       loads, stores) in the middle of the NEXT stage, under its MFMAs.  The bias rides in the accumulator init.
// ---------------------------------------------------------------------------------------------------
constexpr int WS_K = 384, WS_PITCH = 512;                    // halves per LDS row (1 KB)
constexpr int WS_SROWS = 32;                                 // rows per stage: 2 m-tiles
constexpr int WS_NBUF = 4;
// NT = n-tiles (of 16 columns) per wave.  NT = 6: a workgroup owns 384 columns (4 waves x 96).  NT = 3: 192 columns --
// N = 384 then runs as TWO column groups that walk the same rows on the same XCD (the second read of an activation row
// hits L2) and every CU pulls only half of W through L2 in its prologue.
constexpr int WS_ROW_BYTES = WS_PITCH * 2;
constexpr int WS_STAGE_BYTES = WS_SROWS * WS_ROW_BYTES;      // 32 KB
template <int NT> struct WsGeom {
  static constexpr int CPR = 2 * NT;                         // 16-byte chunks per scratch row (16 NT columns)
  static constexpr int NCHT = NT;                            // chunks per lane of a 32-row stage tile (32 CPR / 64)
  static constexpr int EP_PITCH = 32 * NT + 16;              // bytes per scratch row: 16 NT halves + 16 B pad
  static constexpr int EP_BYTES = 32 * EP_PITCH;             // per wave
  static constexpr int LDS_BYTES = WS_NBUF * WS_STAGE_BYTES + 4 * EP_BYTES;
};

__device__ uint4 g_ws_sink[64 * 4];                          // 64 B per lane: where the stores of rows >= M go

#ifdef WS_TRACE
__device__ unsigned long long g_ws_trace[512][4][32];
#define WS_T(i) do { if (lane == 0 && (i) < 32) g_ws_trace[blockIdx.y * gridDim.x + blockIdx.x][wave][i] = wall_clock64(); } while (0)
#else
#define WS_T(i) do {} while (0)
#endif
#define WS_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define WS_DSWRITE8(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")

struct WsOut {                 // everything the epilogue needs (uniform)
  void* out; int64_t ldo; const _Float16* gate; int64_t ldg; _Float16* out16; int64_t ld16;
  int epilogue, n_split, N, n_base; int64_t M;
};
struct WsE2 {                  // registers of the deferred epilogue half
  h8 ep[6]; f4 o0[6], o1[6]; h8 g[6];
  uint32_t sabs[6], eoff[6];   // (plain-store kernel only) resident scratch addresses / output byte offsets of the 6 chunks
};

// Schedule constants.  Vector-memory instructions per stage body: 8 DMA rows + E2 (RMW: 3 loads and 3 stores per chunk,
// else 1 store per chunk).  E2 runs in G groups of NCH chunks: loads at k-step SL (before that step's DMA row), scratch
// read-back at SR, combine and store at SC (after that step's DMA row).  The 96-column read-modify-write kernel splits
// it in two to halve its registers.
template <bool RMW, int NT> struct WsCount {
  static constexpr int NCHT = WsGeom<NT>::NCHT;
  static constexpr int E2_STORES = RMW ? 3 * NCHT : NCHT;
  static constexpr int BODY = 8 + (RMW ? 3 * NCHT : 0) + E2_STORES;
  static constexpr int G = (RMW && NT == 6) ? 2 : 1, NCH = NCHT / G;
  static constexpr int SL(int g) { return g == 0 ? 0 : 5; }
  static constexpr int SR(int g) { return RMW ? (G == 2 ? (g == 0 ? 3 : 8) : 7) : 3; }
  static constexpr int SC(int g) { return RMW ? (G == 2 ? (g == 0 ? 5 : 10) : 9) : 5; }
  // DMA rows (k-steps 0..7, one each) issued between the loads of group g and its combine step
  static constexpr int VM_YOUNGER(int g) { return G == 2 ? (g == 0 ? 6 : 2) : 8; }
  // younger than the last DMA row (k-step 7) of stage t when body t starts: what body t-3 still issued after it (the
  // stores of the last RMW group), and bodies t-2, t-1 in full.  vmcnt is a 6-bit counter: a smaller number only waits longer.
  static constexpr int YOUNGER = 2 * BODY + (RMW ? 3 * NCH : 0);
  static constexpr int WAIT = YOUNGER < 63 ? YOUNGER : 63;
  // LDS reads run RING - 1 k-steps ahead of their MFMAs.  The read-modify-write kernel has no registers left for more
  // than one step (12 MFMAs = 192 cycles of cover for an LDS read).
  static constexpr int RING = RMW ? 2 : 3;
};

// LDS reads of k-step T (both m-tiles) into register slot T % RING
template <int T, int RING>
__device__ __forceinline__ void ws_read(h8 (&fa)[RING][2], const uint32_t (&aq)[4], uint32_t poff) {
  const uint32_t a = aq[T & 3] + poff;
  WS_DSREAD(fa[T % RING][0], a, (T >> 2) * 256);
  WS_DSREAD(fa[T % RING][1], a, (T >> 2) * 256 + 16 * WS_ROW_BYTES);
}

// chunk i (of NT) of a 32 x 16 NT stage tile handled by this lane: 16 B = 8 columns, row-contiguous across lanes
template <int NT>
__device__ __forceinline__ void ws_chunk(int lane, int i, int& row, int& col8) {
  const int c = lane + 64 * i;
  row = c / WsGeom<NT>::CPR;
  col8 = c - WsGeom<NT>::CPR * row;
}

// E2 / L: residual and gate rows of the pending stage (3 loads per chunk, always).  Issued by hand: the compiler's own
// vmcnt bookkeeping gives up on the mix of LDS-DMA, loads and stores and waits for vmcnt(0), i.e. for the whole DMA pipeline.
#define WS_GLOAD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
template <int NT, int C0, int NCH>
__device__ __forceinline__ void ws_e2_load(WsE2& e, const WsOut& o, int64_t m0, int lane, const _Float16* zero) {
  asm volatile("" : "+v"(lane));      // (address arithmetic is recomputed where it is used, not hoisted and spilled)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int row, col8;
    ws_chunk<NT>(lane, C0 + i, row, col8);
    int64_t m = m0 + row;
    m = m < o.M ? m : o.M - 1;
    const int n = o.n_base + col8 * 8;
    const float* src = reinterpret_cast<const float*>(o.out) + m * o.ldo + n;
    WS_GLOAD16(e.o0[i], src);
    WS_GLOAD16(e.o1[i], src + 4);
    const _Float16* gsrc = o.epilogue == DPVO_EPI_GATED ? o.gate + m * o.ldg + n : zero;
    WS_GLOAD16(e.g[i], gsrc);
  }
}
// ... and waited for by hand: YOUNGER vector-memory instructions were issued after them
template <int YOUNGER>
__device__ __forceinline__ void ws_e2_wait3(WsE2& e) {
  asm volatile("s_waitcnt vmcnt(%9)"
               : "+v"(e.o0[0]), "+v"(e.o0[1]), "+v"(e.o0[2]), "+v"(e.o1[0]), "+v"(e.o1[1]), "+v"(e.o1[2]), "+v"(e.g[0]),
                 "+v"(e.g[1]), "+v"(e.g[2])
               : "n"(YOUNGER));
}
// E2 / R: the stage tile back from the wave's scratch (NCH ds_read_b128, always)
template <bool RMW, int NT, int C0, int NCH>
__device__ __forceinline__ void ws_e2_read(WsE2& e, uint32_t scratch, int lane) {
  if constexpr (RMW) {
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int row, col8;
      ws_chunk<NT>(lane, C0 + i, row, col8);
      WS_DSREAD(e.ep[i], scratch + row * WsGeom<NT>::EP_PITCH + col8 * 16, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i) WS_DSREAD(e.ep[i], e.sabs[C0 + i], 0);
  }
}
template <int NCH>
__device__ __forceinline__ void ws_e2_tie(WsE2& e) {      // orders the uses of ep behind the lgkmcnt wait that retired the reads
  if constexpr (NCH == 6)
    asm volatile("" : "+v"(e.ep[0]), "+v"(e.ep[1]), "+v"(e.ep[2]), "+v"(e.ep[3]), "+v"(e.ep[4]), "+v"(e.ep[5]));
  else
    asm volatile("" : "+v"(e.ep[0]), "+v"(e.ep[1]), "+v"(e.ep[2]));
}
// E2 / C of the plain-store kernel inside the stage loop: whole stages only (the ragged last stage of the matrix is always
// the last stage of its workgroup and goes through the general path after the loop); "no stage pending" stores to the sink
template <int NT>
__device__ __forceinline__ void ws_e2_store_fast(WsE2& e, const WsOut& o, int64_t m0, int lane) {
  const bool none = m0 >= o.M;
  char* sb = none ? reinterpret_cast<char*>(g_ws_sink)
                  : reinterpret_cast<char*>(reinterpret_cast<_Float16*>(o.out) + m0 * o.ldo + o.n_base);
#pragma unroll
  for (int i = 0; i < WsGeom<NT>::NCHT; ++i) {
    const uint32_t off = none ? (uint32_t)lane * 64u : e.eoff[i];
    *reinterpret_cast<h8*>(sb + off) = e.ep[i];
  }
}
// E2 / C: combine and store (1 or 3 stores per chunk, always; rows >= M go to the sink)
template <bool RMW, int NT, int C0, int NCH>
__device__ __forceinline__ void ws_e2_store(WsE2& e, const WsOut& o, int64_t m0, int lane) {
  asm volatile("" : "+v"(lane));
  char* sink = reinterpret_cast<char*>(g_ws_sink) + lane * 64;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    int row, col8;
    ws_chunk<NT>(lane, C0 + i, row, col8);
    const int64_t m = m0 + row;
    const bool ok = m < o.M;
    const int n = o.n_base + col8 * 8;
    if constexpr (RMW) {
      h8 hv = e.ep[i];
      if (o.epilogue == DPVO_EPI_GATED) {
#pragma unroll
        for (int q = 0; q < 8; ++q) hv[q] = e.g[i][q] * hv[q];
      }
      f4 a = e.o0[i], b = e.o1[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] += (float)hv[q]; b[q] += (float)hv[4 + q]; }
      float* dst = ok ? reinterpret_cast<float*>(o.out) + m * o.ldo + n : reinterpret_cast<float*>(sink);
      *reinterpret_cast<f4*>(dst) = a;
      *reinterpret_cast<f4*>(dst + 4) = b;
      h8 o16;
#pragma unroll
      for (int q = 0; q < 4; ++q) { o16[q] = (_Float16)a[q]; o16[4 + q] = (_Float16)b[q]; }
      _Float16* d16 = (ok && o.out16) ? o.out16 + m * o.ld16 + n : reinterpret_cast<_Float16*>(sink + 32);
      *reinterpret_cast<h8*>(d16) = o16;
    } else {
      _Float16* dst = ok ? reinterpret_cast<_Float16*>(o.out) + m * o.ldo + n : reinterpret_cast<_Float16*>(sink);
      *reinterpret_cast<h8*>(dst) = e.ep[i];
    }
  }
}
// the whole E2 of a stage outside the pipeline (after the stage loop)
template <bool RMW, int NT>
__device__ __forceinline__ void ws_e2_flush(WsE2& e, const WsOut& o, int64_t m0, uint32_t scratch, int lane,
                                            const _Float16* zero) {
  using C = WsCount<RMW, NT>;
#pragma unroll
  for (int g = 0; g < C::G; ++g) {
    if (g == 0) {
      if constexpr (RMW) ws_e2_load<NT, 0, C::NCH>(e, o, m0, lane, zero);
      ws_e2_read<RMW, NT, 0, C::NCH>(e, scratch, lane);
    } else {
      if constexpr (RMW) ws_e2_load<NT, C::NCH, C::NCH>(e, o, m0, lane, zero);
      ws_e2_read<RMW, NT, C::NCH, C::NCH>(e, scratch, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    ws_e2_tie<C::NCH>(e);
    if constexpr (RMW) ws_e2_wait3<0>(e);
    if (g == 0) ws_e2_store<RMW, NT, 0, C::NCH>(e, o, m0, lane);
    else ws_e2_store<RMW, NT, C::NCH, C::NCH>(e, o, m0, lane);
  }
}

// One LDS-DMA row: lanes 0..47 fetch the 48 16-byte chunks of a 768-byte row, chunk c of a row with swizzle key r lands
// in slot (c & ~15) | ((c ^ r) & 15) = c ^ (r & 15) -- the DMA destination is lane-linear, so lane l fetches chunk
// l ^ (r & 15): one xor on the resident byte offset lane16 = 16 min(lane, 47) (lanes 48..63 land in the row's padding),
// added to a wave-uniform row base.  grow < 0 reads the (1 KB) zero row.  With a single wave per SIMD every dependent
// address instruction is exposed issue latency, hence the care.
__device__ __forceinline__ void ws_dma_row(const _Float16* base, int64_t ld, int32_t grow, int r, uint32_t lds_row,
                                           uint32_t lane16, const _Float16* zero) {
  const char* sb = reinterpret_cast<const char*>(grow >= 0 ? base + (int64_t)grow * ld : zero);
  const uint32_t voff = lane16 ^ (uint32_t)((r & 15) << 4);
  __builtin_amdgcn_global_load_lds((gbl_void_t*)(sb + voff), (lds_void_t*)(uintptr_t)lds_row, 16, 0, 0);
}

// E1: row mt*16 + m16 of the stage, columns j*16 + kg*4 .. +3 -> scratch.  ACT: 0 none, 1 relu, 2 sigmoid
template <int ACT, int NT>
__device__ __forceinline__ void ws_e1(const f4 (&acc)[2][NT], uint32_t scratch, int m16, int kg) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      h4 hv;
#pragma unroll
      for (int q = 0; q < 4; ++q) hv[q] = (_Float16)acc[mt][j][q];
      if constexpr (ACT == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = hv[q] > (_Float16)0 ? hv[q] : (_Float16)0;
      } else if constexpr (ACT == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = (_Float16)sigmoidf_((float)hv[q]);
      }
      const uint32_t wa = scratch + (mt * 16 + m16) * WsGeom<NT>::EP_PITCH + kg * 8;
      switch (j) {
        case 0: WS_DSWRITE8(wa, hv, 0); break;
        case 1: WS_DSWRITE8(wa, hv, 32); break;
        case 2: WS_DSWRITE8(wa, hv, 64); break;
        case 3: WS_DSWRITE8(wa, hv, 96); break;
        case 4: WS_DSWRITE8(wa, hv, 128); break;
        default: WS_DSWRITE8(wa, hv, 160); break;
      }
    }
}

struct WsDma {                 // next-stage DMA parameters (all wave-uniform)
  const _Float16* A; int64_t lda; int32_t row[8]; int wave; uint32_t lds; const _Float16* zero; uint32_t lane16;
};

// k-steps S .. 11 of one stage
template <int S, bool RMW, int NT>
__device__ __forceinline__ void ws_steps(f4 (&acc)[2][NT], const h8 (&wreg)[NT][12],
                                         h8 (&fa)[WsCount<RMW, NT>::RING][2],
                                         const uint32_t (&aq)[4], uint32_t poff, WsE2& e2, const WsOut& o, int64_t pend_m0,
                                         uint32_t scratch, int lane, const WsDma& d) {
  if constexpr (S < 12) {
    using C = WsCount<RMW, NT>;
    constexpr int RING = C::RING, D = RING - 1, NCH = C::NCH;
    constexpr int GL = (S == C::SL(0)) ? 0 : (C::G > 1 && S == C::SL(1)) ? 1 : -1;     // group whose loads go here
    constexpr int GR = (S == C::SR(0)) ? 0 : (C::G > 1 && S == C::SR(1)) ? 1 : -1;     // ... read-back
    constexpr int GC = (S == C::SC(0)) ? 0 : (C::G > 1 && S == C::SC(1)) ? 1 : -1;     // ... combine + store
    if constexpr (RMW && GL == 0) ws_e2_load<NT, 0, NCH>(e2, o, pend_m0, lane, d.zero);
    if constexpr (GR >= 0) ws_e2_read<RMW, NT, (GR > 0 ? NCH : 0), NCH>(e2, scratch, lane);
    if constexpr (S + D < 12) ws_read<S + D, RING>(fa, aq, poff);
    // LDS reads younger than those of step S: steps S+1 .. S+D (2 each) and the NCH scratch reads of a read-back issued
    // at one of the steps S-D+1 .. S (they sit in front of that step's own reads)
    constexpr int ahead = (11 - S) < D ? (11 - S) : D;
    constexpr bool r0 = S >= C::SR(0) && S < C::SR(0) + D;
    constexpr bool r1 = C::G > 1 && S >= C::SR(1) && S < C::SR(1) + D;
    constexpr int younger = 2 * ahead + ((r0 || r1) ? NCH : 0);
    static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
    static_assert(C::SC(0) >= C::SR(0) + D && C::SC(1) >= C::SR(1) + D, "read-back must be retired before the combine step");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[S % RING][0]), "+v"(fa[S % RING][1]) : "n"(younger));
    if constexpr (GC >= 0) ws_e2_tie<NCH>(e2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[j][S], fa[S % RING][mt], acc[mt][j], 0, 0, 0);
#ifndef WS_NO_DMA
    if constexpr (S < 8)       // stage t+3, row S of this wave's 8
      ws_dma_row(d.A, d.lda, d.row[S], d.wave * 8 + S, d.lds + S * WS_ROW_BYTES, d.lane16, d.zero);
#endif
#ifndef WS_NO_STORE
    if constexpr (GC >= 0) {
      // recompute the addresses instead of keeping pointers alive since the loads
      int m0lo = __builtin_amdgcn_readfirstlane((int)(pend_m0 & 0xffffffff));
      int m0hi = __builtin_amdgcn_readfirstlane((int)(pend_m0 >> 32));
      asm volatile("" : "+s"(m0lo), "+s"(m0hi));
      const int64_t m0s = ((int64_t)m0hi << 32) | (uint32_t)m0lo;
      if constexpr (RMW) {
        ws_e2_wait3<C::VM_YOUNGER(GC)>(e2);
        ws_e2_store<true, NT, (GC > 0 ? NCH : 0), NCH>(e2, o, m0s, lane);
      } else {
        ws_e2_store_fast<NT>(e2, o, m0s, lane);
      }
    }
    if constexpr (RMW && GL == 1) ws_e2_load<NT, NCH, NCH>(e2, o, pend_m0, lane, d.zero);
#endif
    ws_steps<S + 1, RMW, NT>(acc, wreg, fa, aq, poff, e2, o, pend_m0, scratch, lane, d);
  }
}

template <bool RMW, int NT>
__global__ __launch_bounds__(256) void linear_ws_kernel(const _Float16* __restrict__ A, int64_t lda,
                                                        const int32_t* __restrict__ rows,
                                                        const _Float16* __restrict__ W, int64_t ldw,
                                                        const _Float16* __restrict__ bias, WsOut o) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ws_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kg = lane >> 4;
  constexpr int WGCOLS = 64 * NT;                           // output columns per workgroup
  o.n_base = blockIdx.y * WGCOLS + wave * (16 * NT);
  const int n_base = o.n_base;
  const int64_t M = o.M;
  const int64_t nstages = (M + WS_SROWS - 1) / WS_SROWS;
  const int64_t gstep = gridDim.x;
  const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_row);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_t*)ws_smem;
  const uint32_t scratch = lds0 + WS_NBUF * WS_STAGE_BYTES + wave * WsGeom<NT>::EP_BYTES;
  const uint32_t lane16 = (uint32_t)(lane < 48 ? lane : 47) << 4;
  WS_T(0);

  // ---- L2 warm-up: the workgroup's 64 NT rows of W are 384 NT cache lines, fire and forget
  constexpr int NWARM = (384 * NT + 255) / 256;
  unsigned sinkreg[NWARM];              // (allocated until the first tile wait below: the loads land asynchronously)
  {
#pragma unroll
    for (int i = 0; i < NWARM; ++i) {
      int line = tid + 256 * i;
      line = line < 384 * NT ? line : 0;
      const int row = line / 6;
      const int part = line - 6 * row;
      const _Float16* p = W + (int64_t)(blockIdx.y * WGCOLS + row) * ldw + part * 64;
      asm volatile("global_load_dword %0, %1, off" : "=v"(sinkreg[i]) : "v"(p));
      if (i == 0) WS_T(27);
    }
  }
  WS_T(28);
  // source rows of the 8 stage rows this wave stages (stage rows 8w .. 8w+7); -1 = zero row.  The gather indices come
  // through SCALAR loads: a vector load consumed d stages later would cap the DMA pipeline at depth d (vmcnt is in-order).
  // (32-bit scalar arithmetic throughout: M < 2^30 is checked by the launcher; a stage past the end starts at >= M)
  const int M32 = (int)M;
  auto stage_rows_raw = [&](int64_t st, int32_t (&raw)[8]) {      // 8 s_load_dword, NOT waited for
    const int g0 = (int)st * WS_SROWS + wave * 8;
    if (rows) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t off = (g0 + i < M32) ? (uint32_t)(g0 + i) * 4u : 0u;
        asm volatile("s_load_dword %0, %1, %2" : "=s"(raw[i]) : "s"(rows), "s"(off));
      }
    }
  };
  auto stage_rows_wait = [&](int32_t (&raw)[8]) {
    if (rows)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(raw[0]), "+s"(raw[1]), "+s"(raw[2]), "+s"(raw[3]), "+s"(raw[4]), "+s"(raw[5]),
                   "+s"(raw[6]), "+s"(raw[7]));
  };
  auto stage_rows_fix = [&](int64_t st, const int32_t (&raw)[8], int32_t (&r)[8]) {
    const int g0 = (int)st * WS_SROWS + wave * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (g0 + i < M32) ? (rows ? raw[i] : g0 + i) : -1;
  };
  int64_t st = blockIdx.x;
  // bias of this lane's accumulator columns n_base + 16 j + 4 kg .. +3 (by hand as well: these loads and the warm-up
  // are simply older than the first W tile in the in-order queue, the tile-0 wait below retires them)
  h4 breg[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const _Float16* bp = bias ? bias + n_base + j * 16 + kg * 4 : zero;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(breg[j]) : "v"(bp));
  }
  WS_T(1);

  // LDS byte address of this lane's MFMA fragment for k-step s of row (lane & 15): chunk 4s + kg lives in slot
  // (c & ~15) | ((c ^ row) & 15) = 16 (s >> 2) + ((kg ^ m16) ^ 4 (s & 3)); m-tile and k-quarter are immediates.
  uint32_t aq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) aq[q] = lds0 + m16 * WS_ROW_BYTES + (((kg ^ m16) ^ (4 * q)) << 4);

  // ---- prologue: the wave's 96 columns x 384 k of W, tile by tile (16 rows) through its own 8-row slices of the stage
  // buffers: tile j -> rows 0..7 in buffer 2 (j & 1), rows 8..15 in buffer 2 (j & 1) + 1; two tiles in flight.  The slices
  // are private to the wave (it later fills the same ones with activation rows), so no barrier is needed.
  h8 wreg[NT][12];
  const uint32_t slice = lds0 + wave * 8 * WS_ROW_BYTES;
  auto w_tile = [&](int j) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      ws_dma_row(W, ldw, n_base + j * 16 + i, i, slice + (2 * (j & 1) + (i >> 3)) * WS_STAGE_BYTES + (i & 7) * WS_ROW_BYTES, lane16,
                 zero);
  };
  auto a_stage = [&](int b, int64_t stg) {
    int32_t raw[8], r[8];
    stage_rows_raw(stg, raw);
    stage_rows_wait(raw);
    stage_rows_fix(stg, raw, r);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      ws_dma_row(A, lda, r[i], wave * 8 + i, slice + b * WS_STAGE_BYTES + i * WS_ROW_BYTES, lane16, zero);
  };
  const uint32_t wq0 = slice + (m16 >> 3) * WS_STAGE_BYTES + (m16 & 7) * WS_ROW_BYTES;
  WS_T(29);
  w_tile(0);
  WS_T(30);
  w_tile(1);
  WS_T(31);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    // Issue order: T0, T1, X0, X1, ... where X_i (issued after tile i has been read out of its buffer pair) is tile i+2,
    // or, once the tiles are exhausted, the activation stages that live in the freed pair (pair 0: stages 0 and 1 = 16
    // rows, pair 1: stage 2 = 8 rows).  Younger than tile j when it is awaited: T1 for j = 0, else X_{j-1}.
    {
      const int i = j - 1;
      const int younger = (j == 0 || i + 2 < NT || (i & 1) == 0) ? 16 : 8;
      if (younger == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (j == 0) {     // the warm-up and bias loads are older than tile 0: retired by now
#pragma unroll
      for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(breg[i]));
#pragma unroll
      for (int i = 0; i < NWARM; ++i) asm volatile("" : "+v"(sinkreg[i]));
    }
    WS_T(2 + j);
    const uint32_t wo = wq0 + (j & 1) * 2 * WS_STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      h8 v;
      const uint32_t a = wo + (((kg ^ m16) ^ (4 * (s & 3))) << 4);
      switch (s >> 2) {
        case 0: WS_DSREAD(v, a, 0); break;
        case 1: WS_DSREAD(v, a, 256); break;
        default: WS_DSREAD(v, a, 512); break;
      }
      wreg[j][s] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(wreg[j][0]), "+v"(wreg[j][1]), "+v"(wreg[j][2]), "+v"(wreg[j][3]), "+v"(wreg[j][4]), "+v"(wreg[j][5]),
                   "+v"(wreg[j][6]), "+v"(wreg[j][7]), "+v"(wreg[j][8]), "+v"(wreg[j][9]), "+v"(wreg[j][10]), "+v"(wreg[j][11])
                 :
                 : "memory");
    if (j + 2 < NT) {
      w_tile(j + 2);
    } else if ((j & 1) == 0) {         // buffers 0, 1 are free: activation stages 0 and 1
      a_stage(0, st);
      a_stage(1, st + gstep);
    } else {                           // buffers 2, 3: stage 2
      a_stage(2, st + 2 * gstep);
    }
  }
  WS_T(8);

  // activation of this wave's 96 columns (RELU_SIG: n_split is a multiple of 96)
  const int act = (o.epilogue == DPVO_EPI_RELU || (o.epilogue == DPVO_EPI_RELU_SIG && n_base < o.n_split)) ? 1
                  : (o.epilogue == DPVO_EPI_SIGMOID || o.epilogue == DPVO_EPI_RELU_SIG) ? 2 : 0;
  WsE2 e2;
  if constexpr (!RMW) {
#pragma unroll
    for (int i = 0; i < WsGeom<NT>::NCHT; ++i) {
      int row, col8;
      ws_chunk<NT>(lane, i, row, col8);
      e2.sabs[i] = scratch + row * WsGeom<NT>::EP_PITCH + col8 * 16;
      e2.eoff[i] = (uint32_t)((row * o.ldo + col8 * 8) * 2);
    }
  }
  WsDma d{A, lda, {0, 0, 0, 0, 0, 0, 0, 0}, wave, 0, zero, lane16};
  {
    int32_t raw[8];
    stage_rows_raw(st + 3 * gstep, raw);
    stage_rows_wait(raw);
    stage_rows_fix(st + 3 * gstep, raw, d.row);
  }
  int64_t pend_m0 = M;                 // "no rows": loads clamp to row M-1, stores go to the sink
  int it = 0;
  for (; st < nstages; st += gstep, ++it) {
    // stage `it` has landed when at most WAIT younger instructions are outstanding.  The first two stages were issued by
    // the prologue with only 16 rows behind stage 0.
    // (NT even: stages were issued 0, 1, 2 -> 16 rows behind stage 0; NT odd: 2, 0, 1 -> 8 rows behind stage 0)
    if (it < 2) { if constexpr ((NT & 1) == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WsCount<RMW, NT>::WAIT) : "memory");
    __builtin_amdgcn_s_barrier();
    WS_T(9 + 3 * it);
    const int buf = it & 3;
    d.lds = slice + (uint32_t)((it + 3) & 3) * WS_STAGE_BYTES;
    f4 acc[2][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f4 b4;
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = (float)breg[j][q];
      acc[0][j] = b4;
      acc[1][j] = b4;
    }
    const uint32_t poff = (uint32_t)buf * WS_STAGE_BYTES;
    constexpr int RING = WsCount<RMW, NT>::RING;
    h8 fa[RING][2];
    ws_read<0, RING>(fa, aq, poff);
    if constexpr (RING > 2) ws_read<1, RING>(fa, aq, poff);
    ws_steps<0, RMW, NT>(acc, wreg, fa, aq, poff, e2, o, pend_m0, scratch, lane, d);
    WS_T(10 + 3 * it);
    int32_t raw_next[8];
    stage_rows_raw(st + 4 * gstep, raw_next);      // scalar loads fly under E1
    // E1: activation (uniform per wave), one rounding to f16, transpose through the scratch
    if (act == 1) ws_e1<1, NT>(acc, scratch, m16, kg);
    else if (act == 2) ws_e1<2, NT>(acc, scratch, m16, kg);
    else ws_e1<0, NT>(acc, scratch, m16, kg);
    pend_m0 = st * WS_SROWS;
    stage_rows_wait(raw_next);
    stage_rows_fix(st + 4 * gstep, raw_next, d.row);
    WS_T(11 + 3 * it);
  }
  // the last stage still owes its E2
  if (pend_m0 < M) {
    ws_e2_flush<RMW, NT>(e2, o, pend_m0, scratch, lane, zero);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the speculative DMA of stages past the end targets our LDS
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over D = 384 with optional fused inputs: one wave per row, 6 elements per lane.
// ---------------------------------------------------------------------------------------------------
template <bool X_F32>
__global__ __launch_bounds__(256) void layernorm384_kernel(const void* __restrict__ x, const _Float16* __restrict__ add1,
                                                           const int64_t* __restrict__ add1_rows, int64_t add1_mod,
                                                           const int32_t* __restrict__ add1_rows32,
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
    else if (add1_rows32) r1 = add1_rows32[row];            // (group index of the plan: the SoftAgg expand, net.py:88)
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
                                                    float* __restrict__ weight, const float* __restrict__ coords,
                                                    int pp, float* __restrict__ target, int64_t E) {
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
    if (target) {     // target = coords[..., P//2, P//2] + delta.float()   (dpvo.py:340)
      target[2 * row + 0] = coords[(row * 2 + 0) * pp + pp / 2] + delta[2 * row + 0];
      target[2 * row + 1] = coords[(row * 2 + 1) * pp + pp / 2] + delta[2 * row + 1];
    }
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
  } else if (a_dtype == DPVO_F16 && K == WS_K && (N % 384) == 0 && M >= 4096 && M < (1 << 30) &&
             (epilogue != DPVO_EPI_RELU_SIG || (n_split % 96) == 0)) {
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
      if (n_cu <= 0) n_cu = 256;
    }
    const bool rmw = epilogue == DPVO_EPI_RESADD || epilogue == DPVO_EPI_GATED;
    // column groups of 192 (NT = 3): N = 384 runs as two, N = 768 as four -- half the W prologue per CU, the twins'
    // activation reads hit L2 (measured 29.9 -> 26.3 us and 48.3 -> 44.6 us); 384-column groups (NT = 6) otherwise
    const bool half = (N % 192) == 0 && (epilogue != DPVO_EPI_RELU_SIG || (n_split % 48) == 0);
    const int wgcols = half ? 192 : 384;
    const int ngrp = N / wgcols;
    const int64_t nst = cdiv64(M, WS_SROWS);
    int gx = n_cu / ngrp;
    if (gx > nst) gx = (int)nst;
    if (gx < 1) gx = 1;
    WsOut wo{out, ldo, (const _Float16*)gate, ldg, (_Float16*)out16, ld16, epilogue, n_split, N, 0, M};
    auto launch = [&](auto kern, size_t sh) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipLaunchKernelGGL(kern, dim3(gx, ngrp), dim3(256), sh, (hipStream_t)stream, (const _Float16*)A, lda, rows,
                         (const _Float16*)W, ldw, (const _Float16*)bias, wo);
    };
    if (half) {
      if (rmw) launch(linear_ws_kernel<true, 3>, (size_t)WsGeom<3>::LDS_BYTES);
      else launch(linear_ws_kernel<false, 3>, (size_t)WsGeom<3>::LDS_BYTES);
    } else {
      if (rmw) launch(linear_ws_kernel<true, 6>, (size_t)WsGeom<6>::LDS_BYTES);
      else launch(linear_ws_kernel<false, 6>, (size_t)WsGeom<6>::LDS_BYTES);
    }
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

namespace {
int launch_layernorm(const void* x, int x_dtype, const void* add1, const int64_t* add1_rows, int64_t add1_mod,
                     const int32_t* add1_rows32, const void* add2, const float* gamma, const float* beta, float eps,
                     float* y_f32, void* y_f16, int relu_f16, int64_t M, int D, void* stream) {
  if (M < 0) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (D != 384) return DPVO_E_UNSUPPORTED;
  if (!x || !gamma || !beta || (!y_f32 && !y_f16)) return DPVO_E_INVALID;
  const dim3 grid((unsigned)cdiv64(M, 4));
  if (x_dtype == DPVO_F32)
    hipLaunchKernelGGL(layernorm384_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)add1,
                       add1_rows, add1_mod, add1_rows32, (const _Float16*)add2, gamma, beta, eps, y_f32, (_Float16*)y_f16,
                       relu_f16, M);
  else if (x_dtype == DPVO_F16)
    hipLaunchKernelGGL(layernorm384_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)add1,
                       add1_rows, add1_mod, add1_rows32, (const _Float16*)add2, gamma, beta, eps, y_f32, (_Float16*)y_f16,
                       relu_f16, M);
  else
    return DPVO_E_UNSUPPORTED;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
}  // namespace

extern "C" int dpvo_layernorm(const void* x, int x_dtype, const void* add1, const int64_t* add1_rows, int64_t add1_mod,
                              const void* add2, const float* gamma, const float* beta, float eps, float* y_f32,
                              void* y_f16, int relu_f16, int64_t M, int D, void* stream) {
  return launch_layernorm(x, x_dtype, add1, add1_rows, add1_mod, nullptr, add2, gamma, beta, eps, y_f32, y_f16, relu_f16, M, D,
                          stream);
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

extern "C" int dpvo_heads_target(const float* net, const void* Wd, const void* bd, const void* Ww, const void* bw,
                                 const float* coords, int P, float* delta, float* weight, float* target, int64_t E, int D,
                                 void* stream) {
  if (E < 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (D != 384) return DPVO_E_UNSUPPORTED;
  if (!net || !Wd || !bd || !Ww || !bw || !delta || !weight) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  hipLaunchKernelGGL(heads_kernel, dim3((unsigned)cdiv64(E, 4)), dim3(256), 0, (hipStream_t)stream, net,
                     (const _Float16*)Wd, (const _Float16*)bd, (const _Float16*)Ww, (const _Float16*)bw, delta, weight, coords,
                     P * P, target, E);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_heads(const float* net, const void* Wd, const void* bd, const void* Ww, const void* bw, float* delta,
                          float* weight, int64_t E, int D, void* stream) {
  return dpvo_heads_target(net, Wd, bd, Ww, bw, nullptr, 0, delta, weight, nullptr, E, D, stream);
}

// ---------------------------------------------------------------------------------------------------
// The whole update operator (net.py:74-92) as ONE C-ABI call: the same 31 launches the host wrapper would issue one by
// one through ctypes.  At > 500 frames/sec the Python side of those launches (~0.5 ms per frame) is what bounds the
// frame rate, not the kernels.
// ---------------------------------------------------------------------------------------------------
namespace {
inline size_t upd_al(size_t x) { return (x + 255) & ~(size_t)255; }
struct UpdWs { size_t h1, h2, x16, fg, y, hy, total; };
inline void upd_ws_layout(int64_t E, int64_t maxg, UpdWs* w) {
  const size_t e = (size_t)(E > 0 ? E : 1), g = (size_t)(maxg > 0 ? maxg : 1);
  size_t o = 0;
  w->h1 = o; o += upd_al(e * 384 * 2);
  w->h2 = o; o += upd_al(e * 384 * 2);
  w->x16 = o; o += upd_al(e * 384 * 2);
  w->fg = o; o += upd_al(e * 768 * 2);
  w->y = o; o += upd_al(g * 384 * 2);
  w->hy = o; o += upd_al(g * 384 * 2);
  w->total = o;
}
}  // namespace

extern "C" size_t dpvo_update_workspace_bytes(int64_t E, int64_t max_groups) {
  if (E < 0 || max_groups < 0) return 0;
  UpdWs w;
  upd_ws_layout(E, max_groups, &w);
  return w.total;
}

extern "C" int dpvo_update_forward(const dpvo_update_params_t* p, const float* net, const void* inp, const int64_t* inp_rows,
                                   int64_t inp_mod, const void* corr, int64_t ld_corr, const int32_t* plan,
                                   int64_t n_patches_ub, int64_t n_pairs_ub, const float* coords, int P, float* net_out,
                                   float* delta, float* weight, float* target, int64_t E, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8)) return DPVO_E_UNSUPPORTED;
  const int64_t maxg = n_patches_ub > n_pairs_ub ? n_patches_ub : n_pairs_ub;
  UpdWs L;
  upd_ws_layout(E, maxg, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* w = (char*)ws;
  void *h1 = w + L.h1, *h2 = w + L.h2, *x16 = w + L.x16, *fg = w + L.fg, *y = w + L.y, *hy = w + L.hy;
  float* x = net_out;
  int rc;
#define UPD(call) do { rc = (call); if (rc) return rc; } while (0)
  // net = net + inp + self.corr(corr); net = self.norm(net)                                  (net.py:77-78)
  UPD(dpvo_linear(corr, DPVO_F16, ld_corr, nullptr, p->c0_w, 896, p->c0_b, h1, 384, nullptr, 0, nullptr, 0, DPVO_EPI_RELU, 0, E,
                  384, 896, stream));
  UPD(dpvo_linear(h1, DPVO_F16, 384, nullptr, p->c2_w, 384, p->c2_b, h2, 384, nullptr, 0, nullptr, 0, DPVO_EPI_NONE, 0, E, 384,
                  384, stream));
  UPD(dpvo_layernorm(h2, DPVO_F16, nullptr, nullptr, 0, nullptr, p->cln_g, p->cln_b, 1e-3f, nullptr, h1, 1, E, 384, stream));
  UPD(dpvo_linear(h1, DPVO_F16, 384, nullptr, p->c5_w, 384, p->c5_b, h2, 384, nullptr, 0, nullptr, 0, DPVO_EPI_NONE, 0, E, 384,
                  384, stream));
  UPD(dpvo_layernorm(net, DPVO_F32, inp, inp_rows, inp_mod, h2, p->norm_g, p->norm_b, 1e-3f, x, x16, 0, E, 384, stream));
  // net = net + c1(mask_ix * net[:,ix]); net = net + c2(mask_jx * net[:,jx])                  (net.py:80-85)
  for (int k = 0; k < 2; ++k) {
    const void *W0 = k ? p->c2n_w0 : p->c1_w0, *b0 = k ? p->c2n_b0 : p->c1_b0, *W2 = k ? p->c2n_w2 : p->c1_w2,
               *b2 = k ? p->c2n_b2 : p->c1_b2;
    const int32_t* rows = plan + (k ? PL.jx : PL.ix);
    UPD(dpvo_linear(x16, DPVO_F16, 384, rows, W0, 384, b0, h1, 384, nullptr, 0, nullptr, 0, DPVO_EPI_RELU, 0, E, 384, 384, stream));
    UPD(dpvo_linear(h1, DPVO_F16, 384, nullptr, W2, 384, b2, x, 384, nullptr, 0, x16, 384, DPVO_EPI_RESADD, 0, E, 384, 384, stream));
  }
  // net = net + agg_kk(net, kk); net = net + agg_ij(net, ii*12345 + jj)                        (net.py:87-88)
  for (int k = 0; k < 2; ++k) {
    const void *Wfg = k ? p->aij_wfg : p->akk_wfg, *bfg = k ? p->aij_bfg : p->akk_bfg, *Wh = k ? p->aij_wh : p->akk_wh,
               *bh = k ? p->aij_bh : p->akk_bh;
    const int32_t* perm = plan + (k ? PL.perm_p : PL.perm_k);
    const int32_t* off = plan + (k ? PL.pair_off : PL.patch_off);
    const int32_t* cnt = plan + PL.counts + k;
    const int32_t* grp = plan + (k ? PL.pu : PL.ku);
    int64_t ng = k ? n_pairs_ub : n_patches_ub;
    if (ng < 1) ng = 1;
    if (ng > E) ng = E;
    UPD(dpvo_linear(x16, DPVO_F16, 384, nullptr, Wfg, 384, bfg, fg, 768, nullptr, 0, nullptr, 0, DPVO_EPI_NONE, 0, E, 768, 384, stream));
    UPD(dpvo_softagg(fg, 768, perm, off, cnt, ng, y, 384, stream));
    UPD(dpvo_linear(y, DPVO_F16, 384, nullptr, Wh, 384, bh, hy, 384, nullptr, 0, nullptr, 0, DPVO_EPI_NONE, 0, ng, 384, 384, stream));
    // (the expand + residual of agg_ij is folded into the LayerNorm that follows: same f32 sum, one pass less over net)
    if (k == 0) UPD(dpvo_gather_add(x, hy, grp, x16, E, 384, stream));
  }
  // net = self.gru(net): 2 x (LayerNorm, x + gate(x) * res(x))                                (net.py:90)
  for (int k = 0; k < 2; ++k) {
    const float *g = k ? p->g1_g : p->g0_g, *b = k ? p->g1_b : p->g0_b;
    const void *Wrg = k ? p->g1_wrg : p->g0_wrg, *brg = k ? p->g1_brg : p->g0_brg, *W2 = k ? p->g1_w2 : p->g0_w2,
               *b2 = k ? p->g1_b2 : p->g0_b2;
    UPD(launch_layernorm(x, DPVO_F32, k == 0 ? hy : nullptr, nullptr, 0, k == 0 ? plan + PL.pu : nullptr, nullptr, g, b, 1e-3f, x,
                         h1, 0, E, 384, stream));
    UPD(dpvo_linear(h1, DPVO_F16, 384, nullptr, Wrg, 384, brg, fg, 768, nullptr, 0, nullptr, 0, DPVO_EPI_RELU_SIG, 384, E, 768, 384,
                    stream));
    UPD(dpvo_linear(fg, DPVO_F16, 768, nullptr, W2, 384, b2, x, 384, (const _Float16*)fg + 384, 768, nullptr, 0, DPVO_EPI_GATED, 0, E,
                    384, 384, stream));
  }
  UPD(dpvo_heads_target(x, p->d_w, p->d_b, p->w_w, p->w_b, target ? coords : nullptr, P, delta, weight, target, E, 384, stream));
#undef UPD
  return DPVO_OK;
}

