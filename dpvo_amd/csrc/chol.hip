// chol.hip -- dense Cholesky solve of the damped global-BA system on gfx950.
//
// Replaces the `torch::linalg_cholesky_ex` + `torch::cholesky_solve` pair of the reference's global BA
// (dpvo/fastba/ba_cuda.cu:546-548; caller DPVO.__run_global_BA, dpvo/dpvo.py:312-326): 6N x 6N f32, N = free poses (hundreds
// with LOOP_CLOSURE).  rocSOLVER took 16 ms per factorisation at 6N = 4 794; this is a blocked right-looking factorisation whose
// rank-64 trailing updates run on v_mfma_f32_32x32x2_f32, with no atomics anywhere: the result is bit-repeatable run to run.
//
//   chol_load_kernel     S (n x n, the caller's) -> the work matrix Lw ((nb + 1) x nb blocks of 64 x 64, row-major, ld = 64 nb):
//                        damping S_ii += 1e-4 S_ii + 1 (ba_cuda.cu:546), identity on the padded diagonal, and the right-hand side
//                        as ROW 64 nb.  Factorising the bordered matrix [S b; b^T .] leaves L^-1 b in that row, so the forward
//                        substitution costs nothing extra.
//   per panel k:
//   chol_panel_kernel    one wave per row block below the diagonal block (the border row included): every wave factorises the
//                        64 x 64 diagonal block by itself, in registers (cheaper than a launch + a round trip through memory),
//                        then solves its own block against it: X = A_ik L_kk^-T.  L_kk is kept in Dg[k].
//   chol_step_kernel     (panels k >= 1) one workgroup per 64 x 64 tile (i >= j >= k) of the trailing matrix of panel k - 1:
//                        A_ij -= L_i,k-1 L_j,k-1^T, K = 64 as 32 steps of the 32x32x2 f32 MFMA per wave, operands straight from L2
//                        into the fragment registers; the workgroups of the first tile column go on with panel k (look-ahead).
//   chol_back_kernel     per block column from the last: x_k = L_kk^-T y_k (one wave, in registers), then every workgroup takes one
//                        block j < k: y_j -= L_kj^T x_k.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int NB = 64;
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void chol_load_kernel(const float* __restrict__ S, const float* __restrict__ y, int n,
                                                        float* __restrict__ Lw, int np) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)(np + NB) * np;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(q / np), c = (int)(q - (int64_t)r * np);
    float v = 0.f;
    if (r < n && c < n) {
      v = S[(int64_t)r * n + c];
      if (r == c) { const float t = v * 1e-4f + 1.0f; v = v + t; }       // d.add_(1e-4 * d + 1.0)
    } else if (r == c) {
      v = 1.f;
    } else if (r == np && c < n) {
      v = y[c];
    }
    Lw[q] = v;
  }
}

__device__ __forceinline__ float lane_bcast(float v, int lane) {          // lane must be a compile-time constant after unrolling
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One wave per row block.  Lane i keeps row i of the diagonal block in registers and factorises it column by column (no workgroup
// barrier: 64 fully unrolled steps); the entries above the diagonal go along as don't-cares.  The
// same wave then solves its own block against the factor: lane r keeps row r of A_ik, x_c = b_c / L_cc, b_c' -= x_c L_c'c.
// (A 256-thread version with the block in LDS and two barriers per column took 80 us per panel; this one takes ~20.)
// The panel step on rows that are already in registers: lane t holds row t of the diagonal block in `a` and row t of its own block
// in `b` (float pairs).  Shared by chol_panel_kernel (rows loaded from the work matrix) and chol_step_kernel (rows taken from the LDS
// tiles its own trailing update has just produced).  Lc: [NB][NB] floats of LDS, invs: [NB].
#define A_(c) a[(c) >> 1][(c) & 1]
#define B_(c) b[(c) >> 1][(c) & 1]
__device__ __forceinline__ void panel_body(f2 (&a)[NB / 2], f2 (&b)[NB / 2], float (*Lc)[NB], float* invs, const int t, const bool write_diag,
                                           float* __restrict__ Dg, float* __restrict__ Di, float* __restrict__ Bik, const int ld, const int k) {
  // Column j of L is published in LDS as it is made (Lc[j][i] = L_ij); the other lanes read it back as 16-byte broadcasts, four
  // multipliers per LDS instruction.  Round 5: the LDS round trip is OFF the dependent chain.  Rounds 3-4 wrote column j, waited,
  // read it back and only then updated the trailing columns -- two exposed LDS latencies per column, 250 ns x 64 columns = 16 of the
  // kernel's 19 us.  Now step j applies column j to the NEXT TWO columns through v_readlane (register to register), and the bulk
  // update with column j - 1 (columns >= j + 2), whose broadcast reads were requested at the top of the step and land under the
  // sqrt / rcp chain.  Every column still receives the updates of all earlier columns before it becomes the pivot column:
  // from j - 1 and j - 2 through the readlane path, from everything older through the bulk path at least one step earlier.
  float lprev = 0.f;
#pragma clang loop unroll(full)
  for (int j = 0; j < NB; ++j) {
    f4 v[NB / 4];
    const int q0 = (j + 2) >> 2;
    if (j > 0) {
#pragma unroll
      for (int q = 0; q < NB / 4; ++q) if (q >= q0) v[q] = *reinterpret_cast<const f4*>(&Lc[j - 1][4 * q]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the hardware square root and reciprocal, 1 ulp each: the IEEE sequences are 40 instructions per column)
    const float d = __builtin_amdgcn_sqrtf(lane_bcast(A_(j), j));
    const float inv = __builtin_amdgcn_rcpf(d);
    const float l = (t == j) ? d : A_(j) * inv;      // lanes above the diagonal: don't-care
    A_(j) = l;
    Lc[j][t] = l;
    invs[j] = inv;                                   // (the same value from every lane: no divergent store)
    if (j + 1 < NB) { const int c = j + 1 < NB ? j + 1 : 0; A_(c) -= l * lane_bcast(l, c); asm volatile("" : "+v"(a[c >> 1])); }
    if (j + 2 < NB) { const int c = j + 2 < NB ? j + 2 : 0; A_(c) -= l * lane_bcast(l, c); asm volatile("" : "+v"(a[c >> 1])); }
    __builtin_amdgcn_sched_barrier(0);
    if (j > 0) {
      const f2 lp = {lprev, lprev};
#pragma unroll
      for (int p2 = 0; p2 < NB / 2; ++p2) {
        const int c = 2 * p2;                          // columns c, c + 1; column j + 2 may be the odd one of its pair
        if (c >= j + 2) {
          a[p2] -= lp * (f2){v[p2 >> 1][2 * (p2 & 1)], v[p2 >> 1][2 * (p2 & 1) + 1]};
          asm volatile("" : "+v"(a[p2]));            // (pins the update here: LLVM otherwise sinks it to step c and keeps 2 016 loaded values alive)
        } else if (c + 1 >= j + 2) {
          a[p2][1] -= lprev * v[p2 >> 1][2 * (p2 & 1) + 1];
          asm volatile("" : "+v"(a[p2]));
        }
      }
    }
    lprev = l;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (write_diag) {
    Di[k * NB + t] = invs[t];
    f4* dst = reinterpret_cast<f4*>(Dg + (int64_t)k * NB * NB + t * NB);
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      f4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (4 * q + e <= t) ? A_(4 * q + e) : 0.f;
      dst[q] = v;
    }
  }
  // X = A_ik L_kk^-T, row r of the block per lane: x_c = b_c / L_cc, b_c' -= x_c L_c'c.  The multipliers of step c + 1 (row c + 1 of
  // Lc: the same for every lane) are requested before step c's arithmetic.
  {
    f4 vv[2][NB / 4];
    float iv[2];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) vv[0][q] = *reinterpret_cast<const f4*>(&Lc[0][4 * q]);
    iv[0] = invs[0];
#pragma clang loop unroll(full)
    for (int c = 0; c < NB; ++c) {
      if (c + 1 < NB) {
        const int q1 = (c + 2) >> 2;
#pragma unroll
        for (int q = 0; q < NB / 4; ++q) if (q >= q1) vv[(c + 1) & 1][q] = *reinterpret_cast<const f4*>(&Lc[c + 1 < NB ? c + 1 : 0][4 * q]);
        iv[(c + 1) & 1] = invs[c + 1 < NB ? c + 1 : 0];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float x = B_(c) * iv[c & 1];
      B_(c) = x;
      const f2 x2 = {x, x};
#pragma unroll
      for (int p2 = 0; p2 < NB / 2; ++p2) {
        const int c2 = 2 * p2;
        if (c2 > c) {
          b[p2] -= x2 * (f2){vv[c & 1][p2 >> 1][2 * (p2 & 1)], vv[c & 1][p2 >> 1][2 * (p2 & 1) + 1]};
          asm volatile("" : "+v"(b[p2]));
        } else if (c2 + 1 > c) {
          b[p2][1] -= x * vv[c & 1][p2 >> 1][2 * (p2 & 1) + 1];
          asm volatile("" : "+v"(b[p2]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f4* dst = reinterpret_cast<f4*>(Bik + (int64_t)t * ld);
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) { f4 v = {b[2 * q][0], b[2 * q][1], b[2 * q + 1][0], b[2 * q + 1][1]}; dst[q] = v; }
#undef A_
#undef B_
}

__global__ __launch_bounds__(64) void chol_panel_kernel(float* __restrict__ Lw, float* __restrict__ Dg, float* __restrict__ Di, int ld, int k) {
  const int t = threadIdx.x;
  const int bi = k + 1 + blockIdx.x;
  const float* Akk = Lw + ((int64_t)k * NB) * ld + k * NB;
  float* Bik = Lw + ((int64_t)bi * NB) * ld + k * NB;
  // rows as float PAIRS: the trailing updates are v_pk_fma_f32 (two columns per instruction; the 4 032 scalar FMAs of a panel were
  // 8 of its 18 us, tools/chol_bench.py under rocprofv3)
  f2 a[NB / 2], b[NB / 2];
  {
    const f4* src = reinterpret_cast<const f4*>(Akk + (int64_t)t * ld);
    const f4* srb = reinterpret_cast<const f4*>(Bik + (int64_t)t * ld);
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      const f4 v = src[q]; a[2 * q] = (f2){v[0], v[1]}; a[2 * q + 1] = (f2){v[2], v[3]};
      const f4 u = srb[q]; b[2 * q] = (f2){u[0], u[1]}; b[2 * q + 1] = (f2){u[2], u[3]};
    }
  }
  __shared__ __attribute__((aligned(16))) float Lc[NB][NB];
  __shared__ float invs[NB];
  panel_body(a, b, Lc, invs, t, blockIdx.x == 0, Dg, Di, Bik, ld, k);
}

// tile (bi, bj) of the trailing matrix minus L_{bi,k} L_{bj,k}^T, one 32 x 32 quadrant (qi, qj) per wave: acc[4 g + rr] <-> row
// 8 g + 4 kh + rr, column m of the quadrant (m = lane & 31, kh = lane >> 5)
__device__ __forceinline__ f16v tile_update(const float* __restrict__ Lw, int ld, int k, int bi, int bj, int qi, int qj, int lane) {
  const int m = lane & 31, kh = lane >> 5;
  // fragments: lane (m, kh) holds elements [32 kh, 32 kh + 32) of row m of its quadrant's 32 x 64 operand; MFMA step s
  // consumes element s of every lane (k = s from the lanes with kh = 0, k = 32 + s from the others) -- the same permutation of
  // the summation index on both operands.
  const f4* pa = reinterpret_cast<const f4*>(Lw + ((int64_t)bi * NB + 32 * qi + m) * ld + k * NB + 32 * kh);
  const f4* pb = reinterpret_cast<const f4*>(Lw + ((int64_t)bj * NB + 32 * qj + m) * ld + k * NB + 32 * kh);
  f4 a[8], bb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = pa[q]; bb[q] = pb[q]; }
  const float* C = Lw + ((int64_t)bi * NB + 32 * qi) * ld + bj * NB + 32 * qj + m;
  f16v acc;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[4 * g + rr] = C[(int64_t)(8 * g + 4 * kh + rr) * ld];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-a[q][e], bb[q][e], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ void tile_enum(int b, int r, int& i, int& j) {
  // tile enumeration: lower triangle of the r trailing blocks (rows first), then the border row (block r) against blocks 0..r-1
  const int ntri = r * (r + 1) / 2;
  if (b < ntri) {
    i = (int)((sqrtf(8.f * (float)b + 1.f) - 1.f) * 0.5f);
    while (i * (i + 1) / 2 > b) --i;
    while ((i + 1) * (i + 2) / 2 <= b) ++i;
    j = b - i * (i + 1) / 2;
  } else {
    i = r; j = b - ntri;
  }
}

// Look-ahead (round 5): ONE launch per panel instead of two.  The trailing update with panel k - 1 and the factorisation of panel k
// used to be two dependent launches (chol_update_kernel: one workgroup per 64 x 64 tile, A_ij -= L_ik L_jk^T as 32 steps of the
// 32x32x2 f32 MFMA per wave, then chol_panel_kernel); here the workgroups of the update's first tile column (tiles (i, 0), i >= 1:
// exactly the blocks panel k needs) update their own tile AND, redundantly, the diagonal tile into LDS instead of memory, and their
// wave 0 goes straight on with the panel step on those rows (panel_body); every other workgroup is the plain update.  Panel k is
// factored while the rest of update k - 1 is still running.  Same operands, same order of operations per entry as the two-launch
// path: bit-identical.  Measured against it on one box (profiles/r05_d_chol_lookahead.txt): n = 294 / 630 / 1 194 / 2 394 / 4 794:
// 123 / 250 / 489 / 1 099 / 2 972 -> 122 / 247 / 478 / 1 055 / 2 821 us -- 1-5 %: back-to-back launches of one stream already
// overlap most of a launch boundary, and the step kernel carries the panel path's 206 registers and 51 KB of LDS on every workgroup.
constexpr int TP = NB + 4;           // LDS tile pitch in floats: 16-byte aligned rows, a lane's own row conflict free for ds_read_b128
__global__ __launch_bounds__(256) void chol_step_kernel(float* __restrict__ Lw, float* __restrict__ Dg, float* __restrict__ Di, int ld, int k, int r) {
  // update with panel kp = k - 1 over its r trailing blocks; panel k = first trailing block column
  const int kp = k - 1;
  int i, j;
  tile_enum(blockIdx.x, r, i, j);
  const int bi = kp + 1 + i, bj = kp + 1 + j;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qi = w >> 1, qj = w & 1;
  const int m = lane & 31, kh = lane >> 5;
  if (j != 0) {
    if (bi == bj && qi < qj) return;
    const f16v acc = tile_update(Lw, ld, kp, bi, bj, qi, qj, lane);
    float* C = Lw + ((int64_t)bi * NB + 32 * qi) * ld + bj * NB + 32 * qj + m;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) C[(int64_t)(8 * g + 4 * kh + rr) * ld] = acc[4 * g + rr];
    return;
  }
  if (i == 0) return;                                       // the diagonal tile itself: every column workgroup forms it for itself
  __shared__ __attribute__((aligned(16))) float T0[NB][TP];          // diagonal tile (k, k) after the update
  __shared__ __attribute__((aligned(16))) float T1[NB][TP];          // own tile (bi, k) after the update
  __shared__ __attribute__((aligned(16))) float Lc[NB][NB];
  __shared__ float invs[NB];
  {
    const f16v acc = tile_update(Lw, ld, kp, bi, k, qi, qj, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) T1[32 * qi + 8 * g + 4 * kh + rr][32 * qj + m] = acc[4 * g + rr];
  }
  if (qi >= qj) {
    const f16v acc = tile_update(Lw, ld, kp, k, k, qi, qj, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) T0[32 * qi + 8 * g + 4 * kh + rr][32 * qj + m] = acc[4 * g + rr];
  }
  __syncthreads();
  if (w != 0) return;
  const int t = lane;
  f2 a[NB / 2], b[NB / 2];
#pragma unroll
  for (int q = 0; q < NB / 4; ++q) {
    const f4 v = *reinterpret_cast<const f4*>(&T0[t][4 * q]); a[2 * q] = (f2){v[0], v[1]}; a[2 * q + 1] = (f2){v[2], v[3]};
    const f4 u = *reinterpret_cast<const f4*>(&T1[t][4 * q]); b[2 * q] = (f2){u[0], u[1]}; b[2 * q + 1] = (f2){u[2], u[3]};
  }
  panel_body(a, b, Lc, invs, t, i == 1, Dg, Di, Lw + ((int64_t)bi * NB) * ld + k * NB, ld, k);
}

__global__ __launch_bounds__(256) void chol_back_kernel(const float* __restrict__ Lw, const float* __restrict__ Dg, const float* __restrict__ Di, int ld, float* __restrict__ yv,
                                                        float* __restrict__ x_out, int n, int k) {
  __shared__ float A[NB][NB + 1];
  __shared__ float xs[NB];
  __shared__ float part[4][NB];
  const int t = threadIdx.x;
  for (int idx = t; idx < NB * NB; idx += 256) A[idx >> 6][idx & 63] = Dg[(int64_t)k * NB * NB + idx];
  __syncthreads();
  if (t < NB) {
    float yl = yv[k * NB + t], x = 0.f;
    const float il = Di[k * NB + t];
#pragma unroll
    for (int c = NB - 1; c >= 0; --c) {
      const float xc = lane_bcast(yl, c) * lane_bcast(il, c);
      if (t < c) yl -= A[c][t] * xc;
      if (t == c) x = xc;
    }
    xs[t] = x;
    if (blockIdx.x == 0 && k * NB + t < n) x_out[k * NB + t] = x;
  }
  __syncthreads();
  if ((int)blockIdx.x < k) {
    const int j = blockIdx.x, q = t >> 6, c = t & 63;
    const float* Lkj = Lw + ((int64_t)k * NB + 16 * q) * ld + j * NB + c;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += Lkj[(int64_t)r * ld] * xs[16 * q + r];
    part[q][c] = s;
    __syncthreads();
    if (t < NB) yv[j * NB + t] -= (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  }
}

// The whole back substitution as ONE launch of one workgroup (round 5), for systems of up to kBackAllMax block columns -- the bench leg's
// global BA has 10-11 (profiles/r05_e_lc_timeline.txt: 11 chol_back_kernel launches x 7 us per solve, each a dependent launch whose
// work is a 64-step chain in one wave and a few 64 x 64 products).  8 waves: per block column k, wave 0 solves x_k = L_kk^-T y_k exactly
// as chol_back_kernel does, then the two 256-thread groups take the blocks j < k (j = group, group + 2, ...): y_j -= L_kj^T x_k with the same
// four 16-row partial sums in the same order -- bit-identical to the launch-per-column path (tests/test_gpu_chol.py compares the two).
// y lives in LDS for the whole solve; the operands of ALL of a column's blocks are requested before the first is consumed (one
// memory round trip per column), and the next column's diagonal factor is fetched under the products.
constexpr int kBackAllMax = 12, kBackGroups = 2, kBackPer = kBackAllMax / kBackGroups;
__global__ __launch_bounds__(256 * kBackGroups) void chol_back_all_kernel(const float* __restrict__ Lw, const float* __restrict__ Dg,
                                                                          const float* __restrict__ Di, int ld, const float* __restrict__ yv,
                                                                          float* __restrict__ x_out, int n, int nb) {
  constexpr int NT = 256 * kBackGroups, PER = NB * NB / NT;
  __shared__ float A[NB][NB + 1];
  __shared__ float xs[NB];
  __shared__ float part[kBackGroups][4][NB];
  __shared__ float ys[kBackAllMax * NB];
  const int t = threadIdx.x, g = t >> 8, tt = t & 255, q = tt >> 6, c = tt & 63;
  for (int i = t; i < nb * NB; i += NT) ys[i] = yv[i];
  float an[PER];                                                 // the diagonal factor of the column about to be solved
#pragma unroll
  for (int u = 0; u < PER; ++u) an[u] = Dg[(int64_t)(nb - 1) * NB * NB + t + NT * u];
  for (int k = nb - 1; k >= 0; --k) {
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int idx = t + NT * u; A[idx >> 6][idx & 63] = an[u]; }
    __syncthreads();
    if (k > 0) {
#pragma unroll
      for (int u = 0; u < PER; ++u) an[u] = Dg[(int64_t)(k - 1) * NB * NB + t + NT * u];
    }
    // operands of this column's products (they do not depend on x_k): up to kBackPer blocks per group, 16 rows each
    float lv[kBackPer][16];
#pragma unroll
    for (int b = 0; b < kBackPer; ++b) {
      const int j = g + kBackGroups * b;
      if (j < k) {
        const float* Lkj = Lw + ((int64_t)k * NB + 16 * q) * ld + j * NB + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) { lv[b][r] = *Lkj; Lkj += ld; }
      }
    }
    if (t < NB) {
      float yl = ys[k * NB + t], x = 0.f;
      const float il = Di[k * NB + t];
#pragma unroll
      for (int cc = NB - 1; cc >= 0; --cc) {
        const float xc = lane_bcast(yl, cc) * lane_bcast(il, cc);
        if (t < cc) yl -= A[cc][t] * xc;
        if (t == cc) x = xc;
      }
      xs[t] = x;
      if (k * NB + t < n) x_out[k * NB + t] = x;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < kBackPer; ++b) {
      const int j = g + kBackGroups * b;
      if (kBackGroups * b < k) {                                  // (uniform over the workgroup: the barriers below are taken by everyone)
        if (j < k) {
          float s = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) s += lv[b][r] * xs[16 * q + r];
          part[g][q][c] = s;
        }
        __syncthreads();
        if (j < k && tt < NB) ys[j * NB + tt] -= (part[g][0][tt] + part[g][1][tt]) + (part[g][2][tt] + part[g][3][tt]);
        __syncthreads();
      }
    }
  }
}

}  // namespace

extern "C" size_t dpvo_gba_solve_workspace_bytes(int n) {
  const size_t np = (size_t)((n > 0 ? n : 1) + NB - 1) / NB * NB;
  return ((np + NB) * np + np * NB + np) * sizeof(float);
}

extern "C" int dpvo_gba_solve(const float* S, const float* y, int n, float* dX, void* ws, size_t ws_bytes,
                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n <= 0 || !S || !y || !dX || !ws) return DPVO_E_INVALID;
  if (ws_bytes < dpvo_gba_solve_workspace_bytes(n)) return DPVO_E_WORKSPACE;
  const int nb = (n + NB - 1) / NB, np = nb * NB;
  float* Lw = (float*)ws;
  float* Dg = Lw + (int64_t)(np + NB) * np;               // the diagonal blocks' factors [nb][64][64]
  float* Di = Dg + (int64_t)np * NB;                       // 1 / L_jj
  const int64_t total = (int64_t)(np + NB) * np;
  const int lg = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(chol_load_kernel, dim3(lg), dim3(256), 0, stream, S, y, n, Lw, np);
  // panel 0 by itself, then ONE launch per panel: the trailing update with panel k - 1 and, in its first tile column, panel k
  hipLaunchKernelGGL(chol_panel_kernel, dim3(nb), dim3(64), 0, stream, Lw, Dg, Di, np, 0);
  for (int k = 1; k < nb; ++k) {
    const int r = nb - k;                                   // trailing blocks of panel k - 1
    hipLaunchKernelGGL(chol_step_kernel, dim3(r * (r + 1) / 2 + r), dim3(256), 0, stream, Lw, Dg, Di, np, k, r);
  }
  float* yv = Lw + (int64_t)np * np;
  // back substitution: one launch of one workgroup for small systems, else one launch per block column (same operations in the same
  // order: round 5 compared the two bit for bit on n = 6 .. 768 before the switch that forced the latter was removed)
  if (nb <= kBackAllMax)
    hipLaunchKernelGGL(chol_back_all_kernel, dim3(1), dim3(256 * kBackGroups), 0, stream, Lw, Dg, Di, np, yv, dX, n, nb);
  else
    for (int k = nb - 1; k >= 0; --k)
      hipLaunchKernelGGL(chol_back_kernel, dim3(k > 0 ? k : 1), dim3(256), 0, stream, Lw, Dg, Di, np, yv, dX, n, k);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
