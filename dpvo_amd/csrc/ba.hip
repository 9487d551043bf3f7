// ba.hip -- fastba bundle adjustment for gfx950 (dense Schur path, N = t1 - t0 <= 20 free poses).
//
// Replaces cuda_ba.forward (reference dpvo/fastba/ba_cuda.cu:433-582): the per-edge residual / Jacobian
// kernel with ~340 float atomics per edge onto a 60x60 tile (:232-376), ~25 ATen launches + cuSOLVER
// potrf/potrs per iteration (:519-565) and the two retraction kernels (:178-229).
//
// MI355X design -- five short launches per Gauss-Newton iteration, no float atomics, bit-reproducible:
//   1. ba_pair_kernel     two waves per (i,j) frame pair (CSR from the plan; ~96 edges): per-edge residuals and
//                         Jacobians; the pair's 13x13 Gram block  sum_rows w [Ji Jj r]^T [Ji Jj r]  (= Hii, Hij, Hjj,
//                         vi, vj) is reduced on the matrix core with v_mfma_f32_16x16x4_f32 (exact f32 fma chain)
//                         from an LDS row image -> pairbuf[g][16x16]; per-edge depth terms (c, u, Ei, Ej) -> edgebuf.
//   2. ba_patch_kernel    a block owns 32 patches (CSR by patch), 8 lanes gather one patch's edges (two per lane and trip): builds
//                         the Schur column e_k (6N) in LDS, stores Q, u, e_k, and the block's partial  sum_k Q_k e_k e_k^T,
//                         sum_k Q_k u_k e_k  (16 x 16 tiles on v_mfma_f32_16x16x4_f32) -> spart[block].  The same launch carries
//                         one extra block per free pose that assembles its block row of B and v from pairbuf in a fixed order
//                         (match list + direct pair tables, ba_bpart_body) -> Bbuf: independent work.
//   3. ba_assemble_kernel six blocks per free pose, 16 lanes per entry of a row of S / y: B, v (Bbuf) minus the Schur
//                         partials (all of an entry's loads in flight together), plus the reference's damping
//                         S += I*(1e-4*S + 1)  -> Sg, yg.
//   4. ba_solve60_kernel  n6 <= 60: one workgroup, 6x6-blocked right-looking Cholesky of [S; y^T] in LDS, blocked backward
//                         substitution -> dX;  ba_solve_kernel (n6 <= 120): left-looking Cholesky in LDS (one barrier per
//                         column), column-oriented triangular solves.
//   5. ba_retr_kernel     dZ_k = Q_k (u_k - e_k . dX), depth retraction with the reference's clamps; pose
//                         retraction Exp(dX)*T.
#include "ba_common.h"

namespace {
using namespace ba;

// ---------------------------------------------------------------------------------------------------
// 2. per-patch kernel: Schur columns + partial Schur products; block = 32 patches x 8 lanes
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ba_patch_body(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                              const int32_t* __restrict__ perm_k, const int32_t* __restrict__ patch_off,
                                              const int32_t* __restrict__ n_patches, const float* __restrict__ edgebuf,
                                              float lmbda, int t0, int N, float* __restrict__ Qbuf, float* __restrict__ ubuf,
                                              float* __restrict__ Ecol, int64_t ldE, float* __restrict__ spart, int ch) {
  __shared__ float col[kPatchChunk][kMaxDim + 1];
  __shared__ float qv[kPatchChunk], uv[kPatchChunk];
  const int n6 = 6 * N;
  const int tid = threadIdx.x;
  // (the patch's edge range is fetched together with the patch count, not behind it: patch_off has an entry for every k a launch
  //  sized from the E upper bound can reach -- it is E + 1 ints inside the plan buffer -- and the values are only used when k < np)
  const int b0s = patch_off[ch * kPatchChunk + (tid >> 3)], b1s = patch_off[ch * kPatchChunk + (tid >> 3) + 1];
  const int np = *n_patches;
  const int nent = n6 * n6 + n6;                       // S entries followed by y entries
  for (int a = tid; a < kPatchChunk * (kMaxDim + 1); a += 256) (&col[0][0])[a] = 0.f;
  __syncthreads();
  {
    const int pl = tid >> 3, sub = tid & 7;
    const int k = ch * kPatchChunk + pl;
    float C = 0.f, u = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0};
    int ix = -1;
    if (k < np) {
      const int b0 = b0s, b1 = b1s;
      // two edges per lane and trip (a patch has ~14 edges: one trip of the offset -> edge id -> record chain instead of two);
      // the sums keep the order p, p + 8
      for (int p = b0 + sub; p < b1; p += 16) {
        const bool two = p + 8 < b1;
        const int e0 = perm_k[p], e1 = two ? perm_k[p + 8] : e0;
        const f4* eb0 = reinterpret_cast<const f4*>(edgebuf + (int64_t)e0 * kEdgeStride);
        const f4* eb1 = reinterpret_cast<const f4*>(edgebuf + (int64_t)e1 * kEdgeStride);
        const f4 a0 = eb0[0], a1 = eb0[1], a2 = eb0[2], a3 = eb0[3];
        const f4 c0 = eb1[0], c1 = eb1[1], c2 = eb1[2], c3 = eb1[3];
        const int i0 = (int)ii[e0], j0 = (int)jj[e0], i1 = (int)ii[e1], j1 = (int)jj[e1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !two) break;
          const f4 v0 = h ? c0 : a0, v1 = h ? c1 : a1, v2 = h ? c2 : a2, v3 = h ? c3 : a3;
          C += v0[0]; u += v0[1];
          Ei[0] += v0[2]; Ei[1] += v0[3]; Ei[2] += v1[0]; Ei[3] += v1[1]; Ei[4] += v1[2]; Ei[5] += v1[3];
          ix = (h ? i1 : i0) - t0;
          const int jx = (h ? j1 : j0) - t0;
          if (jx >= 0 && jx < N) {
            // distinct (patch, j) edges own distinct slots; duplicates (if any) are folded by the LDS add
            atomicAdd(&col[pl][6 * jx + 0], v2[0]); atomicAdd(&col[pl][6 * jx + 1], v2[1]);
            atomicAdd(&col[pl][6 * jx + 2], v2[2]); atomicAdd(&col[pl][6 * jx + 3], v2[3]);
            atomicAdd(&col[pl][6 * jx + 4], v3[0]); atomicAdd(&col[pl][6 * jx + 5], v3[1]);
          }
        }
      }
    }
    // fixed-order butterfly over the 8 lanes of a patch
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      C += __shfl_xor(C, o); u += __shfl_xor(u, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) Ei[a] += __shfl_xor(Ei[a], o);
      ix = max(ix, __shfl_xor(ix, o));
    }
    __syncthreads();      // all Ej adds of this block are done before the i-side is merged
    if (sub == 0) {
      if (k < np) {
        if (ix >= 0 && ix < N)
#pragma unroll
          for (int a = 0; a < 6; ++a) col[pl][6 * ix + a] += Ei[a];
        const float Q = 1.0f / (C + lmbda);               // ba_cuda.cu:519
        qv[pl] = Q; uv[pl] = u;
        Qbuf[k] = Q; ubuf[k] = u;
      } else {
        qv[pl] = 0.f; uv[pl] = 0.f;
      }
    }
  }
  __syncthreads();
  // store the columns for the dZ back-substitution, row-major [n6][ldE] (coalesced over patches)
  for (int a = tid; a < kPatchChunk * n6; a += 256) {
    const int r = a / kPatchChunk, pl = a - r * kPatchChunk;
    const int k = ch * kPatchChunk + pl;
    if (k < np) Ecol[(int64_t)r * ldE + k] = col[pl][r];
  }
  // partial S and y of this block on the matrix core:  [S | y] (n6 x (n6 + 1)) = col^T (Q col | Q u),  16 x 16 tiles, K = the 32
  // patches in 8 steps of v_mfma_f32_16x16x4_f32 (an exact f32 fma chain in patch order).  (As 3 660 scalar dot products of
  // length 32 this was 350 k LDS reads per block: 5 of the kernel's 21 us.)
  float* sp = spart + (int64_t)ch * kSEntries;
  {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int nti = (n6 + 15) >> 4, ntj = (n6 + 1 + 15) >> 4;
    for (int T = wave; T < nti * ntj; T += 4) {
      const int ti = T / ntj, tj = T - ti * ntj;
      const int ra = 16 * ti + li, cb = 16 * tj + li;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < kPatchChunk / 4; ++t) {
        const int pl = 4 * t + lk;
        const float a = ra < n6 ? col[pl][ra] : 0.f;
        const float q = qv[pl];
        const float b = cb < n6 ? q * col[pl][cb] : (cb == n6 ? q * uv[pl] : 0.f);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * lk + r;
        if (row < n6) {
          if (cb < n6) sp[row * n6 + cb] = acc[r];
          else if (cb == n6) sp[n6 * n6 + row] = acc[r];
        }
      }
    }
  }
  (void)nent;
}

// ---------------------------------------------------------------------------------------------------
// 2b. B part: one block per free pose p = block row p of B (6 x n6) and v[6p..6p+5]
//    wave 0: diagonal block + y: lanes stride over the pairs, per-lane partial sums, fixed-order butterfly;
//    waves 1..3: off-diagonal blocks (two binary searches in the LDS copy of the sorted pair list);
//    then all threads: subtract the Schur partials (4 lanes per entry) and apply the damping.
// ---------------------------------------------------------------------------------------------------
constexpr int kDiagSlices = 6;       // 42 entries x 6 interleaved slices of the match list = 252 threads
constexpr int kMaxMatch = 256;       // pairs that touch one pose (<= 2 x the frames a pose shares edges with; ~50 in the tracker)

// B part of block row p (and v of pose p) from the pair blocks, in a fixed order -> Bbuf[p][6][kMaxDim + 1] (column kMaxDim
// holds v).  Independent of the per-patch kernel, whose launch it shares.
//   phase 0  one scan of the pair list: the pairs with i == f or j == f go into a match list in pair order (wave ballots: the
//            order is the pair order whatever the timing), and into two direct tables tabA[q] / tabB[q] = the pair (f, t0 + q) /
//            (t0 + q, f) (round 3 found them by two binary searches per matrix entry: 18 dependent LDS reads);
//   phase 1  the diagonal block + v as 42 entries x 6 interleaved slices of the match list (252 threads, four matches per trip),
//            the off-diagonal blocks two entries per thread;
//   phase 2  the six slices added in order, the row stored.
// (Round 3's version -- per-thread 42-entry partial sums over a strided pair scan, 42 wave reductions, binary searches -- was the
//  longest workgroup of the launch: 17 us against 7-12 for the per-patch blocks, tools/ba_trace.sh.)
__device__ __forceinline__ void ba_bpart_body(const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ n_pairs,
                                              const float* __restrict__ pairbuf, int t0, int N, int p, float* __restrict__ Bbuf) {
  __shared__ int mlist[kMaxMatch], mflag[kMaxMatch];
  __shared__ int tabA[kMaxN], tabB[kMaxN];
  __shared__ int wcnt[4], s_nm;
  __shared__ float rowS[6][kMaxDim];      // block row p of B
  __shared__ float dsl[kDiagSlices][42];  // diagonal block (36) + v (6), one row per slice of the match list
  const int n6 = 6 * N;
  const int ng = *n_pairs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = t0 + p;
  const int2* pg = reinterpret_cast<const int2*>(pair_ij);
  if (tid < kMaxN) { tabA[tid] = -1; tabB[tid] = -1; }
  if (tid == 0) s_nm = 0;
  for (int a = tid; a < 6 * kMaxDim; a += 256) (&rowS[0][0])[a] = 0.f;
  __syncthreads();
  // ---- phase 0
  for (int g0 = 0; g0 < ng; g0 += 256) {
    const int g = g0 + tid;
    int2 ij = {-1, -1};
    if (g < ng) ij = pg[g];
    const bool hit = g < ng && (ij.x == f || ij.y == f);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(bal);
    __syncthreads();
    int base = s_nm;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    if (hit) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1));
      if (pos < kMaxMatch) { mlist[pos] = g; mflag[pos] = (ij.x == f ? 1 : 0) | (ij.y == f ? 2 : 0); }
      if (ij.x == f) { const int q = ij.y - t0; if (q >= 0 && q < N) tabA[q] = g; }
      if (ij.y == f) { const int q = ij.x - t0; if (q >= 0 && q < N) tabB[q] = g; }
    }
    __syncthreads();
    if (tid == 0) s_nm += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int nm = s_nm;
  if (nm <= kMaxMatch) {
    // ---- phase 1: diagonal block + v.  Thread (sl, en): slice sl = tid / 42 of the matches (m = sl, sl + 4, ...), entry en
    if (tid < 42 * kDiagSlices) {
      const int sl = tid / 42, en = tid - 42 * sl;
      const int a = en < 36 ? en / 6 : en - 36, b = en < 36 ? en - 6 * (en / 6) : 0;
      // the entry's offset inside a pair block on the i-side / the j-side, and the two cross terms of a self edge
      const int oi = en < 36 ? a * 16 + b : a * 16 + 12, oj = en < 36 ? (6 + a) * 16 + 6 + b : (6 + a) * 16 + 12;
      const int ox1 = a * 16 + 6 + b, ox2 = b * 16 + 6 + a;
      const float si = en < 36 ? 1.f : -1.f;                                      // v -= w r Ji, v += w r Jj
      float acc = 0.f;
      for (int m0 = sl; m0 < nm; m0 += 4 * kDiagSlices) {                         // four matches per trip, their loads in flight together
        float vi[4], vj[4], vx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + u * kDiagSlices;
          vi[u] = vj[u] = vx[u] = 0.f;
          if (m < nm) {
            const int g = mlist[m], fl = mflag[m];
            const float* pb = pairbuf + (int64_t)g * kPairStride;
            if (fl & 1) vi[u] = pb[oi];
            if (fl & 2) vj[u] = pb[oj];
            if (fl == 3 && en < 36) vx[u] = -pb[ox1] - pb[ox2];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc += si * vi[u]; acc += vj[u]; acc += vx[u]; }
      }
      dsl[sl][en] = acc;
    }
  } else {
    // (more pairs at one pose than the list holds: one thread per entry walks the whole pair list)
    if (tid < 42) {
      const int en = tid, a = en < 36 ? en / 6 : en - 36, b = en < 36 ? en - 6 * (en / 6) : 0;
      float acc = 0.f;
      for (int g = 0; g < ng; ++g) {
        const int2 ij = pg[g];
        if (ij.x != f && ij.y != f) continue;
        const float* pb = pairbuf + (int64_t)g * kPairStride;
        if (en < 36) {
          if (ij.x == f) acc += pb[a * 16 + b];
          if (ij.y == f) acc += pb[(6 + a) * 16 + 6 + b];
          if (ij.x == f && ij.y == f) acc += -pb[a * 16 + 6 + b] - pb[b * 16 + 6 + a];
        } else {
          if (ij.x == f) acc -= pb[a * 16 + 12];
          if (ij.y == f) acc += pb[(6 + a) * 16 + 12];
        }
      }
      dsl[0][en] = acc;
      for (int k = 1; k < kDiagSlices; ++k) dsl[k][en] = 0.f;
    }
  }
  // off-diagonal blocks: entry (q, a, b) of block row p gets  -w Ji Jj^T  of the pair (p, q) and  -w Jj Ji^T  of the pair (q, p)   (:342-345)
  for (int ent = tid; ent < (N - 1) * 36; ent += 256) {
    int q = ent / 36; const int r = ent - q * 36;
    if (q >= p) q += 1;
    const int a = r / 6, b = r - a * 6;
    const int g1 = tabA[q], g2 = tabB[q];
    float sv = 0.f;
    if (g1 >= 0) sv += -pairbuf[(int64_t)g1 * kPairStride + a * 16 + 6 + b];
    if (g2 >= 0) sv += -pairbuf[(int64_t)g2 * kPairStride + b * 16 + 6 + a];
    rowS[a][6 * q + b] = sv;
  }
  __syncthreads();
  // ---- phase 2
  float* dst = Bbuf + (int64_t)p * 6 * (kMaxDim + 1);
  if (tid < 42) {
    float sv = dsl[0][tid];
#pragma unroll
    for (int k = 1; k < kDiagSlices; ++k) sv += dsl[k][tid];
    if (tid < 36) rowS[tid / 6][6 * p + tid % 6] = sv; else dst[(tid - 36) * (kMaxDim + 1) + kMaxDim] = sv;
  }
  __syncthreads();
  for (int a = tid; a < 6 * n6; a += 256) { const int r = a / n6, c = a - r * n6; dst[r * (kMaxDim + 1) + c] = rowS[r][c]; }
}

#ifdef BA_TRACE
// instrumentation build (tools/ba_trace.sh): start / end of every workgroup of the last ba_patch_kernel launch (100 MHz wall clock)
__device__ unsigned long long ba_trace_buf[1024][2];
#endif
__global__ __launch_bounds__(256) void ba_patch_kernel(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                       const int32_t* __restrict__ perm_k,
                                                       const int32_t* __restrict__ patch_off,
                                                       const int32_t* __restrict__ n_patches,
                                                       const float* __restrict__ edgebuf, float lmbda, int t0, int N,
                                                       float* __restrict__ Qbuf, float* __restrict__ ubuf,
                                                       float* __restrict__ Ecol, int64_t ldE, float* __restrict__ spart,
                                                       int patch_blocks, const int32_t* __restrict__ pair_ij,
                                                       const int32_t* __restrict__ n_pairs, const float* __restrict__ pairbuf,
                                                       float* __restrict__ Bbuf) {
#ifdef BA_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 1024) ba_trace_buf[blockIdx.x][0] = wall_clock64();
#endif
  if ((int)blockIdx.x < patch_blocks)
    ba_patch_body(ii, jj, perm_k, patch_off, n_patches, edgebuf, lmbda, t0, N, Qbuf, ubuf, Ecol, ldE, spart, blockIdx.x);
  else
    ba_bpart_body(pair_ij, n_pairs, pairbuf, t0, N, (int)blockIdx.x - patch_blocks, Bbuf);
#ifdef BA_TRACE
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 1024) { ba_trace_buf[blockIdx.x][1] = wall_clock64(); if (blockIdx.x == 0) ba_trace_buf[1023][0] = patch_blocks; }
#endif
}
#ifdef BA_TRACE
}  // namespace
extern "C" int dpvo_debug_ba_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ba_trace_buf), sizeof(ba_trace_buf)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// ---------------------------------------------------------------------------------------------------
// 3. assemble kernel: block (p, ra) = row 6p + ra of S and y[6p + ra]: B (from Bbuf) minus the Schur partials, damping
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ba_assemble_kernel(const float* __restrict__ Bbuf, const float* __restrict__ spart,
                                                           int n_spart, int N, float* __restrict__ Sg, float* __restrict__ yg) {
  const int n6 = 6 * N;
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const float* brow = Bbuf + ((int64_t)p * 6 + blockIdx.y) * (kMaxDim + 1);
  // Schur complement (ba_cuda.cu:557-558) + damping (:560).  blockIdx.y = row ra of the pose's block row: its n6 entries of
  // S and its entry of y, SIXTEEN lanes per entry: lane `sub` adds the partial matrices sub, sub + 16, ... (eight loads in flight
  // per trip: the ~70 partials of an entry are one round trip instead of the five of the 4-lane version), then a fixed-order
  // butterfly over the 16 lanes.
  const int ra = blockIdx.y;
  const int nrow = n6 + 1;
  const int sub = tid & 15;
  for (int e0 = 0; e0 < nrow; e0 += 64) {
    const int e4 = e0 + (tid >> 4);
    const int gent = (e4 < n6) ? (6 * p + ra) * n6 + e4 : n6 * n6 + 6 * p + ra;
    float sc = 0.f;
    if (e4 < nrow)
      for (int b0 = sub; b0 < n_spart; b0 += 128) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + 16 * u;
          v[u] = b < n_spart ? spart[(int64_t)b * kSEntries + gent] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sc += v[u];
      }
    sc += __shfl_xor(sc, 1);
    sc += __shfl_xor(sc, 2);
    sc += __shfl_xor(sc, 4);
    sc += __shfl_xor(sc, 8);
    if (e4 < nrow && sub == 0) {
      if (e4 < n6) {
        float sv = brow[e4] - sc;
        if (6 * p + ra == e4) sv += 1e-4f * sv + 1.0f;                // S += I * (1e-4 * S + 1.0)
        Sg[gent] = sv;
      } else {
        yg[6 * p + ra] = brow[kMaxDim] - sc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 4a. solve kernel for n6 <= 60 (the default window: N <= 10 free poses): blocked right-looking Cholesky of the
//     AUGMENTED matrix [S; y^T] in LDS with 6x6 pose blocks, one workgroup of 256 threads.  Per block step
//       panel    (wave 0, one lane per remaining row, the y row included): the diagonal block is read by every lane
//                and factorised redundantly, each row below solves  x L_bb^T = A[r][block]  in registers;
//       trailing (all four waves): A[r][c] -= <A[r][block], A[c][block]> for the columns to the right.
//     Carrying y along as row n6 makes the forward substitution part of the factorisation; the backward substitution is
//     blocked the same way (lane c owns z[c]; a 6x6 triangular solve per block, then one rank-6 update of the lanes to
//     the left).  ~3.5 k instructions on the longest wave and 2 barriers per block step, instead of one wave issuing
//     9.5 k straight-line instructions (the previous register-resident version: 27 us, and 100 us on boxes whose
//     shader clock stays low while a single wave runs).  Loops are rolled: the kernel is 4 KB of code.
// ---------------------------------------------------------------------------------------------------
constexpr int kLd60 = 62;      // row stride of the 61 x 60 LDS matrix in floats (8-byte aligned rows)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef BA_TRACE
__device__ unsigned long long ba_solve_trace[40];
#define SVT(i) do { if (threadIdx.x == 0) ba_solve_trace[i] = wall_clock64(); } while (0)
#else
#define SVT(i) do {} while (0)
#endif
__device__ __forceinline__ void solve60_body(const float* __restrict__ Sg, const float* __restrict__ yg, int N, float* __restrict__ dX,
                                             int32_t* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) float A[61 * kLd60];
  __shared__ float Lbb[10][28];              // per block: L_bb (lower, packed by rows: 21) and 1 / diag (6)
  __shared__ float zs[64];
  __shared__ int s_bad;
  const int n6 = 6 * N, tid = threadIdx.x;
  SVT(0);
  // lower triangle of S and y as row n6: thread (ty, tx) of a 16 x 16 grid owns the elements (ty + 16 i, tx + 16 j); all 16
  // loads are issued before the first LDS store (as a nested loop they were 16 dependent global round trips)
  {
    const int ty = tid >> 4, tx = tid & 15;
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ty + 16 * i, c = tx + 16 * j;
        const bool on = r <= n6 && c < n6 && (c <= r);
        v[i][j] = on ? (r < n6 ? Sg[r * n6 + c] : yg[c]) : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ty + 16 * i, c = tx + 16 * j;
        if (r <= n6 && c < n6 && c <= r) A[r * kLd60 + c] = v[i][j];
      }
  }
  if (tid == 0) s_bad = 0;
  __syncthreads();
  SVT(1);
  for (int B = 0; B < N; ++B) {
    const int o = 6 * B;
    const int r = o + tid;
    if (r <= n6) {                             // panel: rows o..n6 (n6 - o + 1 <= 61 lanes of wave 0)
      // every LDS operand first (27 independent reads), then the dependent arithmetic: interleaved, each read was a
      // ~100-cycle round trip inside the factorisation chain
      float D[6][6], x[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int c = 0; c <= i; ++c) D[i][c] = A[(o + i) * kLd60 + o + c];
      float* ar = A + r * kLd60 + o;
      const bool below = r >= o + 6;
#pragma unroll
      for (int c = 0; c < 6; ++c) x[c] = below ? ar[c] : 0.f;
      float Lb[6][6], inv[6];
      int bad = 0;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float d = D[c][c];
#pragma unroll
        for (int k = 0; k < c; ++k) d -= Lb[c][k] * Lb[c][k];
        if (!(d > 0.f) && bad == 0) bad = o + c + 1;
        inv[c] = __builtin_amdgcn_rsqf(d);           // 1 ulp; the pivot itself is only needed as d * rsq(d)
        Lb[c][c] = d * inv[c];
#pragma unroll
        for (int r2 = c + 1; r2 < 6; ++r2) {
          float v = D[r2][c];
#pragma unroll
          for (int k = 0; k < c; ++k) v -= Lb[r2][k] * Lb[c][k];
          Lb[r2][c] = v * inv[c];
        }
      }
      if (below) {                             // rows below the block (nobody writes the block's own rows in this step)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          float v = x[c];
#pragma unroll
          for (int k = 0; k < c; ++k) v -= x[k] * Lb[c][k];
          x[c] = v * inv[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) ar[c] = x[c];
      }
      if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
          for (int c = 0; c <= i; ++c) Lbb[B][i * (i + 1) / 2 + c] = Lb[i][c];
          Lbb[B][21 + i] = inv[i];
        }
        if (bad && s_bad == 0) s_bad = bad;
      }
    }
    if (B < 10) SVT(2 + 3 * B);
    __syncthreads();
    if (B < 10) SVT(3 + 3 * B);
    // trailing update: rows rr (0..m, row m = y), columns cc <= min(rr, m - 1), relative to o + 6.  Thread (ty, tx) of a
    // 16 x 16 grid owns the elements (ty + 16 i, tx + 16 j): all its operand rows are fetched first (independent 8-byte LDS
    // reads), then the dot products, then the read-modify-writes.  Only the 16 x 16 blocks that exist at this step and touch the
    // lower triangle are visited (j <= i, 16 i <= m, 16 j < m: uniform branches) -- visiting all sixteen with predicates made
    // this phase 0.8-1.1 us of every block step whatever m, 8 of the kernel's 19 us (tools/ba_trace.sh); 42 of the 144 blocks of a
    // 60 x 60 solve are left.
    const int m = n6 - o - 6;
    if (m > 0) {
      const float* base = A + (o + 6) * kLd60 + o;
      const int ty = tid >> 4, tx = tid & 15;
      typedef float f2v __attribute__((ext_vector_type(2)));
      f2v ar[4][3], ac[4][3];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (16 * i <= m) {
          const int rr = ty + 16 * i;
          const f2v* pr = reinterpret_cast<const f2v*>(base + (rr <= m ? rr : m) * kLd60);       // (rows and o are even: 8-byte aligned)
#pragma unroll
          for (int k = 0; k < 3; ++k) ar[i][k] = pr[k];
        }
        if (16 * i < m) {
          const int cc = tx + 16 * i;
          const f2v* pc = reinterpret_cast<const f2v*>(base + (cc < m ? cc : m - 1) * kLd60);
#pragma unroll
          for (int k = 0; k < 3; ++k) ac[i][k] = pc[k];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          if (16 * i <= m && 16 * j < m) {
            const int rr = ty + 16 * i, cc = tx + 16 * j;
            if (rr <= m && cc < m && cc <= rr) {
              float* el = A + (o + 6 + rr) * kLd60 + o + 6 + cc;
              const float sum = ar[i][0][0] * ac[j][0][0] + ar[i][0][1] * ac[j][0][1] + ar[i][1][0] * ac[j][1][0] + ar[i][1][1] * ac[j][1][1] +
                                ar[i][2][0] * ac[j][2][0] + ar[i][2][1] * ac[j][2][1];
              *el = *el - sum;
            }
          }
        }
    }
    __syncthreads();
    if (B < 10) SVT(4 + 3 * B);
  }
  // backward substitution L^T x = z (z = row n6), wave 0, lane c owns z[c]
  if (tid < 64) {
    float z = (tid < n6) ? A[n6 * kLd60 + tid] : 0.f;
    float xv = 0.f;
    for (int B = N - 1; B >= 0; --B) {
      const int o = 6 * B;
      zs[tid] = z;
      wave_lds_sync();
      // operands first (the block's factor, its right-hand side, this lane's column of the block rows), then the chain
      float Lv[27], sv[6], col[6];
#pragma unroll
      for (int q = 0; q < 27; ++q) Lv[q] = Lbb[B][q];
#pragma unroll
      for (int c = 0; c < 6; ++c) { sv[c] = zs[o + c]; col[c] = (tid < o) ? A[(o + c) * kLd60 + tid] : 0.f; }
      float x[6];
#pragma unroll
      for (int c = 5; c >= 0; --c) {
        float v = sv[c];
#pragma unroll
        for (int k = c + 1; k < 6; ++k) v -= Lv[k * (k + 1) / 2 + c] * x[k];
        x[c] = v * Lv[21 + c];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) xv = (tid == o + k) ? x[k] : xv;
#pragma unroll
      for (int k = 0; k < 6; ++k) z -= col[k] * x[k];
      wave_lds_sync();
    }
    if (tid < n6) dX[tid] = xv;
    if (tid == 0 && info) *info = s_bad;
    SVT(32);
  }
}
__global__ __launch_bounds__(256) void ba_solve60_kernel(const float* __restrict__ Sg, const float* __restrict__ yg, int N,
                                                         float* __restrict__ dX, int32_t* __restrict__ info) {
  solve60_body(Sg, yg, N, dX, info);
}
#ifdef BA_TRACE
}  // namespace
extern "C" int dpvo_debug_ba_solve_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ba_solve_trace), sizeof(ba_solve_trace)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// ---------------------------------------------------------------------------------------------------
// 4. solve kernel: one workgroup, thread r owns row r.  n6 <= 64: a single wave, no barriers at all
//    (LDS traffic of one wave is program-ordered; pivots travel by v_readlane); n6 <= 120: two waves.
// ---------------------------------------------------------------------------------------------------
// 4b. generic solve kernel (n6 <= 120)
constexpr int kLdS = kMaxDim + 4;       // 124: 16-byte aligned rows, conflict-free for ds_read_b128

template <bool ONE_WAVE>
__device__ __forceinline__ void blk_sync() {
  if constexpr (ONE_WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <bool ONE_WAVE>
__global__ __launch_bounds__(ONE_WAVE ? 64 : 128) void ba_solve_kernel(const float* __restrict__ Sg,
                                                                        const float* __restrict__ yg, int N,
                                                                        float* __restrict__ dX,
                                                                        int32_t* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) float S[kMaxDim * kLdS];
  __shared__ float xs[kMaxDim];
  constexpr int NT = ONE_WAVE ? 64 : 128;
  const int n6 = 6 * N;
  const int tid = threadIdx.x;
  for (int a = tid; a < n6 * n6; a += NT) S[(a / n6) * kLdS + (a % n6)] = Sg[a];
  blk_sync<ONE_WAVE>();
  int bad = 0;
  const float* rowt = S + tid * kLdS;
  // left-looking Cholesky (lower): L[t][j] = (S[t][j] - sum_k L[t][k] L[j][k]) / L[j][j]; every thread recomputes
  // the pivot from the same operands in the same order -> identical value, one sync per column.
  for (int j = 0; j < n6; ++j) {
    const float* rowj = S + j * kLdS;
    float d = rowj[j];
    float v = (tid < n6) ? rowt[j] : 0.f;
    int k = 0;
    for (; k + 4 <= j; k += 4) {
      const f4 a = *reinterpret_cast<const f4*>(rowj + k);
      const f4 b = (tid < n6) ? *reinterpret_cast<const f4*>(rowt + k) : (f4)0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { d -= a[r] * a[r]; v -= b[r] * a[r]; }
    }
    for (; k < j; ++k) {
      const float a = rowj[k];
      d -= a * a;
      if (tid < n6) v -= rowt[k] * a;
    }
    if (!(d > 0.f) && bad == 0) bad = j + 1;                       // linalg_cholesky_ex info (ignored by the reference)
    const float djj = sqrtf(d);
    blk_sync<ONE_WAVE>();                                          // all reads of column j / row j done
    if (tid > j && tid < n6) S[tid * kLdS + j] = v / djj;
    if (tid == j) S[j * kLdS + j] = djj;
    blk_sync<ONE_WAVE>();
  }
  // triangular solves, column oriented: thread r owns b_r
  float b = (tid < n6) ? yg[tid] : 0.f;
  if constexpr (ONE_WAVE) {
    for (int k = 0; k < n6; ++k) {                                 // L z = y
      const float xk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), k)) / S[k * kLdS + k];
      if (tid == k) b = xk;
      if (tid > k && tid < n6) b -= rowt[k] * xk;
    }
    for (int k = n6 - 1; k >= 0; --k) {                            // L^T x = z
      const float xk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), k)) / S[k * kLdS + k];
      if (tid == k) b = xk;
      if (tid < k) b -= S[k * kLdS + tid] * xk;
    }
    if (tid < n6) dX[tid] = b;
  } else {
    for (int k = 0; k < n6; ++k) {
      if (tid == k) xs[k] = b / S[k * kLdS + k];
      __syncthreads();
      if (tid > k && tid < n6) b -= rowt[k] * xs[k];
    }
    __syncthreads();
    b = (tid < n6) ? xs[tid] : 0.f;
    __syncthreads();
    for (int k = n6 - 1; k >= 0; --k) {
      if (tid == k) xs[k] = b / S[k * kLdS + k];
      __syncthreads();
      if (tid < k) b -= S[k * kLdS + tid] * xs[k];
    }
    __syncthreads();
    if (tid < n6) dX[tid] = xs[tid];
  }
  if (tid == 0 && info) *info = bad;
}

// ---------------------------------------------------------------------------------------------------
// 5. depth back-substitution + retractions (patch_retr_kernel :209-229, dZ :563; pose_retr_kernel :178-206)
// ---------------------------------------------------------------------------------------------------
__global__ void ba_retr_kernel(float* __restrict__ poses, float* __restrict__ patches, const int32_t* __restrict__ kx,
                               const int32_t* __restrict__ n_patches, const float* __restrict__ Qbuf,
                               const float* __restrict__ ubuf, const float* __restrict__ Ecol, int64_t ldE,
                               const float* __restrict__ dX, int t0, int N, int P) {
  __shared__ float sdx[kMaxDim];
  const int np = *n_patches;
  const int PP = P * P, n6 = 6 * N;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  for (int a = threadIdx.x; a < n6; a += blockDim.x) sdx[a] = dX[a];
  __syncthreads();
  for (int k = gt; k < np; k += gridDim.x * blockDim.x) {
    // four summation chains (a mod 4), the column fetched 20 entries at a time: only 54 waves exist here, so the kernel's
    // time is its number of dependent round trips (3 for the usual n6 = 60; it was 15 with 4 loads per trip, 12 us)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int a = 0;
    for (; a + 20 <= n6; a += 20) {
      float e[20];
#pragma unroll
      for (int u = 0; u < 20; ++u) e[u] = Ecol[(int64_t)(a + u) * ldE + k];
#pragma unroll
      for (int u = 0; u < 20; u += 4) {
        s0 += e[u + 0] * sdx[a + u + 0];
        s1 += e[u + 1] * sdx[a + u + 1];
        s2 += e[u + 2] * sdx[a + u + 2];
        s3 += e[u + 3] * sdx[a + u + 3];
      }
    }
    for (; a + 4 <= n6; a += 4) {
      s0 += Ecol[(int64_t)(a + 0) * ldE + k] * sdx[a + 0];
      s1 += Ecol[(int64_t)(a + 1) * ldE + k] * sdx[a + 1];
      s2 += Ecol[(int64_t)(a + 2) * ldE + k] * sdx[a + 2];
      s3 += Ecol[(int64_t)(a + 3) * ldE + k] * sdx[a + 3];
    }
    for (; a < n6; ++a) s0 += Ecol[(int64_t)a * ldE + k] * sdx[a];
    const float s = (s0 + s1) + (s2 + s3);
    const float dZ = Qbuf[k] * (ubuf[k] - s);
    float* pk = patches + (int64_t)kx[k] * 3 * PP + 2 * PP;
    float d = pk[0] + dZ;
    d = (d > 20.0f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int a2 = 0; a2 < PP; ++a2) pk[a2] = d;
  }
  if (gt < N) {
    float* p = poses + 7 * (int64_t)(t0 + gt);
    const float tt[3] = {p[0], p[1], p[2]}, qq[4] = {p[3], p[4], p[5], p[6]};
    const float xi[6] = {dX[6 * gt], dX[6 * gt + 1], dX[6 * gt + 2], dX[6 * gt + 3], dX[6 * gt + 4], dX[6 * gt + 5]};
    float t1v[3], q1v[4];
    retrSE3(xi, tt, qq, t1v, q1v);
    p[0] = t1v[0]; p[1] = t1v[1]; p[2] = t1v[2]; p[3] = q1v[0]; p[4] = q1v[1]; p[5] = q1v[2]; p[6] = q1v[3];
  }
}

struct BaWs {
  size_t pairbuf, edgebuf, Qbuf, ubuf, Ecol, spart, Sg, yg, dX, Bbuf, total;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

inline void ba_ws_layout(int64_t E, int N, BaWs* L) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  const size_t n6 = (size_t)(6 * (N > 0 ? N : 1));
  const size_t nblk = (n + kPatchChunk - 1) / kPatchChunk;
  size_t o = 0;
  L->pairbuf = o; o += al(n * kPairStride * 4);
  L->edgebuf = o; o += al(n * kEdgeStride * 4);
  L->Qbuf = o; o += al(n * 4);
  L->ubuf = o; o += al(n * 4);
  L->Ecol = o; o += al(n * n6 * 4);
  L->spart = o; o += al(nblk * (size_t)kSEntries * 4);
  L->Sg = o; o += al((size_t)kMaxDim * kMaxDim * 4);
  L->yg = o; o += al(kMaxDim * 4);
  L->dX = o; o += al(kMaxDim * 4);
  L->Bbuf = o; o += al((size_t)kMaxN * 6 * (kMaxDim + 1) * 4);
  L->total = o;
}

}  // namespace

extern "C" size_t dpvo_ba_workspace_bytes(int64_t E, int n_free_poses) {
  if (E < 0 || n_free_poses < 0 || n_free_poses > kMaxN) return 0;
  BaWs L;
  ba_ws_layout(E, n_free_poses, &L);
  return L.total;
}

extern "C" int dpvo_ba(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
                       float lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, const int32_t* plan,
                       int64_t n_patches_hint, int64_t n_pairs_hint, int64_t E, int P, int t0, int t1, int iterations,
                       int32_t* info, void* ws, size_t ws_bytes, void* stream) {
  if (E < 0 || P <= 0 || t1 < t0 || iterations < 0) return DPVO_E_INVALID;
  const int N = t1 - t0;
  if (6 * N > kMaxDim) return DPVO_E_UNSUPPORTED;
  if (E == 0 || iterations == 0) return DPVO_OK;
  if (!poses || !patches || !intrinsics || !target || !weight || !ii || !jj || !kk || !plan || !ws) return DPVO_E_INVALID;
  BaWs L;
  ba_ws_layout(E, N, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* w = (char*)ws;
  float* pairbuf = (float*)(w + L.pairbuf);
  float* edgebuf = (float*)(w + L.edgebuf);
  float* Qbuf = (float*)(w + L.Qbuf);
  float* ubuf = (float*)(w + L.ubuf);
  float* Ecol = (float*)(w + L.Ecol);
  float* spart = (float*)(w + L.spart);
  float* Sg = (float*)(w + L.Sg);
  float* yg = (float*)(w + L.yg);
  float* dX = (float*)(w + L.dX);
  float* Bbuf = (float*)(w + L.Bbuf);
  hipStream_t st = (hipStream_t)stream;
  const int32_t* n_patches = plan + PL.counts + 0;
  const int32_t* n_pairs = plan + PL.counts + 1;
  // exact group counts are a launch-size hint (the plan's device-side counts stay authoritative):
  // unknown (<= 0) falls back to the E upper bound.
  const int64_t np_h = (n_patches_hint > 0 && n_patches_hint <= E) ? n_patches_hint : E;
  const int64_t ng_h = (n_pairs_hint > 0 && n_pairs_hint <= E) ? n_pairs_hint : E;
  const unsigned pair_grid = (unsigned)(ng_h < 65535 ? ng_h : 65535);
  const unsigned patch_blocks = (unsigned)((np_h + kPatchChunk - 1) / kPatchChunk);
  for (int itr = 0; itr < iterations; ++itr) {
    hipLaunchKernelGGL(ba_pair_kernel, dim3(pair_grid), dim3(128), 0, st, poses, patches, intrinsics, target, weight, kk,
                       plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, n_pairs, pairbuf, edgebuf, P);
    // per-patch blocks + (N > 0) one B-part block per free pose in the same launch: they are independent
    hipLaunchKernelGGL(ba_patch_kernel, dim3(patch_blocks + (unsigned)N), dim3(256), 0, st, ii, jj, plan + PL.perm_k,
                       plan + PL.patch_off, n_patches, edgebuf, lmbda, t0, N, Qbuf, ubuf, Ecol, np_h, spart, (int)patch_blocks,
                       plan + PL.pair_ij, n_pairs, pairbuf, Bbuf);
    // (assemble + solve as ONE launch -- the last assemble workgroup to finish runs the solve -- was measured: 93 -> 128 us per call.
    //  Workgroups of one launch that hand data to each other need agent-scope release / acquire, i.e. write-backs and invalidations of
    //  the per-XCD L2s; a launch boundary does the same once and costs ~5 us.)
    if (N > 0) {
      hipLaunchKernelGGL(ba_assemble_kernel, dim3(N, 6), dim3(1024), 0, st, Bbuf, spart, (int)patch_blocks, N, Sg, yg);
      if (6 * N <= 60)
        hipLaunchKernelGGL(ba_solve60_kernel, dim3(1), dim3(256), 0, st, Sg, yg, N, dX, info ? info + itr : nullptr);
      else if (6 * N <= 64)
        hipLaunchKernelGGL(ba_solve_kernel<true>, dim3(1), dim3(64), 0, st, Sg, yg, N, dX, info ? info + itr : nullptr);
      else
        hipLaunchKernelGGL(ba_solve_kernel<false>, dim3(1), dim3(128), 0, st, Sg, yg, N, dX, info ? info + itr : nullptr);
    }
    // (64-thread workgroups: np patches are only ~54 waves, spread them over as many CUs)
    hipLaunchKernelGGL(ba_retr_kernel, dim3((unsigned)((np_h + 63) / 64)), dim3(64), 0, st, poses, patches,
                       plan + PL.kx, n_patches, Qbuf, ubuf, Ecol, np_h, dX, t0, N, P);
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
