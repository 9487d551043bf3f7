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
//   2. ba_patch_kernel    a block owns 32 patches (CSR by patch), 8 lanes gather one patch's edges: builds the Schur
//                         column e_k (6N) in LDS, stores Q, u, e_k, and the block's partial  sum_k Q_k e_k e_k^T,
//                         sum_k Q_k u_k e_k  -> spart[block].
//   3. ba_assemble_kernel one thread per entry of S / y: B, v from pairbuf in a fixed order, minus the Schur
//                         partials, plus the reference's damping S += I*(1e-4*S + 1)  -> Sg, yg.
//   4. ba_solve_kernel    one workgroup: left-looking Cholesky of the <=120x120 system in LDS (one barrier per
//                         column), column-oriented triangular solves -> dX.
//   5. ba_retr_kernel     dZ_k = Q_k (u_k - e_k . dX), depth retraction with the reference's clamps; pose
//                         retraction Exp(dX)*T.
#include "common.h"

namespace {

constexpr int kMaxN = 20;            // free poses on the dense path
constexpr int kMaxDim = 6 * kMaxN;   // 120
constexpr int kPairStride = 256;     // 16x16 Gram block G = sum_rows w a a^T, a = [Ji(6) Jj(6) r 0 0 0]
constexpr int kEdgeStride = 16;      // c, u, Ei[6], Ej[6], pad
constexpr int kPatchChunk = 32;
constexpr int kSEntries = kMaxDim * kMaxDim + kMaxDim;

// ---- device math of ba_cuda.cu:36-174 (no quaternion normalisation, same operation order) ----
__device__ __forceinline__ void actSO3(const float* q, const float* X, float* Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
__device__ __forceinline__ void adjSE3(const float* t, const float* q, const float* X, float* Y) {
  const float qinv[4] = {-q[0], -q[1], -q[2], q[3]};
  actSO3(qinv, &X[0], &Y[0]);
  actSO3(qinv, &X[3], &Y[3]);
  float u[3], v[3];
  u[0] = t[2] * X[1] - t[1] * X[2];
  u[1] = t[0] * X[2] - t[2] * X[0];
  u[2] = t[1] * X[0] - t[0] * X[1];
  actSO3(qinv, u, v);
  Y[3] += v[0]; Y[4] += v[1]; Y[5] += v[2];
}
__device__ __forceinline__ void relSE3(const float* ti, const float* qi, const float* tj, const float* qj, float* tij,
                                       float* qij) {
  qij[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  qij[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  qij[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  qij[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  actSO3(qij, ti, tij);
  tij[0] = tj[0] - tij[0]; tij[1] = tj[1] - tij[1]; tij[2] = tj[2] - tij[2];
}
__device__ __forceinline__ void expSO3(const float* phi, float* q) {
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta_p4 = theta_sq * theta_sq;
  const float theta = sqrtf(theta_sq);
  float imag, real;
  if (theta_sq < 1e-8f) {
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}
__device__ __forceinline__ void crossInplace(const float* a, float* b) {
  const float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}
__device__ __forceinline__ void expSE3(const float* xi, float* t, float* q) {
  expSO3(xi + 3, q);
  float tau[3] = {xi[0], xi[1], xi[2]};
  const float phi[3] = {xi[3], xi[4], xi[5]};
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta_sq);
  t[0] = tau[0]; t[1] = tau[1]; t[2] = tau[2];
  if (theta > 1e-4f) {
    const float a = (1 - cosf(theta)) / theta_sq;
    crossInplace(phi, tau);
    t[0] += a * tau[0]; t[1] += a * tau[1]; t[2] += a * tau[2];
    const float b = (theta - sinf(theta)) / (theta * theta_sq);
    crossInplace(phi, tau);
    t[0] += b * tau[0]; t[1] += b * tau[1]; t[2] += b * tau[2];
  }
}
__device__ __forceinline__ void retrSE3(const float* xi, const float* t, const float* q, float* t1, float* q1) {
  float dt[3] = {0, 0, 0};
  float dq[4] = {0, 0, 0, 1};
  expSE3(xi, dt, dq);
  q1[0] = dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1];
  q1[1] = dq[3] * q[1] + dq[1] * q[3] + dq[2] * q[0] - dq[0] * q[2];
  q1[2] = dq[3] * q[2] + dq[2] * q[3] + dq[0] * q[1] - dq[1] * q[0];
  q1[3] = dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2];
  actSO3(dq, t, t1);
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
}

// ---------------------------------------------------------------------------------------------------
// 1. per-pair kernel: 128 threads (2 waves) per frame pair
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void ba_pair_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                      const float* __restrict__ intr, const float* __restrict__ target,
                                                      const float* __restrict__ weight, const int64_t* __restrict__ kk,
                                                      const int32_t* __restrict__ perm_p, const int32_t* __restrict__ pair_off,
                                                      const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ n_pairs,
                                                      float* __restrict__ pairbuf, float* __restrict__ edgebuf, int P) {
  __shared__ float Arow[2][128][17];     // per wave: 128 residual rows x 16 columns (+1 pad)
  __shared__ float Wrow[2][128];
  __shared__ float comb[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ng = *n_pairs;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];     // intrinsics[0] (ba_cuda.cu:253-259)
  const int PP = P * P, ctr = (P / 2) * P + P / 2;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int i = pair_ij[2 * g], j = pair_ij[2 * g + 1];
    const float* pi = poses + 7 * (int64_t)i; const float* pj = poses + 7 * (int64_t)j;
    const float ti[3] = {pi[0], pi[1], pi[2]}, tj[3] = {pj[0], pj[1], pj[2]};
    const float qi[4] = {pi[3], pi[4], pi[5], pi[6]}, qj[4] = {pj[3], pj[4], pj[5], pj[6]};
    float tij[3], qij[4];
    relSE3(ti, qi, tj, qj, tij, qij);

    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int b0 = pair_off[g], b1 = pair_off[g + 1];
    for (int c0 = b0 + wave * 64; c0 < b1; c0 += 128) {
      const int p = c0 + lane;
      float rowv[2][13];
      float wv[2] = {0.f, 0.f};
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int a = 0; a < 13; ++a) rowv[r2][a] = 0.f;
      if (p < b1) {
        const int e = perm_p[p];
        const float* pk = patches + kk[e] * 3 * PP;
        float Xi[4], Xj[4];
        Xi[0] = (pk[ctr] - cx) / fx;
        Xi[1] = (pk[PP + ctr] - cy) / fy;
        Xi[2] = 1.0f;
        Xi[3] = pk[2 * PP + ctr];
        actSO3(qij, Xi, Xj);
        Xj[3] = Xi[3];
        Xj[0] += Xi[3] * tij[0]; Xj[1] += Xi[3] * tij[1]; Xj[2] += Xi[3] * tij[2];
        const float X = Xj[0], Y = Xj[1], Z = Xj[2], W = Xj[3];
        const float d = (Z >= 0.2f) ? 1.0f / Z : 0.0f;
        const float d2 = d * d;
        const float x1 = fx * (X / Z) + cx;
        const float y1 = fy * (Y / Z) + cy;
        const float rx = target[2 * (int64_t)e + 0] - x1;
        const float ry = target[2 * (int64_t)e + 1] - y1;
        const bool in_bounds = (sqrtf(rx * rx + ry * ry) < 128.0f) && (Z > 0.2f) && (x1 > -64.0f) && (y1 > -64.0f) &&
                               (x1 < 2 * cx + 64.0f) && (y1 < 2 * cy + 64.0f);
        const float mask = in_bounds ? 1.0f : 0.0f;
        float ce = 0.f, ue = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0}, Ej[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          float Jj[6], Ji[6], Jz, r, w;
          if (row == 0) {
            r = rx; w = mask * weight[2 * (int64_t)e + 0];
            Jz = fx * (tij[0] * d - tij[2] * (X * d2));
            Jj[0] = fx * W * d; Jj[1] = 0.f; Jj[2] = fx * -X * W * d2;
            Jj[3] = fx * -X * Y * d2; Jj[4] = fx * (1 + X * X * d2); Jj[5] = fx * -Y * d;
          } else {
            r = ry; w = mask * weight[2 * (int64_t)e + 1];
            Jz = fy * (tij[1] * d - tij[2] * (Y * d2));
            Jj[0] = 0.f; Jj[1] = fy * W * d; Jj[2] = fy * -Y * W * d2;
            Jj[3] = fy * (-1 - Y * Y * d2); Jj[4] = fy * (X * Y * d2); Jj[5] = fy * X * d;
          }
          adjSE3(tij, qij, Jj, Ji);
          // a non-finite residual row with zero weight must not poison the Gram block (0 * inf = NaN)
          const bool live = (w != 0.f);
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            rowv[row][a] = live ? Ji[a] : 0.f;
            rowv[row][6 + a] = live ? Jj[a] : 0.f;
            Ei[a] += live ? -w * Jz * Ji[a] : 0.f;
            Ej[a] += live ? w * Jz * Jj[a] : 0.f;
          }
          rowv[row][12] = live ? r : 0.f;
          wv[row] = w;
          ce += live ? w * Jz * Jz : 0.f;
          ue += live ? w * r * Jz : 0.f;
        }
        float* eb = edgebuf + (int64_t)e * kEdgeStride;
        eb[0] = ce; eb[1] = ue;
#pragma unroll
        for (int a = 0; a < 6; ++a) { eb[2 + a] = Ei[a]; eb[8 + a] = Ej[a]; }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
#pragma unroll
        for (int a = 0; a < 13; ++a) Arow[wave][2 * lane + r2][a] = rowv[r2][a];
#pragma unroll
        for (int a = 13; a < 16; ++a) Arow[wave][2 * lane + r2][a] = 0.f;
        Wrow[wave][2 * lane + r2] = wv[r2];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // G += (w a)^T a over the wave's 128 rows, 4 rows per MFMA: A operand lane l = A[i=l&15][k=l>>4]
      const int cidx = lane & 15, ksub = lane >> 4;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) {
        const int k = 4 * t + ksub;
        const float a = Arow[wave][k][cidx];
        const float w = Wrow[wave][k];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w * a, a, acc, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // combine the two waves in a fixed order, store G[16][16]: lane holds G[4*(l>>4)+r][l&15]
    if (wave == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) comb[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
      float* pb = pairbuf + (int64_t)g * kPairStride;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = (4 * (lane >> 4) + r) * 16 + (lane & 15);
        pb[idx] = acc[r] + comb[idx];
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// 2. per-patch kernel: Schur columns + partial Schur products; block = 32 patches x 8 lanes
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_patch_kernel(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                       const int32_t* __restrict__ perm_k,
                                                       const int32_t* __restrict__ patch_off,
                                                       const int32_t* __restrict__ n_patches,
                                                       const float* __restrict__ edgebuf, float lmbda, int t0, int N,
                                                       float* __restrict__ Qbuf, float* __restrict__ ubuf,
                                                       float* __restrict__ Ecol, int64_t ldE,
                                                       float* __restrict__ spart) {
  __shared__ float col[kPatchChunk][kMaxDim + 1];
  __shared__ float qv[kPatchChunk], uv[kPatchChunk];
  const int n6 = 6 * N;
  const int np = *n_patches;
  const int tid = threadIdx.x;
  const int nent = n6 * n6 + n6;                       // S entries followed by y entries
  const int ch = blockIdx.x;
  for (int a = tid; a < kPatchChunk * (kMaxDim + 1); a += 256) (&col[0][0])[a] = 0.f;
  __syncthreads();
  {
    const int pl = tid >> 3, sub = tid & 7;
    const int k = ch * kPatchChunk + pl;
    float C = 0.f, u = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0};
    int ix = -1;
    if (k < np) {
      const int b0 = patch_off[k], b1 = patch_off[k + 1];
      for (int p = b0 + sub; p < b1; p += 8) {
        const int e = perm_k[p];
        const f4* eb = reinterpret_cast<const f4*>(edgebuf + (int64_t)e * kEdgeStride);
        const f4 v0 = eb[0], v1 = eb[1], v2 = eb[2], v3 = eb[3];
        C += v0[0]; u += v0[1];
        Ei[0] += v0[2]; Ei[1] += v0[3]; Ei[2] += v1[0]; Ei[3] += v1[1]; Ei[4] += v1[2]; Ei[5] += v1[3];
        ix = (int)ii[e] - t0;
        const int jx = (int)jj[e] - t0;
        if (jx >= 0 && jx < N) {
          // distinct (patch, j) edges own distinct slots; duplicates (if any) are folded by the LDS add
          atomicAdd(&col[pl][6 * jx + 0], v2[0]); atomicAdd(&col[pl][6 * jx + 1], v2[1]);
          atomicAdd(&col[pl][6 * jx + 2], v2[2]); atomicAdd(&col[pl][6 * jx + 3], v2[3]);
          atomicAdd(&col[pl][6 * jx + 4], v3[0]); atomicAdd(&col[pl][6 * jx + 5], v3[1]);
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
  // partial S and y of this block
  float* sp = spart + (int64_t)blockIdx.x * kSEntries;
  for (int ent = tid; ent < nent; ent += 256) {
    float sum = 0.f;
    if (ent < n6 * n6) {
      const int a = ent / n6, b = ent - a * n6;
#pragma unroll 8
      for (int pl = 0; pl < kPatchChunk; ++pl) sum += qv[pl] * col[pl][a] * col[pl][b];
    } else {
      const int a = ent - n6 * n6;
#pragma unroll 8
      for (int pl = 0; pl < kPatchChunk; ++pl) sum += qv[pl] * uv[pl] * col[pl][a];
    }
    sp[ent] = sum;
  }
}

// ---------------------------------------------------------------------------------------------------
// 3. assemble kernel: one block per free pose p = block row p of S (6 x n6) and y[6p..6p+5]
//    wave 0: diagonal block + y: lanes stride over the pairs, per-lane partial sums, fixed-order butterfly;
//    waves 1..3: off-diagonal blocks (two binary searches in the LDS copy of the sorted pair list);
//    then all threads: subtract the Schur partials (4 lanes per entry) and apply the damping.
// ---------------------------------------------------------------------------------------------------
constexpr int kMaxPairsLds = 4096;

__device__ __forceinline__ int find_pair(const int2* pl, int ng, int i, int j) {
  int lo = 0, hi = ng - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int2 v = pl[mid];
    if (v.x == i && v.y == j) return mid;
    if (v.x < i || (v.x == i && v.y < j)) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

__global__ __launch_bounds__(256) void ba_assemble_kernel(const int32_t* __restrict__ pair_ij,
                                                          const int32_t* __restrict__ n_pairs,
                                                          const float* __restrict__ pairbuf,
                                                          const float* __restrict__ spart, int n_spart, int t0, int N,
                                                          float* __restrict__ Sg, float* __restrict__ yg) {
  __shared__ int2 plist[kMaxPairsLds];
  __shared__ float rowS[6][kMaxDim];      // block row p of B (then S)
  __shared__ float rowy[6];
  const int n6 = 6 * N;
  const int ng = *n_pairs;
  const int ngl = ng < kMaxPairsLds ? ng : kMaxPairsLds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x, f = t0 + p;
  const int2* pg = reinterpret_cast<const int2*>(pair_ij);
  for (int g = tid; g < ngl; g += 256) plist[g] = pg[g];
  for (int a = tid; a < 6 * kMaxDim; a += 256) (&rowS[0][0])[a] = 0.f;
  __syncthreads();
  const int2* pl = (ng <= kMaxPairsLds) ? plist : pg;   // (global fallback for very large graphs)
  if (wave == 0) {
    float acc[42];
#pragma unroll
    for (int a = 0; a < 42; ++a) acc[a] = 0.f;
    for (int g = lane; g < ng; g += 64) {
      const int2 ij = pl[g];
      const float* pb = pairbuf + (int64_t)g * kPairStride;
      if (ij.x == f) {                                            // i-side: + w Ji Ji^T, v -= w r Ji
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int b = 0; b < 6; ++b) acc[a * 6 + b] += pb[a * 16 + b];
          acc[36 + a] -= pb[a * 16 + 12];
        }
      }
      if (ij.y == f) {                                            // j-side: + w Jj Jj^T, v += w r Jj
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int b = 0; b < 6; ++b) acc[a * 6 + b] += pb[(6 + a) * 16 + 6 + b];
          acc[36 + a] += pb[(6 + a) * 16 + 12];
        }
      }
      if (ij.x == f && ij.y == f) {                               // self edge: both cross terms land here
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b) acc[a * 6 + b] += -pb[a * 16 + 6 + b] - pb[b * 16 + 6 + a];
      }
    }
#pragma unroll
    for (int a = 0; a < 42; ++a) {
      const float s = wave_sum(acc[a]);
      if (lane == 0) { if (a < 36) rowS[a / 6][6 * p + a % 6] = s; else rowy[a - 36] = s; }
    }
  } else {
    for (int ent = tid - 64; ent < (N - 1) * 36; ent += 192) {
      int q = ent / 36; const int r = ent - q * 36;
      if (q >= p) q += 1;
      const int a = r / 6, b = r - a * 6;
      const int g1 = find_pair(pl, ng, f, t0 + q);     // i = p, j = q: block (ix,jx) gets -w Ji Jj^T   (:342-345)
      const int g2 = find_pair(pl, ng, t0 + q, f);     // i = q, j = p: block (jx,ix) gets -w Jj Ji^T
      float s = 0.f;
      if (g1 >= 0) s += -pairbuf[(int64_t)g1 * kPairStride + a * 16 + 6 + b];
      if (g2 >= 0) s += -pairbuf[(int64_t)g2 * kPairStride + b * 16 + 6 + a];
      rowS[a][6 * q + b] = s;
    }
  }
  __syncthreads();
  // Schur complement (ba_cuda.cu:557-558) + damping (:560): entry = 6 x n6 of S, then 6 of y; 4 lanes per entry
  const int nrow = 6 * n6 + 6;
  for (int base = 0; base < nrow; base += 64) {
    const int e4 = base + (tid >> 2), sub = tid & 3;
    float sc = 0.f;
    int gent = 0;
    if (e4 < nrow) {
      gent = (e4 < 6 * n6) ? (6 * p + e4 / n6) * n6 + (e4 % n6) : n6 * n6 + 6 * p + (e4 - 6 * n6);
      for (int b = sub; b < n_spart; b += 4) sc += spart[(int64_t)b * kSEntries + gent];
    }
    sc += __shfl_xor(sc, 1);
    sc += __shfl_xor(sc, 2);
    if (e4 < nrow && sub == 0) {
      if (e4 < 6 * n6) {
        const int ra = e4 / n6, rb = e4 - ra * n6;
        float s = rowS[ra][rb] - sc;
        if (6 * p + ra == rb) s += 1e-4f * s + 1.0f;              // S += I * (1e-4 * S + 1.0)
        Sg[gent] = s;
      } else {
        yg[6 * p + (e4 - 6 * n6)] = rowy[e4 - 6 * n6] - sc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 4. solve kernel: one workgroup, thread r owns row r.  n6 <= 64: a single wave, no barriers at all
//    (LDS traffic of one wave is program-ordered; pivots travel by v_readlane); n6 <= 120: two waves.
// ---------------------------------------------------------------------------------------------------
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
  const int np = *n_patches;
  const int PP = P * P, n6 = 6 * N;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = gt; k < np; k += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int a = 0; a < n6; ++a) s += Ecol[(int64_t)a * ldE + k] * dX[a];
    const float dZ = Qbuf[k] * (ubuf[k] - s);
    float* pk = patches + (int64_t)kx[k] * 3 * PP + 2 * PP;
    float d = pk[0] + dZ;
    d = (d > 20.0f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int a = 0; a < PP; ++a) pk[a] = d;
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
  size_t pairbuf, edgebuf, Qbuf, ubuf, Ecol, spart, Sg, yg, dX, total;
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
    hipLaunchKernelGGL(ba_patch_kernel, dim3(patch_blocks), dim3(256), 0, st, ii, jj, plan + PL.perm_k,
                       plan + PL.patch_off, n_patches, edgebuf, lmbda, t0, N, Qbuf, ubuf, Ecol, np_h, spart);
    if (N > 0) {
      hipLaunchKernelGGL(ba_assemble_kernel, dim3(N), dim3(256), 0, st, plan + PL.pair_ij, n_pairs, pairbuf, spart,
                         (int)patch_blocks, t0, N, Sg, yg);
      if (6 * N <= 64)
        hipLaunchKernelGGL(ba_solve_kernel<true>, dim3(1), dim3(64), 0, st, Sg, yg, N, dX, info ? info + itr : nullptr);
      else
        hipLaunchKernelGGL(ba_solve_kernel<false>, dim3(1), dim3(128), 0, st, Sg, yg, N, dX, info ? info + itr : nullptr);
    }
    hipLaunchKernelGGL(ba_retr_kernel, dim3((unsigned)((np_h + 255) / 256)), dim3(256), 0, st, poses, patches,
                       plan + PL.kx, n_patches, Qbuf, ubuf, Ecol, np_h, dX, t0, N, P);
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
