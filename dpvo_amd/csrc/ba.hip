// ba.hip -- fastba bundle adjustment for gfx950 (dense Schur path, N = t1 - t0 <= 20 free poses).
//
// Replaces cuda_ba.forward (reference dpvo/fastba/ba_cuda.cu:433-582): the per-edge residual / Jacobian
// kernel with ~340 float atomics per edge onto a 60x60 tile (:232-376), ~25 ATen launches + cuSOLVER
// potrf/potrs per iteration (:519-565) and the two retraction kernels (:178-229).
//
// MI355X design -- four launches per Gauss-Newton iteration, no float atomics, bit-reproducible:
//   1. ba_pair_kernel   one wave per (i,j) frame pair (CSR from the plan; ~96 edges): per-edge residuals and
//                       Jacobians, DPP/shuffle wave reduction of the pair's Hii / Hjj / Hij / vi / vj blocks
//                       -> pairbuf[g][96]; per-edge depth terms (c, u, Ei[6], Ej[6]) -> edgebuf[e][16].
//   2. ba_patch_kernel  fixed grid, a block owns chunks of 32 patches (CSR by patch): builds the Schur
//                       column e_k (6N) of each patch in LDS, stores Q, u, e_k, and accumulates the block's
//                       partial  sum_k Q_k e_k e_k^T  and  sum_k Q_k u_k e_k  in registers -> spart[block].
//   3. ba_solve_kernel  one workgroup: assembles B and v from pairbuf in a fixed order, subtracts the Schur
//                       partials, applies the reference's damping S += I*(1e-4*S + 1), Cholesky-factorises
//                       the <=120x120 system in LDS, solves, writes dX and retracts the poses (Exp(dX)*T).
//   4. ba_retr_kernel   dZ_k = Q_k (u_k - e_k . dX), depth retraction with the reference's clamps.
#include "common.h"

namespace {

constexpr int kMaxN = 20;            // free poses on the dense path
constexpr int kMaxDim = 6 * kMaxN;   // 120
constexpr int kPairStride = 96;      // 36 Hii + 36 Hjj(unused upper dup) ... see layout below
constexpr int kEdgeStride = 16;
constexpr int kPatchChunk = 32;
constexpr int kPatchBlocks = 64;     // fixed grid of the patch kernel (deterministic partial count)

// pairbuf layout per pair: [0,21) Hii upper-tri, [21,42) Hjj upper-tri, [42,78) Hij full 6x6 (row = i-side),
// [78,84) vi, [84,90) vj
constexpr int kHii = 0, kHjj = 21, kHij = 42, kVi = 78, kVj = 84;

__device__ __forceinline__ int tri(int a, int b) {   // a <= b < 6
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

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
// 1. per-pair kernel: one wave per frame pair
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ba_pair_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                     const float* __restrict__ intr, const float* __restrict__ target,
                                                     const float* __restrict__ weight, const int64_t* __restrict__ kk,
                                                     const int32_t* __restrict__ perm_p, const int32_t* __restrict__ pair_off,
                                                     const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ n_pairs,
                                                     float* __restrict__ pairbuf, float* __restrict__ edgebuf, int P) {
  const int lane = threadIdx.x;
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

    float acc[90];
#pragma unroll
    for (int a = 0; a < 90; ++a) acc[a] = 0.f;

    const int b0 = pair_off[g], b1 = pair_off[g + 1];
    for (int p = b0 + lane; p < b1; p += 64) {
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
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int b = a; b < 6; ++b) {
            acc[kHii + tri(a, b)] += w * Ji[a] * Ji[b];
            acc[kHjj + tri(a, b)] += w * Jj[a] * Jj[b];
          }
#pragma unroll
          for (int b = 0; b < 6; ++b) acc[kHij + a * 6 + b] += w * Ji[a] * Jj[b];
          acc[kVi + a] += w * r * Ji[a];
          acc[kVj + a] += w * r * Jj[a];
          Ei[a] += -w * Jz * Ji[a];
          Ej[a] += w * Jz * Jj[a];
        }
        ce += w * Jz * Jz;
        ue += w * r * Jz;
      }
      float* eb = edgebuf + (int64_t)e * kEdgeStride;
      eb[0] = ce; eb[1] = ue;
#pragma unroll
      for (int a = 0; a < 6; ++a) { eb[2 + a] = Ei[a]; eb[8 + a] = Ej[a]; }
    }
    float* pb = pairbuf + (int64_t)g * kPairStride;
#pragma unroll
    for (int a = 0; a < 90; ++a) {
      const float s = wave_sum(acc[a]);
      if (lane == 0) pb[a] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// 2. per-patch kernel: Schur columns + partial Schur products
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_patch_kernel(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                                       const int32_t* __restrict__ perm_k,
                                                       const int32_t* __restrict__ patch_off,
                                                       const int32_t* __restrict__ n_patches,
                                                       const float* __restrict__ edgebuf, float lmbda, int t0, int N,
                                                       float* __restrict__ Qbuf, float* __restrict__ ubuf,
                                                       float* __restrict__ Ecol, float* __restrict__ spart) {
  __shared__ float col[kPatchChunk][kMaxDim + 1];
  __shared__ float qv[kPatchChunk], uv[kPatchChunk];
  const int n6 = 6 * N;
  const int np = *n_patches;
  const int tid = threadIdx.x;
  const int nent = n6 * n6 + n6;                       // S entries followed by y entries
  constexpr int kPer = (kMaxDim * kMaxDim + kMaxDim + 255) / 256;   // 57
  float part[kPer];
#pragma unroll
  for (int a = 0; a < kPer; ++a) part[a] = 0.f;

  const int nchunks = (np + kPatchChunk - 1) / kPatchChunk;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    for (int a = tid; a < kPatchChunk * (kMaxDim + 1); a += 256) (&col[0][0])[a] = 0.f;
    __syncthreads();
    if (tid < kPatchChunk) {
      const int k = ch * kPatchChunk + tid;
      float C = 0.f, u = 0.f;
      if (k < np) {
        const int b0 = patch_off[k], b1 = patch_off[k + 1];
        for (int p = b0; p < b1; ++p) {
          const int e = perm_k[p];
          const float* eb = edgebuf + (int64_t)e * kEdgeStride;
          C += eb[0]; u += eb[1];
          const int ix = (int)ii[e] - t0, jx = (int)jj[e] - t0;
          if (ix >= 0 && ix < N)
#pragma unroll
            for (int a = 0; a < 6; ++a) col[tid][6 * ix + a] += eb[2 + a];
          if (jx >= 0 && jx < N)
#pragma unroll
            for (int a = 0; a < 6; ++a) col[tid][6 * jx + a] += eb[8 + a];
        }
        const float Q = 1.0f / (C + lmbda);               // ba_cuda.cu:519
        qv[tid] = Q; uv[tid] = u;
        Qbuf[k] = Q; ubuf[k] = u;
      } else {
        qv[tid] = 0.f; uv[tid] = 0.f;
      }
    }
    __syncthreads();
    // store the columns (patch-major) for the dZ back-substitution
    for (int a = tid; a < kPatchChunk * n6; a += 256) {
      const int pl = a / n6, r = a - pl * n6;
      const int k = ch * kPatchChunk + pl;
      if (k < np) Ecol[(int64_t)k * n6 + r] = col[pl][r];
    }
    // partial S and y
#pragma unroll
    for (int s = 0; s < kPer; ++s) {
      const int ent = tid + 256 * s;
      if (ent < nent) {
        float sum = 0.f;
        if (ent < n6 * n6) {
          const int a = ent / n6, b = ent - a * n6;
          for (int pl = 0; pl < kPatchChunk; ++pl) sum += qv[pl] * col[pl][a] * col[pl][b];
        } else {
          const int a = ent - n6 * n6;
          for (int pl = 0; pl < kPatchChunk; ++pl) sum += qv[pl] * uv[pl] * col[pl][a];
        }
        part[s] += sum;
      }
    }
    __syncthreads();
  }
  float* sp = spart + (int64_t)blockIdx.x * (kMaxDim * kMaxDim + kMaxDim);
#pragma unroll
  for (int s = 0; s < kPer; ++s) {
    const int ent = tid + 256 * s;
    if (ent < nent) sp[ent] = part[s];
  }
}

// ---------------------------------------------------------------------------------------------------
// 3. solve kernel: one workgroup
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_pair(const int32_t* pair_ij, int ng, int i, int j) {
  int lo = 0, hi = ng - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int pi = pair_ij[2 * mid], pj = pair_ij[2 * mid + 1];
    if (pi == i && pj == j) return mid;
    if (pi < i || (pi == i && pj < j)) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

__global__ __launch_bounds__(512) void ba_solve_kernel(float* __restrict__ poses, const int32_t* __restrict__ pair_ij,
                                                       const int32_t* __restrict__ n_pairs, const float* __restrict__ pairbuf,
                                                       const float* __restrict__ spart, int n_spart, int t0, int N,
                                                       float* __restrict__ dX, int32_t* __restrict__ info) {
  __shared__ float S[kMaxDim * (kMaxDim + 1)];
  __shared__ float y[kMaxDim];
  __shared__ int bad;
  const int n6 = 6 * N, ld = kMaxDim + 1;
  const int ng = *n_pairs;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) bad = 0;

  // ---- assemble B (into S) and v (into y) from the pair blocks, fixed summation order
  for (int ent = tid; ent < N * N * 36; ent += nt) {
    const int blk = ent / 36, r = ent - blk * 36;
    const int p = blk / N, q = blk - p * N;
    const int a = r / 6, b = r - a * 6;
    float s = 0.f;
    if (p == q) {
      const int f = t0 + p;
      const int ta = a <= b ? tri(a, b) : tri(b, a);
      for (int g = 0; g < ng; ++g) {
        const int gi = pair_ij[2 * g], gj = pair_ij[2 * g + 1];
        const float* pb = pairbuf + (int64_t)g * kPairStride;
        if (gi == f) s += pb[kHii + ta];
        if (gj == f) s += pb[kHjj + ta];
        if (gi == f && gj == f) s += -pb[kHij + a * 6 + b] - pb[kHij + b * 6 + a];
      }
    } else {
      const int g1 = find_pair(pair_ij, ng, t0 + p, t0 + q);     // i = p, j = q: block (ix,jx) gets -Ji Jj^T
      const int g2 = find_pair(pair_ij, ng, t0 + q, t0 + p);     // i = q, j = p: block (jx,ix) gets -Jj Ji^T
      if (g1 >= 0) s += -pairbuf[(int64_t)g1 * kPairStride + kHij + a * 6 + b];
      if (g2 >= 0) s += -pairbuf[(int64_t)g2 * kPairStride + kHij + b * 6 + a];
    }
    S[(6 * p + a) * ld + 6 * q + b] = s;
  }
  for (int ent = tid; ent < n6; ent += nt) {
    const int p = ent / 6, a = ent - p * 6;
    const int f = t0 + p;
    float s = 0.f;
    for (int g = 0; g < ng; ++g) {
      const int gi = pair_ij[2 * g], gj = pair_ij[2 * g + 1];
      const float* pb = pairbuf + (int64_t)g * kPairStride;
      if (gi == f) s += -pb[kVi + a];
      if (gj == f) s += pb[kVj + a];
    }
    y[ent] = s;
  }
  __syncthreads();
  // ---- Schur complement: S = B - sum_k Q e e^T, y = v - sum_k Q u e   (ba_cuda.cu:557-558)
  const int pstride = kMaxDim * kMaxDim + kMaxDim;
  for (int ent = tid; ent < n6 * n6 + n6; ent += nt) {
    float s = 0.f;
    for (int b = 0; b < n_spart; ++b) s += spart[(int64_t)b * pstride + ent];
    if (ent < n6 * n6) { const int a = ent / n6, c = ent - a * n6; S[a * ld + c] -= s; }
    else y[ent - n6 * n6] -= s;
  }
  __syncthreads();
  // ---- damping  S += I * (1e-4 * S + 1.0)   (:560)
  for (int a = tid; a < n6; a += nt) S[a * ld + a] += 1e-4f * S[a * ld + a] + 1.0f;
  __syncthreads();
  // ---- in-place lower Cholesky (right-looking), info like linalg_cholesky_ex (ignored by the reference)
  for (int j = 0; j < n6; ++j) {
    if (tid == 0) {
      const float d = S[j * ld + j];
      if (!(d > 0.f) && bad == 0) bad = j + 1;
      S[j * ld + j] = sqrtf(d);
    }
    __syncthreads();
    const float djj = S[j * ld + j];
    for (int i2 = j + 1 + tid; i2 < n6; i2 += nt) S[i2 * ld + j] /= djj;
    __syncthreads();
    const int rem = n6 - j - 1;
    for (int ent = tid; ent < rem * rem; ent += nt) {
      const int r = j + 1 + ent / rem, c = j + 1 + ent % rem;
      if (c <= r) S[r * ld + c] -= S[r * ld + j] * S[c * ld + j];
    }
    __syncthreads();
  }
  // ---- forward / backward substitution (single wave; n6 <= 120)
  if (tid < 64) {
    volatile float* yv = y;
    for (int i2 = 0; i2 < n6; ++i2) {
      float s = 0.f;
      for (int k = tid; k < i2; k += 64) s += S[i2 * ld + k] * yv[k];
      s = wave_sum(s);
      if (tid == 0) yv[i2] = (yv[i2] - s) / S[i2 * ld + i2];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    for (int i2 = n6 - 1; i2 >= 0; --i2) {
      float s = 0.f;
      for (int k = i2 + 1 + tid; k < n6; k += 64) s += S[k * ld + i2] * yv[k];
      s = wave_sum(s);
      if (tid == 0) yv[i2] = (yv[i2] - s) / S[i2 * ld + i2];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
  }
  __syncthreads();
  for (int a = tid; a < n6; a += nt) dX[a] = y[a];
  if (tid == 0 && info) *info = bad;
  // ---- pose retraction (pose_retr_kernel, :178-206)
  if (tid < N) {
    float* p = poses + 7 * (int64_t)(t0 + tid);
    const float tt[3] = {p[0], p[1], p[2]}, qq[4] = {p[3], p[4], p[5], p[6]};
    const float xi[6] = {y[6 * tid], y[6 * tid + 1], y[6 * tid + 2], y[6 * tid + 3], y[6 * tid + 4], y[6 * tid + 5]};
    float t1v[3], q1v[4];
    retrSE3(xi, tt, qq, t1v, q1v);
    p[0] = t1v[0]; p[1] = t1v[1]; p[2] = t1v[2]; p[3] = q1v[0]; p[4] = q1v[1]; p[5] = q1v[2]; p[6] = q1v[3];
  }
}

// ---------------------------------------------------------------------------------------------------
// 4. depth back-substitution + retraction (patch_retr_kernel, :209-229; dZ :563)
// ---------------------------------------------------------------------------------------------------
__global__ void ba_retr_kernel(float* __restrict__ patches, const int32_t* __restrict__ kx,
                               const int32_t* __restrict__ n_patches, const float* __restrict__ Qbuf,
                               const float* __restrict__ ubuf, const float* __restrict__ Ecol,
                               const float* __restrict__ dX, int n6, int P) {
  const int np = *n_patches;
  const int PP = P * P;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < np; k += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int a = 0; a < n6; ++a) s += Ecol[(int64_t)k * n6 + a] * dX[a];
    const float dZ = Qbuf[k] * (ubuf[k] - s);
    float* pk = patches + (int64_t)kx[k] * 3 * PP + 2 * PP;
    float d = pk[0] + dZ;
    d = (d > 20.0f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int a = 0; a < PP; ++a) pk[a] = d;
  }
}

struct BaWs {
  size_t pairbuf, edgebuf, Qbuf, ubuf, Ecol, spart, dX, total;
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

inline void ba_ws_layout(int64_t E, int N, BaWs* L) {
  const size_t n = (size_t)(E > 0 ? E : 1);
  const size_t n6 = (size_t)(6 * (N > 0 ? N : 1));
  size_t o = 0;
  L->pairbuf = o; o += al(n * kPairStride * 4);
  L->edgebuf = o; o += al(n * kEdgeStride * 4);
  L->Qbuf = o; o += al(n * 4);
  L->ubuf = o; o += al(n * 4);
  L->Ecol = o; o += al(n * n6 * 4);
  L->spart = o; o += al((size_t)kPatchBlocks * (kMaxDim * kMaxDim + kMaxDim) * 4);
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
                       int64_t E, int P, int t0, int t1, int iterations, int32_t* info, void* ws, size_t ws_bytes,
                       void* stream) {
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
  float* dX = (float*)(w + L.dX);
  hipStream_t st = (hipStream_t)stream;
  const int32_t* n_patches = plan + PL.counts + 0;
  const int32_t* n_pairs = plan + PL.counts + 1;
  const unsigned pair_grid = (unsigned)(E < 4096 ? (E > 0 ? E : 1) : 4096);
  for (int itr = 0; itr < iterations; ++itr) {
    hipLaunchKernelGGL(ba_pair_kernel, dim3(pair_grid), dim3(64), 0, st, poses, patches, intrinsics, target, weight, kk,
                       plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, n_pairs, pairbuf, edgebuf, P);
    hipLaunchKernelGGL(ba_patch_kernel, dim3(kPatchBlocks), dim3(256), 0, st, ii, jj, plan + PL.perm_k, plan + PL.patch_off,
                       n_patches, edgebuf, lmbda, t0, N, Qbuf, ubuf, Ecol, spart);
    if (N > 0)
      hipLaunchKernelGGL(ba_solve_kernel, dim3(1), dim3(512), 0, st, poses, plan + PL.pair_ij, n_pairs, pairbuf, spart,
                         kPatchBlocks, t0, N, dX, info ? info + itr : nullptr);
    hipLaunchKernelGGL(ba_retr_kernel, dim3(64), dim3(256), 0, st, patches, plan + PL.kx, n_patches, Qbuf, ubuf, Ecol, dX,
                       6 * N, P);
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
