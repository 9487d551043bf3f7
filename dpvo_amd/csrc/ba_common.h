// ba_common.h -- device math of fastba (reference dpvo/fastba/ba_cuda.cu:36-174) and the per-pair linearisation
// kernel shared by the dense (ba.hip) and the block-sparse global (ba_global.hip) bundle-adjustment paths.
#pragma once
#include "common.h"

namespace ba {


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
__global__ __launch_bounds__(128) static void ba_pair_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
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
        // four aligned 16-byte stores (the row is 64 B; 14 scalar stores get merged into 16-byte stores at 8-byte alignment)
        f4* eb = reinterpret_cast<f4*>(edgebuf + (int64_t)e * kEdgeStride);
        eb[0] = (f4){ce, ue, Ei[0], Ei[1]};
        eb[1] = (f4){Ei[2], Ei[3], Ei[4], Ei[5]};
        eb[2] = (f4){Ej[0], Ej[1], Ej[2], Ej[3]};
        eb[3] = (f4){Ej[4], Ej[5], 0.f, 0.f};
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


}  // namespace ba
