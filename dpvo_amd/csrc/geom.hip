// geom.hip -- SE3 forward ops, fused reprojection, flow magnitude and point cloud for gfx950.
//
// Replaces the lietorch_backends SE3 forward kernels (reference dpvo/lietorch/src/lietorch_gpu.cu:21-30,
// 47-56,73-82,101-110,225-236 with the math of include/se3.h:34-56,124-142 and include/so3.h:31-60,
// 115-208) and the ~12-launch elementwise chain of pops.transform (dpvo/projective_ops.py:19-68) that
// DPVO.reproject runs every update (dpvo/dpvo.py:209-213).  All f32, one thread per element / edge
// pixel; these kernels are latency-trivial (E*9 threads, <200 B per edge) -- the point is removing
// launches and the materialised 9E x 7 broadcast of lietorch/broadcasting.py:26-29.
#include "common.h"
#include "se3_dev.h"

namespace {

__device__ __forceinline__ void hat3(const float* p, float* M) {
  M[0] = 0; M[1] = -p[2]; M[2] = p[1]; M[3] = p[2]; M[4] = 0; M[5] = -p[0]; M[6] = -p[1]; M[7] = p[0]; M[8] = 0;
}
__device__ __forceinline__ void mat3mul(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float s = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
      C[3 * i + j] = s;
    }
}

constexpr float kEps = 1e-6f;     // common.h:7 of lietorch

__global__ void se3_inv_kernel(const float* X, float* Y, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    store_pose(Y + 7 * i, se3_inv(load_pose(X + 7 * i)));
}
__global__ void se3_mul_kernel(const float* X, const float* Y, float* Z, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    store_pose(Z + 7 * i, se3_mul(load_pose(X + 7 * i), load_pose(Y + 7 * i)));
}
__global__ void se3_act4_kernel(const float* X, const float* p, float* q, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi[4] = {p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}, o[4];
    se3_act4(load_pose(X + 7 * i), pi, o);
    q[4 * i] = o[0]; q[4 * i + 1] = o[1]; q[4 * i + 2] = o[2]; q[4 * i + 3] = o[3];
  }
}
// se3.h:133-142 + so3.h:152-190
__global__ void se3_exp_kernel(const float* A, float* X, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* a = A + 6 * i;
    const float tau[3] = {a[0], a[1], a[2]}, phi[3] = {a[3], a[4], a[5]};
    const float theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const float theta = sqrtf(theta2);
    float imag, real;
    if (theta < kEps) {
      const float theta4 = theta2 * theta2;
      imag = 0.5f - (1.0f / 48.0f) * theta2 + (1.0f / 3840.0f) * theta4;
      real = 1.0f - (1.0f / 8.0f) * theta2 + (1.0f / 384.0f) * theta4;
    } else {
      imag = sinf(.5f * theta) / theta;
      real = cosf(.5f * theta);
    }
    Pose P;
    P.q = qnormalize({imag * phi[0], imag * phi[1], imag * phi[2], real});
    float Phi[9], Phi2[9];
    hat3(phi, Phi); mat3mul(Phi, Phi, Phi2);
    const float c1 = (theta < kEps) ? 0.5f - (1.0f / 24.0f) * theta2 : (1.0f - cosf(theta)) / theta2;
    const float c2 = (theta < kEps) ? (1.0f / 6.0f) - (1.0f / 120.0f) * theta2 : (theta - sinf(theta)) / (theta2 * theta);
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      float s = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) s += (((r == c) ? 1.0f : 0.0f) + c1 * Phi[3 * r + c] + c2 * Phi2[3 * r + c]) * tau[c];
      t[r] = s;
    }
    P.t = {t[0], t[1], t[2]};
    store_pose(X + 7 * i, P);
  }
}
// se3.h:124-131 + so3.h:115-150,192-208
__global__ void se3_log_kernel(const float* X, float* A, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const Pose P = load_pose(X + 7 * i);
    const float sn = P.q.x * P.q.x + P.q.y * P.q.y + P.q.z * P.q.z;
    const float w = P.q.w;
    float f;
    if (sn < kEps * kEps) {
      const float sw = w * w;
      f = 2.0f / w - (2.0f / 3.0f) * sn / (w * sw);
    } else {
      const float nn = sqrtf(sn);
      if (fabsf(w) < kEps) f = (w > 0) ? 3.14159265358979323846f / nn : -3.14159265358979323846f / nn;
      else f = 2.0f * atanf(nn / w) / nn;
    }
    const float phi[3] = {f * P.q.x, f * P.q.y, f * P.q.z};
    float Phi[9], Phi2[9];
    hat3(phi, Phi); mat3mul(Phi, Phi, Phi2);
    const float theta2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    const float theta = sqrtf(theta2), ht = 0.5f * theta;
    const float c2 = (theta < kEps) ? (1.0f / 12.0f) : (1.0f - theta * cosf(ht) / (2.0f * sinf(ht))) / (theta * theta);
    const float t[3] = {P.t.x, P.t.y, P.t.z};
    float* a = A + 6 * i;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      float s = 0;
#pragma unroll
      for (int c = 0; c < 3; c++) s += (((r == c) ? 1.0f : 0.0f) - 0.5f * Phi[3 * r + c] + c2 * Phi2[3 * r + c]) * t[c];
      a[r] = s;
    }
    a[3] = phi[0]; a[4] = phi[1]; a[5] = phi[2];
  }
}

// One thread per (edge, patch pixel).  clamp_z = 1: pops.transform semantics (projective_ops.py:43);
// clamp_z = 0: the exported cuda_ba.reproject (ba_cuda.cu:379-429): raw Z, intrinsics[0], no quaternion
// normalisation (relSE3 :74-85).
__global__ void reproject_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                 const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                 const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                                 float* __restrict__ coords, int64_t E, int P, int clamp_z) {
  reproject_body(poses, patches, intr, ii, jj, kk, coords, E, P, clamp_z, blockIdx.x, gridDim.x);
}

__global__ void flow_mag_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                const int64_t* __restrict__ jj, const int64_t* __restrict__ kk, float beta,
                                float* __restrict__ flow, float* __restrict__ valid, int64_t E, int P) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x)
    edge_flow(poses, patches, intr, ii[e], jj[e], kk[e], beta, P, flow + e, valid + e);
}

// PatchGraph.edges_loop's candidate test (patchgraph.py:56-72), one workgroup per candidate frame pair (target j = j0 + b / n_i, source
// frame f = i0 + b % n_i): thread p takes the CENTRE pixel of patch k = f M + p (the reference hands pops.flow_mag patches[..., 1, 1]),
// i = ix[k]; flow and validity exactly as flow_mag_kernel computes them for a 1 x 1 patch (same flow_pixel).  Then the reference's
// reduction: val = nvalid > 0.5; sum (flow * val) and sum val over the M patches; flow_mag = sum / max(count, 1) if count > 0.75 M,
// else +inf.  Fixed reduction tree (wave butterflies, then the waves in order): bit-repeatable.  The result may be pinned host memory.
// ONE kernel for both callers, so that both compute the same bits: dpvo_loop_flow passes the candidate ranges; dpvo_frame_update's tail
// (decision != NULL) derives them on the device from the frame count the keyframe step leaves behind, n_eval = n - decision + 1, and
// puts {n_eval, number of pairs} in front of the magnitudes (the launch is sized for decision == 0; surplus workgroups leave).
struct LoopFlowArgs {
  const float *poses, *patches, *intr; const int64_t* ix;
  int64_t j0, i0, n_i; int M, P; float beta; float* out;
  const int32_t* decision; int n, removal_window, keyframe_index, freq, max_age;
};
__global__ __launch_bounds__(256) void loop_flow_kernel(const LoopFlowArgs A) {
  __shared__ float red[2][4];
  int64_t j0 = A.j0, i0 = A.i0, n_i = A.n_i;
  float* out = A.out;
  const int64_t b = blockIdx.x;
  if (A.decision) {
    const int64_t n_eval = (int64_t)A.n - (*A.decision != 0 ? 1 : 0) + 1, l = n_eval - A.removal_window;
    const int64_t n_j = A.freq - A.keyframe_index;
    j0 = n_eval - A.freq;
    i0 = l - A.max_age > 0 ? l - A.max_age : 0;
    n_i = l - i0;
    const int64_t pairs = (l > 0 && n_j > 0 && j0 >= 0) ? n_j * n_i : 0;
    if (b == 0 && threadIdx.x == 0) { out[0] = (float)n_eval; out[1] = (float)pairs; }
    if (b >= pairs) return;
    out += 2;
  }
  const int64_t j = j0 + b / n_i, f = i0 + b % n_i;
  const int M = A.M, PP = A.P * A.P, c = (A.P / 2) * A.P + A.P / 2;
  float s = 0.f, cnt = 0.f;
  for (int p = threadIdx.x; p < M; p += 256) {            // (M <= 256 in every configuration: one pass)
    const int64_t k = f * M + p;
    const FlowPair F = flow_pair(A.poses, A.intr, A.ix[k], j);
    const float* pk = A.patches + k * 3 * PP + c;
    float fl = 0.f, v = 0.f;
    flow_pixel(F, pk[0], pk[PP], pk[2 * PP], A.beta, fl, v);
    const float val = v > 0.5f ? 1.f : 0.f;
    s += fl * val;                                          // (a product, as in the reference: inf * 0 = NaN stays NaN)
    cnt += val;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], C = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    out[b] = C > (float)M * 0.75f ? S / fmaxf(C, 1.f) : __builtin_inff();
  }
}

// scan variant: no plan needed, one 1024-thread block walks every edge
__global__ __launch_bounds__(1024) void motionmag_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                         const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                                         const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                                                         int64_t E, int P, int64_t qi, int64_t qj, float beta,
                                                         float* __restrict__ out) {
  __shared__ float red[4][16];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t e = threadIdx.x; e < E; e += 1024) {
    const int64_t i = ii[e], j = jj[e];
    const bool fwd = (i == qi && j == qj), bwd = (i == qj && j == qi);
    if (fwd || bwd) {
      float f, v;
      edge_flow(poses, patches, intr, i, j, kk[e], beta, P, &f, &v);
      if (fwd) { s[0] += f; s[1] += 1.f; }
      if (bwd) { s[2] += f; s[3] += 1.f; }
    }
  }
  block_reduce4(s, red, out);
}

__global__ __launch_bounds__(256) void motionmag_plan_kernel(MotionPlanArgs A) {
  motionmag_plan_body(A.poses, A.patches, A.intr, A.kk, A.perm_p, A.pair_off, A.pair_ij, A.n_pairs, A.P, A.qi, A.qj, A.beta, A.out, A.status, A.flow);
}

// pops.point_cloud centre pixel, dpvo.py:358-360.
__device__ __forceinline__ void point_cloud_body(const float* __restrict__ poses, const float* __restrict__ patches,
                                                 const float* __restrict__ intr, const int64_t* __restrict__ ix,
                                                 float* __restrict__ points, int64_t m, int P, int64_t bid, int64_t nblk) {
  const int PP = P * P, c = (P / 2) * P + P / 2;
  for (int64_t k = bid * (int64_t)blockDim.x + threadIdx.x; k < m; k += nblk * blockDim.x) {
    const int64_t f = ix[k];
    const Pose Ginv = se3_inv(load_pose(poses + 7 * f));
    const float* K = intr + 4 * f;
    const float* pk = patches + k * 3 * PP;
    const float X0[4] = {(pk[c] - K[2]) / K[0], (pk[PP + c] - K[3]) / K[1], 1.0f, pk[2 * PP + c]};
    float X1[4];
    se3_act4(Ginv, X0, X1);
    points[3 * k + 0] = X1[0] / X1[3]; points[3 * k + 1] = X1[1] / X1[3]; points[3 * k + 2] = X1[2] / X1[3];
  }
}

__global__ void point_cloud_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                   const float* __restrict__ intr, const int64_t* __restrict__ ix,
                                   float* __restrict__ points, int64_t m, int P) {
  point_cloud_body(poses, patches, intr, ix, points, m, P, blockIdx.x, gridDim.x);
}
// point cloud + flow test in ONE launch (the tail of a frame: both read the poses / patches the bundle adjustment has just
// written and nothing else depends on either): the last block is the flow test, the others the point cloud
__global__ __launch_bounds__(256) void point_cloud_motionmag_kernel(const float* __restrict__ poses, const float* __restrict__ patches,
                                                                    const float* __restrict__ intr, const int64_t* __restrict__ ix,
                                                                    float* __restrict__ points, int64_t m, int P, MotionPlanArgs A) {
  if (blockIdx.x == gridDim.x - 1)
    motionmag_plan_body(A.poses, A.patches, A.intr, A.kk, A.perm_p, A.pair_off, A.pair_ij, A.n_pairs, A.P, A.qi, A.qj, A.beta, A.out, A.status, A.flow);
  else
    point_cloud_body(poses, patches, intr, ix, points, m, P, blockIdx.x, gridDim.x - 1);
}

inline unsigned grid_for(int64_t n) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// PatchGraph.normalize (patchgraph.py:84-90) as two launches instead of eight torch kernels (a 43 us strided reduction among them,
// profiles/r05_e_lc_timeline.txt): scale = mean of the depth channel of the first n frames' patches, depths /= scale, translations
// *= scale, every pose multiplied from the right by the inverse of (the scaled) pose 0.
//   normalize_sum_kernel    per-block partial sums of the depths in f64 (fixed order: bit-repeatable), and a copy of pose 0
//   normalize_apply_kernel  every block adds the partials up in block order (the same scale everywhere), then takes its share
// scratch: [0] = scale (f32), [1..7] = pose 0 as it was, then the partial sums as doubles from byte 64 on.
constexpr int kNormBlocks = 256;
__global__ __launch_bounds__(1024) void normalize_sum_kernel(const float* __restrict__ poses, const float* __restrict__ patches, int64_t total,
                                                             int PP, float* __restrict__ scratch) {
  __shared__ double red[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  double acc = 0.0;
  for (int64_t q = (int64_t)blockIdx.x * 1024 + t; q < total; q += (int64_t)gridDim.x * 1024) {
    const int64_t k = q / PP; const int a = (int)(q - k * PP);
    acc += (double)patches[(k * 3 + 2) * PP + a];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (t == 0) {
    double b = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) b += red[i];
    reinterpret_cast<double*>(scratch + 16)[blockIdx.x] = b;
  }
  if (blockIdx.x == 0 && t < 7) scratch[1 + t] = poses[t];
}

__global__ __launch_bounds__(1024) void normalize_apply_kernel(float* __restrict__ poses, float* __restrict__ patches, int n, int64_t total,
                                                               int PP, int nblk_sum, float* __restrict__ scratch) {
  const int t = threadIdx.x;
  double sum = 0.0;
  for (int b = 0; b < nblk_sum; ++b) sum += reinterpret_cast<const double*>(scratch + 16)[b];      // (uniform: every thread the same order)
  const float s = (float)(sum / (double)total);
  if (blockIdx.x == 0 && t == 0) scratch[0] = s;
  for (int64_t q = (int64_t)blockIdx.x * 1024 + t; q < total; q += (int64_t)gridDim.x * 1024) {
    const int64_t k = q / PP; const int a = (int)(q - k * PP);
    float* d = patches + (k * 3 + 2) * PP + a;
    *d = *d / s;
  }
  // poses_[:n] = SE3(poses_[:n] with t *= s) * SE3(poses_[0] with t *= s).inv()
  Pose X0 = load_pose(scratch + 1);
  X0.t.x *= s; X0.t.y *= s; X0.t.z *= s;
  Pose I0 = se3_inv(X0);
  I0.q = qnormalize(I0.q);                                         // (as a stored inverse read back by the product: se3.h:34 normalises on construction)
  for (int64_t i = (int64_t)blockIdx.x * 1024 + t; i < n; i += (int64_t)gridDim.x * 1024) {
    Pose X = load_pose(poses + 7 * i);
    X.t.x *= s; X.t.y *= s; X.t.z *= s;
    store_pose(poses + 7 * i, se3_mul(X, I0));
  }
}

}  // namespace

extern "C" size_t dpvo_normalize_scratch_bytes(void) { return 64 + kNormBlocks * sizeof(double); }

extern "C" int dpvo_normalize(float* poses, float* patches, int n, int M, int P, float* scratch, void* stream) {
  if (n <= 0 || M <= 0 || P <= 0 || !poses || !patches || !scratch) return DPVO_E_INVALID;
  const int PP = P * P;
  const int64_t total = (int64_t)n * M * PP;
  const int64_t g = cdiv64(total, 1024);
  const int nb = (int)(g > kNormBlocks ? kNormBlocks : g);
  hipLaunchKernelGGL(normalize_sum_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, poses, patches, total, PP, scratch);
  hipLaunchKernelGGL(normalize_apply_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, poses, patches, n, total, PP, nb, scratch);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

#define SE3_ENTRY(name, kernel, ...)                                                              \
  if (n < 0) return DPVO_E_INVALID;                                                               \
  if (n == 0) return DPVO_OK;                                                                     \
  hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);  \
  DPVO_LAUNCH_CHECK();                                                                            \
  return DPVO_OK;

extern "C" int dpvo_se3_inv(const float* X, float* Y, int64_t n, void* stream) { SE3_ENTRY(inv, se3_inv_kernel, X, Y, n) }
extern "C" int dpvo_se3_mul(const float* X, const float* Y, float* Z, int64_t n, void* stream) { SE3_ENTRY(mul, se3_mul_kernel, X, Y, Z, n) }
extern "C" int dpvo_se3_act4(const float* X, const float* p, float* q, int64_t n, void* stream) { SE3_ENTRY(act4, se3_act4_kernel, X, p, q, n) }
extern "C" int dpvo_se3_exp(const float* a, float* X, int64_t n, void* stream) { SE3_ENTRY(exp, se3_exp_kernel, a, X, n) }
extern "C" int dpvo_se3_log(const float* X, float* a, int64_t n, void* stream) { SE3_ENTRY(log, se3_log_kernel, X, a, n) }

extern "C" int dpvo_reproject(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                              const int64_t* jj, const int64_t* kk, float* coords, int64_t E, int P, int clamp_z,
                              void* stream) {
  if (E < 0 || P <= 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !coords) return DPVO_E_INVALID;
  hipLaunchKernelGGL(reproject_kernel, dim3(grid_for(E * P * P)), dim3(256), 0, (hipStream_t)stream, poses, patches,
                     intrinsics, ii, jj, kk, coords, E, P, clamp_z);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_loop_flow(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, int64_t j0,
                              int64_t n_j, int64_t i0, int64_t n_i, int M, int P, float beta, float* flow_mag, void* stream) {
  if (n_j < 0 || n_i < 0 || M <= 0 || P <= 0 || j0 < 0 || i0 < 0) return DPVO_E_INVALID;
  if (n_j == 0 || n_i == 0) return DPVO_OK;
  if (!poses || !patches || !intrinsics || !ix || !flow_mag || n_j * n_i > 0x7fffffffll) return DPVO_E_INVALID;
  const LoopFlowArgs A = {poses, patches, intrinsics, ix, j0, i0, n_i, M, P, beta, flow_mag, nullptr, 0, 0, 0, 0, 0};
  hipLaunchKernelGGL(loop_flow_kernel, dim3((unsigned)(n_j * n_i)), dim3(256), 0, (hipStream_t)stream, A);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
// (internal, common.h) the same test for the frame count the keyframe step of THIS call leaves behind: dpvo_frame_update_t.loop_out
extern "C" int dpvo_loop_flow_next(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                                   const int32_t* decision, int n, int removal_window, int keyframe_index, int freq, int max_age, int M,
                                   int P, float beta, float* out, void* stream) {
  if (!poses || !patches || !intrinsics || !ix || !decision || !out || M <= 0 || P <= 0 || n < 1 || max_age < 0) return DPVO_E_INVALID;
  const int64_t l = (int64_t)n + 1 - removal_window, n_j = freq - keyframe_index;         // (the larger of the two possible ranges: no drop)
  const int64_t n_i = l > 0 ? l - (l - max_age > 0 ? l - max_age : 0) : 0;
  const int64_t grid = (n_j > 0 && n_i > 0) ? n_j * n_i : 1;                               // (one workgroup writes the header even when empty)
  if (grid > 0x7fffffffll) return DPVO_E_INVALID;
  const LoopFlowArgs A = {poses, patches, intrinsics, ix, 0, 0, 1, M, P, beta, out, decision, n, removal_window, keyframe_index, freq, max_age};
  hipLaunchKernelGGL(loop_flow_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, A);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_flow_mag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                             const int64_t* jj, const int64_t* kk, float beta, float* flow, float* valid, int64_t E,
                             int P, void* stream) {
  if (E < 0 || P <= 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!poses || !patches || !intrinsics || !ii || !jj || !kk || !flow || !valid) return DPVO_E_INVALID;
  hipLaunchKernelGGL(flow_mag_kernel, dim3(grid_for(E)), dim3(256), 0, (hipStream_t)stream, poses, patches, intrinsics,
                     ii, jj, kk, beta, flow, valid, E, P);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_point_cloud(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                                float* points, int64_t m, int P, void* stream) {
  if (m < 0 || P <= 0) return DPVO_E_INVALID;
  if (m == 0) return DPVO_OK;
  if (!poses || !patches || !intrinsics || !ix || !points) return DPVO_E_INVALID;
  hipLaunchKernelGGL(point_cloud_kernel, dim3(grid_for(m)), dim3(256), 0, (hipStream_t)stream, poses, patches,
                     intrinsics, ix, points, m, P);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_motionmag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                              const int64_t* jj, const int64_t* kk, const int32_t* plan, int64_t E, int P, int64_t i,
                              int64_t j, float beta, float* out4, void* stream) {
  return dpvo_motionmag_status(poses, patches, intrinsics, ii, jj, kk, plan, E, P, i, j, beta, out4, nullptr, stream);
}

extern "C" int dpvo_motionmag_status(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                                     const int64_t* jj, const int64_t* kk, const int32_t* plan, int64_t E, int P, int64_t i,
                                     int64_t j, float beta, float* out4, float* status4, void* stream) {
  if (E < 0 || P <= 0 || !out4) return DPVO_E_INVALID;
  if (status4 && !(plan && E > 0)) return DPVO_E_INVALID;
  if (E > 0 && (!poses || !patches || !intrinsics || !ii || !jj || !kk)) return DPVO_E_INVALID;
  if (plan && E > 0) {
    dpvo_plan_layout_t PL;
    dpvo_plan_layout(E, &PL);
    const MotionPlanArgs A = {poses, patches, intrinsics, kk, plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, plan + PL.counts + 1,
                              P, (int)i, (int)j, beta, out4, status4, plan + PL.flow};
    hipLaunchKernelGGL(motionmag_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A);
  } else
  hipLaunchKernelGGL(motionmag_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, poses, patches, intrinsics, ii, jj,
                     kk, E, P, i, j, beta, out4);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_point_cloud_motionmag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                                          float* points, int64_t m, const int64_t* kk, const int32_t* plan, int64_t E, int P,
                                          int64_t i, int64_t j, float beta, float* out4, float* status4, void* stream) {
  if (m <= 0 || E <= 0 || P <= 0 || !poses || !patches || !intrinsics || !ix || !points || !kk || !plan || !out4) return DPVO_E_INVALID;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  const MotionPlanArgs A = {poses, patches, intrinsics, kk, plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, plan + PL.counts + 1,
                            P, (int)i, (int)j, beta, out4, status4, plan + PL.flow};
  hipLaunchKernelGGL(point_cloud_motionmag_kernel, dim3(grid_for(m) + 1), dim3(256), 0, (hipStream_t)stream, poses, patches,
                     intrinsics, ix, points, m, P, A);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
