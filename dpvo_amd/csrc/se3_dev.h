// se3_dev.h -- SE3 device helpers shared by geom.hip and track.hip (math of lietorch include/se3.h:34-56, so3.h:31-60).
// Translation units that include this are built WITHOUT packed-FP32 VALU ops (csrc/Makefile NOPK): the compiler packs the
// cross products below into the operand-select form that faults on MI355X.
#pragma once
#include "common.h"

namespace {

struct Quat { float x, y, z, w; };
struct Vec3 { float x, y, z; };
struct Pose { Vec3 t; Quat q; };

// so3.h:31-37: every SO3 construction normalises the quaternion
__device__ __forceinline__ Quat qnormalize(Quat q) {
  const float n = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  return {q.x / n, q.y / n, q.z / n, q.w / n};
}
__device__ __forceinline__ Quat qmul(Quat a, Quat b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// so3.h:55-60
__device__ __forceinline__ Vec3 qrot(Quat q, Vec3 p) {
  float ux = q.y * p.z - q.z * p.y, uy = q.z * p.x - q.x * p.z, uz = q.x * p.y - q.y * p.x;
  ux += ux; uy += uy; uz += uz;
  return {p.x + q.w * ux + (q.y * uz - q.z * uy), p.y + q.w * uy + (q.z * ux - q.x * uz),
          p.z + q.w * uz + (q.x * uy - q.y * ux)};
}
__device__ __forceinline__ Pose load_pose(const float* p) {   // se3.h:34
  Pose X;
  X.t = {p[0], p[1], p[2]};
  X.q = qnormalize({p[3], p[4], p[5], p[6]});
  return X;
}
__device__ __forceinline__ void store_pose(float* p, Pose X) {
  p[0] = X.t.x; p[1] = X.t.y; p[2] = X.t.z; p[3] = X.q.x; p[4] = X.q.y; p[5] = X.q.z; p[6] = X.q.w;
}
__device__ __forceinline__ Pose se3_inv(Pose X) {              // se3.h:36-38
  Pose Y;
  Y.q = qnormalize({-X.q.x, -X.q.y, -X.q.z, X.q.w});
  Vec3 r = qrot(Y.q, X.t);
  Y.t = {-r.x, -r.y, -r.z};
  return Y;
}
__device__ __forceinline__ Pose se3_mul(Pose A, Pose B) {      // se3.h:45-47
  Pose C;
  C.q = qnormalize(qmul(A.q, B.q));
  Vec3 r = qrot(A.q, B.t);
  C.t = {A.t.x + r.x, A.t.y + r.y, A.t.z + r.z};
  return C;
}
__device__ __forceinline__ void se3_act4(Pose X, const float* p, float* o) {   // se3.h:53-56
  Vec3 r = qrot(X.q, {p[0], p[1], p[2]});
  o[0] = r.x + X.t.x * p[3]; o[1] = r.y + X.t.y * p[3]; o[2] = r.z + X.t.z * p[3]; o[3] = p[3];
}


// ---- flow test of DPVO.keyframe (dpvo.py:257-270), shared by geom.hip (dpvo_motionmag*) and track.hip (the keyframe step computes it itself)
// pops.flow_mag (projective_ops.py:120-130) for one edge: mean over the PxP pixels of
// beta*|x(Gij) - x(Gii)| + (1-beta)*|x(t-only) - x(Gii)|, and the number of valid pixels (Z > 0.2).
__device__ __forceinline__ void edge_flow(const float* __restrict__ poses, const float* __restrict__ patches,
                                          const float* __restrict__ intr, int64_t i, int64_t j, int64_t k, float beta,
                                          int P, float* flow, float* nvalid) {
  const int PP = P * P;
  const Pose Gi = load_pose(poses + 7 * i), Gj = load_pose(poses + 7 * j);
  const Pose Gi_inv = se3_inv(Gi);
  const Pose Gij = se3_mul(Gj, Gi_inv);
  const Pose Gii = se3_mul(Gi, Gi_inv);
  Pose Gt; Gt.t = Gij.t; Gt.q = {0.f, 0.f, 0.f, 1.f};          // tonly (:62-63)
  const float* Ki = intr + 4 * i; const float* Kj = intr + 4 * j;
  const float* pk = patches + k * 3 * PP;
  float fsum = 0.f, vsum = 0.f;
  for (int a = 0; a < PP; ++a) {
    const float X0[4] = {(pk[a] - Ki[2]) / Ki[0], (pk[PP + a] - Ki[3]) / Ki[1], 1.0f, pk[2 * PP + a]};
    float A0[4], A1[4], A2[4];
    se3_act4(Gii, X0, A0); se3_act4(Gij, X0, A1); se3_act4(Gt, X0, A2);
    const float d0 = 1.0f / fmaxf(A0[2], 0.1f), d1 = 1.0f / fmaxf(A1[2], 0.1f), d2 = 1.0f / fmaxf(A2[2], 0.1f);
    const float c0x = Ki[0] * (d0 * A0[0]) + Ki[2], c0y = Ki[1] * (d0 * A0[1]) + Ki[3];
    const float c1x = Kj[0] * (d1 * A1[0]) + Kj[2], c1y = Kj[1] * (d1 * A1[1]) + Kj[3];
    const float c2x = Kj[0] * (d2 * A2[0]) + Kj[2], c2y = Kj[1] * (d2 * A2[1]) + Kj[3];
    const float f1 = sqrtf((c1x - c0x) * (c1x - c0x) + (c1y - c0y) * (c1y - c0y));
    const float f2 = sqrtf((c2x - c0x) * (c2x - c0x) + (c2y - c0y) * (c2y - c0y));
    fsum += beta * f1 + (1.0f - beta) * f2;
    vsum += (A1[2] > 0.2f) ? 1.0f : 0.0f;
  }
  *flow = fsum / (float)PP;
  *nvalid = vsum;
}


// DPVO.motionmag(i,j) + motionmag(j,i) (dpvo.py:257-264,269) in one launch: sums and counts of the per-edge
// pixel-mean flow (pops.flow_mag(...).mean() averages over edges x pixels; every edge has PxP pixels) over the
// edges (i->j) and (j->i).  out = {sum_ij, n_ij, sum_ji, n_ji}.  Fixed-order tree reductions (deterministic).
__device__ __forceinline__ void block_reduce4(float (&s)[4], float (*red)[1024], float* out) {
#pragma unroll
  for (int a = 0; a < 4; ++a) red[a][threadIdx.x] = s[a];
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int a = 0; a < 4; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) out[threadIdx.x] = red[threadIdx.x][0];
}


// plan variant: the two frame pairs are looked up in the plan's pair list and only their
// ~2 x 96 edges are touched (the scan above reads all E index triples: 60 us at E = 47 712 vs ~6 us here)
struct MotionPlanArgs {
  const float *poses, *patches, *intr; const int64_t* kk;
  const int32_t *perm_p, *pair_off, *pair_ij, *n_pairs;
  int P, qi, qj; float beta; float *out, *status;
};
__device__ __forceinline__ void motionmag_plan_body(const float* __restrict__ poses, const float* __restrict__ patches,
                                                    const float* __restrict__ intr, const int64_t* __restrict__ kk,
                                                    const int32_t* __restrict__ perm_p,
                                                    const int32_t* __restrict__ pair_off,
                                                    const int32_t* __restrict__ pair_ij,
                                                    const int32_t* __restrict__ n_pairs, int P, int qi, int qj,
                                                    float beta, float* __restrict__ out,
                                                    float* __restrict__ status) {
  __shared__ float red[4][1024];
  __shared__ int found[2];
  const int ng = *n_pairs;
  // the plan's counters [n_patches, n_pairs, 0, ids-outside-the-window flag] ride along with the frame's only read-back
  if (status && threadIdx.x < 4) status[threadIdx.x] = (float)n_pairs[(int)threadIdx.x - 1];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  // both pairs located by ONE parallel scan of the pair list (a binary search is ~9 dependent global round trips per
  // pair: 12 of this kernel's 21 us); pairs are unique, so at most one thread writes each slot
  if (threadIdx.x < 2) found[threadIdx.x] = -1;
  __syncthreads();
  for (int g = threadIdx.x; g < ng; g += blockDim.x) {
    const int pi = pair_ij[2 * g], pj = pair_ij[2 * g + 1];
    if (pi == qi && pj == qj) found[0] = g;
    if (pi == qj && pj == qi) found[1] = g;
  }
  __syncthreads();
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const int a = dir ? qj : qi, b = dir ? qi : qj;
    const int g = found[dir];
    if (g >= 0) {
      for (int p = pair_off[g] + threadIdx.x; p < pair_off[g + 1]; p += blockDim.x) {
        float f, v;
        edge_flow(poses, patches, intr, a, b, kk[perm_p[p]], beta, P, &f, &v);
        s[2 * dir] += f; s[2 * dir + 1] += 1.f;
      }
    }
  }
  block_reduce4(s, red, out);
}

}  // namespace
