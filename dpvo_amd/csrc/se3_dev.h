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


// ---- reprojection (pops.transform, projective_ops.py:43), shared by geom.hip (dpvo_reproject) and graph.hip (it can ride in the
//      plan's histogram launch): one thread per (edge, patch pixel), blocks bid of nblk
__device__ __forceinline__ void reproject_body(const float* __restrict__ poses, const float* __restrict__ patches,
                                 const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                 const int64_t* __restrict__ jj, const int64_t* __restrict__ kk,
                                 float* __restrict__ coords, int64_t E, int P, int clamp_z, int64_t bid, int64_t nblk) {
  const int PP = P * P;
  const int64_t total = E * PP;
  for (int64_t n = bid * (int64_t)blockDim.x + threadIdx.x; n < total; n += nblk * blockDim.x) {
    const int64_t e = n / PP;
    const int a = (int)(n - e * PP);
    const int64_t i = ii[e], j = jj[e], k = kk[e];
    const float* pk = patches + k * 3 * PP;
    float x1, y1;
    if (clamp_z) {
      const Pose Gij = se3_mul(load_pose(poses + 7 * j), se3_inv(load_pose(poses + 7 * i)));
      const float* Ki = intr + 4 * i; const float* Kj = intr + 4 * j;
      const float X0[4] = {(pk[a] - Ki[2]) / Ki[0], (pk[PP + a] - Ki[3]) / Ki[1], 1.0f, pk[2 * PP + a]};
      float X1[4];
      se3_act4(Gij, X0, X1);
      const float d = 1.0f / fmaxf(X1[2], 0.1f);
      x1 = Kj[0] * (d * X1[0]) + Kj[2];
      y1 = Kj[1] * (d * X1[1]) + Kj[3];
    } else {
      const float* pi = poses + 7 * i; const float* pj = poses + 7 * j;
      const Quat qi = {pi[3], pi[4], pi[5], pi[6]}, qj = {pj[3], pj[4], pj[5], pj[6]};
      Quat qij;
      qij.x = -qj.w * qi.x + qj.x * qi.w - qj.y * qi.z + qj.z * qi.y;
      qij.y = -qj.w * qi.y + qj.y * qi.w - qj.z * qi.x + qj.x * qi.z;
      qij.z = -qj.w * qi.z + qj.z * qi.w - qj.x * qi.y + qj.y * qi.x;
      qij.w = qj.w * qi.w + qj.x * qi.x + qj.y * qi.y + qj.z * qi.z;
      const Vec3 r = qrot(qij, {pi[0], pi[1], pi[2]});
      const float tij[3] = {pj[0] - r.x, pj[1] - r.y, pj[2] - r.z};
      const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
      const float X0[4] = {(pk[a] - cx) / fx, (pk[PP + a] - cy) / fy, 1.0f, pk[2 * PP + a]};
      const Vec3 R = qrot(qij, {X0[0], X0[1], X0[2]});
      const float X = R.x + X0[3] * tij[0], Y = R.y + X0[3] * tij[1], Z = R.z + X0[3] * tij[2];
      x1 = fx * (X / Z) + cx;
      y1 = fy * (Y / Z) + cy;
    }
    coords[(e * 2 + 0) * PP + a] = x1;
    coords[(e * 2 + 1) * PP + a] = y1;
  }
}


// ---- flow test of DPVO.keyframe (dpvo.py:257-270), shared by geom.hip (dpvo_motionmag*) and track.hip (the keyframe step computes it itself)
// pops.flow_mag (projective_ops.py:120-130) for one edge: mean over the PxP pixels of
// beta*|x(Gij) - x(Gii)| + (1-beta)*|x(t-only) - x(Gii)|, and the number of valid pixels (Z > 0.2).
// the three transforms of one frame pair and its intrinsics (loaded once per pair, not per edge)
struct FlowPair { Pose Gii, Gij, Gt; float Ki[4], Kj[4]; };
__device__ __forceinline__ FlowPair flow_pair(const float* __restrict__ poses, const float* __restrict__ intr, int64_t i, int64_t j) {
  FlowPair F;
  const Pose Gi = load_pose(poses + 7 * i), Gj = load_pose(poses + 7 * j);
#pragma unroll
  for (int a = 0; a < 4; ++a) { F.Ki[a] = intr[4 * i + a]; F.Kj[a] = intr[4 * j + a]; }
  const Pose Gi_inv = se3_inv(Gi);
  F.Gij = se3_mul(Gj, Gi_inv);
  F.Gii = se3_mul(Gi, Gi_inv);
  F.Gt.t = F.Gij.t; F.Gt.q = {0.f, 0.f, 0.f, 1.f};        // tonly (:62-63)
  return F;
}
__device__ __forceinline__ void flow_pixel(const FlowPair& F, float x, float y, float dd, float beta, float& fsum, float& vsum) {
  const float* Ki = F.Ki; const float* Kj = F.Kj;
  const float X0[4] = {(x - Ki[2]) / Ki[0], (y - Ki[3]) / Ki[1], 1.0f, dd};
  float A0[4], A1[4], A2[4];
  se3_act4(F.Gii, X0, A0); se3_act4(F.Gij, X0, A1); se3_act4(F.Gt, X0, A2);
  const float d0 = 1.0f / fmaxf(A0[2], 0.1f), d1 = 1.0f / fmaxf(A1[2], 0.1f), d2 = 1.0f / fmaxf(A2[2], 0.1f);
  const float c0x = Ki[0] * (d0 * A0[0]) + Ki[2], c0y = Ki[1] * (d0 * A0[1]) + Ki[3];
  const float c1x = Kj[0] * (d1 * A1[0]) + Kj[2], c1y = Kj[1] * (d1 * A1[1]) + Kj[3];
  const float c2x = Kj[0] * (d2 * A2[0]) + Kj[2], c2y = Kj[1] * (d2 * A2[1]) + Kj[3];
  const float f1 = sqrtf((c1x - c0x) * (c1x - c0x) + (c1y - c0y) * (c1y - c0y));
  const float f2 = sqrtf((c2x - c0x) * (c2x - c0x) + (c2y - c0y) * (c2y - c0y));
  fsum += beta * f1 + (1.0f - beta) * f2;
  vsum += (A1[2] > 0.2f) ? 1.0f : 0.0f;
}
// one edge of a pair: P == 3 (every configuration of the reference) has its 27 patch values fetched up front -- as a loop over
// a run-time P x P the nine pixels were nine dependent global round trips, 9 of the flow test's 17 us
__device__ __forceinline__ void edge_flow_pair(const FlowPair& F, const float* __restrict__ patches, int64_t k, float beta, int P,
                                               float* flow, float* nvalid) {
  const int PP = P * P;
  const float* pk = patches + k * 3 * PP;
  float fsum = 0.f, vsum = 0.f;
  if (P == 3) {
    float v[27];
#pragma unroll
    for (int a = 0; a < 27; ++a) v[a] = pk[a];
#pragma unroll
    for (int a = 0; a < 9; ++a) flow_pixel(F, v[a], v[9 + a], v[18 + a], beta, fsum, vsum);
  } else {
    for (int a = 0; a < PP; ++a) flow_pixel(F, pk[a], pk[PP + a], pk[2 * PP + a], beta, fsum, vsum);
  }
  *flow = fsum / (float)PP;
  *nvalid = vsum;
}
__device__ __forceinline__ void edge_flow(const float* __restrict__ poses, const float* __restrict__ patches,
                                          const float* __restrict__ intr, int64_t i, int64_t j, int64_t k, float beta,
                                          int P, float* flow, float* nvalid) {
  const FlowPair F = flow_pair(poses, intr, i, j);
  edge_flow_pair(F, patches, k, beta, P, flow, nvalid);
}


// DPVO.motionmag(i,j) + motionmag(j,i) (dpvo.py:257-264,269) in one launch: sums and counts of the per-edge
// pixel-mean flow (pops.flow_mag(...).mean() averages over edges x pixels; every edge has PxP pixels) over the
// edges (i->j) and (j->i).  out = {sum_ij, n_ij, sum_ji, n_ji}.  Fixed order (deterministic, and the same for every block size
// as long as a thread holds the same edges): butterfly inside a wave, then the waves in order.
__device__ __forceinline__ void block_reduce4(float (&s)[4], float (*red)[16], float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float v = s[a];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[a][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float v = 0.f;
    for (int w = 0; w < nw; ++w) v += red[threadIdx.x][w];
    out[threadIdx.x] = v;
  }
}


// plan variant: the two frame pairs are looked up in the plan's pair list and only their
// ~2 x 96 edges are touched (the scan above reads all E index triples: 60 us at E = 47 712 vs ~6 us here)
struct MotionPlanArgs {
  const float *poses, *patches, *intr; const int64_t* kk;
  const int32_t *perm_p, *pair_off, *pair_ij, *n_pairs;
  int P, qi, qj; float beta; float *out, *status;
  const int32_t* flow;      // dpvo_plan_layout_t.flow of the same plan, or NULL
};
constexpr int kFlowPx = 256;       // pixel-parallel flow test: at most this many edges per direction (else one thread per edge)
__device__ __forceinline__ void motionmag_plan_body(const float* __restrict__ poses, const float* __restrict__ patches,
                                                    const float* __restrict__ intr, const int64_t* __restrict__ kk,
                                                    const int32_t* __restrict__ perm_p,
                                                    const int32_t* __restrict__ pair_off,
                                                    const int32_t* __restrict__ pair_ij,
                                                    const int32_t* __restrict__ n_pairs, int P, int qi, int qj,
                                                    float beta, float* __restrict__ out,
                                                    float* __restrict__ status, const int32_t* __restrict__ flow = nullptr) {
  __shared__ float red[4][16];
  __shared__ int found[2];
  __shared__ int klist[2][kFlowPx];                     // patch ids of the pair's edges, both directions, in edge order
  __shared__ FlowPair fpair[2];
  __shared__ float pxl[2][kFlowPx * 9];                 // per-pixel flow terms
  const int tid = threadIdx.x, nt = blockDim.x;
  // ---- 1. the two edge lists.  The plan may carry them (dpvo_plan_build_window_flow): header, patch ids and the plan's counters
  //         travel together -- one round trip instead of the four dependent ones of the walk (pair list -> offsets -> edge -> patch id)
  int hq[4] = {-1, -1, 0, 0}, kf = 0, kb = 0;
  if (flow) {
#pragma unroll
    for (int a = 0; a < 4; ++a) hq[a] = flow[a];
    if (tid < DPVO_PLAN_FLOW_CAP && tid < kFlowPx) { kf = flow[4 + tid]; kb = flow[4 + DPVO_PLAN_FLOW_CAP + tid]; }
  }
  const int ng = *n_pairs;
  const int outside = n_pairs[2];                       // ids outside the window the plan was sized for: its bins are not to be trusted
  // the plan's counters [n_patches, n_pairs, 0, ids-outside-the-window flag] ride along with the frame's only read-back
  if (status && tid < 4) status[tid] = (float)n_pairs[tid - 1];
  int n0, n1, p0[2] = {0, 0};
  const bool listed = flow && qi >= 0 && hq[0] == qi && hq[1] == qj && !outside && hq[2] >= 0 && hq[3] >= 0 &&
                      hq[2] <= DPVO_PLAN_FLOW_CAP && hq[3] <= DPVO_PLAN_FLOW_CAP && hq[2] <= kFlowPx && hq[3] <= kFlowPx;
  if (listed) {
    n0 = hq[2]; n1 = hq[3];
    if (tid < kFlowPx) { klist[0][tid] = kf; klist[1][tid] = kb; }
  } else {
    // both pairs located by ONE parallel scan of the pair list (a binary search is ~9 dependent global round trips per
    // pair); pairs are unique, so at most one thread writes each slot
    if (tid < 2) found[tid] = -1;
    __syncthreads();
    for (int g = tid; g < ng; g += nt) {
      const int pi = pair_ij[2 * g], pj = pair_ij[2 * g + 1];
      if (pi == qi && pj == qj) found[0] = g;
      if (pi == qj && pj == qi) found[1] = g;
    }
    __syncthreads();
    int cnt[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int g = found[d];
      cnt[d] = 0;
      if (g >= 0) { p0[d] = pair_off[g]; cnt[d] = pair_off[g + 1] - p0[d]; }
    }
    n0 = cnt[0]; n1 = cnt[1];
    if (n0 <= kFlowPx && n1 <= kFlowPx) {
      if (tid < n0) klist[0][tid] = (int)kk[perm_p[p0[0] + tid]];
      if (tid < n1) klist[1][tid] = (int)kk[perm_p[p0[1] + tid]];
    }
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (P == 3 && n0 <= kFlowPx && n1 <= kFlowPx && n0 <= nt && n1 <= nt) {
    // ---- 2. one work item per (edge, pixel): a thread per EDGE ran ~3 500 dependent VALU instructions (IEEE divisions, square
    //         roots, three SE3 actions per pixel) on one wave per SIMD -- 10 of the flow test's 14 us.  The pair's transforms are
    //         built once (threads 0 and 64: two waves) and shared through LDS; a pair that has edges has valid frame ids.
    __syncthreads();                                   // (the lists)
    const int items = 9 * (n0 + n1);
    float px[8][3];                                    // (256-thread callers: up to 18 x 256 items -> a few per thread, fetched up front)
    int cnt = 0;
    for (int w = tid; w < items && cnt < 8; w += nt, ++cnt) {
      const int d = w >= 9 * n0, l = w - (d ? 9 * n0 : 0), e = l / 9, a = l - 9 * e;
      const float* pk = patches + (int64_t)klist[d][e] * 27;
      px[cnt][0] = pk[a]; px[cnt][1] = pk[9 + a]; px[cnt][2] = pk[18 + a];
    }
    if (tid == 0 && n0 > 0) fpair[0] = flow_pair(poses, intr, qi, qj);
    if (tid == 64 % nt && n1 > 0) fpair[1] = flow_pair(poses, intr, qj, qi);
    __syncthreads();
    cnt = 0;
    for (int w = tid; w < items; w += nt, ++cnt) {
      const int d = w >= 9 * n0, l = w - (d ? 9 * n0 : 0);
      float x, y, dd;
      if (cnt < 8) { x = px[cnt][0]; y = px[cnt][1]; dd = px[cnt][2]; }
      else { const int e = l / 9, a = l - 9 * e; const float* pk = patches + (int64_t)klist[d][e] * 27; x = pk[a]; y = pk[9 + a]; dd = pk[18 + a]; }
      float f = 0.f, v = 0.f;
      flow_pixel(fpair[d], x, y, dd, beta, f, v);
      pxl[d][l] = f;
    }
    __syncthreads();
    // per-edge pixel sums in pixel order, thread t = edge t of either direction (the order the per-edge loop had)
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (tid < (d ? n1 : n0)) {
        float fs = 0.f;
#pragma unroll
        for (int a = 0; a < 9; ++a) fs += pxl[d][9 * tid + a];
        s[2 * d] += fs / 9.0f; s[2 * d + 1] += 1.f;
      }
  } else {
    // ---- 2'. general patch size / very wide pairs: one thread per edge
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int nd = d ? n1 : n0;
      if (tid < nd) {
        const FlowPair FP = flow_pair(poses, intr, d ? qj : qi, d ? qi : qj);
        for (int t = tid; t < nd; t += nt) {
          const int64_t k = (listed || (n0 <= kFlowPx && n1 <= kFlowPx)) ? (int64_t)klist[d][t] : kk[perm_p[p0[d] + t]];
          float f, v;
          edge_flow_pair(FP, patches, k, beta, P, &f, &v);
          s[2 * d] += f; s[2 * d + 1] += 1.f;
        }
      }
    }
  }
  block_reduce4(s, red, out);
}

}  // namespace
