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

}  // namespace
