// common.h -- shared helpers for the gfx950 kernels of libdpvo_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/dpvo_hip.h"

#define DPVO_LAUNCH_CHECK()                              \
  do {                                                   \
    hipError_t e__ = hipGetLastError();                  \
    if (e__ != hipSuccess) return (int)e__;              \
  } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bilinear blend of a 2x2 window exactly as the reference's Python evaluates it (altcorr/correlation.py:62-66):
// x00 = (1-dy)*(1-dx)*v00 ... ; out = x00 + x01 + x10 + x11, every product and sum rounded to f32 on its own.  The pragma
// keeps the compiler from contracting a*b+c into an FMA (HIP's __fmul_rn / __fadd_rn are plain operators and do not), which it
// otherwise decides per translation unit -- and differently with and without packed-FP32 ops -- so that two kernels computing
// "the same" blend would differ in the last bit.
__device__ __forceinline__ float blend4_ref(float dx, float dy, float v00, float v01, float v10, float v11) {
#pragma clang fp contract(off)
  const float ey = 1.f - dy, ex = 1.f - dx;
  const float x00 = ey * ex * v00, x01 = ey * dx * v01, x10 = dy * ex * v10, x11 = dy * dx * v11;
  return ((x00 + x01) + x10) + x11;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// F.avg_pool2d(fmap, 4, 4) on an NHWC f16 map, 8 channels per thread (shared by encoder.hip and the fused frame kernel)
__device__ __forceinline__ void pool4_nhwc_body(const _Float16* __restrict__ in, _Float16* __restrict__ out, int h, int w, int C,
                                                 int64_t bid, int64_t nblk) {
  const int h4 = h / 4, w4 = w / 4, c8 = C / 8;
  const int64_t total = (int64_t)h4 * w4 * c8;
  for (int64_t q = bid * (int64_t)blockDim.x + threadIdx.x; q < total; q += nblk * blockDim.x) {
    const int ch = (int)(q % c8);
    const int64_t pix = q / c8;
    const int px = (int)(pix % w4), py = (int)(pix / w4);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const h8 v = *reinterpret_cast<const h8*>(in + ((int64_t)(py * 4 + a) * w + px * 4 + b) * C + ch * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
      }
    h8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (_Float16)(acc[k] / 16.0f);
    *reinterpret_cast<h8*>(out + pix * C + ch * 8) = o;
  }
}

// ---- internal entry points (between the translation units of this library; not part of include/dpvo_hip.h) -----------------------
extern "C" int dpvo_plan_window_counters(int64_t E, void* ws, size_t ws_bytes, int32_t** ptr, int64_t* count);
extern "C" int dpvo_plan_build_window_job(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                                          size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo,
                                          int64_t n_patches_win, int64_t qi, int64_t qj, int counters_cleared, const float* r_poses,
                                          const float* r_patches, const float* r_intr, float* r_coords, int r_P, void* stream);
extern "C" int dpvo_loop_flow_next(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                                   const int32_t* decision, int n, int removal_window, int keyframe_index, int freq, int max_age, int M,
                                   int P, float beta, float* out, void* stream);
extern "C" int dpvo_frame_state_part_clear(dpvo_frame_state_t* p, int part, int32_t* clear_ptr, int64_t clear_count, void* stream);

