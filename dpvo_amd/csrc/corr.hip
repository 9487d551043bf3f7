// corr.hip -- altcorr forward for gfx950.
//
// Replaces cuda_corr.forward / patchify_forward (reference dpvo/altcorr/correlation_kernel.cu:16-47,
// 82-136,193-233) and the two-level stack of DPVO.corr (dpvo/dpvo.py:200-207).
//
// corr_pyramid_kernel (the hot one): one wave64 per edge, both pyramid levels.
//   The 9 patch-pixel templates [9 x 128ch] form the A operand (16 x 32-wide MFMA rows, 7 rows of
//   zero padding) and are loaded ONCE per edge for both levels.  The target features inside the
//   bounding box of the nine 8x8 windows form the B operand: lane (n = lane&15, kg = lane>>4) reads
//   the 16 contiguous bytes of channels [(4s+kg)*8, +8) of window position n straight from the
//   channels-last pyramid into its MFMA fragment (every byte of the window is read exactly once, in
//   full 64-byte sectors; no LDS staging of features is needed because the window is used by one
//   MFMA chain only).  v_mfma_f32_16x16x32_f16 accumulates the [9 x positions] raw dot products in
//   f32; they go to LDS, and each lane then blends 7 of the 441 bilinear outputs per level and
//   writes them as packed (level0, level1) half2 words -> one coalesced 1764-byte row per edge in
//   the exact [E,882] layout the update operator's first Linear expects.
//   HBM/L2-bound by design: ~52.9 KB of algorithmic traffic per edge for 0.3 MFLOP.
#include "corr_dev.h"

// WIDE: ld_out == 896 and a 16-byte aligned output (what the tracker allocates): the row incl. its 14 zero padding columns leaves as
// 112 sixteen-byte pieces, two per lane, instead of seven 4-byte pieces per lane and a padding loop.
template <int OCC, bool WIDE>
__global__ __launch_bounds__(64, OCC) void corr_pyramid_kernel(
    const _Float16* __restrict__ gmap, const _Float16* __restrict__ fmap0, const _Float16* __restrict__ fmap1,
    const float* __restrict__ coords, const int64_t* __restrict__ us, const int64_t* __restrict__ vs,
    const int32_t* __restrict__ order, _Float16* __restrict__ out, int64_t ld_out, int64_t E, int H0, int W0,
    int H1, int W1, int N1, int N2) {
  __shared__ CorrShared sm;
  const int lane = threadIdx.x;
  if (lane < 14) sm.orow[2 * CORR_NOUT + lane] = (_Float16)0;       // the padding columns 882..895: written once, never overwritten
  const int64_t nblk = order ? ((E + 7) >> 3) << 3 : E;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int64_t e = blk;
    if (order) {
      // XCD-aware walk of the hinted order: workgroup b runs on XCD b % 8 (observed, a speed hint only), so XCD x
      // is handed the contiguous slice [x*per, (x+1)*per) of `order` and its L2 sees one target frame at a time.
      const int64_t per = (E + 7) >> 3;
      const int64_t v = (blk & 7) * per + (blk >> 3);
      if (v >= E) continue;
      e = order[v];
    }
    CORR_T(0);
    // ring-buffer indices (dpvo.py:202-203: ii % (M * pmem), jj % mem), reduced here instead of by two elementwise launches
    // (non-negative and < 2^31 by the entry's contract: unsigned remainders)
    const int64_t u = (unsigned)us[e] % (unsigned)N1, v = (unsigned)vs[e] % (unsigned)N2;
    // A fragments: template pixel m = lane&15 (<9), channels [(4s+kg)*8, +8)
    h8 a[4];
    {
      const int m = lane & 15, kg = lane >> 4;
      if (m < CORR_NPIX) {
        const h8* src = reinterpret_cast<const h8*>(gmap + ((int64_t)u * CORR_NPIX + m) * CORR_C) + kg;
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = src[4 * s];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = (h8)(_Float16)0;
      }
    }
    float cx = 0.f, cy = 0.f;
    if (lane < CORR_NPIX) {
      cx = coords[e * 18 + lane];
      cy = coords[e * 18 + 9 + lane];
    }
#ifdef FU_TRACE
    { float s0 = cx + (float)a[0][0]; asm volatile("" :: "v"(s0)); }        // (stamp 1 = indices, coordinates and templates have landed)
#endif
    CORR_T(1);
    corr_level(a, fmap0 + (int64_t)v * H0 * W0 * CORR_C, H0, W0, cx, cy, sm, lane, 0);
    CORR_T(3);
    corr_level(a, fmap1 + (int64_t)v * H1 * W1 * CORR_C, H1, W1, cx * 0.25f, cy * 0.25f, sm, lane, 1);
    CORR_T(5);
    if constexpr (WIDE) {
      const u4* src = reinterpret_cast<const u4*>(sm.orow);
      u4* dst = reinterpret_cast<u4*>(out + e * ld_out);
      dst[lane] = src[lane];
      if (lane < CORR_ROW_BYTES / 16 - 64) dst[64 + lane] = src[64 + lane];
    } else {
      // coalesced row store: 441 packed (level0, level1) words
      const uint32_t* src = reinterpret_cast<const uint32_t*>(sm.orow);
      uint32_t* dst = reinterpret_cast<uint32_t*>(out + e * ld_out);
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const int q = lane + 64 * s;
        if (q < CORR_NOUT) dst[q] = src[q];
      }
      // zero the padding columns [882, ld_out)
      for (int64_t c = 2 * CORR_NOUT + lane; c < ld_out; c += 64) out[e * ld_out + c] = (_Float16)0;
    }
    __syncthreads();
    CORR_T(6);
  }
}

// ---------------------------------------------------------------------------------------------------
// Generic (any stride, f16/f32, any C/P/radius) single-level kernel: API-parity path of cuda_corr.forward.
// One thread per output element (e, a, b, i0, j0): 4 window dot products + bilinear.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void corr_generic_kernel(const T* __restrict__ f1, int64_t s1n, int64_t s1c, int64_t s1i, int64_t s1j,
                                    const T* __restrict__ f2, int64_t s2n, int64_t s2c, int64_t s2h, int64_t s2w,
                                    const float* __restrict__ coords, float cscale, const int64_t* __restrict__ us,
                                    const int64_t* __restrict__ vs, T* __restrict__ out, int64_t E, int C, int P,
                                    int H2, int W2, int radius) {
  const int Dm = 2 * radius + 1;
  const int64_t total = E * Dm * Dm * P * P;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = n;
    const int j0 = t % P; t /= P;
    const int i0 = t % P; t /= P;
    const int b = t % Dm; t /= Dm;
    const int a = t % Dm; t /= Dm;
    const int64_t e = t;
    const float x = coords[((e * 2 + 0) * P + i0) * P + j0] * cscale;
    const float y = coords[((e * 2 + 1) * P + i0) * P + j0] * cscale;
    const int fx = safe_floor_int(x), fy = safe_floor_int(y);
    const float dx = x - floorf(x), dy = y - floorf(y);
    const T* p1 = f1 + us[e] * s1n + i0 * s1i + j0 * s1j;
    const T* p2 = f2 + vs[e] * s2n;
    float c[2][2];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int i1 = fy + (a + aa - radius), j1 = fx + (b + bb - radius);
        float s = 0.f;
        if (i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2) {
          const T* q2 = p2 + i1 * s2h + j1 * s2w;
          for (int ch = 0; ch < C; ++ch) s += (float)p1[ch * s1c] * (float)q2[ch * s2c];
        }
        c[aa][bb] = s;
      }
    float o = (1.f - dx) * (1.f - dy) * c[0][0];
    o += dx * (1.f - dy) * c[0][1];
    o += (1.f - dx) * dy * c[1][0];
    o += dx * dy * c[1][1];
    out[n] = (T)o;
  }
}

template <typename T>
__global__ void patchify_kernel(const T* __restrict__ net, int64_t sc, int64_t sh, int64_t sw,
                                const float* __restrict__ coords, T* __restrict__ out, int64_t M, int C, int H, int W,
                                int radius) {
  const int D = 2 * radius + 2;
  const int64_t total = M * C * D * D;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = n;
    const int b = t % D; t /= D;
    const int a = t % D; t /= D;
    const int c = t % C; t /= C;
    const int64_t m = t;
    const int i = safe_floor_int(coords[2 * m + 1]) + (a - radius);
    const int j = safe_floor_int(coords[2 * m + 0]) + (b - radius);
    T v = (T)0.f;
    if (i >= 0 && i < H && j >= 0 && j < W) v = net[c * sc + i * sh + j * sw];
    out[n] = v;
  }
}


// patchify + the Python bilinear blend of altcorr.patchify (correlation.py:51-68) in one launch:
// out[m][c][a][b] = blend of the (2R+2)^2 integer-floor window (zeros when OOB) with the centroid's (dx,dy).
template <typename T>
__global__ void patchify_bilinear_kernel(const T* __restrict__ net, int64_t sc, int64_t sh, int64_t sw,
                                         const float* __restrict__ coords, T* __restrict__ out, int64_t M, int C, int H,
                                         int W, int radius) {
  const int d = 2 * radius + 1;
  const int64_t total = M * C * d * d;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < total; n += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = n;
    const int b = t % d; t /= d;
    const int a = t % d; t /= d;
    const int c = t % C; t /= C;
    const int64_t m = t;
    const float x = coords[2 * m + 0], y = coords[2 * m + 1];
    const float dx = x - floorf(x), dy = y - floorf(y);
    const int i0 = safe_floor_int(y) + (a - radius), j0 = safe_floor_int(x) + (b - radius);
    float v[2][2];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int i = i0 + aa, j = j0 + bb;
        v[aa][bb] = (i >= 0 && i < H && j >= 0 && j < W) ? (float)net[c * sc + i * sh + j * sw] : 0.f;
      }
    const float o = blend4_ref(dx, dy, v[0][0], v[0][1], v[1][0], v[1][1]);
    out[n] = (T)o;
  }
}

extern "C" int dpvo_corr_pyramid_forward(const void* gmap, const void* fmap0, const void* fmap1, const float* coords,
                                         const int64_t* us, const int64_t* vs, const int32_t* order, void* out,
                                         int64_t ld_out, int64_t E, int C, int P, int64_t N1, int64_t N2, int H0,
                                         int W0, int H1, int W1, int radius, void* stream) {
  if (C != CORR_C || P != CORR_P || radius != CORR_R) return DPVO_E_UNSUPPORTED;
  if (E < 0 || ld_out < 2 * CORR_NOUT || (ld_out & 1)) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!gmap || !fmap0 || !fmap1 || !coords || !us || !vs || !out) return DPVO_E_INVALID;
  if (N1 <= 0 || N2 <= 0 || N1 > 0x7fffffff || N2 > 0x7fffffff) return DPVO_E_INVALID;
  // with an order hint the grid is padded to 8 slices of ceil(E/8) so that the XCD remap is a bijection onto [0,E)
  const int64_t grid = order ? ((E + 7) >> 3) << 3 : E;
  // three waves per SIMD (launch bound of the kernel): 2 / 3 measure the same, 4 and 5 are slower (263 -> 275 -> 332 us, round 2).
  // Rounds 1-3 read DPVO_CORR_OCC from the environment once per process into a static; a library entry has no business doing
  // either: one instantiation, no state.
  if (ld_out == CORR_ROW_BYTES / 2 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL((corr_pyramid_kernel<3, true>), dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)gmap, (const _Float16*)fmap0, (const _Float16*)fmap1, coords, us, vs, order,
                       (_Float16*)out, ld_out, E, H0, W0, H1, W1, (int)N1, (int)N2);
  else
    hipLaunchKernelGGL((corr_pyramid_kernel<3, false>), dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream,
                       (const _Float16*)gmap, (const _Float16*)fmap0, (const _Float16*)fmap1, coords, us, vs, order,
                       (_Float16*)out, ld_out, E, H0, W0, H1, W1, (int)N1, (int)N2);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_corr_forward(const void* fmap1, const int64_t* s1, const void* fmap2, const int64_t* s2,
                                 const float* coords, float coord_scale, const int64_t* us, const int64_t* vs,
                                 void* out, int dtype, int64_t E, int C, int P, int64_t N1, int64_t N2, int H2, int W2,
                                 int radius, void* stream) {
  if (E < 0 || C <= 0 || P <= 0 || radius < 0 || !s1 || !s2) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!fmap1 || !fmap2 || !coords || !us || !vs || !out) return DPVO_E_INVALID;
  if (N1 <= 0 || N2 <= 0 || N1 > 0x7fffffff || N2 > 0x7fffffff) return DPVO_E_INVALID;
  const int Dm = 2 * radius + 1;
  const int64_t total = E * Dm * Dm * P * P;
  const int64_t grid = cdiv64(total, 256) < 65536 * 4 ? cdiv64(total, 256) : 65536 * 4;
  if (dtype == DPVO_F16) {
    hipLaunchKernelGGL(corr_generic_kernel<_Float16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)fmap1, s1[0], s1[1], s1[2], s1[3], (const _Float16*)fmap2, s2[0], s2[1], s2[2],
                       s2[3], coords, coord_scale, us, vs, (_Float16*)out, E, C, P, H2, W2, radius);
  } else if (dtype == DPVO_F32) {
    hipLaunchKernelGGL(corr_generic_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)fmap1, s1[0], s1[1], s1[2], s1[3], (const float*)fmap2, s2[0], s2[1], s2[2],
                       s2[3], coords, coord_scale, us, vs, (float*)out, E, C, P, H2, W2, radius);
  } else {
    return DPVO_E_UNSUPPORTED;
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_patchify_forward(const void* net, const int64_t* sn, const float* coords, void* out, int dtype,
                                     int64_t M, int C, int H, int W, int radius, void* stream) {
  if (M < 0 || C <= 0 || radius < 0 || !sn) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (!net || !coords || !out) return DPVO_E_INVALID;
  const int D = 2 * radius + 2;
  const int64_t total = M * C * D * D;
  const int64_t grid = cdiv64(total, 256) < 65536 ? cdiv64(total, 256) : 65536;
  if (dtype == DPVO_F16)
    hipLaunchKernelGGL(patchify_kernel<_Float16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)net, sn[0], sn[1], sn[2], coords, (_Float16*)out, M, C, H, W, radius);
  else if (dtype == DPVO_F32)
    hipLaunchKernelGGL(patchify_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)net, sn[0], sn[1], sn[2], coords, (float*)out, M, C, H, W, radius);
  else
    return DPVO_E_UNSUPPORTED;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_patchify_bilinear(const void* net, const int64_t* sn, const float* coords, void* out, int dtype,
                                      int64_t M, int C, int H, int W, int radius, void* stream) {
  if (M < 0 || C <= 0 || radius < 0 || !sn) return DPVO_E_INVALID;
  if (M == 0) return DPVO_OK;
  if (!net || !coords || !out) return DPVO_E_INVALID;
  const int d = 2 * radius + 1;
  const int64_t total = M * C * d * d;
  const int64_t grid = cdiv64(total, 256) < 65536 ? cdiv64(total, 256) : 65536;
  if (dtype == DPVO_F16)
    hipLaunchKernelGGL(patchify_bilinear_kernel<_Float16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)net, sn[0], sn[1], sn[2], coords, (_Float16*)out, M, C, H, W, radius);
  else if (dtype == DPVO_F32)
    hipLaunchKernelGGL(patchify_bilinear_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)net, sn[0], sn[1], sn[2], coords, (float*)out, M, C, H, W, radius);
  else
    return DPVO_E_UNSUPPORTED;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

#ifdef FU_TRACE
extern "C" int dpvo_debug_corr_trace_buffer(void* buf) {      // trace builds only: device buffer of 65536*8 u64, or NULL
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_corr_trace), &p, sizeof(p));
}
#endif
