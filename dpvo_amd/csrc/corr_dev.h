// corr_dev.h -- device building blocks of corr_pyramid_kernel (corr.hip): the bounding-box MFMA pass and the per-level blend.  Shared
// with tools/probes/corr_variant.hip, the measurement kernel whose levels can be switched off / made cache-hot (results wrong on
// purpose; never part of the product library).
#pragma once
#include "common.h"

#ifdef FU_TRACE
// per-edge timeline (100 MHz wall clock) of corr_pyramid_kernel for tools/corr_trace.py: [edge slot 65536][8 stamps]
__device__ unsigned long long* g_corr_trace = nullptr;
#define CORR_T(i) do { if (g_corr_trace && threadIdx.x == 0 && blockIdx.x < 65536) g_corr_trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define CORR_T(i) do {} while (0)
#endif

#define CORR_C 128
#define CORR_P 3
#define CORR_R 3
#define CORR_D 8
#define CORR_NPIX 9
#define CORR_NOUT 441          // 7*7*9 outputs per level
#define CORR_MAXPOS 144        // bounding boxes up to 144 positions (e.g. 12x12) take the single-pass path

__device__ __forceinline__ int safe_floor_int(float v) {
  float f = floorf(v);
  // non-finite / huge coordinates: clamp so that every window is out of bounds (reference result is 0
  // for the dot products; dx = v - floor(v) still propagates NaN exactly as the reference does).
  if (!(f > -1.0e6f)) f = -1.0e6f;   // also catches NaN
  if (f > 1.0e6f) f = 1.0e6f;
  return (int)f;
}

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h8 as_h8(u4 v) { return __builtin_bit_cast(h8, v); }

// raw[m * np + pos] for m < 9, pos < np = bw*bh: dot(template m, feature(y0+pos/bw, x0+pos%bw)), 0 if OOB.
// Features are fetched with bounds-checked buffer loads (out-of-image / padding lanes get an offset
// beyond num_records and read zeros without a branch); CORR_U tiles (= 16 x 16 B per lane) are put in
// flight before the first MFMA consumes them.
#define CORR_U 4
__device__ __forceinline__ void corr_bbox_mfma(const h8 (&a)[4], __amdgpu_buffer_rsrc_t rsrc, int H, int W, int x0,
                                               int y0, int bw, int bh, float* __restrict__ raw, int lane) {
  const int np = bw * bh;
  const int n = lane & 15, kg = lane >> 4;
  const int ntiles = (np + 15) >> 4;
  const float inv_bw = 1.0f / (float)bw;   // pos / bw via float: exact for pos < 2^12 (distance to an integer >= 0.5/bw)
  for (int t0 = 0; t0 < ntiles; t0 += CORR_U) {
    u4 b[CORR_U][4];
#pragma unroll
    for (int u = 0; u < CORR_U; ++u) {
      const int pos = (t0 + u) * 16 + n;
      const int py = (int)(((float)pos + 0.5f) * inv_bw), px = pos - py * bw;
      const int y = y0 + py, x = x0 + px;
      const bool ok = (pos < np) && (y >= 0) && (y < H) && (x >= 0) && (x < W);
      const unsigned voff = ok ? (unsigned)(((y * W + x) * CORR_C + kg * 8) * 2) : 0x80000000u;
#pragma unroll
      for (int s = 0; s < 4; ++s) b[u][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + s * 64, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < CORR_U; ++u) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], as_h8(b[u][s]), acc, 0, 0, 0);
      // D[row = 4*(lane>>4)+r][col = lane&15]; rows are the patch pixels (only 0..8 are real)
      const int pos = (t0 + u) * 16 + n;
      if (pos < np) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * kg + r;
          if (m < CORR_NPIX) raw[m * np + pos] = acc[r];
        }
      }
    }
  }
}

// One level: blends this level's 441 outputs from the raw volume into the LDS row image
// orow[q*2 + level] (f16), q = ((x*7+y)*3+i0)*3+j0.
__device__ __forceinline__ void corr_level(const h8 (&a)[4], const _Float16* fmap, int H, int W,
                                           float cx, float cy, float* __restrict__ raw, int* __restrict__ meta_i,
                                           float* __restrict__ meta_f, int lane, _Float16* __restrict__ orow,
                                           int level) {
  // lanes 0..8 own one patch pixel each
  int fx = safe_floor_int(cx), fy = safe_floor_int(cy);
  float dx = cx - floorf(cx), dy = cy - floorf(cy);
  int mnx = (lane < CORR_NPIX) ? fx : INT_MAX, mxx = (lane < CORR_NPIX) ? fx : INT_MIN;
  int mny = (lane < CORR_NPIX) ? fy : INT_MAX, mxy = (lane < CORR_NPIX) ? fy : INT_MIN;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, o)); mxx = max(mxx, __shfl_xor(mxx, o));
    mny = min(mny, __shfl_xor(mny, o)); mxy = max(mxy, __shfl_xor(mxy, o));
  }
  mnx = __builtin_amdgcn_readfirstlane(mnx); mxx = __builtin_amdgcn_readfirstlane(mxx);
  mny = __builtin_amdgcn_readfirstlane(mny); mxy = __builtin_amdgcn_readfirstlane(mxy);
  const int64_t bw64 = (int64_t)mxx - mnx + CORR_D, bh64 = (int64_t)mxy - mny + CORR_D;
  const bool single = (bw64 * bh64 <= CORR_MAXPOS);

  if (lane < CORR_NPIX) { meta_f[lane] = dx; meta_f[16 + lane] = dy; }
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)fmap, (short)0, H * W * CORR_C * 2, 0x00020000);

  if (single) {
    const int bw = (int)bw64, bh = (int)bh64;
    const int x0 = mnx - CORR_R, y0 = mny - CORR_R;
    if (lane < CORR_NPIX) { meta_i[lane] = fx - mnx; meta_i[16 + lane] = fy - mny; }
    corr_bbox_mfma(a, rsrc, H, W, x0, y0, bw, bh, raw, lane);
    __syncthreads();
    CORR_T(2 + 2 * level);
    const int np = bw * bh;
    // Blend (correlation_kernel.cu:221-230): lane = (patch pixel p = lane % 9, window column bx = lane / 9) walks the 7 window
    // rows ay of its column.  Its four weights are computed once, every raw row is read once (the lower pair of row ay is the
    // upper pair of row ay + 1), no index arithmetic is left inside the loop -- rounds 1-3 gave every lane 7 arbitrary outputs
    // (q = lane + 64 s: two integer divisions, six LDS look-ups and the weight products PER OUTPUT): ~280 of the kernel's 815
    // VALU instructions per edge and level, in a kernel that tools/corr_variants.sh shows to be bound by exactly those (with every
    // window load served cache-hot it still takes 187 of its 270 us).  Same products, same order of additions: same bits.
    if (lane < 63) {
      const int p = lane % 9, bx = lane / 9;
      const float ddx = meta_f[p], ddy = meta_f[16 + p];
      const float w00 = (1.f - ddx) * (1.f - ddy), w01 = ddx * (1.f - ddy), w10 = (1.f - ddx) * ddy, w11 = ddx * ddy;
      const float* rp = raw + p * np + meta_i[16 + p] * bw + meta_i[p] + bx;
      _Float16* op = orow + 2 * (bx * 63 + p) + level;          // q = (bx * 7 + ay) * 9 + p
      float a0 = rp[0], a1 = rp[1];
#pragma unroll
      for (int ay = 0; ay < 7; ++ay) {
        rp += bw;
        const float b0 = rp[0], b1 = rp[1];
        float o = w00 * a0;
        o += w01 * a1;
        o += w10 * b0;
        o += w11 * b1;
        op[18 * ay] = (_Float16)o;
        a0 = b0; a1 = b1;
      }
    }
    __syncthreads();
  } else {
    // scattered windows (extreme scale change / far out of bounds): one 8x8 box per patch pixel
#pragma unroll 1
    for (int p = 0; p < CORR_NPIX; ++p) {
      const int pfx = __shfl(fx, p), pfy = __shfl(fy, p);
      const int x0 = pfx - CORR_R, y0 = pfy - CORR_R;
      const bool any_in = (x0 + CORR_D > 0) && (x0 < W) && (y0 + CORR_D > 0) && (y0 < H);
      if (any_in) corr_bbox_mfma(a, rsrc, H, W, x0, y0, CORR_D, CORR_D, raw, lane);
      __syncthreads();
      if (lane < 49) {                       // 49 outputs of pixel p: lane = bx*7 + ay
        const int bx = lane / 7, ay = lane - bx * 7;
        const float ddx = meta_f[p], ddy = meta_f[16 + p];
        float c00 = 0.f, c01 = 0.f, c10 = 0.f, c11 = 0.f;
        if (any_in) {
          const float* rp = raw + p * 64 + ay * CORR_D + bx;
          c00 = rp[0]; c01 = rp[1]; c10 = rp[CORR_D]; c11 = rp[CORR_D + 1];
        }
        float o = (1.f - ddx) * (1.f - ddy) * c00;
        o += ddx * (1.f - ddy) * c01;
        o += (1.f - ddx) * ddy * c10;
        o += ddx * ddy * c11;
        orow[2 * ((bx * 7 + ay) * 9 + p) + level] = (_Float16)o;
      }
      __syncthreads();
    }
  }
}

