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

template <int OCC>
__global__ __launch_bounds__(64, OCC) void corr_pyramid_kernel(
    const _Float16* __restrict__ gmap, const _Float16* __restrict__ fmap0, const _Float16* __restrict__ fmap1,
    const float* __restrict__ coords, const int64_t* __restrict__ us, const int64_t* __restrict__ vs,
    const int32_t* __restrict__ order, _Float16* __restrict__ out, int64_t ld_out, int64_t E, int H0, int W0,
    int H1, int W1, int N1, int N2) {
  __shared__ __attribute__((aligned(16))) float raw[CORR_NPIX * CORR_MAXPOS];
  __shared__ __attribute__((aligned(16))) _Float16 orow[2 * CORR_NOUT + 2];
  __shared__ int meta_i[32];
  __shared__ float meta_f[32];
  const int lane = threadIdx.x;
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
    const int64_t u = (int)us[e] % N1, v = (int)vs[e] % N2;
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
    // CORR_VARIANT (measurement builds only, tools/corr_variants.sh; results are wrong on purpose): 1 = level 0 only, 2 = level 1
    // only, 4 = level 1 reads one fixed cache-hot window (what a level-1 tile shared through LDS could save at best), 5 = both do
#if !defined(CORR_VARIANT) || CORR_VARIANT != 2
  #if defined(CORR_VARIANT) && CORR_VARIANT == 5
    corr_level(a, fmap0, H0, W0, 20.f + (cx - floorf(cx)), 20.f + (cy - floorf(cy)), raw, meta_i, meta_f, lane, orow, 0);
  #else
    corr_level(a, fmap0 + (int64_t)v * H0 * W0 * CORR_C, H0, W0, cx, cy, raw, meta_i, meta_f, lane, orow, 0);
  #endif
#endif
    CORR_T(3);
#if !defined(CORR_VARIANT) || CORR_VARIANT != 1
  #if defined(CORR_VARIANT) && (CORR_VARIANT == 4 || CORR_VARIANT == 5)
    corr_level(a, fmap1, H1, W1, 10.f + (cx * 0.25f - floorf(cx * 0.25f)), 10.f + (cy * 0.25f - floorf(cy * 0.25f)), raw, meta_i, meta_f, lane, orow, 1);
  #else
    corr_level(a, fmap1 + (int64_t)v * H1 * W1 * CORR_C, H1, W1, cx * 0.25f, cy * 0.25f, raw, meta_i, meta_f, lane, orow, 1);
  #endif
#endif
    CORR_T(5);
    // coalesced row store: 441 packed (level0, level1) words
    const uint32_t* src = reinterpret_cast<const uint32_t*>(orow);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + e * ld_out);
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int q = lane + 64 * s;
      if (q < CORR_NOUT) dst[q] = src[q];
    }
    // zero the padding columns [882, ld_out)
    for (int64_t c = 2 * CORR_NOUT + lane; c < ld_out; c += 64) out[e * ld_out + c] = (_Float16)0;
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
  hipLaunchKernelGGL(corr_pyramid_kernel<3>, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream,
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
