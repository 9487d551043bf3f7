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
#define CORR_RS 12             // floats per POSITION of the raw volume: patch pixels 0..8, then three padding rows (written, never read)
#define CORR_RAWPOS 192         // 12 MFMA tiles of 16 positions: the tile loop runs in batches of CORR_U = 4 and stores every tile it runs
#define CORR_ROW_BYTES 1792    // one output row incl. the zero padding columns 882..895 when ld_out == 896

// The workgroup's LDS (one wave = one edge).  raw[pos][m]: dot(template m, feature at bounding-box position pos), position-major so
// that a lane's four accumulator rows of one MFMA tile are ONE ds_write_b128 (rounds 1-5: raw[m][pos], four exec-masked ds_write_b32 per
// tile -- the store predicates alone were ~250 of the kernel's ~760 scalar instructions per edge and level: round 6, §3.1).
struct CorrShared {
  float raw[CORR_RAWPOS * CORR_RS];                          // 9 216 B
  __attribute__((aligned(16))) _Float16 orow[CORR_ROW_BYTES / 2];      // the [882] row, both levels interleaved, + 14 zero columns
  __attribute__((aligned(16))) f4 meta[16];                  // per patch pixel: {dx, dy, bit pattern of its window's first position, 0}
};

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

// Bounding box of the nine patch pixels (lanes 0..8; the other lanes of row 0 carry the identities): min / max over lanes 0..15 of
// row 0, result in lane 15 -- sixteen v_{min,max}_i32 with a DPP row shift each (a lane without a source is not written: it keeps its
// own value) instead of sixteen ds_bpermute round trips with their index arithmetic.  Hand-placed: the compiler expands
// __builtin_amdgcn_update_dpp into mov + mov_dpp + min with s_nops between dependent steps; here the four independent chains are
// interleaved, so every DPP read is three instructions behind the write it depends on (two wait states required).
__device__ __forceinline__ void bbox_reduce(int& mnx, int& mxx, int& mny, int& mxy) {
  asm volatile(
      "s_nop 1\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %3, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_min_i32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_i32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(mnx), "+v"(mxx), "+v"(mny), "+v"(mxy));
}

// Which position and which bytes a lane fetches for the MFMA B fragment of a tile.  MODE 0 is the product: lane (n = lane & 15, kg = lane >> 4)
// fetches the 16-byte channel chunk 4 s + kg of position 16 tile + n -- what v_mfma_f32_16x16x32_f16 wants in that lane.  The measurement
// kernel (tools/probes/corr_variant.hip) specialises other modes -- other lane -> address maps, wrong results on purpose -- to time what the
// vector L1 makes of them; the product library instantiates MODE 0 only.
template <int MODE> struct CorrLoad {
  static constexpr bool ALIGN4 = false;          // bounding-box columns aligned to groups of four
  static constexpr int STEP = 64;                // bytes between a lane's four loads of a tile
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + n; }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) {
    return (unsigned)(__mul24(y, W) + x) * (CORR_C * 2) + kg * 16;       // (24-bit multiply: the caller's `any_in` keeps |y| < 2^11, W < 2^15)
  }
};

// raw[pos][m] for pos < np = bw*bh (and the padding positions of the last tile), m < 12: dot(template m, feature(y0+pos/bw, x0+pos%bw)),
// 0 if out of the image.  Features are fetched with bounds-checked buffer loads: a lane whose position is outside the image or is tile
// padding gets the offset 0x8000'0000 (beyond num_records, and far from the 2^32 wrap of offset + instruction offset + 16) and reads
// zeros without a branch.  CORR_U tiles (= 16 x 16 B per lane) are put in flight before the first MFMA consumes them.
#define CORR_U 4
template <int MODE = 0>
__device__ __forceinline__ void corr_bbox_mfma(const h8 (&a)[4], __amdgpu_buffer_rsrc_t rsrc, int H, int W, int x0, int y0, int bw, int np,
                                               bool any_in, float* __restrict__ raw, int lane) {
  const int n = lane & 15, kg = lane >> 4;
  const int ntiles = (np + 15) >> 4;
  float* sp = raw + n * CORR_RS + 4 * kg;      // this lane's four accumulator rows m = 4 kg .. 4 kg + 3 of tile 0
  if (!any_in) {                               // (uniform) the whole box is outside the image: zeros, no loads
    if (kg < 3)
      for (int t = 0; t < ntiles; ++t) *reinterpret_cast<f4*>(sp + t * 16 * CORR_RS) = (f4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  const float inv_bw = 1.0f / (float)bw;       // pos / bw via float: exact for pos < 2^12 (distance to an integer >= 0.5/bw)
  const float half_inv = 0.5f * inv_bw;
  for (int t0 = 0; t0 < ntiles; t0 += CORR_U) {
    u4 b[CORR_U][4];
#pragma unroll
    for (int u = 0; u < CORR_U; ++u) {
      const int pos = CorrLoad<MODE>::pos(t0 + u, n, lane);
      const int py = (int)fmaf((float)pos, inv_bw, half_inv), px = pos - __mul24(py, bw);
      const int y = y0 + py, x = x0 + px;
      const bool ok = (pos < np) & ((unsigned)x < (unsigned)W) & ((unsigned)y < (unsigned)H);
      // (the address is computed for every lane and the invalid ones are redirected by mask arithmetic: as `ok ? address : 0x80000000`
      //  the compiler computes the address under an exec mask -- two scalar instructions and a branch shadow per tile)
      const unsigned addr = CorrLoad<MODE>::addr(y, x, W, kg, lane);
      const unsigned keep = ok ? 0xffffffffu : 0u;
      const unsigned voff = (addr & keep) | (~keep & 0x80000000u);
#pragma unroll
      for (int s = 0; s < 4; ++s) b[u][s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + s * CorrLoad<MODE>::STEP, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < CORR_U; ++u) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], as_h8(b[u][s]), acc, 0, 0, 0);
      // D[row = 4*(lane>>4)+r][col = lane&15]; rows are the patch pixels (0..8 real, 9..11 padding; 12..15 not stored)
      if (kg < 3) *reinterpret_cast<f4*>(sp + (t0 + u) * 16 * CORR_RS) = acc;
    }
  }
}

// One level: blends this level's 441 outputs from the raw volume into the LDS row image
// orow[q*2 + level] (f16), q = ((x*7+y)*3+i0)*3+j0.
template <int MODE = 0>
__device__ __forceinline__ void corr_level(const h8 (&a)[4], const _Float16* fmap, int H, int W, float cx, float cy, CorrShared& sm,
                                           int lane, int level) {
  float* __restrict__ raw = sm.raw;
  _Float16* __restrict__ orow = sm.orow;
  // lanes 0..8 own one patch pixel each
  const int fx = safe_floor_int(cx), fy = safe_floor_int(cy);
  const float dx = cx - floorf(cx), dy = cy - floorf(cy);
  const bool pl = lane < CORR_NPIX;
  int rnx = pl ? fx : INT_MAX, rxx = pl ? fx : INT_MIN, rny = pl ? fy : INT_MAX, rxy = pl ? fy : INT_MIN;
  bbox_reduce(rnx, rxx, rny, rxy);
  const int mnx = __builtin_amdgcn_readlane(rnx, 15), mxx = __builtin_amdgcn_readlane(rxx, 15);
  const int mny = __builtin_amdgcn_readlane(rny, 15), mxy = __builtin_amdgcn_readlane(rxy, 15);
  constexpr bool A4 = CorrLoad<MODE>::ALIGN4;       // (measurement modes only: columns aligned to groups of four, the width rounded up)
  const int x0 = (mnx - CORR_R) & (A4 ? ~3 : ~0), y0 = mny - CORR_R;
  const int bw = A4 ? ((mxx - CORR_R + CORR_D + 3) & ~3) - x0 : mxx - mnx + CORR_D, bh = mxy - mny + CORR_D;      // (|coordinates| <= 1e6: no overflow)
  const bool single = (bw <= CORR_MAXPOS) & (bh <= CORR_MAXPOS) && (bw * bh <= (A4 ? CORR_RAWPOS : CORR_MAXPOS));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)fmap, (short)0, H * W * CORR_C * 2, 0x00020000);

  if (single) {
    const int np = bw * bh;
    if (pl) sm.meta[lane] = (f4){dx, dy, __int_as_float((fy - mny) * bw + (fx - CORR_R - x0)), 0.f};
    const bool any_in = (x0 + bw > 0) & (x0 < W) & (y0 + bh > 0) & (y0 < H);
    corr_bbox_mfma<MODE>(a, rsrc, H, W, x0, y0, bw, np, any_in, raw, lane);
    __syncthreads();
    CORR_T(2 + 2 * level);
    // Blend (correlation_kernel.cu:221-230): lane = (patch pixel p = lane % 9, window column bx = lane / 9) walks the 7 window
    // rows ay of its column.  Its four weights are computed once, every raw row is read once (the lower pair of row ay is the
    // upper pair of row ay + 1), no index arithmetic is left inside the loop.  Same products, same order of additions as rounds 1-5.
    if (lane < 63) {
      const int p = lane % 9, bx = lane / 9;
      const f4 m = sm.meta[p];
      const float ddx = m[0], ddy = m[1];
      const float w00 = (1.f - ddx) * (1.f - ddy), w01 = ddx * (1.f - ddy), w10 = (1.f - ddx) * ddy, w11 = ddx * ddy;
      const float* rp = raw + (__float_as_int(m[2]) + bx) * CORR_RS + p;
      const int rstep = bw * CORR_RS;
      _Float16* op = orow + 2 * (bx * 63 + p) + level;          // q = (bx * 7 + ay) * 9 + p
      float a0 = rp[0], a1 = rp[CORR_RS];
#pragma unroll
      for (int ay = 0; ay < 7; ++ay) {
        rp += rstep;
        const float b0 = rp[0], b1 = rp[CORR_RS];
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
    if (pl) sm.meta[lane] = (f4){dx, dy, 0.f, 0.f};
#pragma unroll 1
    for (int p = 0; p < CORR_NPIX; ++p) {
      const int pfx = __builtin_amdgcn_readlane(fx, p), pfy = __builtin_amdgcn_readlane(fy, p);
      const int x0 = pfx - CORR_R, y0 = pfy - CORR_R;
      const bool any_in = (x0 + CORR_D > 0) & (x0 < W) & (y0 + CORR_D > 0) & (y0 < H);
      corr_bbox_mfma<MODE>(a, rsrc, H, W, x0, y0, CORR_D, CORR_D * CORR_D, any_in, raw, lane);
      __syncthreads();
      if (lane < 49) {                       // 49 outputs of pixel p: lane = bx*7 + ay
        const int bx = lane / 7, ay = lane - bx * 7;
        const f4 m = sm.meta[p];
        const float ddx = m[0], ddy = m[1];
        const float* rp = raw + (ay * CORR_D + bx) * CORR_RS + p;
        const float c00 = rp[0], c01 = rp[CORR_RS], c10 = rp[CORR_D * CORR_RS], c11 = rp[(CORR_D + 1) * CORR_RS];
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

