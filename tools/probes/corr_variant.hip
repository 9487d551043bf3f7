// tools/probes/corr_variant.hip -- MEASUREMENT kernel: corr_pyramid_kernel (dpvo_amd/csrc/corr.hip) with parts switched off or made
// cache-hot, to bound what a different organisation could gain (profiles/r04_corr_variants.txt).  Results are WRONG ON
// PURPOSE for every variant but 0.  Its own translation unit since round 5 (the branches used to live in the product's
// corr.hip): tools/corr_variants.sh links it, beside the product objects, into dpvo_amd/libdpvo_hip_corrvar.so, and
// tools/corr_bench.py calls dpvo_corr_pyramid_variant through ctypes when CORR_VARIANT is set.  Never part of libdpvo_hip.so.
//   variant 0: the product's arithmetic (must reproduce the product kernel's checksum);  1: level 0 only;  2: level 1 only;
//   4: level 1 reads one fixed cache-hot window (the most a level-1 tile shared through LDS could deliver);  5: both levels do;
//   6: the per-edge FIXED work made free -- ring indices computed instead of loaded, the nine templates taken from patch 0 (cache-hot),
//      coordinates still loaded: the most that several edges of one patch per workgroup sharing the index / template round trips
//       could save;  7: as 6, and the output row is not written (the most a correlation fused into the update
//      operator's first kernel could save on THIS side of the fusion);
//   8 .. 12: other lane -> address maps of the window loads (the CorrLoad specialisations below): what the vector L1 makes of the same
//      bytes requested differently (round 6; profiles/r06_d_corr_diet_and_quad_probe.txt).
#include "../../dpvo_amd/csrc/corr_dev.h"

// ---- other lane -> address maps of the window loads (CorrLoad<MODE>, corr_dev.h), results wrong on purpose: what does the vector L1 make of them?
template <> struct CorrLoad<1> {       // variant 8: four consecutive lanes fetch 64 contiguous bytes of ONE position
  static constexpr bool ALIGN4 = false; static constexpr int STEP = 64;
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + (lane >> 2); }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) { return (unsigned)(__mul24(y, W) + x) * 256 + (lane & 3) * 16; }
};
template <> struct CorrLoad<2> {       // variant 10: sixteen consecutive lanes fetch the 256 contiguous bytes of ONE position
  static constexpr bool ALIGN4 = false; static constexpr int STEP = 1024;
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + (lane >> 4); }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) { return (unsigned)(__mul24(y, W) + x) * 256 + (lane & 15) * 16; }
};
template <> struct CorrLoad<3> {       // variant 9: the addressing of an x4-interleaved map [y][x / 4][chunk 16][x % 4][8 halves], boxes aligned to 4 columns
  static constexpr bool ALIGN4 = true; static constexpr int STEP = 256;
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + n; }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) { return (unsigned)(__mul24(y, W >> 2) + (x >> 2)) * 1024 + kg * 64 + (x & 3) * 16; }
};
template <> struct CorrLoad<4> {       // variant 11: the same with a group stride of 1 088 bytes (no power-of-two stride between the groups of a quarter wave)
  static constexpr bool ALIGN4 = true; static constexpr int STEP = 256;
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + n; }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) { return (unsigned)(__mul24(y, W >> 2) + (x >> 2)) * 1088 + kg * 64 + (x & 3) * 16; }
};
template <> struct CorrLoad<5> {       // variant 12: an x16-blocked map [y][x / 16][kg 4][(x / 4) % 4][s 4][x % 4][8 halves]: variant 8's address pattern made legal, boxes aligned to 4
  static constexpr bool ALIGN4 = true; static constexpr int STEP = 64;
  static __device__ __forceinline__ int pos(int tile, int n, int lane) { return tile * 16 + n; }
  static __device__ __forceinline__ unsigned addr(int y, int x, int W, int kg, int lane) { return (unsigned)(__mul24(y, W >> 4) + (x >> 4)) * 4096 + kg * 1024 + ((x >> 2) & 3) * 256 + (x & 3) * 16; }
};

template <int VARIANT>
__global__ __launch_bounds__(64, 3) void corr_pyramid_variant_kernel(
    const _Float16* __restrict__ gmap, const _Float16* __restrict__ fmap0, const _Float16* __restrict__ fmap1,
    const float* __restrict__ coords, const int64_t* __restrict__ us, const int64_t* __restrict__ vs, _Float16* __restrict__ out,
    int64_t ld_out, int64_t E, int H0, int W0, int H1, int W1, int N1, int N2) {
  __shared__ CorrShared sm;
  constexpr int Q = VARIANT == 8 ? 1 : VARIANT == 10 ? 2 : VARIANT == 9 ? 3 : VARIANT == 11 ? 4 : VARIANT == 12 ? 5 : 0;
  const int lane = threadIdx.x;
  if (lane < 14) sm.orow[2 * CORR_NOUT + lane] = (_Float16)0;
  for (int64_t e = blockIdx.x; e < E; e += gridDim.x) {
    int64_t u, v;
    if constexpr (VARIANT == 6 || VARIANT == 7) { u = 0; v = (int)(e / 1326) % N2; }      // (1 326 = E / 36: same frames, no dependent loads)
    else { u = (unsigned)us[e] % (unsigned)N1; v = (unsigned)vs[e] % (unsigned)N2; }
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
    if constexpr (VARIANT != 2) {
      if constexpr (VARIANT == 5)
        corr_level(a, fmap0, H0, W0, 20.f + (cx - floorf(cx)), 20.f + (cy - floorf(cy)), sm, lane, 0);
      else
        corr_level<Q>(a, fmap0 + (int64_t)v * H0 * W0 * CORR_C, H0, W0, cx, cy, sm, lane, 0);
    }
    if constexpr (VARIANT != 1) {
      if constexpr (VARIANT == 4 || VARIANT == 5)
        corr_level(a, fmap1, H1, W1, 10.f + (cx * 0.25f - floorf(cx * 0.25f)), 10.f + (cy * 0.25f - floorf(cy * 0.25f)), sm, lane, 1);
      else
        corr_level<Q>(a, fmap1 + (int64_t)v * H1 * W1 * CORR_C, H1, W1, cx * 0.25f, cy * 0.25f, sm, lane, 1);
    }
    if constexpr (VARIANT == 7) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(sm.orow);
      uint32_t* dst = reinterpret_cast<uint32_t*>(out + e * ld_out);
      if (src[lane] == 0x7fff7fffu) dst[lane] = 1;       // (keeps the blend alive without the row store)
    } else {                                             // (ld_out == 896, 16-byte aligned: the product's wide row store)
      const u4* src = reinterpret_cast<const u4*>(sm.orow);
      u4* dst = reinterpret_cast<u4*>(out + e * ld_out);
      dst[lane] = src[lane];
      if (lane < CORR_ROW_BYTES / 16 - 64) dst[64 + lane] = src[64 + lane];
    }
    __syncthreads();
  }
}

// same arguments as dpvo_corr_pyramid_forward (include/dpvo_hip.h) without the order hint, plus the variant
extern "C" int dpvo_corr_pyramid_variant(const void* gmap, const void* fmap0, const void* fmap1, const float* coords, const int64_t* us,
                                         const int64_t* vs, void* out, int64_t ld_out, int64_t E, int64_t N1, int64_t N2, int H0, int W0,
                                         int H1, int W1, int variant, void* stream) {
  if (E <= 0 || !gmap || !fmap0 || !fmap1 || !coords || !us || !vs || !out) return DPVO_E_INVALID;
  if (ld_out != CORR_ROW_BYTES / 2 || ((uintptr_t)out & 15)) return DPVO_E_UNSUPPORTED;
#define CV(V)                                                                                                                    \
  hipLaunchKernelGGL(corr_pyramid_variant_kernel<V>, dim3((unsigned)E), dim3(64), 0, (hipStream_t)stream, (const _Float16*)gmap,  \
                     (const _Float16*)fmap0, (const _Float16*)fmap1, coords, us, vs, (_Float16*)out, ld_out, E, H0, W0, H1, W1,  \
                     (int)N1, (int)N2)
  switch (variant) {
    case 0: CV(0); break;
    case 1: CV(1); break;
    case 2: CV(2); break;
    case 4: CV(4); break;
    case 5: CV(5); break;
    case 6: CV(6); break;
    case 7: CV(7); break;
    case 8: CV(8); break;
    case 9: CV(9); break;
    case 10: CV(10); break;
    case 11: CV(11); break;
    case 12: CV(12); break;
    default: return DPVO_E_UNSUPPORTED;
  }
#undef CV
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
