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
//      operator's first kernel could save on THIS side of the fusion).
#include "../../dpvo_amd/csrc/corr_dev.h"

template <int VARIANT>
__global__ __launch_bounds__(64, 3) void corr_pyramid_variant_kernel(
    const _Float16* __restrict__ gmap, const _Float16* __restrict__ fmap0, const _Float16* __restrict__ fmap1,
    const float* __restrict__ coords, const int64_t* __restrict__ us, const int64_t* __restrict__ vs, _Float16* __restrict__ out,
    int64_t ld_out, int64_t E, int H0, int W0, int H1, int W1, int N1, int N2) {
  __shared__ __attribute__((aligned(16))) float raw[CORR_NPIX * CORR_MAXPOS];
  __shared__ __attribute__((aligned(16))) _Float16 orow[2 * CORR_NOUT + 2];
  __shared__ int meta_i[32];
  __shared__ float meta_f[32];
  const int lane = threadIdx.x;
  for (int64_t e = blockIdx.x; e < E; e += gridDim.x) {
    int64_t u, v;
    if constexpr (VARIANT == 6 || VARIANT == 7) { u = 0; v = (int)(e / 1326) % N2; }      // (1 326 = E / 36: same frames, no dependent loads)
    else { u = (int)us[e] % N1; v = (int)vs[e] % N2; }
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
        corr_level(a, fmap0, H0, W0, 20.f + (cx - floorf(cx)), 20.f + (cy - floorf(cy)), raw, meta_i, meta_f, lane, orow, 0);
      else
        corr_level(a, fmap0 + (int64_t)v * H0 * W0 * CORR_C, H0, W0, cx, cy, raw, meta_i, meta_f, lane, orow, 0);
    }
    if constexpr (VARIANT != 1) {
      if constexpr (VARIANT == 4 || VARIANT == 5)
        corr_level(a, fmap1, H1, W1, 10.f + (cx * 0.25f - floorf(cx * 0.25f)), 10.f + (cy * 0.25f - floorf(cy * 0.25f)), raw, meta_i, meta_f,
                   lane, orow, 1);
      else
        corr_level(a, fmap1 + (int64_t)v * H1 * W1 * CORR_C, H1, W1, cx * 0.25f, cy * 0.25f, raw, meta_i, meta_f, lane, orow, 1);
    }
    const uint32_t* src = reinterpret_cast<const uint32_t*>(orow);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + e * ld_out);
    if constexpr (VARIANT == 7) {
      if (src[lane] == 0x7fff7fffu) dst[lane] = 1;       // (keeps the blend alive without the row store)
    } else {
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const int q = lane + 64 * s;
        if (q < CORR_NOUT) dst[q] = src[q];
      }
      for (int64_t c = 2 * CORR_NOUT + lane; c < ld_out; c += 64) out[e * ld_out + c] = (_Float16)0;
    }
    __syncthreads();
  }
}

// same arguments as dpvo_corr_pyramid_forward (include/dpvo_hip.h) without the order hint, plus the variant
extern "C" int dpvo_corr_pyramid_variant(const void* gmap, const void* fmap0, const void* fmap1, const float* coords, const int64_t* us,
                                         const int64_t* vs, void* out, int64_t ld_out, int64_t E, int64_t N1, int64_t N2, int H0, int W0,
                                         int H1, int W1, int variant, void* stream) {
  if (E <= 0 || !gmap || !fmap0 || !fmap1 || !coords || !us || !vs || !out) return DPVO_E_INVALID;
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
    default: return DPVO_E_UNSUPPORTED;
  }
#undef CV
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
