// ba_global.hip -- block-sparse global bundle adjustment for gfx950 (reference: eff_impl=True).
//
// Replaces the EfficentE machinery of the reference (dpvo/fastba/block_e.cu:43-300: host-built index tables with
// unordered_set / Eigen lookups per call, EEt_kernel / Ev_kernel / Etv_kernel with 36 float atomics per (i,j1,j2,patch)
// tuple) and the eff_impl branch of cuda_ba (ba_cuda.cu:475-478,538-550).  Used by DPVO.__run_global_BA
// (dpvo/dpvo.py:312-326, LOOP_CLOSURE=True) where the free poses number in the hundreds.
//
// Same algebra as the dense path, with E stored exactly as the reference's E_lookup: one [M x 6] block per frame pair
// (i,j) (patch slot = kk % M) plus one "self" block per source frame (the -w Jz Ji side, summed over all edges of the
// patch).  Everything is indexed by the device-built graph plan (pairs sorted by (i,j), so the pairs of one source
// frame are a contiguous run) -- no host tables.
//   gba_scatter_kernel   edge records -> Ecol[pair][slot][6]                 (plain stores; atomics only fold duplicates)
//   gba_patch_kernel     per patch (CSR): C, u, Ei -> Q, u, Eself[frame][slot][6]
//   gba_assemble_kernel  per pair: B blocks and v into the dense S / y       (float atomics, like the reference)
//   gba_schur_kernel     per (source frame, block a, block b): S -= sum_slot Q ea eb^T, y -= sum_slot Q u ea; the sum
//                        over the frame's M patch slots is reduced IN the wave, then 36 atomics per block pair
//                        (the reference issues 36 atomics per block pair PER PATCH)
//   dpvo_gba_solve       S += I*(1e-4*S+1); blocked Cholesky + both substitutions on the device (chol.hip)
//   gba_retr_kernel      dZ = Q (u - e^T dX), depth + pose retraction
#include "ba_common.h"

namespace {
using namespace ba;

struct GbaWs {
  size_t pairbuf, edgebuf, Q, u, Ecol, Eself, total;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline void gba_layout(int64_t E, int64_t n_pairs, int64_t n_frames, int M, GbaWs* L) {
  const size_t n = (size_t)(E > 0 ? E : 1), g = (size_t)(n_pairs > 0 ? n_pairs : 1), f = (size_t)(n_frames > 0 ? n_frames : 1);
  size_t o = 0;
  L->pairbuf = o; o += al(g * kPairStride * 4);
  L->edgebuf = o; o += al(n * kEdgeStride * 4);
  L->Q = o; o += al(f * M * 4);
  L->u = o; o += al(f * M * 4);
  L->Ecol = o; o += al(g * M * 6 * 4);
  L->Eself = o; o += al(f * M * 6 * 4);
  L->total = o;
}

// Ecol[pu[e]][kk[e] % M][0..5] += Ej_e
__global__ void gba_scatter_kernel(const int64_t* __restrict__ kk, const int32_t* __restrict__ pu,
                                   const float* __restrict__ edgebuf, float* __restrict__ Ecol, int64_t E, int M) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(kk[e] % M);
    float* dst = Ecol + ((int64_t)pu[e] * M + slot) * 6;
    const float* eb = edgebuf + e * kEdgeStride + 8;
#pragma unroll
    for (int a = 0; a < 6; ++a) atomicAdd(dst + a, eb[a]);
  }
}

// per patch k (index into kx): Q, u and the i-side block, stored by (frame - f0, slot)
__global__ void gba_patch_kernel(const int32_t* __restrict__ perm_k, const int32_t* __restrict__ patch_off,
                                 const int32_t* __restrict__ kx, const int32_t* __restrict__ n_patches,
                                 const float* __restrict__ edgebuf, float lmbda, int M, int f0, int n_frames,
                                 float* __restrict__ Q, float* __restrict__ U, float* __restrict__ Eself) {
  const int np = *n_patches;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < np; k += gridDim.x * blockDim.x) {
    float C = 0.f, u = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0};
    for (int p = patch_off[k]; p < patch_off[k + 1]; ++p) {
      const float* eb = edgebuf + (int64_t)perm_k[p] * kEdgeStride;
      C += eb[0]; u += eb[1];
#pragma unroll
      for (int a = 0; a < 6; ++a) Ei[a] += eb[2 + a];
    }
    const int patch = kx[k];
    const int fr = patch / M - f0, slot = patch % M;
    if (fr < 0 || fr >= n_frames) continue;
    const int64_t o = (int64_t)fr * M + slot;
    Q[o] = 1.0f / (C + lmbda);
    U[o] = u;
#pragma unroll
    for (int a = 0; a < 6; ++a) Eself[o * 6 + a] = Ei[a];
  }
}

// B and v from the pair Gram blocks (ba_cuda.cu:335-349,363-368); one wave per pair, lanes over the 36 entries
__global__ __launch_bounds__(64) void gba_assemble_kernel(const int32_t* __restrict__ pair_ij,
                                                          const int32_t* __restrict__ n_pairs,
                                                          const float* __restrict__ pairbuf, int t0, int N,
                                                          float* __restrict__ S, float* __restrict__ y) {
  const int ng = *n_pairs;
  const int lane = threadIdx.x;
  const int64_t n6 = 6 * (int64_t)N;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int ix = pair_ij[2 * g] - t0, jx = pair_ij[2 * g + 1] - t0;
    const bool fi = ix >= 0 && ix < N, fj = jx >= 0 && jx < N;
    const float* pb = pairbuf + (int64_t)g * kPairStride;
    if (lane < 36) {
      const int a = lane / 6, b = lane - 6 * a;
      if (fi) atomicAdd(&S[(6 * ix + a) * n6 + 6 * ix + b], pb[a * 16 + b]);
      if (fj) atomicAdd(&S[(6 * jx + a) * n6 + 6 * jx + b], pb[(6 + a) * 16 + 6 + b]);
      if (fi && fj) {
        atomicAdd(&S[(6 * ix + a) * n6 + 6 * jx + b], -pb[a * 16 + 6 + b]);
        atomicAdd(&S[(6 * jx + b) * n6 + 6 * ix + a], -pb[a * 16 + 6 + b]);
      }
    } else if (lane < 42) {
      const int a = lane - 36;
      if (fi) atomicAdd(&y[6 * ix + a], -pb[a * 16 + 12]);
      if (fj) atomicAdd(&y[6 * jx + a], pb[(6 + a) * 16 + 12]);
    }
  }
}

__device__ __forceinline__ int lower_bound_i(const int32_t* pair_ij, int ng, int f) {
  int lo = 0, hi = ng;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (pair_ij[2 * mid] < f) lo = mid + 1; else hi = mid; }
  return lo;
}

// Schur complement.  grid.x = source frame (f0 + blockIdx.x), grid.y strides over block a; 4 waves stride over block b.
// Block index b in [0, P]: b < P -> pair ga+b (pose slot = j of that pair), b == P -> the self block (pose slot = i).
__global__ __launch_bounds__(256) void gba_schur_kernel(const int32_t* __restrict__ pair_ij,
                                                        const int32_t* __restrict__ n_pairs,
                                                        const float* __restrict__ Q, const float* __restrict__ U,
                                                        const float* __restrict__ Ecol, const float* __restrict__ Eself,
                                                        int M, int f0, int t0, int N, float* __restrict__ S,
                                                        float* __restrict__ y) {
  const int ng = *n_pairs;
  const int fr = blockIdx.x, f = f0 + fr;
  const int ga = lower_bound_i(pair_ij, ng, f), gb = lower_bound_i(pair_ij, ng, f + 1);
  const int P = gb - ga;
  if (P == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n6 = 6 * (int64_t)N;
  const float* Qf = Q + (int64_t)fr * M;
  const float* Uf = U + (int64_t)fr * M;
  auto blk_ptr = [&](int b) { return b < P ? Ecol + (int64_t)(ga + b) * M * 6 : Eself + (int64_t)fr * M * 6; };
  auto blk_pose = [&](int b) { return (b < P ? pair_ij[2 * (ga + b) + 1] : f) - t0; };
  for (int a = blockIdx.y; a <= P; a += gridDim.y) {
    const int pa = blk_pose(a);
    if (pa < 0 || pa >= N) continue;
    const float* Ea = blk_ptr(a);
    for (int b = a + wave; b <= P + 1; b += 4) {
      // b == P + 1: the right-hand side  y[pa] -= sum_slot Q u ea
      float acc[36];
#pragma unroll
      for (int q = 0; q < 36; ++q) acc[q] = 0.f;
      if (b <= P) {
        const int pb_ = blk_pose(b);
        if (pb_ < 0 || pb_ >= N) continue;
        const float* Eb = blk_ptr(b);
        for (int s = lane; s < M; s += 64) {
          const float q = Qf[s];
          float ea[6], eb[6];
#pragma unroll
          for (int r = 0; r < 6; ++r) { ea[r] = Ea[s * 6 + r]; eb[r] = Eb[s * 6 + r]; }
#pragma unroll
          for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += q * ea[r] * eb[c];
        }
#pragma unroll
        for (int q = 0; q < 36; ++q) acc[q] = wave_sum(acc[q]);
        if (lane < 36) {
          const int r = lane / 6, c = lane - 6 * r;
          float v = 0.f;
#pragma unroll
          for (int q = 0; q < 36; ++q) v = (q == lane) ? acc[q] : v;
          atomicAdd(&S[(6 * pa + r) * n6 + 6 * pb_ + c], -v);
          if (b != a) atomicAdd(&S[(6 * pb_ + c) * n6 + 6 * pa + r], -v);
        }
      } else {
        for (int s = lane; s < M; s += 64) {
          const float qu = Qf[s] * Uf[s];
#pragma unroll
          for (int r = 0; r < 6; ++r) acc[r] += qu * Ea[s * 6 + r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) acc[r] = wave_sum(acc[r]);
        if (lane < 6) {
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < 6; ++r) v = (r == lane) ? acc[r] : v;
          atomicAdd(&y[6 * pa + lane], -v);
        }
      }
    }
  }
}

// dZ = Q (u - sum_blocks e_block . dX[pose(block)]) per (frame, slot) that owns edges; then retractions
__global__ __launch_bounds__(128) void gba_retr_kernel(float* __restrict__ poses, float* __restrict__ patches,
                                                       const int32_t* __restrict__ pair_ij,
                                                       const int32_t* __restrict__ n_pairs, const int32_t* __restrict__ kx,
                                                       const int32_t* __restrict__ n_patches, const float* __restrict__ Q,
                                                       const float* __restrict__ U, const float* __restrict__ Ecol,
                                                       const float* __restrict__ Eself, const float* __restrict__ dX,
                                                       int M, int f0, int n_frames, int t0, int N, int P) {
  const int ng = *n_pairs, np = *n_patches;
  const int PP = P * P;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = gt; k < np; k += gridDim.x * blockDim.x) {
    const int patch = kx[k];
    const int f = patch / M, fr = f - f0, slot = patch % M;
    if (fr < 0 || fr >= n_frames) continue;
    const int ga = lower_bound_i(pair_ij, ng, f), gb = lower_bound_i(pair_ij, ng, f + 1);
    float s = 0.f;
    const int ix = f - t0;
    if (ix >= 0 && ix < N) {
      const float* e = Eself + ((int64_t)fr * M + slot) * 6;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += e[r] * dX[6 * ix + r];
    }
    for (int g = ga; g < gb; ++g) {
      const int jx = pair_ij[2 * g + 1] - t0;
      if (jx < 0 || jx >= N) continue;
      const float* e = Ecol + ((int64_t)g * M + slot) * 6;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += e[r] * dX[6 * jx + r];
    }
    const int64_t o = (int64_t)fr * M + slot;
    const float dZ = Q[o] * (U[o] - s);
    float* pk = patches + (int64_t)patch * 3 * PP + 2 * PP;
    float d = pk[0] + dZ;
    d = (d > 20.0f) ? 1.0f : d;
    d = fmaxf(d, 1e-4f);
    for (int a = 0; a < PP; ++a) pk[a] = d;
  }
  for (int i = gt; i < N; i += gridDim.x * blockDim.x) {
    float* p = poses + 7 * (int64_t)(t0 + i);
    const float tt[3] = {p[0], p[1], p[2]}, qq[4] = {p[3], p[4], p[5], p[6]};
    const float xi[6] = {dX[6 * i], dX[6 * i + 1], dX[6 * i + 2], dX[6 * i + 3], dX[6 * i + 4], dX[6 * i + 5]};
    float t1v[3], q1v[4];
    retrSE3(xi, tt, qq, t1v, q1v);
    p[0] = t1v[0]; p[1] = t1v[1]; p[2] = t1v[2]; p[3] = q1v[0]; p[4] = q1v[1]; p[5] = q1v[2]; p[6] = q1v[3];
  }
}

}  // namespace

extern "C" size_t dpvo_gba_workspace_bytes(int64_t E, int64_t n_pairs, int64_t n_frames, int M) {
  if (E < 0 || n_pairs < 0 || n_frames < 0 || M <= 0) return 0;
  GbaWs L;
  gba_layout(E, n_pairs, n_frames, M, &L);
  return L.total;
}

// Linearise: fills S [6N,6N] (B - E Q E^T, WITHOUT the damping) and y [6N] (v - E Q u); both must be zeroed by the
// caller.  Frames f0 .. f0+n_frames-1 are the source frames that own patches (n_frames*M Q/u slots).
extern "C" int dpvo_gba_linearize(const float* poses, const float* patches, const float* intrinsics, const float* target,
                                  const float* weight, float lmbda, const int64_t* ii, const int64_t* jj,
                                  const int64_t* kk, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                                  int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, float* S, float* y,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (E <= 0 || P <= 0 || M <= 0 || t1 <= t0 || n_frames <= 0 || n_pairs_h <= 0 || n_patches_h <= 0) return DPVO_E_INVALID;
  if (!poses || !patches || !intrinsics || !target || !weight || !ii || !jj || !kk || !plan || !S || !y || !ws) return DPVO_E_INVALID;
  GbaWs L;
  gba_layout(E, n_pairs_h, n_frames, M, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* w = (char*)ws;
  float* pairbuf = (float*)(w + L.pairbuf);
  float* edgebuf = (float*)(w + L.edgebuf);
  float* Q = (float*)(w + L.Q);
  float* U = (float*)(w + L.u);
  float* Ecol = (float*)(w + L.Ecol);
  float* Eself = (float*)(w + L.Eself);
  hipStream_t st = (hipStream_t)stream;
  const int N = t1 - t0;
  const int32_t* n_patches = plan + PL.counts + 0;
  const int32_t* n_pairs = plan + PL.counts + 1;
  hipError_t e = hipMemsetAsync(w + L.Q, 0, L.total - L.Q, st);     // Q, u, Ecol, Eself
  if (e != hipSuccess) return (int)e;
  const unsigned pair_grid = (unsigned)(n_pairs_h < 65535 ? n_pairs_h : 65535);
  hipLaunchKernelGGL(ba_pair_kernel, dim3(pair_grid), dim3(128), 0, st, poses, patches, intrinsics, target, weight, kk,
                     plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, n_pairs, pairbuf, edgebuf, P);
  hipLaunchKernelGGL(gba_scatter_kernel, dim3((unsigned)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096)), dim3(256), 0,
                     st, kk, plan + PL.pu, edgebuf, Ecol, E, M);
  hipLaunchKernelGGL(gba_patch_kernel, dim3((unsigned)((n_patches_h + 255) / 256)), dim3(256), 0, st, plan + PL.perm_k,
                     plan + PL.patch_off, plan + PL.kx, n_patches, edgebuf, lmbda, M, f0, n_frames, Q, U, Eself);
  hipLaunchKernelGGL(gba_assemble_kernel, dim3(pair_grid), dim3(64), 0, st, plan + PL.pair_ij, n_pairs, pairbuf, t0, N,
                     S, y);
  hipLaunchKernelGGL(gba_schur_kernel, dim3((unsigned)n_frames, 8), dim3(256), 0, st, plan + PL.pair_ij, n_pairs, Q, U,
                     Ecol, Eself, M, f0, t0, N, S, y);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_gba_retract(float* poses, float* patches, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                                int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, const float* dX, void* ws,
                                size_t ws_bytes, void* stream) {
  if (E <= 0 || P <= 0 || M <= 0 || t1 <= t0 || n_frames <= 0 || n_pairs_h <= 0 || n_patches_h <= 0) return DPVO_E_INVALID;
  if (!poses || !patches || !plan || !dX || !ws) return DPVO_E_INVALID;
  GbaWs L;
  gba_layout(E, n_pairs_h, n_frames, M, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* w = (char*)ws;
  const int N = t1 - t0;
  const int64_t work = n_patches_h > N ? n_patches_h : N;
  hipLaunchKernelGGL(gba_retr_kernel, dim3((unsigned)((work + 127) / 128)), dim3(128), 0, (hipStream_t)stream, poses,
                     patches, plan + PL.pair_ij, plan + PL.counts + 1, plan + PL.kx, plan + PL.counts + 0,
                     (const float*)(w + L.Q), (const float*)(w + L.u), (const float*)(w + L.Ecol),
                     (const float*)(w + L.Eself), dX, M, f0, n_frames, t0, N, P);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
