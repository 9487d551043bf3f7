// graph.hip -- device-resident patch-graph index structures ("plan") for gfx950.
//
// Replaces, without any host round trip:
//   * fastba.neighbors  (reference dpvo/fastba/ba.cpp:59-97: D2H copy, per-patch std::stable_sort by jj, H2D)
//   * torch::_unique(kk, sorted, inverse) of cuda_ba (dpvo/fastba/ba_cuda.cu:447-449)
//   * torch.unique(..., return_inverse=True) of SoftAgg for kk and ii*12345+jj (dpvo/blocks.py:41, net.py:87-88)
// by two stable LSD radix sorts (rocPRIM) of 64-bit composite keys (kk<<24|jj and ii<<24|jj; edge id as the
// value, so ties keep edge order exactly like stable_sort) followed by flag / scan / scatter kernels.
// Integer results are bit-exact with the oracle (tests/test_graph_*).
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace {

constexpr int kKeyShift = 24;   // jj (frame index) < 2^24; BUFFER_SIZE is 4096 in the reference (config.py:6)

__global__ void make_keys_kernel(const int64_t* __restrict__ hi, const int64_t* __restrict__ lo,
                                 uint64_t* __restrict__ keys, int32_t* __restrict__ vals, int64_t E) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    keys[e] = ((uint64_t)hi[e] << kKeyShift) | (uint64_t)(lo[e] & ((1 << kKeyShift) - 1));
    vals[e] = (int32_t)e;
  }
}

// flags[p] = 1 where a new group (same_hi: group by the high part only; else by the whole key) starts
__global__ void flag_kernel(const uint64_t* __restrict__ keys, int32_t* __restrict__ flags, int64_t E, int by_hi) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    int f = 1;
    if (p > 0) {
      const uint64_t a = by_hi ? (keys[p] >> kKeyShift) : keys[p];
      const uint64_t b = by_hi ? (keys[p - 1] >> kKeyShift) : keys[p - 1];
      f = (a != b);
    }
    flags[p] = f;
  }
}

// rank[p] = inclusive scan of flags; scatter group structures.
__global__ void scatter_patch_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ perm,
                                     const int32_t* __restrict__ flags, const int32_t* __restrict__ rank,
                                     int32_t* __restrict__ ku, int32_t* __restrict__ kx, int32_t* __restrict__ off,
                                     int32_t* __restrict__ ix, int32_t* __restrict__ jx, int32_t* __restrict__ count,
                                     int64_t E) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t g = rank[p] - 1;
    const int32_t e = perm[p];
    ku[e] = g;
    if (flags[p]) { off[g] = (int32_t)p; kx[g] = (int32_t)(keys[p] >> kKeyShift); }
    ix[e] = flags[p] ? -1 : perm[p - 1];
    jx[e] = (p + 1 < E && !flags[p + 1]) ? perm[p + 1] : -1;
    if (p == E - 1) { off[g + 1] = (int32_t)E; *count = g + 1; }
  }
}

__global__ void scatter_pair_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ perm,
                                    const int32_t* __restrict__ flags, const int32_t* __restrict__ rank,
                                    int32_t* __restrict__ pu, int32_t* __restrict__ off, int32_t* __restrict__ pair_ij,
                                    int32_t* __restrict__ count, int64_t E) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t g = rank[p] - 1;
    pu[perm[p]] = g;
    if (flags[p]) {
      off[g] = (int32_t)p;
      pair_ij[2 * g + 0] = (int32_t)(keys[p] >> kKeyShift);
      pair_ij[2 * g + 1] = (int32_t)(keys[p] & ((1 << kKeyShift) - 1));
    }
    if (p == E - 1) { off[g + 1] = (int32_t)E; *count = g + 1; }
  }
}

__global__ void widen_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b, int64_t* __restrict__ A,
                             int64_t* __restrict__ B, int64_t E) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    A[e] = a[e]; B[e] = b[e];
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct WsLayout {
  size_t keys_in, keys_out, vals_in, flags, rank, temp, temp_bytes, total;
};

int ws_layout(int64_t E, WsLayout* L) {
  size_t sort_bytes = 0, scan_bytes = 0;
  const size_t n = (size_t)(E > 0 ? E : 1);
  hipError_t e1 = rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                            (int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64, (hipStream_t)0);
  hipError_t e2 = rocprim::inclusive_scan(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, n,
                                          rocprim::plus<int32_t>(), (hipStream_t)0);
  if (e1 != hipSuccess) return (int)e1;
  if (e2 != hipSuccess) return (int)e2;
  size_t o = 0;
  L->keys_in = o; o += align256(n * 8);
  L->keys_out = o; o += align256(n * 8);
  L->vals_in = o; o += align256(n * 4);
  L->flags = o; o += align256(n * 4);
  L->rank = o; o += align256(n * 4);
  L->temp = o;
  L->temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  o += align256(L->temp_bytes);
  L->total = o;
  return 0;
}

// sorts (hi<<24|lo, edge id), leaves sorted keys in ws.keys_out, perm in `perm`, flags/rank in ws
int sort_and_rank(const int64_t* hi, const int64_t* lo, int64_t E, int32_t* perm, char* ws, const WsLayout& L,
                  int by_hi, hipStream_t st) {
  uint64_t* keys_in = (uint64_t*)(ws + L.keys_in);
  uint64_t* keys_out = (uint64_t*)(ws + L.keys_out);
  int32_t* vals_in = (int32_t*)(ws + L.vals_in);
  int32_t* flags = (int32_t*)(ws + L.flags);
  int32_t* rank = (int32_t*)(ws + L.rank);
  hipLaunchKernelGGL(make_keys_kernel, dim3(grid_for(E)), dim3(256), 0, st, hi, lo, keys_in, vals_in, E);
  size_t tb = L.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs((void*)(ws + L.temp), tb, keys_in, keys_out, vals_in, perm, (size_t)E, 0, 64, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(flag_kernel, dim3(grid_for(E)), dim3(256), 0, st, keys_out, flags, E, by_hi);
  tb = L.temp_bytes;
  e = rocprim::inclusive_scan((void*)(ws + L.temp), tb, flags, rank, (size_t)E, rocprim::plus<int32_t>(), st);
  if (e != hipSuccess) return (int)e;
  return 0;
}

}  // namespace

extern "C" int dpvo_plan_layout(int64_t E, dpvo_plan_layout_t* L) {
  if (E < 0 || !L) return DPVO_E_INVALID;
  int64_t o = 0;
  const int64_t n = E > 0 ? E : 1;
  L->perm_k = o; o += n;
  L->ku = o; o += n;
  L->kx = o; o += n;
  L->patch_off = o; o += n + 1;
  L->ix = o; o += n;
  L->jx = o; o += n;
  L->perm_p = o; o += n;
  L->pu = o; o += n;
  L->pair_off = o; o += n + 1;
  L->pair_ij = o; o += 2 * n;
  L->counts = o; o += 4;
  L->total_ints = o;
  return DPVO_OK;
}

extern "C" size_t dpvo_plan_workspace_bytes(int64_t E) {
  WsLayout L;
  if (E < 0 || ws_layout(E, &L) != 0) return 0;
  return L.total;
}

extern "C" int dpvo_plan_build(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                               void* ws, size_t ws_bytes, void* stream) {
  if (E < 0 || !plan) return DPVO_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  dpvo_plan_layout_t P;
  dpvo_plan_layout(E, &P);
  if (E == 0) {
    hipError_t e = hipMemsetAsync(plan + P.counts, 0, 4 * sizeof(int32_t), st);
    return e == hipSuccess ? DPVO_OK : (int)e;
  }
  if (!ii || !jj || !kk || !ws) return DPVO_E_INVALID;
  WsLayout L;
  int rc = ws_layout(E, &L);
  if (rc) return rc;
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  char* w = (char*)ws;
  hipError_t e = hipMemsetAsync(plan + P.counts, 0, 4 * sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  // --- per-patch structure: sort by (kk, jj, edge)
  rc = sort_and_rank(kk, jj, E, plan + P.perm_k, w, L, /*by_hi=*/1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_patch_kernel, dim3(grid_for(E)), dim3(256), 0, st, (const uint64_t*)(w + L.keys_out),
                     plan + P.perm_k, (const int32_t*)(w + L.flags), (const int32_t*)(w + L.rank), plan + P.ku,
                     plan + P.kx, plan + P.patch_off, plan + P.ix, plan + P.jx, plan + P.counts + 0, E);
  // --- per-frame-pair structure: sort by (ii, jj, edge)
  rc = sort_and_rank(ii, jj, E, plan + P.perm_p, w, L, /*by_hi=*/0, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_pair_kernel, dim3(grid_for(E)), dim3(256), 0, st, (const uint64_t*)(w + L.keys_out),
                     plan + P.perm_p, (const int32_t*)(w + L.flags), (const int32_t*)(w + L.rank), plan + P.pu,
                     plan + P.pair_off, plan + P.pair_ij, plan + P.counts + 1, E);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

// cuda_ba.neighbors API parity: ws must hold dpvo_neighbors_workspace_bytes(E).
extern "C" size_t dpvo_neighbors_workspace_bytes(int64_t E) {
  WsLayout L;
  if (E < 0 || ws_layout(E, &L) != 0) return 0;
  return L.total + align256((size_t)(E > 0 ? E : 1) * 4) * 6 + 512;
}

extern "C" int dpvo_neighbors(const int64_t* kk, const int64_t* jj, int64_t* ix, int64_t* jx, int64_t E, void* ws,
                              size_t ws_bytes, void* stream) {
  if (E < 0) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!kk || !jj || !ix || !jx || !ws) return DPVO_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  WsLayout L;
  int rc = ws_layout(E, &L);
  if (rc) return rc;
  const size_t extra = align256((size_t)E * 4) * 6 + 512;
  if (ws_bytes < L.total + extra) return DPVO_E_WORKSPACE;
  char* w = (char*)ws;
  int32_t* base = (int32_t*)(w + L.total);
  const size_t stride = align256((size_t)E * 4) / 4;
  int32_t *perm = base, *ku = base + stride, *kx = base + 2 * stride, *ix32 = base + 3 * stride,
          *jx32 = base + 4 * stride, *off = base + 5 * stride;   // off needs E+1 <= stride*... (E+1)*4 <= align256(4E)+256
  rc = sort_and_rank(kk, jj, E, perm, w, L, 1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scatter_patch_kernel, dim3(grid_for(E)), dim3(256), 0, st, (const uint64_t*)(w + L.keys_out), perm,
                     (const int32_t*)(w + L.flags), (const int32_t*)(w + L.rank), ku, kx, off, ix32, jx32,
                     off + E + 1 /*count scratch*/, E);
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(E)), dim3(256), 0, st, ix32, jx32, ix, jx, E);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
