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
#include "se3_dev.h"      // (reproject_body: the reprojection can ride in the histogram launch; this unit is built without packed-FP32 ops)
#include <rocprim/rocprim.hpp>

namespace {

constexpr int kKeyShift = 24;   // generic path: jj (frame index) < 2^24; BUFFER_SIZE is 4096 in the reference (config.py:6)

// keys of BOTH sorts in one pass: kA = kk<<shift | jj, kB = ii<<shift | jj, value = edge id
template <typename K>
__global__ void make_keys_kernel(const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                 const int64_t* __restrict__ kk, K* __restrict__ keysA, K* __restrict__ keysB,
                                 int32_t* __restrict__ vals, int64_t E, int shift, int32_t* __restrict__ flow) {
  if (flow && blockIdx.x == 0 && threadIdx.x < 4) flow[threadIdx.x] = threadIdx.x < 2 ? -1 : 0;      // (no flow-test list from this builder)
  const K lomask = ((K)1 << shift) - 1;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const K lo = (K)jj[e] & lomask;
    if (keysA) keysA[e] = ((K)kk[e] << shift) | lo;
    if (keysB) keysB[e] = ((K)ii[e] << shift) | lo;
    vals[e] = (int32_t)e;
  }
}

// Group structures of a sorted key array in ONE launch (flag + scan + scatter), 1024 positions per workgroup.  A workgroup
// gets the rank of its first position by counting the group starts of ALL preceding positions itself (coalesced, L2
// resident: ~E/2048 loads per thread) instead of waiting for a scan pass -- no inter-workgroup communication at all.
// by_hi: a group is a run of equal high parts (patch), else of equal whole keys (frame pair).
//   PATCH: ku[e] = group, kx[g] = patch id, off[g] = first position, ix/jx[e] = previous / next edge of the patch
//   PAIR : pu[e] = group, pair_ij[g] = (i, j), off[g]
// Round 5: part (a) used to re-scan the sorted keys -- E / 2048 loads per thread on average, E / 1024 in the last workgroup: fine for
// the tracker's 47 k active edges (18 us), quadratic for the global BA's 330 k active + inactive ones (2 x 95 us per global-BA frame,
// profiles/r05_lc_timeline.txt).  group_count_kernel leaves the number of group starts of every 1024-position block; a workgroup
// now adds up the counts of the blocks before it (<= E / 1024 values).  Same ranks, same outputs.
template <typename K, bool PATCH>
__global__ __launch_bounds__(1024) void group_count_kernel(const K* __restrict__ keys, int32_t* __restrict__ blk_count, int64_t E, int shift) {
  __shared__ int32_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t p = (int64_t)blockIdx.x * 1024 + t;
  auto gkey = [&](int64_t q) -> K { return PATCH ? (K)(keys[q] >> shift) : keys[q]; };
  const bool start = p < E && (p == 0 || gkey(p) != gkey(p - 1));
  const unsigned long long bal = __ballot(start);
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  if (t == 0) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += wsum[i];
    blk_count[blockIdx.x] = c;
  }
}

template <typename K, bool PATCH>
__global__ __launch_bounds__(1024) void group_kernel(const K* __restrict__ keys, const int32_t* __restrict__ perm,
                                                     int32_t* __restrict__ gu, int32_t* __restrict__ gx,
                                                     int32_t* __restrict__ off, int32_t* __restrict__ ix,
                                                     int32_t* __restrict__ jx, int32_t* __restrict__ count,
                                                     int32_t* __restrict__ zero2, int64_t E, int shift,
                                                     const int32_t* __restrict__ blk_count) {
  __shared__ int32_t wsum[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  auto gkey = [&](int64_t p) -> K { return PATCH ? (K)(keys[p] >> shift) : keys[p]; };
  auto is_start = [&](int64_t p) -> bool { return p == 0 || gkey(p) != gkey(p - 1); };
  // (a) group starts before this workgroup's first position
  const int64_t first = (int64_t)blockIdx.x * 1024;
  int32_t c = 0;
  for (int b = t; b < (int)blockIdx.x; b += 1024) c += blk_count[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) wsum[wv] = c;
  __syncthreads();
  int32_t base = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) base += wsum[i];
  __syncthreads();
  // (b) ranks inside the workgroup
  const int64_t p = first + t;
  const bool valid = p < E;
  const bool start = valid && is_start(p);
  const unsigned long long bal = __ballot(start);
  const int32_t wpre = __popcll(bal & ((1ull << lane) - 1));
  if (lane == 0) wsum[wv] = __popcll(bal);
  __syncthreads();
  int32_t wbase = 0, total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { wbase += i < wv ? wsum[i] : 0; total += wsum[i]; }
  const int32_t g = base + wbase + wpre + (start ? 1 : 0) - 1;
  // (c) scatter
  if (valid) {
    const int32_t e = perm[p];
    gu[e] = g;
    const K lomask = ((K)1 << shift) - 1;
    if (start) {
      off[g] = (int32_t)p;
      if (PATCH) gx[g] = (int32_t)(keys[p] >> shift);
      else { gx[2 * g + 0] = (int32_t)(keys[p] >> shift); gx[2 * g + 1] = (int32_t)(keys[p] & lomask); }
    }
    if (PATCH) {
      ix[e] = start ? -1 : perm[p - 1];
      jx[e] = (p + 1 < E && gkey(p + 1) == gkey(p)) ? perm[p + 1] : -1;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && t == 0) {
    off[base + total] = (int32_t)E;
    *count = base + total;
    if (zero2) { zero2[0] = 0; zero2[1] = 0; }
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
  size_t keys_a, keys_b, keys_out, vals_in, temp, temp_bytes, win, total;
};

// ---------------------------------------------------------------------------------------------------
// Windowed plan build (dpvo_plan_build_window): when the caller can bound the frame and patch ids of all edges by a small
// window (the tracker can: source frames within REMOVAL_WINDOW, targets within PATCH_LIFETIME of them), both orderings
// are ONE stable counting-sort pass each -- by patch (<= 4096 bins), then each patch's few edges ranked by (jj, edge) by
// brute force; by frame pair (<= 2048 bins) -- sharing their launches: per-tile histograms, one scan, one scatter, then the
// group structures straight from the bins (group index = number of non-empty bins before).  5 launches instead of 14 (two rocPRIM device radix sorts cost ~45 us each at
// E = 47 k whatever the key width, almost all of it launch-to-launch latency).
// ---------------------------------------------------------------------------------------------------
constexpr int kWinA = 4096, kWinB = 2048, kWinBins = kWinA + kWinB, kWinTile = 1024;

struct WinArgs {
  const int64_t *ii, *jj, *kk; int64_t E;
  int frame_lo, nfw, patch_lo, npw;     // windows: frame ids in [frame_lo, frame_lo + nfw), patch ids in [patch_lo, patch_lo + npw)
  int shift;                            // composite keys: (hi << shift) | jj, as in the radix path
  int32_t* T;                           // [tiles][kWinBins] tile histograms -> exclusive offsets
  int32_t* binstart;                    // [kWinBins + 2]: segment starts of the A bins ([0, npw]) and of the B bins (kWinA + [0, nfw*nfw])
  int32_t* tot;                         // [kWinBins]: per-bin totals (zeroed before the histogram kernel)
  uint32_t *tmp_key; int32_t* tmp_e;    // by-patch order before the in-patch ranking
  int32_t *perm_k, *perm_p, *flag;      // flag: ids outside the window were seen (lives next to tot, copied to counts[3])
  int32_t* gid;                         // [kWinBins]: number of non-empty bins before a bin = its group index
  int32_t *ku, *kx, *patch_off, *ix, *jx, *pu, *pair_off, *pair_ij, *counts;
  int32_t* flow; int qi, qj;            // flow-test list of the pair (qi, qj) / (qj, qi): dpvo_plan_layout_t.flow
};

__device__ __forceinline__ void win_bins(const WinArgs& W, int64_t e, int& ba, int& bb, bool& bad) {
  const int k = (int)(W.kk[e] - W.patch_lo), i = (int)(W.ii[e] - W.frame_lo), j = (int)(W.jj[e] - W.frame_lo);
  bad = k < 0 || k >= W.npw || i < 0 || i >= W.nfw || j < 0 || j >= W.nfw;
  ba = k < 0 ? 0 : (k >= W.npw ? W.npw - 1 : k);
  const int ic = i < 0 ? 0 : (i >= W.nfw ? W.nfw - 1 : i), jc = j < 0 ? 0 : (j >= W.nfw ? W.nfw - 1 : j);
  bb = ic * W.nfw + jc;
}

__global__ __launch_bounds__(1024) void win_zero_kernel(int32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// an independent job that may share the histogram launch (its blocks follow the tiles'): the reprojection of the same edges
struct ReprojJob { const float *poses, *patches, *intr; float* coords; int P, nblk; };
__global__ __launch_bounds__(1024) void win_hist_kernel(WinArgs W, int tiles, ReprojJob R) {
  if ((int)blockIdx.x >= tiles) {
    reproject_body(R.poses, R.patches, R.intr, W.ii, W.jj, W.kk, R.coords, W.E, R.P, 1, (int)blockIdx.x - tiles, R.nblk);
    return;
  }
  __shared__ int32_t h[kWinBins];
  if (blockIdx.x == 0 && threadIdx.x < 4) W.flow[threadIdx.x] = threadIdx.x == 0 ? W.qi : (threadIdx.x == 1 ? W.qj : 0);   // flow-list header
  const int nb_b = W.nfw * W.nfw;
  for (int b = threadIdx.x; b < kWinBins; b += 1024) h[b] = 0;
  __syncthreads();
  const int64_t e = (int64_t)blockIdx.x * kWinTile + threadIdx.x;
  if (e < W.E) {
    int ba, bb; bool bad;
    win_bins(W, e, ba, bb, bad);
    if (bad) *W.flag = 1;
    atomicAdd(&h[ba], 1);
    atomicAdd(&h[kWinA + bb], 1);
  }
  __syncthreads();
  int32_t* T = W.T + (int64_t)blockIdx.x * kWinBins;
  for (int b = threadIdx.x; b < W.npw; b += 1024) { T[b] = h[b]; if (h[b]) atomicAdd(&W.tot[b], h[b]); }
  for (int b = threadIdx.x; b < nb_b; b += 1024) { T[kWinA + b] = h[kWinA + b]; if (h[kWinA + b]) atomicAdd(&W.tot[kWinA + b], h[kWinA + b]); }
}

// T[tile][bin] -> exclusive prefix over the tiles, plus for every bin its segment start (binstart) and its group index
// (gid = number of non-empty bins before it).  One workgroup per 1024 bins of an ordering (blockIdx.x < chunks_a: by-patch
// bins, else pair bins); what lies before the chunk is summed from the per-bin totals the histogram kernel accumulated
// (W.tot), so the workgroups do not wait for each other.  The last bin of each ordering also writes the group count and
// the closing offset.
__global__ __launch_bounds__(1024) void win_scan_kernel(WinArgs W, int tiles, int chunks_a) {
  __shared__ int32_t wsum[16], wocc[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int part = (int)blockIdx.x >= chunks_a;
  const int c0 = 1024 * (part ? (int)blockIdx.x - chunks_a : (int)blockIdx.x);
  const int nb = part ? W.nfw * W.nfw : W.npw, b0 = part ? kWinA : 0;
  // bins before this chunk
  int32_t pre = 0, preo = 0;
  for (int q = t; q < c0; q += 1024) { const int32_t v = W.tot[b0 + q]; pre += v; preo += v != 0; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { pre += __shfl_xor(pre, o); preo += __shfl_xor(preo, o); }
  if (lane == 0) { wsum[wv] = pre; wocc[wv] = preo; }
  __syncthreads();
  int32_t carry = 0, carryo = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { carry += wsum[k]; carryo += wocc[k]; }
  __syncthreads();
  // own bin: prefix over the tiles (loads in independent batches of 8)
  const int b = c0 + t;
  int32_t tot = 0;
  if (b < nb) {
    int32_t* col = W.T + b0 + b;
    for (int tl = 0; tl < tiles; tl += 8) {
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (tl + u < tiles) ? col[(int64_t)(tl + u) * kWinBins] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (tl + u < tiles) col[(int64_t)(tl + u) * kWinBins] = tot;
        tot += v[u];
      }
    }
  }
  int32_t x = tot, xo = tot != 0;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t y = __shfl_up(x, o), yo = __shfl_up(xo, o);
    if (lane >= o) { x += y; xo += yo; }
  }
  if (lane == 63) { wsum[wv] = x; wocc[wv] = xo; }
  __syncthreads();
  int32_t wbase = 0, obase = 0;
  for (int k = 0; k < wv; ++k) { wbase += wsum[k]; obase += wocc[k]; }
  if (b < nb) {
    W.binstart[b0 + b] = carry + wbase + x - tot;
    W.gid[b0 + b] = carryo + obase + xo - (tot != 0);
  }
  if (b == nb - 1) {
    const int32_t ng = carryo + obase + xo, E = carry + wbase + x;
    W.binstart[b0 + nb] = E;
    (part ? W.pair_off : W.patch_off)[ng] = E;
    W.counts[part] = ng;
    if (part) { W.counts[2] = 0; W.counts[3] = *W.flag; }
  }
}

// rank of a lane among the lanes of its wave with the same bin (bits: width of the bin index), and the size of that group
__device__ __forceinline__ void wave_match(int bin, bool valid, int bits, int lane, int& rank, int& cnt) {
  unsigned long long m = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const unsigned long long bal = __ballot((bin >> b) & 1);
    m &= ((bin >> b) & 1) ? bal : ~bal;
  }
  rank = __popcll(m & ((1ull << lane) - 1));
  cnt = __popcll(m);
}

__global__ __launch_bounds__(1024) void win_scatter_kernel(WinArgs W) {
  __shared__ int32_t run[kWinBins];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int b = t; b < kWinBins; b += 1024) run[b] = 0;
  const int64_t e = (int64_t)blockIdx.x * kWinTile + t;
  const bool valid = e < W.E;
  int ba = 0, bb = 0; bool bad;
  if (valid) win_bins(W, e, ba, bb, bad);
  int ra, ca, rb, cb;
  wave_match(ba, valid, 12, lane, ra, ca);
  wave_match(bb, valid, 11, lane, rb, cb);
  int la = 0, lb = 0;
  // the 16 waves of the tile take their turn in order: stable ranks inside the tile
  for (int w = 0; w < 16; ++w) {
    __syncthreads();
    if (wv == w && valid) {
      la = run[ba] + ra;
      lb = run[kWinA + bb] + rb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wv == w && valid) {
      if (ra == 0) run[ba] += ca;
      if (rb == 0) run[kWinA + bb] += cb;
    }
  }
  if (valid) {
    const int32_t* T = W.T + (int64_t)blockIdx.x * kWinBins;
    const int pa = W.binstart[ba] + T[ba] + la;
    const int pb = W.binstart[kWinA + bb] + T[kWinA + bb] + lb;
    W.tmp_key[pa] = ((uint32_t)W.kk[e] << W.shift) | ((uint32_t)W.jj[e] & ((1u << W.shift) - 1));
    W.tmp_e[pa] = (int32_t)e;
    W.perm_p[pb] = (int32_t)e;
    // frame-pair group structures: the bins are in (i, j) order already
    const int g = W.gid[kWinA + bb];
    W.pu[e] = g;
    const int bs = W.binstart[kWinA + bb];
    const int32_t fi = (int32_t)W.ii[e], fj = (int32_t)W.jj[e];
    if (pb == bs) {
      W.pair_off[g] = pb;
      W.pair_ij[2 * g] = fi;
      W.pair_ij[2 * g + 1] = fj;
    }
    // the flow test's pair: its edges in edge order (the scatter is stable), and how many there are
    if (W.qi >= 0) {
      const int dirn = (fi == W.qi && fj == W.qj) ? 0 : ((fi == W.qj && fj == W.qi) ? 1 : -1);
      if (dirn >= 0) {
        const int r = pb - bs;
        if (r < DPVO_PLAN_FLOW_CAP) W.flow[4 + dirn * DPVO_PLAN_FLOW_CAP + r] = (int32_t)W.kk[e];
        if (r == 0) W.flow[2 + dirn] = W.binstart[kWinA + bb + 1] - bs;
      }
    }
  }
}

// by-patch order -> (patch, jj, edge) order: every element ranks itself inside its patch's segment and finds its
// predecessor / successor there (fastba.neighbors); per-patch group structures from the bins
__global__ void win_segsort_kernel(WinArgs W) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= W.E) return;
  const uint32_t key = W.tmp_key[p];
  const int32_t e = W.tmp_e[p];
  int k = (int)(key >> W.shift) - W.patch_lo;
  k = k < 0 ? 0 : (k >= W.npw ? W.npw - 1 : k);
  const int s = W.binstart[k], t = W.binstart[k + 1];
  int rank = 0;
  uint32_t pk = 0, nk = 0xffffffffu; int32_t pe = -1, ne = -1;        // closest element below / above in (key, edge) order
  for (int q = s; q < t; ++q) {
    const uint32_t kq = W.tmp_key[q];
    const int32_t eq = W.tmp_e[q];
    const bool below = (kq < key) || (kq == key && eq < e);
    const bool above = (kq > key) || (kq == key && eq > e);
    rank += below;
    if (below && (pe < 0 || kq > pk || (kq == pk && eq > pe))) { pk = kq; pe = eq; }
    if (above && (ne < 0 || kq < nk || (kq == nk && eq < ne))) { nk = kq; ne = eq; }
  }
  int64_t pos = (int64_t)s + rank;
  if (pos >= W.E) pos = W.E - 1;          // (only reachable with ids outside the promised window: stay inside the arrays)
  W.perm_k[pos] = e;
  W.ix[e] = pe; W.jx[e] = ne;
  const int g = W.gid[k];
  W.ku[e] = g;
  if (rank == 0) { W.kx[g] = (int32_t)(key >> W.shift); W.patch_off[g] = s; }
}

// ---------------------------------------------------------------------------------------------------
// Wide plan build (dpvo_plan_build_wide, round 5): the same two orderings when the ids are bounded by the tracker's frame count only
// (LOOP_CLOSURE: long-range edges among the active ones; the global BA's plan of all active + inactive edges) -- too many bins for
// per-tile histograms in LDS, so the bins live in memory: one bin per patch id and one per (i, j) frame pair.  Histogram (integer
// atomics: order-independent), two-level scan over the bins, placement by counting DOWN the same counters (any order inside a bin),
// then every element ranks itself inside its bin's segment by (jj, edge) resp. by edge -- which is what makes the result the stable
// sort's, independent of the order the atomics landed in.  6 launches; the two rocPRIM radix sorts it replaces are 16 launches at
// E = 50 k and 48 at E = 300 k (170 / 450 us of every global-BA frame, profiles/r05_e_lc_timeline.txt).  Consecutive edges usually
// share a bin (an append lists the 96 patches of a frame against one target, or one patch against 13 targets), so a run of equal
// bins inside a wave issues ONE atomic for the whole run.
// ---------------------------------------------------------------------------------------------------
constexpr int64_t kWideMaxBins = (int64_t)1 << 22;

struct WideArgs {
  const int64_t *ii, *jj, *kk; int64_t E;
  int nf, np;                           // promised: ii, jj in [0, nf), kk in [0, np)
  int nbA, nbB, chunksA, chunksB;       // bins of the two orderings (np, nf * nf) and their 1024-bin chunks
  int32_t* tot;                         // [nbA + nbB] per-bin counts (zeroed; counted down to zero again by the scatter), then the flag
  int32_t* flag;                        // ids outside the promise were seen (-> counts[3])
  int32_t* csum;                        // [chunksA + chunksB][2]: edges / non-empty bins of every chunk
  int32_t *startA, *startB;             // [nbA + 1], [nbB + 1]: segment starts
  int32_t *gidA, *gidB;                 // number of non-empty bins before a bin = its group index
  uint32_t* tmpA_j; int32_t *tmpA_e, *tmpB_e;      // bin order before the ranking
  int32_t *perm_k, *perm_p, *ku, *kx, *patch_off, *ix, *jx, *pu, *pair_off, *pair_ij, *counts, *flow;
};

__device__ __forceinline__ void wide_bins(const WideArgs& W, int64_t e, int& ba, int& bb, bool& bad) {
  const int64_t k = W.kk[e], i = W.ii[e], j = W.jj[e];
  bad = k < 0 || k >= W.np || i < 0 || i >= W.nf || j < 0 || j >= W.nf;
  ba = (int)(k < 0 ? 0 : (k >= W.np ? W.np - 1 : k));
  const int ic = (int)(i < 0 ? 0 : (i >= W.nf ? W.nf - 1 : i)), jc = (int)(j < 0 ? 0 : (j >= W.nf ? W.nf - 1 : j));
  bb = ic * W.nf + jc;
}

// Runs of equal bins among the valid lanes of a wave (the valid lanes are a prefix of the wave): head = first lane of its run,
// hl = the head lane of this lane's run, cnt = length of the run (meaningful in the head lane).  Every lane of the wave calls this.
__device__ __forceinline__ void wave_runs(int bin, bool valid, int lane, bool& head, int& hl, int& cnt) {
  const int prev = __shfl_up(bin, 1);
  head = valid && (lane == 0 || prev != bin);
  const unsigned long long heads = __ballot(head), vmask = __ballot(valid);
  const unsigned long long upto = (2ull << lane) - 1;              // bits 0 .. lane (lane 63: all ones)
  const unsigned long long below = heads & upto, above = heads & ~upto;
  hl = below ? 63 - __clzll((long long)below) : 0;
  const int nvalid = __popcll(vmask);
  const int next = above ? __ffsll((long long)above) - 1 : nvalid;
  cnt = next - lane;
}

__global__ __launch_bounds__(256) void wide_hist_kernel(WideArgs W) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = e < W.E;
  int ba = 0, bb = 0; bool bad = false;
  if (valid) wide_bins(W, e, ba, bb, bad);
  if (bad) *W.flag = 1;
  bool head; int hl, cnt;
  wave_runs(ba, valid, lane, head, hl, cnt);
  if (head) atomicAdd(&W.tot[ba], cnt);
  wave_runs(bb, valid, lane, head, hl, cnt);
  if (head) atomicAdd(&W.tot[W.nbA + bb], cnt);
}

// edges and non-empty bins of every 1024-bin chunk (blockIdx.x < chunksA: by-patch bins, else pair bins)
__global__ __launch_bounds__(1024) void wide_chunk_kernel(WideArgs W) {
  __shared__ int32_t wsum[16], wocc[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int part = (int)blockIdx.x >= W.chunksA;
  const int c = part ? (int)blockIdx.x - W.chunksA : (int)blockIdx.x;
  const int nb = part ? W.nbB : W.nbA, b0 = part ? W.nbA : 0;
  const int b = c * 1024 + t;
  int32_t v = b < nb ? W.tot[b0 + b] : 0, o = v != 0;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { v += __shfl_xor(v, s); o += __shfl_xor(o, s); }
  if (lane == 0) { wsum[wv] = v; wocc[wv] = o; }
  __syncthreads();
  if (t == 0) {
    int32_t a = 0, q = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a += wsum[k]; q += wocc[k]; }
    W.csum[2 * blockIdx.x] = a; W.csum[2 * blockIdx.x + 1] = q;
  }
}

// segment start and group index of every bin; the last bin of each ordering also writes the group count and the closing offset
__global__ __launch_bounds__(1024) void wide_scan_kernel(WideArgs W) {
  __shared__ int32_t wsum[16], wocc[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int part = (int)blockIdx.x >= W.chunksA;
  const int c = part ? (int)blockIdx.x - W.chunksA : (int)blockIdx.x;
  const int nb = part ? W.nbB : W.nbA, b0 = part ? W.nbA : 0;
  if (blockIdx.x == 0 && t < 4) W.flow[t] = t < 2 ? -1 : 0;        // (no flow-test list from this builder)
  // chunks of this ordering before this one
  const int32_t* cs = W.csum + 2 * (part ? W.chunksA : 0);
  int32_t pre = 0, preo = 0;
  for (int q = t; q < c; q += 1024) { pre += cs[2 * q]; preo += cs[2 * q + 1]; }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { pre += __shfl_xor(pre, s); preo += __shfl_xor(preo, s); }
  if (lane == 0) { wsum[wv] = pre; wocc[wv] = preo; }
  __syncthreads();
  int32_t carry = 0, carryo = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { carry += wsum[k]; carryo += wocc[k]; }
  __syncthreads();
  const int b = c * 1024 + t;
  const int32_t tot = b < nb ? W.tot[b0 + b] : 0;
  int32_t x = tot, xo = tot != 0;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int32_t y = __shfl_up(x, s), yo = __shfl_up(xo, s);
    if (lane >= s) { x += y; xo += yo; }
  }
  if (lane == 63) { wsum[wv] = x; wocc[wv] = xo; }
  __syncthreads();
  int32_t wbase = 0, obase = 0;
  for (int k = 0; k < wv; ++k) { wbase += wsum[k]; obase += wocc[k]; }
  int32_t* start = part ? W.startB : W.startA;
  if (b < nb) {
    start[b] = carry + wbase + x - tot;
    (part ? W.gidB : W.gidA)[b] = carryo + obase + xo - (tot != 0);
  }
  if (b == nb - 1) {
    const int32_t ng = carryo + obase + xo, E = carry + wbase + x;
    start[nb] = E;
    (part ? W.pair_off : W.patch_off)[ng] = E;
    W.counts[part] = ng;
    if (part) { W.counts[2] = 0; W.counts[3] = *W.flag; }
  }
}

__global__ __launch_bounds__(256) void wide_scatter_kernel(WideArgs W) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = e < W.E;
  int ba = 0, bb = 0; bool bad = false;
  if (valid) wide_bins(W, e, ba, bb, bad);
  bool head; int hl, cnt;
  wave_runs(ba, valid, lane, head, hl, cnt);
  int base = head ? atomicSub(&W.tot[ba], cnt) - cnt : 0;           // the run's slots inside the bin's segment: [base, base + cnt)
  base = __shfl(base, hl);
  if (valid) {
    const int pa = W.startA[ba] + base + (lane - hl);
    W.tmpA_j[pa] = (uint32_t)W.jj[e];
    W.tmpA_e[pa] = (int32_t)e;
  }
  wave_runs(bb, valid, lane, head, hl, cnt);
  base = head ? atomicSub(&W.tot[W.nbA + bb], cnt) - cnt : 0;
  base = __shfl(base, hl);
  if (valid) W.tmpB_e[W.startB[bb] + base + (lane - hl)] = (int32_t)e;
}

// thread p < E: position p of the by-patch order; thread E + p: position p of the by-pair order.  Every element ranks itself
// inside its bin's segment ((jj, edge) resp. edge: a total order, so the ranks are a permutation of the segment) and writes the
// group structures of its ordering; the first of a segment also writes the group's entry.
__global__ __launch_bounds__(256) void wide_rank_kernel(WideArgs W) {
  const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gt < W.E) {
    const int32_t e = W.tmpA_e[gt];
    const uint32_t key = W.tmpA_j[gt];
    const int64_t kraw = W.kk[e];
    const int k = (int)(kraw < 0 ? 0 : (kraw >= W.np ? W.np - 1 : kraw));
    const int s = W.startA[k], t = W.startA[k + 1];
    int rank = 0;
    uint32_t pk = 0, nk = 0xffffffffu; int32_t pe = -1, ne = -1;          // closest element below / above in (jj, edge) order
    for (int q = s; q < t; ++q) {
      const uint32_t kq = W.tmpA_j[q];
      const int32_t eq = W.tmpA_e[q];
      const bool below = (kq < key) || (kq == key && eq < e);
      const bool above = (kq > key) || (kq == key && eq > e);
      rank += below;
      if (below && (pe < 0 || kq > pk || (kq == pk && eq > pe))) { pk = kq; pe = eq; }
      if (above && (ne < 0 || kq < nk || (kq == nk && eq < ne))) { nk = kq; ne = eq; }
    }
    int64_t pos = (int64_t)s + rank;
    if (pos >= W.E) pos = W.E - 1;
    W.perm_k[pos] = e;
    W.ix[e] = pe; W.jx[e] = ne;
    const int g = W.gidA[k];
    W.ku[e] = g;
    if (rank == 0) { W.kx[g] = (int32_t)kraw; W.patch_off[g] = s; }
  } else if (gt < 2 * W.E) {
    const int32_t e = W.tmpB_e[gt - W.E];
    int ba, bb; bool bad;
    wide_bins(W, e, ba, bb, bad);
    const int s = W.startB[bb], t = W.startB[bb + 1];
    int rank = 0;
    for (int q = s; q < t; ++q) rank += W.tmpB_e[q] < e;
    int64_t pos = (int64_t)s + rank;
    if (pos >= W.E) pos = W.E - 1;
    W.perm_p[pos] = e;
    const int g = W.gidB[bb];
    W.pu[e] = g;
    if (rank == 0) {
      W.pair_off[g] = s;
      W.pair_ij[2 * g] = (int32_t)W.ii[e];
      W.pair_ij[2 * g + 1] = (int32_t)W.jj[e];
    }
  }
}

struct WideWs { size_t tot, csum, startA, startB, gidA, gidB, tmpA_j, tmpA_e, tmpB_e, total; int nbA, nbB, chunksA, chunksB; };

// 0 = ok, else the id ranges are outside what the wide build takes (the caller then uses the radix build)
int wide_layout(int64_t E, int64_t n_frames, int64_t n_patch_ids, WideWs* L) {
  if (E < 0 || E >= ((int64_t)1 << 30) || n_frames <= 0 || n_patch_ids <= 0 || n_frames > 2048 || n_patch_ids > kWideMaxBins ||
      n_frames * n_frames > kWideMaxBins)
    return 1;
  const size_t n = (size_t)(E > 0 ? E : 1);
  L->nbA = (int)n_patch_ids; L->nbB = (int)(n_frames * n_frames);
  L->chunksA = (L->nbA + 1023) / 1024; L->chunksB = (L->nbB + 1023) / 1024;
  size_t o = 0;
  L->tot = o; o += align256(((size_t)L->nbA + L->nbB + 1) * 4);
  L->csum = o; o += align256((size_t)(L->chunksA + L->chunksB) * 2 * 4);
  L->startA = o; o += align256(((size_t)L->nbA + 1) * 4);
  L->startB = o; o += align256(((size_t)L->nbB + 1) * 4);
  L->gidA = o; o += align256((size_t)L->nbA * 4);
  L->gidB = o; o += align256((size_t)L->nbB * 4);
  L->tmpA_j = o; o += align256(n * 4);
  L->tmpA_e = o; o += align256(n * 4);
  L->tmpB_e = o; o += align256(n * 4);
  L->total = o;
  return 0;
}

// (sized for the 64-bit generic path; the 32-bit path uses a prefix of every buffer)
int ws_layout(int64_t E, WsLayout* L) {
  size_t sort_bytes = 0, sort_bytes32 = 0;
  const size_t n = (size_t)(E > 0 ? E : 1);
  hipError_t e1 = rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                            (int32_t*)nullptr, (int32_t*)nullptr, n, 0, 64, (hipStream_t)0);
  hipError_t e2 = rocprim::radix_sort_pairs(nullptr, sort_bytes32, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                            (int32_t*)nullptr, (int32_t*)nullptr, n, 0, 32, (hipStream_t)0);
  if (e1 != hipSuccess) return (int)e1;
  if (e2 != hipSuccess) return (int)e2;
  size_t o = 0;
  L->keys_a = o; o += align256(n * 8);
  L->keys_b = o; o += align256(n * 8);
  L->keys_out = o; o += align256(n * 8);
  L->vals_in = o; o += align256(n * 4);
  L->temp = o;
  L->temp_bytes = sort_bytes > sort_bytes32 ? sort_bytes : sort_bytes32;
  o += align256(L->temp_bytes);
  L->win = o;                      // windowed path: tile histograms + bin starts (keys_* / vals_in are reused for its arrays)
  o += align256((size_t)cdiv64((int64_t)n, kWinTile) * kWinBins * 4 + (size_t)(3 * kWinBins + 4) * 4 + 256);
  L->total = o;
  return 0;
}

inline int bits_for(int64_t n) {      // bits needed for values in [0, n)
  int b = 1;
  while (b < 63 && ((int64_t)1 << b) < n) ++b;
  return b;
}

// The whole plan for key type K.  which: bit 0 = per-patch structure, bit 1 = per-pair structure.
template <typename K>
int build_plan(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
               const dpvo_plan_layout_t& P, char* w, const WsLayout& L, int shift, int bits_a, int bits_b, int which,
               hipStream_t st) {
  K* keys_a = (K*)(w + L.keys_a);
  K* keys_b = (K*)(w + L.keys_b);
  K* keys_out = (K*)(w + L.keys_out);
  int32_t* vals_in = (int32_t*)(w + L.vals_in);
  int32_t* blk_count = (int32_t*)(w + L.win);      // (the window path's tile histograms: unused here, >= 24 KB per 1024 edges)
  hipLaunchKernelGGL(make_keys_kernel<K>, dim3(grid_for(E)), dim3(256), 0, st, ii, jj, kk, (which & 1) ? keys_a : (K*)nullptr,
                     (which & 2) ? keys_b : (K*)nullptr, vals_in, E, shift, plan + P.flow);
  if (which & 1) {      // sort by (kk, jj, edge)
    size_t tb = L.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs((void*)(w + L.temp), tb, keys_a, keys_out, vals_in, plan + P.perm_k, (size_t)E, 0,
                                             (unsigned)bits_a, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((group_count_kernel<K, true>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const K*)keys_out, blk_count, E, shift);
    hipLaunchKernelGGL((group_kernel<K, true>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const K*)keys_out, plan + P.perm_k, plan + P.ku,
                       plan + P.kx, plan + P.patch_off, plan + P.ix, plan + P.jx, plan + P.counts + 0, plan + P.counts + 2, E,
                       shift, blk_count);
  }
  if (which & 2) {      // sort by (ii, jj, edge)
    size_t tb = L.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs((void*)(w + L.temp), tb, keys_b, keys_out, vals_in, plan + P.perm_p, (size_t)E, 0,
                                             (unsigned)bits_b, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((group_count_kernel<K, false>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const K*)keys_out, blk_count, E, shift);
    hipLaunchKernelGGL((group_kernel<K, false>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const K*)keys_out, plan + P.perm_p, plan + P.pu,
                       plan + P.pair_ij, plan + P.pair_off, (int32_t*)nullptr, (int32_t*)nullptr, plan + P.counts + 1,
                       (int32_t*)nullptr, E, shift, blk_count);
  }
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
  o = (o + 3) & ~(int64_t)3;
  L->flow = o; o += DPVO_PLAN_FLOW_INTS;
  L->total_ints = o;
  return DPVO_OK;
}

extern "C" size_t dpvo_plan_workspace_bytes(int64_t E) {
  WsLayout L;
  if (E < 0 || ws_layout(E, &L) != 0) return 0;
  return L.total;
}

extern "C" int dpvo_plan_build_ranged(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                                      void* ws, size_t ws_bytes, int64_t n_frames, int64_t n_patch_ids, void* stream) {
  if (E < 0 || !plan) return DPVO_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  dpvo_plan_layout_t P;
  dpvo_plan_layout(E, &P);
  if (E == 0) {
    hipError_t e = hipMemsetAsync(plan + P.counts, 0, 4 * sizeof(int32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(plan + P.flow, 0xff, 4 * sizeof(int32_t), st);       // (qi = qj = -1: no flow-test list)
    return e == hipSuccess ? DPVO_OK : (int)e;
  }
  if (!ii || !jj || !kk || !ws) return DPVO_E_INVALID;
  WsLayout L;
  int rc = ws_layout(E, &L);
  if (rc) return rc;
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  // With bounds on the index ranges (ii, jj < n_frames, kk < n_patch_ids) the composite keys usually fit 32 bits and
  // the LSD radix sorts only visit the bits that can be set: 4 digit passes instead of 8.
  if (n_frames > 0 && n_patch_ids > 0) {
    const int lo = bits_for(n_frames), hk = bits_for(n_patch_ids), hi = bits_for(n_frames);
    if (lo + hk <= 32 && lo + hi <= 32)
      rc = build_plan<uint32_t>(ii, jj, kk, E, plan, P, (char*)ws, L, lo, lo + hk, lo + hi, 3, st);
    else if (lo + hk <= 64)
      rc = build_plan<uint64_t>(ii, jj, kk, E, plan, P, (char*)ws, L, lo, lo + hk, lo + hi, 3, st);
    else
      return DPVO_E_UNSUPPORTED;
  } else {
    rc = build_plan<uint64_t>(ii, jj, kk, E, plan, P, (char*)ws, L, kKeyShift, 64, 64, 3, st);
  }
  if (rc) return rc;
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_plan_build_window(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                                      void* ws, size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo,
                                      int64_t n_patches_win, void* stream) {
  return dpvo_plan_build_window_flow(ii, jj, kk, E, plan, ws, ws_bytes, frame_lo, n_frames_win, patch_lo, n_patches_win, -1, -1, stream);
}

extern "C" int dpvo_plan_build_window_flow(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                                           void* ws, size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo,
                                           int64_t n_patches_win, int64_t qi, int64_t qj, void* stream) {
  return dpvo_plan_build_window_job(ii, jj, kk, E, plan, ws, ws_bytes, frame_lo, n_frames_win, patch_lo, n_patches_win, qi, qj, 0, nullptr,
                                    nullptr, nullptr, nullptr, 0, stream);
}

// (internal, csrc/common.h)  where the window build keeps the counters it needs cleared before its histogram launch
extern "C" int dpvo_plan_window_counters(int64_t E, void* ws, size_t ws_bytes, int32_t** ptr, int64_t* count) {
  WsLayout L;
  if (E <= 0 || !ws || !ptr || !count || ws_layout(E, &L) != 0 || ws_bytes < L.total) return DPVO_E_INVALID;
  const int tiles = (int)cdiv64(E, kWinTile);
  int32_t* T = (int32_t*)((char*)ws + L.win);
  *ptr = T + (size_t)tiles * kWinBins + kWinBins + 2;          // = W.tot (flag follows)
  *count = kWinBins + 1;
  return DPVO_OK;
}

// (internal)  the window build with two options of the frame call: counters_cleared != 0 -- the caller has cleared the region
// dpvo_plan_window_counters names (dpvo_frame_update lets the frame-state launch do it: one launch less); r_poses != NULL -- the
// reprojection of the same edges (dpvo_reproject(..., clamp_z = 1) into r_coords) rides in the histogram launch as extra blocks
extern "C" int dpvo_plan_build_window_job(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                                          void* ws, size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo,
                                          int64_t n_patches_win, int64_t qi, int64_t qj, int counters_cleared, const float* r_poses,
                                          const float* r_patches, const float* r_intr, float* r_coords, int r_P, void* stream) {
  if (E < 0 || !plan || frame_lo < 0 || patch_lo < 0 || n_frames_win <= 0 || n_patches_win <= 0) return DPVO_E_INVALID;
  if (n_frames_win * n_frames_win > kWinB || n_patches_win > kWinA || E >= (1 << 24)) return DPVO_E_UNSUPPORTED;
  const int shift = bits_for(frame_lo + n_frames_win);
  if (shift + bits_for(patch_lo + n_patches_win) > 32) return DPVO_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dpvo_plan_layout_t P;
  dpvo_plan_layout(E, &P);
  if (E == 0) {
    hipError_t e = hipMemsetAsync(plan + P.counts, 0, 4 * sizeof(int32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(plan + P.flow, 0xff, 4 * sizeof(int32_t), st);       // (qi = qj = -1: no flow-test list)
    return e == hipSuccess ? DPVO_OK : (int)e;
  }
  if (!ii || !jj || !kk || !ws) return DPVO_E_INVALID;
  WsLayout L;
  int rc = ws_layout(E, &L);
  if (rc) return rc;
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  char* w = (char*)ws;
  const int tiles = (int)cdiv64(E, kWinTile);
  WinArgs W;
  W.ii = ii; W.jj = jj; W.kk = kk; W.E = E;
  W.frame_lo = (int)frame_lo; W.nfw = (int)n_frames_win; W.patch_lo = (int)patch_lo; W.npw = (int)n_patches_win; W.shift = shift;
  W.T = (int32_t*)(w + L.win);
  W.binstart = W.T + (size_t)tiles * kWinBins;
  W.tot = W.binstart + kWinBins + 2;
  W.flag = W.tot + kWinBins;
  W.gid = W.flag + 1;
  // totals + flag = 0 (one small launch: hipMemsetAsync turns this 24 KB region into three fill kernels, 13 us of the frame start)
  W.flow = plan + P.flow;
  const bool with_flow = qi >= 0 && qj >= 0 && qi != qj && qi < (1ll << 30) && qj < (1ll << 30);
  W.qi = with_flow ? (int)qi : -1; W.qj = with_flow ? (int)qj : -1;
  if (!counters_cleared) hipLaunchKernelGGL(win_zero_kernel, dim3((kWinBins + 1 + 1023) / 1024), dim3(1024), 0, st, W.tot, kWinBins + 1);
  W.tmp_key = (uint32_t*)(w + L.keys_a); W.tmp_e = (int32_t*)(w + L.vals_in);
  W.perm_k = plan + P.perm_k; W.perm_p = plan + P.perm_p;
  W.ku = plan + P.ku; W.kx = plan + P.kx; W.patch_off = plan + P.patch_off; W.ix = plan + P.ix; W.jx = plan + P.jx;
  W.pu = plan + P.pu; W.pair_off = plan + P.pair_off; W.pair_ij = plan + P.pair_ij; W.counts = plan + P.counts;
  ReprojJob R = {r_poses, r_patches, r_intr, r_coords, r_P, 0};
  if (r_poses) {
    if (!r_patches || !r_intr || !r_coords || r_P <= 0) return DPVO_E_INVALID;
    const int64_t nb = cdiv64(E * r_P * r_P, 1024);
    R.nblk = (int)(nb < 1 ? 1 : (nb > 16384 ? 16384 : nb));
  }
  hipLaunchKernelGGL(win_hist_kernel, dim3(tiles + R.nblk), dim3(1024), 0, st, W, tiles, R);
  const int chunks_a = (int)cdiv64(W.npw, 1024), chunks_b = (int)cdiv64((int64_t)W.nfw * W.nfw, 1024);
  hipLaunchKernelGGL(win_scan_kernel, dim3(chunks_a + chunks_b), dim3(1024), 0, st, W, tiles, chunks_a);
  hipLaunchKernelGGL(win_scatter_kernel, dim3(tiles), dim3(1024), 0, st, W);
  hipLaunchKernelGGL(win_segsort_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, W);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" size_t dpvo_plan_wide_workspace_bytes(int64_t E, int64_t n_frames, int64_t n_patch_ids) {
  WideWs L;
  if (wide_layout(E, n_frames, n_patch_ids, &L) != 0) return 0;
  return L.total;
}

extern "C" int dpvo_plan_build_wide(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                                    size_t ws_bytes, int64_t n_frames, int64_t n_patch_ids, void* stream) {
  if (E < 0 || !plan || n_frames <= 0 || n_patch_ids <= 0) return DPVO_E_INVALID;
  WideWs L;
  if (wide_layout(E, n_frames, n_patch_ids, &L) != 0) return DPVO_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dpvo_plan_layout_t P;
  dpvo_plan_layout(E, &P);
  if (E == 0) {
    hipError_t e = hipMemsetAsync(plan + P.counts, 0, 4 * sizeof(int32_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(plan + P.flow, 0xff, 4 * sizeof(int32_t), st);       // (qi = qj = -1: no flow-test list)
    return e == hipSuccess ? DPVO_OK : (int)e;
  }
  if (!ii || !jj || !kk || !ws) return DPVO_E_INVALID;
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  char* w = (char*)ws;
  WideArgs W;
  W.ii = ii; W.jj = jj; W.kk = kk; W.E = E;
  W.nf = (int)n_frames; W.np = (int)n_patch_ids;
  W.nbA = L.nbA; W.nbB = L.nbB; W.chunksA = L.chunksA; W.chunksB = L.chunksB;
  W.tot = (int32_t*)(w + L.tot); W.flag = W.tot + (size_t)L.nbA + L.nbB;
  W.csum = (int32_t*)(w + L.csum);
  W.startA = (int32_t*)(w + L.startA); W.startB = (int32_t*)(w + L.startB);
  W.gidA = (int32_t*)(w + L.gidA); W.gidB = (int32_t*)(w + L.gidB);
  W.tmpA_j = (uint32_t*)(w + L.tmpA_j); W.tmpA_e = (int32_t*)(w + L.tmpA_e); W.tmpB_e = (int32_t*)(w + L.tmpB_e);
  W.perm_k = plan + P.perm_k; W.perm_p = plan + P.perm_p;
  W.ku = plan + P.ku; W.kx = plan + P.kx; W.patch_off = plan + P.patch_off; W.ix = plan + P.ix; W.jx = plan + P.jx;
  W.pu = plan + P.pu; W.pair_off = plan + P.pair_off; W.pair_ij = plan + P.pair_ij; W.counts = plan + P.counts;
  W.flow = plan + P.flow;
  const int64_t nz = (int64_t)L.nbA + L.nbB + 1;
  hipLaunchKernelGGL(win_zero_kernel, dim3((unsigned)cdiv64(nz, 1024)), dim3(1024), 0, st, W.tot, (int)nz);
  hipLaunchKernelGGL(wide_hist_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, W);
  hipLaunchKernelGGL(wide_chunk_kernel, dim3((unsigned)(L.chunksA + L.chunksB)), dim3(1024), 0, st, W);
  hipLaunchKernelGGL(wide_scan_kernel, dim3((unsigned)(L.chunksA + L.chunksB)), dim3(1024), 0, st, W);
  hipLaunchKernelGGL(wide_scatter_kernel, dim3((unsigned)cdiv64(E, 256)), dim3(256), 0, st, W);
  hipLaunchKernelGGL(wide_rank_kernel, dim3((unsigned)cdiv64(2 * E, 256)), dim3(256), 0, st, W);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_plan_build(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                               void* ws, size_t ws_bytes, void* stream) {
  return dpvo_plan_build_ranged(ii, jj, kk, E, plan, ws, ws_bytes, 0, 0, stream);
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
  uint64_t* keys_a = (uint64_t*)(w + L.keys_a);
  uint64_t* keys_out = (uint64_t*)(w + L.keys_out);
  int32_t* vals_in = (int32_t*)(w + L.vals_in);
  hipLaunchKernelGGL(make_keys_kernel<uint64_t>, dim3(grid_for(E)), dim3(256), 0, st, (const int64_t*)nullptr, jj, kk, keys_a,
                     (uint64_t*)nullptr, vals_in, E, kKeyShift, (int32_t*)nullptr);
  size_t tb = L.temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs((void*)(w + L.temp), tb, keys_a, keys_out, vals_in, perm, (size_t)E, 0, 64, st);
  if (e != hipSuccess) return (int)e;
  int32_t* blk_count = (int32_t*)(w + L.win);
  hipLaunchKernelGGL((group_count_kernel<uint64_t, true>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const uint64_t*)keys_out, blk_count, E,
                     kKeyShift);
  hipLaunchKernelGGL((group_kernel<uint64_t, true>), dim3((unsigned)cdiv64(E, 1024)), dim3(1024), 0, st, (const uint64_t*)keys_out, perm, ku, kx, off,
                     ix32, jx32, off + E + 1 /*count scratch*/, (int32_t*)nullptr, E, kKeyShift, blk_count);
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(E)), dim3(256), 0, st, ix32, jx32, ix, jx, E);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
