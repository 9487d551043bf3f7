// update_pm2.hip -- the update operator (reference dpvo/net.py:74-92, dpvo/blocks.py:15-48, under autocast dpvo/dpvo.py:332)
// in FOUR launches over 64-row tiles made of WHOLE PATCHES, f32 hidden state in registers from the first layer to the last.
//
// Why (round 3): the seven-launch operator (update_fused.hip) moves 34 KB per edge through memory (measured,
// profiles/r03_a_update_pmc_mem.txt) -- 19 KB of it the f32 hidden state, written and re-read at every kernel boundary --
// and cuts the operator at every exchange between edges.  Two of the three exchanges are local to a PATCH:
//   * c1 / c2 read the previous / next edge of the same patch in target-frame order (fastba.neighbors, ba.cpp:59-97);
//   * agg_kk is a softmax-weighted sum over the edges of a patch (net.py:87).
// With the edges taken in the plan's per-patch order and tiles that hold whole patches both are row shifts / row-segment
// reductions inside the LDS tile.  Round 2's version of this (update_pm.hip, 96-row tiles, greedy consecutive packing) lost to the
// seven launches because 144 state + 144 accumulator registers left only a 4-deep weight ring and 58 spills, and because
// consecutive patches filled the tiles to 85 %.  Here:
//   * 64-row tiles: 96 state + 96 accumulator registers, the full 6-deep weight ring, unrolled k-loops, no spills;
//   * tiles are packed by patch SIZE, not by position (first-fit decreasing over the size histogram: 25 + 25 + 14, 25 + 24 + 15,
//     ... -> 98 % full at the default.yaml steady state); a tile is a list of up to 6 patches, its rows are gathered through the
//     plan's permutation, so which patches share a tile is free;
//   * the state never leaves the registers inside a kernel (the gated residuals included).
//
//     pm2_prepare   size histogram, packing patterns (one thread, O(#patterns)), tile table (parallel)
//     KA2           corr MLP + norm, c1, c2, agg_kk (f, g, segment softmax-sum, h), f | g of agg_ij   -> state image, f | g rows
//     SA            agg_ij softmax-sum over the frame-pair groups (rows addressed through the edge -> tile-row map)
//     KB2           agg_ij.h, 2 x (LayerNorm, gated residual), heads                                  -> net, delta, weight, target
// Memory per edge: corr 1792 + net 1536 + inp 768 in, image 1536 + f | g 1536 out (KA2); 1536 in (SA); image 1536 + y 768 in,
// net 1536 + 16 out (KB2) = ~12.5 KB.  Precision contract and rounding points: exactly those of update_fused.hip.
#include "update_fused_dev.h"
#ifdef PM2_IN_CMP
#include "../../include/dpvo_hip_cmp.h"
#endif

namespace {
namespace fu {

constexpr int PM2_RT = 2, PM2_R = 64, PM2_DW = 6, PM2_SEG = 6;
struct Tile2 { int32_t nrows, nseg; int32_t p[PM2_SEG]; };      // p[k]: index of the k-th patch of the tile in the plan's patch order
enum { H2_NTILES = 0, H2_ERR = 1, H2_INTS = 8 };

// ------------------------------------------------------------------------------------------------ packing
// One workgroup.  (a) histogram of the patch sizes (1 .. 64 edges) and the patch ids listed by size (order within a size class is
// arbitrary -- which patches share a tile does not change a single bit of the result: every row-wise operation is independent of
// the other rows of its tile, and the two group reductions follow the plan's order); (b) thread 0 turns the histogram into
// packing PATTERNS by first-fit decreasing on the counts -- "largest size that still fits" is one count-leading-zeros on a 64-bit
// mask of the non-empty classes, a pattern is applied min(count / uses) times at once, so the loop runs once per distinct pattern
// (~20), not once per patch; (c) every tile reads its patches off the class lists.
struct Pat { int32_t t0, m, nseg; int32_t sz[PM2_SEG], base[PM2_SEG], stride[PM2_SEG]; };
constexpr int PM2_MAXPAT = 192, PM2_MAXPATCH = 4096;

__global__ __launch_bounds__(1024) void pm2_prepare_kernel(const int32_t* __restrict__ patch_off, const int32_t* __restrict__ counts,
                                                           Tile2* __restrict__ tiles, int32_t* __restrict__ hdr, int max_tiles) {
  __shared__ int hist[66], cls_off[66];
  __shared__ int wcnt[16][65];                     // per wave, per size class: patches of that class in the wave's slice
  __shared__ int32_t lists[PM2_MAXPATCH];
  __shared__ Pat pats[PM2_MAXPAT];
  __shared__ int npat, ntile, err;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int np = counts[0];
  for (int i = tid; i < 16 * 65; i += 1024) (&wcnt[0][0])[i] = 0;
  if (tid == 0) { npat = 0; ntile = 0; err = (np > PM2_MAXPATCH || np < 0) ? 1 : 0; }
  __syncthreads();
  const int npc = np > PM2_MAXPATCH ? PM2_MAXPATCH : np;
  // stable counting sort of the patches by size (ties in patch order: the packing, hence the tile a patch lands in, is the same
  // on every run).  Wave w owns the slice [w C, (w + 1) C); a chunk of 64 patches has one or two distinct sizes (patches of a
  // frame have equal sizes), so ranking by "ballot of my size" loops once or twice per chunk.
  const int C = ((npc + 15) / 16 + 63) / 64 * 64;
  const int lo = wv * C, hi = (lo + C < npc) ? lo + C : npc;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      __syncthreads();
      if (tid <= 64) {                                // class totals, class offsets, per-wave bases (exclusive over the waves)
        int tot = 0;
        for (int w = 0; w < 16; ++w) { const int c = wcnt[w][tid > 64 ? 0 : tid]; wcnt[w][tid] = tot; tot += c; }
        hist[tid] = tot;
      }
      __syncthreads();
      if (tid == 0) { int o = 0; for (int s = 0; s <= 64; ++s) { cls_off[s] = o; o += hist[s]; } cls_off[65] = o; }
      __syncthreads();
    }
    for (int p0 = lo; p0 < hi; p0 += 64) {
      const int p = p0 + lane;
      int s = p < hi ? patch_off[p + 1] - patch_off[p] : -1;
      if (p < hi && (s < 1 || s > PM2_R)) { err = 1; s = -1; }
      unsigned long long todo = __ballot(s >= 1);
      while (todo) {
        const int first = __builtin_ctzll(todo);
        const int s0 = __shfl(s, first);
        const unsigned long long same = __ballot(s == s0);
        if (s == s0) {
          const int rank = __popcll(same & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
          if (pass == 1) lists[cls_off[s0] + wcnt[wv][s0] + rank] = p;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == first) wcnt[wv][s0] += __popcll(same);
        __builtin_amdgcn_wave_barrier();
        todo &= ~same;
      }
    }
  }
  __syncthreads();
  // pattern search by wave 0, lane l <-> size class l + 1, everything in registers (a one-thread version with its tables in
  // scratch / LDS took 60 us): first-fit decreasing over the classes that still have patches; a pattern is used for as many
  // tiles as its scarcest class allows.
  if (wv == 0 && !err) {
    int cnt = hist[lane + 1], ptr = 0;
    int nt = 0, np_ = 0;
    bool fail = false;
    while (np_ < PM2_MAXPAT) {
      if (!__ballot(cnt > 0)) break;
      int cap = PM2_R, k = 0, used = 0;
      int sz[PM2_SEG], ub[PM2_SEG];
#pragma unroll
      for (int q = 0; q < PM2_SEG; ++q) {
        sz[q] = 0; ub[q] = 0;
        unsigned long long fit = __ballot(cnt - used > 0);
        if (cap < 64) fit &= (1ull << cap) - 1ull;
        if (fit) {
          const int sp = 64 - __builtin_clzll(fit);
          sz[q] = sp; ub[q] = __shfl(used, sp - 1);
          used += (lane == sp - 1);
          cap -= sp; ++k;
        }
      }
      int m = used > 0 ? cnt / used : 0x7fffffff;
#pragma unroll
      for (int o = 32; o; o >>= 1) { const int v = __shfl_xor(m, o); m = v < m ? v : m; }
      if (k == 0 || m <= 0) { fail = true; break; }
      Pat P;
      P.t0 = nt; P.m = m; P.nseg = k;
#pragma unroll
      for (int q = 0; q < PM2_SEG; ++q) {
        const int sl = sz[q] > 0 ? sz[q] - 1 : 0;
        P.sz[q] = sz[q]; P.base[q] = __shfl(ptr, sl) + ub[q]; P.stride[q] = __shfl(used, sl);
      }
      if (lane == 0) pats[np_] = P;
      cnt -= m * used; ptr += m * used;
      nt += m; ++np_;
    }
    const bool left = __ballot(cnt > 0) != 0;                   // (more patterns than the table holds: never with real graphs)
    if (lane == 0) { if (fail || left) err = 1; npat = np_; ntile = nt; }
  }
  __syncthreads();
  const int nt = ntile < max_tiles ? ntile : max_tiles;
  for (int t = tid; t < nt; t += 1024) {
    int pi = 0;
    while (pi + 1 < npat && pats[pi + 1].t0 <= t) ++pi;
    const Pat& P = pats[pi];
    const int r = t - P.t0;
    Tile2 T;
    T.nseg = P.nseg; T.nrows = 0;
#pragma unroll
    for (int k = 0; k < PM2_SEG; ++k) {
      T.p[k] = 0;
      if (k < P.nseg) { T.p[k] = lists[cls_off[P.sz[k]] + P.base[k] + r * P.stride[k]]; T.nrows += P.sz[k]; }
    }
    tiles[t] = T;
  }
  if (tid == 0) { hdr[H2_NTILES] = nt; hdr[H2_ERR] = (err || ntile > max_tiles) ? 1 : 0; }
}

// SoftAgg over groups whose member rows are addressed through a map (row = pos[perm[p]])
__global__ __launch_bounds__(384) void softagg_pos_kernel(const _Float16* __restrict__ fg, int64_t ldfg,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ pos,
                                                          const int32_t* __restrict__ off, const int32_t* __restrict__ n_groups,
                                                          _Float16* __restrict__ y) {
  __shared__ float part[4][3][384];
  const int ng = *n_groups;
  const int q = threadIdx.x / 96, cq = threadIdx.x - 96 * q;
  for (int g = blockIdx.x; g < ng; g += gridDim.x) {
    const int b = off[g], e = off[g + 1];
    float m[4], s[4], a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -INFINITY; s[r] = 0.f; a[r] = 0.f; }
    for (int p = b + q; p < e; p += 4) {
      const _Float16* rowp = fg + (int64_t)pos[perm[p]] * ldfg + 4 * cq;
      const h4 fx = *reinterpret_cast<const h4*>(rowp), gx = *reinterpret_cast<const h4*>(rowp + D);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g_ = (float)gx[r];
        const float mn = fmaxf(m[r], g_);
        const float sc = __expf(m[r] - mn), wv = __expf(g_ - mn);
        s[r] = s[r] * sc + wv;
        a[r] = a[r] * sc + wv * (float)fx[r];
        m[r] = mn;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { part[q][0][4 * cq + r] = m[r]; part[q][1][4 * cq + r] = s[r]; part[q][2][4 * cq + r] = a[r]; }
    __syncthreads();
    {
      const int c = threadIdx.x;
      float M = part[0][0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k) M = fmaxf(M, part[k][0][c]);
      float S = 0.f, A = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float mk = part[k][0][c];
        const float sc = (mk == -INFINITY) ? 0.f : __expf(mk - M);
        S += part[k][1][c] * sc;
        A += part[k][2][c] * sc;
      }
      y[(int64_t)g * D + c] = (_Float16)(A / S);
    }
    __syncthreads();
  }
}

// k-loop of a 384-wide layer, fully unrolled, with a B-fragment base per row tile (a lane may read a shifted row or the zero row)
template <int RT, int DW>
__device__ __forceinline__ void gemm_ptrs(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* __restrict__ wp, const char* const (&bl)[RT]) {
  h8 bf[2][RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl[r]);
#pragma unroll
  for (int s = 0; s < KS384; ++s) {
    if (s + 1 < KS384) {
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[(s + 1) & 1][r] = *reinterpret_cast<const h8*>(bl[r] + (s + 1) * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[s & 1][r], acc[r][t], 0, 0, 0);
    if (s + DW < KS384) {
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// v = f32(f16(v)) and x += v
template <int RT>
__device__ __forceinline__ void residual_add(f16v (&x)[RT][3], const f16v (&v)[RT][3]) {
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 16; ++k) x[r][t][k] += (float)(_Float16)v[r][t][k];
}

// the LDS tile -> rows [row0, row0 + nrows) of a P-order f16 matrix with leading dimension ld (halves)
template <int RT>
__device__ __forceinline__ void tile_to_rows(const char* act, _Float16* __restrict__ dst, int64_t ld, int64_t row0, int nrows, int tid) {
  constexpr int N = RT * 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int idx = tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
    const h8 v = *reinterpret_cast<const h8*>(act + row * PITCH + ch * 16);
    if (row < nrows) *reinterpret_cast<h8*>(dst + (row0 + row) * ld + ch * 8) = v;
  }
}

// row i of a tile -> edge id (-1: none) and (segment | first-in-patch << 8 | last-in-patch << 9)
__device__ __forceinline__ void tile_row(const Tile2& T, const int32_t* __restrict__ patch_off, const int32_t* __restrict__ perm_k,
                                         const int32_t* __restrict__ ix, const int32_t* __restrict__ jx, int i, int& e, int& f) {
  e = -1; f = (1 << 8) | (1 << 9);
  int start = 0;
#pragma unroll
  for (int k = 0; k < PM2_SEG; ++k) {
    if (k < T.nseg) {
      const int b = patch_off[T.p[k]], n = patch_off[T.p[k] + 1] - b;
      if (i >= start && i < start + n) {
        e = perm_k[b + (i - start)];
        f = k | ((ix[e] < 0) << 8) | ((jx[e] < 0) << 9);
      }
      start += n;
    }
  }
}

struct PA2 {
  Lin c0, c2, c5, c1a, c1b, c2a, c2b, fk, gk, hk, fi, gi;
  const float *cln_g, *cln_b, *norm_g, *norm_b;
  const _Float16* corr; int64_t ld_corr;
  const float* net; const int64_t* net_rows; int64_t n_kept;
  const _Float16* inp; const int64_t* inp_rows; int64_t inp_mod;
  const int32_t *perm_k, *patch_off, *ix, *jx;
  const Tile2* tiles; const int32_t* hdr;
  float* img; _Float16* fg; int32_t* pos;
  int64_t E;
};

struct GeoA2 {
  static constexpr int R = PM2_R;
  static constexpr int T1 = 0, T2 = R * PITCH, RED = 2 * R * PITCH, ZERO = RED + R * 64, META = ZERO + PITCH;
  static constexpr int LNP = META + R * 8 + 16;                  // two LayerNorms: 2 x [gamma | beta] f32
  static constexpr int LDS_BYTES = LNP + 2 * 2 * D * 4;
  static_assert(LDS_BYTES <= 163840, "LDS");
};

__global__ __launch_bounds__(256, 1) void ka2_kernel(const PA2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA2;
  constexpr int R = G::R, RT = PM2_RT, DW = PM2_DW;
  const Lane l = lane_of();
  const int tile = blockIdx.x;
  if (tile >= p.hdr[H2_NTILES]) return;
  char* t1 = smem + G::T1;
  char* t2 = smem + G::T2;
  float* red = reinterpret_cast<float*>(smem + G::RED);
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);          // edge id of row i (-1: no row)
  int32_t* meta_f = meta_e + R;                                            // segment | first << 8 | last << 9
  for (int i = l.tid; i < PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(smem + G::ZERO)[i] = 0u;
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.cln_g[i]; lnp[D + i] = p.cln_b[i]; lnp[2 * D + i] = p.norm_g[i]; lnp[3 * D + i] = p.norm_b[i];
  }
  const Tile2 T = p.tiles[tile];
  if (l.tid < R) {
    int e, f;
    tile_row(T, p.patch_off, p.perm_k, p.ix, p.jx, l.tid, e, f);
    meta_e[l.tid] = e; meta_f[l.tid] = f;
    if (e >= 0) p.pos[e] = tile * R + l.tid;
  }
  __syncthreads();

  f16v acc[RT][3], x[RT][3];
  h8 wf[DW][3];
  Bias bias;
  const char* al = t1 + l.n * PITCH + 16 * l.h;          // this lane's row of row tile 0 (write side / unshifted read side)
  const char* bl0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bl0[r] = al + r * 32 * PITCH;
  const char* bl2[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bl2[r] = bl0[r] + G::T2;
  int eg[RT];                                              // edge id of this lane's rows (clamped to a valid one for loads)
#pragma unroll
  for (int r = 0; r < RT; ++r) { const int e = meta_e[r * 32 + l.n]; eg[r] = e < 0 ? 0 : e; }

  FU_T(5, 0);
  // ---- Linear(882 -> 384) + ReLU: K = 896 streamed from the rows corr[e] in 7 chunks of 128 through two LDS stages
  {
    const h8* wp = w_base(p.c0.w, 56, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.c0.b, l);
    constexpr int NS = RT * 2;
    const _Float16* src[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
      const int e = meta_e[row];
      src[i] = p.corr + (int64_t)(e < 0 ? 0 : e) * p.ld_corr + ch * 8;
    }
    h8 st[2][NS];
    auto load = [&](h8 (&d)[NS], int kc) {
#pragma unroll
      for (int i = 0; i < NS; ++i) d[i] = *reinterpret_cast<const h8*>(src[i] + kc * KCH);
    };
    auto store = [&](const h8 (&d)[NS], int b) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int idx = l.tid + 256 * i, row = idx >> 4, ch = idx & 15;
        *reinterpret_cast<h8*>(smem + b * R * CPITCH + row * CPITCH + ch * 16) = d[i];
      }
    };
    load(st[0], 0);
    load(st[1], 1);
    acc_init<RT>(acc, bias);
    store(st[0], 0);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 7; ++kc) {
      if (kc + 2 < 7) load(st[kc & 1], kc + 2);
      const char* bl = smem + (kc & 1) * R * CPITCH + l.n * CPITCH + 16 * l.h;
      h8 bf[2][RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int s = kc * 8 + ks;
        if (ks + 1 < 8) {
#pragma unroll
          for (int r = 0; r < RT; ++r) bf[(ks + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * CPITCH + (ks + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[ks & 1][r], acc[r][t], 0, 0, 0);
        if (s + DW < 56) {
#pragma unroll
          for (int t = 0; t < 3; ++t) wf[s % DW][t] = wp[((s + DW) * 3 + t) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kc + 1 < 7) store(st[(kc + 1) & 1], (kc + 1) & 1);
      __syncthreads();
    }
  }
  FU_T(5, 1);
  const h8* wp = w_base(p.c2.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.c2.b, l);
  to_lds<RT, 1>(acc, const_cast<char*>(al), l);
  __syncthreads();
  // ---- Linear, LayerNorm, ReLU
  acc_init<RT>(acc, bias);
  gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
  wp = w_base(p.c5.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.c5.b, l);
  round_f16<RT>(acc);
  layernorm_tile_lds<RT>(acc, red, lnp, l);
  to_lds<RT, 1>(acc, const_cast<char*>(al), l);
  __syncthreads();
  FU_T(5, 2);
  // ---- Linear; x = LayerNorm(net + inp + .)      (net.py:77-78)
  {
    acc_init<RT>(x, bias);
    gemm_ptrs<RT, DW>(x, wf, wp, bl0);
    wp = w_base(p.c1a.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.c1a.b, l);
    round_f16<RT>(x);
    // rows net[e] (f32) and inp[kk[e] % mod] (f16) in feature order, one row tile at a time
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int64_t ir = eg[r];
      if (p.inp_rows) { ir = p.inp_rows[eg[r]]; if (p.inp_mod > 0) ir %= p.inp_mod; }
      const int64_t gs = !p.net_rows ? (int64_t)eg[r] : (eg[r] < p.n_kept ? p.net_rows[eg[r]] : -1);
      f4 nv[3][4];
      h4 iv[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
          nv[t][j] = gs >= 0 ? *reinterpret_cast<const f4*>(p.net + gs * D + f) : (f4)0.f;
          iv[t][j] = *reinterpret_cast<const h4*>(p.inp + ir * D + f);
        }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) x[r][t][4 * j + q] = (nv[t][j][q] + (float)iv[t][j][q]) + x[r][t][4 * j + q];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  layernorm_tile_lds<RT>(x, red, lnp + 2 * D, l);
  to_lds<RT, 0>(x, const_cast<char*>(al), l);
  __syncthreads();
  FU_T(5, 3);
  // ---- x += c1(previous edge of the patch); x += c2(next edge of the patch)              (net.py:80-85)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const char* bls[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int f = meta_f[r * 32 + l.n];
      const bool none = k == 0 ? ((f >> 8) & 1) : ((f >> 9) & 1);
      bls[r] = none ? smem + G::ZERO + 16 * l.h : bl0[r] + (k == 0 ? -PITCH : PITCH);
    }
    acc_init<RT>(acc, bias);
    gemm_ptrs<RT, DW>(acc, wf, wp, bls);
    wp = w_base(k == 0 ? p.c1b.w : p.c2b.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, k == 0 ? p.c1b.b : p.c2b.b, l);
    to_lds<RT, 1>(acc, const_cast<char*>(al) + G::T2, l);
    __syncthreads();
    acc_init<RT>(acc, bias);
    gemm_ptrs<RT, DW>(acc, wf, wp, bl2);
    wp = w_base(k == 0 ? p.c2a.w : p.fk.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, k == 0 ? p.c2a.b : p.fk.b, l);
    residual_add<RT>(x, acc);
    to_lds<RT, 0>(x, const_cast<char*>(al), l);          // (T1 was last read before the barrier above)
    __syncthreads();
  }
  // ---- agg_kk: f -> T2, g -> T1, segmented softmax-sum over the rows of every patch, y -> T1 rows [0, nseg)   (blocks.py:40-43)
  FU_T(5, 4);
  acc_init<RT>(acc, bias);
  gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
  wp = w_base(p.gk.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.gk.b, l);
  to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);
  acc_init<RT>(acc, bias);
  gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
  wp = w_base(p.hk.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.hk.b, l);
  __syncthreads();
  to_lds<RT, 0>(acc, const_cast<char*>(al), l);
  __syncthreads();
  if (l.tid < 192) {
  
    // channels 2 tid, 2 tid + 1 (P order; f and g of a channel sit at the same position of T2 / T1)
    const char* gp = t1 + 4 * l.tid;
    const char* fp = t2 + 4 * l.tid;
    float m0 = -INFINITY, m1 = -INFINITY, s0 = 0.f, s1 = 0.f, a0 = 0.f, a1 = 0.f;
    int cur = 0;
    for (int i = 0; i < T.nrows; ++i) {
      const int pl = meta_f[i] & 0xff;
      if (pl != cur) {
        h2 o; o[0] = (_Float16)(a0 / s0); o[1] = (_Float16)(a1 / s1);
        *reinterpret_cast<h2*>(t1 + cur * PITCH + 4 * l.tid) = o;      // row `cur` of g is already consumed (cur <= first row of the patch)
        cur = pl; m0 = m1 = -INFINITY; s0 = s1 = a0 = a1 = 0.f;
      }
      const h2 gv = *reinterpret_cast<const h2*>(gp + i * PITCH), fv = *reinterpret_cast<const h2*>(fp + i * PITCH);
      {
        const float g_ = (float)gv[0], mn = fmaxf(m0, g_), sc = __expf(m0 - mn), wv = __expf(g_ - mn);
        s0 = s0 * sc + wv; a0 = a0 * sc + wv * (float)fv[0]; m0 = mn;
      }
      {
        const float g_ = (float)gv[1], mn = fmaxf(m1, g_), sc = __expf(m1 - mn), wv = __expf(g_ - mn);
        s1 = s1 * sc + wv; a1 = a1 * sc + wv * (float)fv[1]; m1 = mn;
      }
    }
    h2 o; o[0] = (_Float16)(a0 / s0); o[1] = (_Float16)(a1 / s1);
    *reinterpret_cast<h2*>(t1 + cur * PITCH + 4 * l.tid) = o;
  }
  __syncthreads();
  FU_T(5, 6);
  // ---- h on the (<= 6) patch rows, expanded back to the edges: x += h(y)[patch of the row]       (blocks.py:45-48)
  {
    f16v hy[1][3];
    const char* blh[1] = {al};
    acc_init<1>(hy, bias);
    gemm_ptrs<1, DW>(hy, wf, wp, blh);
    wp = w_base(p.fi.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.fi.b, l);
    to_lds<1, 0>(hy, const_cast<char*>(al) + G::T2, l);               // (T2 = f: consumed before the barrier above)
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int pl = meta_f[r * 32 + l.n] & 0xff;
    const char* hp = t2 + pl * PITCH + 16 * l.h;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const h8 v = *reinterpret_cast<const h8*>(hp + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[r][t][8 * c + i] += (float)v[i];
      }
  }
  to_lds<RT, 0>(x, const_cast<char*>(al), l);            // (T1 = y: consumed by the h GEMM before the barrier above)
  img_store<RT>(x, img_ptr<RT>(p.img, tile, l));
  __syncthreads();
  FU_T(5, 7);
  // ---- f | g of agg_ij for every edge row, stored at the tile-major position
  acc_init<RT>(acc, bias);
  gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
  wp = w_base(p.gi.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.gi.b, l);
  to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);               // (T2 = h(y): consumed before the barrier above)
  acc_init<RT>(acc, bias);
  gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
  __syncthreads();
  to_lds<RT, 0>(acc, const_cast<char*>(al), l);
  __syncthreads();
  tile_to_rows<RT>(t2, p.fg, 768, (int64_t)tile * R, T.nrows, l.tid);
  tile_to_rows<RT>(t1, p.fg + D, 768, (int64_t)tile * R, T.nrows, l.tid);
  FU_T(5, 8);
}

struct PB2 {
  Lin h;
  Lin gate[2], res0[2], res2[2];
  const float *ln_g[2], *ln_b[2];
  const _Float16 *d_w, *d_b, *w_w, *w_b;
  const _Float16* y; const int32_t *pu, *perm_k, *patch_off, *ix, *jx;
  const Tile2* tiles; const int32_t* hdr;
  const float* img;
  const float* coords; int pp;
  float *net_out, *delta, *weight, *target;
  int64_t E;
};

__global__ __launch_bounds__(256, 1) void kb2_kernel(const PB2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA2;
  constexpr int R = G::R, RT = PM2_RT, DW = PM2_DW;
  const Lane l = lane_of();
  const int tile = blockIdx.x;
  if (tile >= p.hdr[H2_NTILES]) return;
  char* t1 = smem + G::T1;
  float* red = reinterpret_cast<float*>(smem + G::RED);          // LayerNorm partials, later the heads' [R][4 waves][4] sums (R x 64 B)
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.ln_g[0][i]; lnp[D + i] = p.ln_b[0][i]; lnp[2 * D + i] = p.ln_g[1][i]; lnp[3 * D + i] = p.ln_b[1][i];
  }
  char* al = t1 + l.n * PITCH + 16 * l.h;
  char* gl = al + G::T2;
  const char* bl0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bl0[r] = al + r * 32 * PITCH;
  const Tile2 T = p.tiles[tile];
  if (l.tid < R) {
    int e, f;
    tile_row(T, p.patch_off, p.perm_k, p.ix, p.jx, l.tid, e, f);
    meta_e[l.tid] = e;
  }
  __syncthreads();

  FU_T(6, 0);
  f16v x[RT][3];
  h8 wf[DW][3];
  Bias bias;
  const h8* wp = w_base(p.h.w, KS384, l);
  w_preload<DW>(wf, wp);
  bias_load(bias, p.h.b, l);
  const float* ip = img_ptr<RT>(const_cast<float*>(p.img), tile, l);
  {
    // rows y[pu[e]] of the group table -> T1
    constexpr int N = RT * 6;
    h8 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = l.tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
      const int e = meta_e[row];
      v[i] = e >= 0 ? *reinterpret_cast<const h8*>(p.y + (int64_t)p.pu[e] * D + ch * 8) : (h8)(_Float16)0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int idx = l.tid + 256 * i, row = idx / 48, ch = idx - 48 * row;
      *reinterpret_cast<h8*>(t1 + row * PITCH + ch * 16) = v[i];
    }
  }
  {
    Img<RT> im;
    img_load<RT>(im, ip);                 // lands under the first GEMM
    __syncthreads();
    acc_init<RT>(x, bias);
    gemm_ptrs<RT, DW>(x, wf, wp, bl0);
    round_f16<RT>(x);
    img_add<RT>(x, im);
  }
  FU_T(6, 1);
  // ---- 2 x (LayerNorm, x + gate(x) * res(x)): the f32 state stays in its 96 registers                 (net.py:90, blocks.py:15-29)
#pragma unroll
  for (int Gi = 0; Gi < 2; ++Gi) {
    layernorm_tile_lds<RT>(x, red, lnp + 2 * D * Gi, l);
    wp = w_base(p.gate[Gi].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gate[Gi].b, l);
    to_lds<RT, 0>(x, al, l);
    __syncthreads();
    f16v acc[RT][3];
    acc_init<RT>(acc, bias);
    gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.res0[Gi].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.res0[Gi].b, l);
    to_lds<RT, 2>(acc, gl, l);                                    // gate = sigmoid(.), parked by its own lane
    acc_init<RT>(acc, bias);
    gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.res2[Gi].w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.res2[Gi].b, l);
    __syncthreads();
    to_lds<RT, 1>(acc, al, l);
    __syncthreads();
    acc_init<RT>(acc, bias);
    gemm_ptrs<RT, DW>(acc, wf, wp, bl0);
    // x = x + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int k = 8 * c + i;
            const _Float16 rv = (_Float16)acc[r][t][k];
            x[r][t][k] = x[r][t][k] + (float)(_Float16)(gt[i] * rv);
          }
        }
    if (Gi == 0) __syncthreads();          // (every wave has read the tile of res2's GEMM before the next LayerNorm's to_lds rewrites it)
      FU_T(6, 2 + Gi);
  }
  // ---- hidden state out (rows net_out[e], feature order) and the heads (one MFMA chain per wave over its own 96 features)
  int eg[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) eg[r] = meta_e[r * 32 + l.n];
  f16v hacc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) hacc[r][k] = 0.f;
  {
    const int m = l.n;                              // A row = output row of the 32-row MFMA tile: d0, d1, w0, w1, then zeros
    const _Float16* wrow = m == 0 ? p.d_w : m == 1 ? p.d_w + D : m == 2 ? p.w_w : p.w_w + D;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int base = 96 * l.w + 32 * t + 16 * c + 4 * l.h;      // features base + {0..3} and base + 8 + {0..3}
        h8 af = (h8)(_Float16)0;
        if (m < 4) {
          const h4 lo = *reinterpret_cast<const h4*>(wrow + base), hi = *reinterpret_cast<const h4*>(wrow + base + 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) { af[i] = lo[i]; af[4 + i] = hi[i]; }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          h8 bfr;
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float v = x[r][t][8 * c + i]; bfr[i] = (_Float16)(v > 0.f ? v : 0.f); }
          hacc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bfr, hacc[r], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f4 o4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = x[r][t][4 * j + q];
        if (eg[r] >= 0) *reinterpret_cast<f4*>(p.net_out + (int64_t)eg[r] * D + f) = o4;
      }
    }
  __syncthreads();                                  // (the LayerNorm partials in `red` are dead)
#pragma unroll
  for (int r = 0; r < RT; ++r)
    if (l.h == 0) *reinterpret_cast<f4*>(red + ((r * 32 + l.n) * 4 + l.w) * 4) = (f4){hacc[r][0], hacc[r][1], hacc[r][2], hacc[r][3]};
  __syncthreads();
  if (l.tid < R) {
    const int e = meta_e[l.tid];
    if (e >= 0) {
      f4 s = *reinterpret_cast<const f4*>(red + (l.tid * 4 + 0) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f4 q = *reinterpret_cast<const f4*>(red + (l.tid * 4 + w) * 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) s[o] += q[o];
      }
      const float d0 = (float)(_Float16)(s[0] + (float)p.d_b[0]), d1 = (float)(_Float16)(s[1] + (float)p.d_b[1]);
      const _Float16 h0 = (_Float16)(s[2] + (float)p.w_b[0]), h1 = (_Float16)(s[3] + (float)p.w_b[1]);
      p.delta[2 * (int64_t)e + 0] = d0;
      p.delta[2 * (int64_t)e + 1] = d1;
      p.weight[2 * (int64_t)e + 0] = (float)(_Float16)sigm((float)h0);
      p.weight[2 * (int64_t)e + 1] = (float)(_Float16)sigm((float)h1);
      if (p.target) {
        p.target[2 * (int64_t)e + 0] = p.coords[((int64_t)e * 2 + 0) * p.pp + p.pp / 2] + d0;
        p.target[2 * (int64_t)e + 1] = p.coords[((int64_t)e * 2 + 1) * p.pp + p.pp / 2] + d1;
      }
    }
  }
  FU_T(6, 4);
}

struct WsPm2 { size_t img, fg, y, pos, tiles, hdr, total; int64_t max_tiles; };
inline void ws_layout_pm2(int64_t E, int64_t maxg, WsPm2* w) {
  const size_t e = (size_t)(E > 0 ? E : 1), g = (size_t)(maxg > 0 ? maxg : 1);
  // first-fit decreasing leaves at most one tile half empty: 2 ceil(E / 64) + 2 tiles always suffice
  const int64_t mt = 2 * cdiv64((int64_t)e, PM2_R) + 2;
  w->max_tiles = mt;
  size_t o = 0;
  w->img = o; o += al256((size_t)mt * PM2_R * D * 4);
  w->fg = o; o += al256((size_t)mt * PM2_R * 2 * D * 2);
  w->y = o; o += al256(g * D * 2);
  w->pos = o; o += al256(e * 4);
  w->tiles = o; o += al256((size_t)mt * sizeof(Tile2));
  w->hdr = o; o += al256(H2_INTS * 4);
  w->total = o;
}

}  // namespace fu
}  // namespace

extern "C" size_t dpvo_update_pm2_workspace_bytes(int64_t E, int64_t max_groups) {
  if (E < 0 || max_groups < 0) return 0;
  fu::WsPm2 w;
  fu::ws_layout_pm2(E, max_groups, &w);
  return w.total;
}

// Same contract as dpvo_update_forward_fused_rows (net_rows / n_kept: the deferred compaction of the hidden state; net_out may
// alias net: the first kernel reads every state row before the last one writes any).  status (device int32, may be NULL): 1 if
// the graph does not fit the packing (a patch with more than 64 edges, more than 4096 patches): outputs then unspecified, memory
// safe -- the caller falls back to dpvo_update_forward_fused_rows.
extern "C" int dpvo_update_forward_pm2(const dpvo_update_fused_params_t* p, const float* net, const int64_t* net_rows, int64_t n_kept,
                                       const void* inp, const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr,
                                       const int32_t* plan, int64_t n_patches_ub, int64_t n_pairs_ub, const float* coords, int P,
                                       float* net_out, float* delta, float* weight, float* target, int64_t E, void* ws,
                                       size_t ws_bytes, int32_t* status, void* stream) {
  using namespace fu;
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (net_rows && (n_kept < 0 || n_kept > E)) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8) || E >= (1ll << 30)) return DPVO_E_UNSUPPORTED;
  for (int i = 0; i < DPVO_UF_NLIN; ++i)
    if (!p->w[i] || !p->b[i]) return DPVO_E_INVALID;
  const int64_t maxg = n_patches_ub > n_pairs_ub ? n_patches_ub : n_pairs_ub;
  WsPm2 L;
  ws_layout_pm2(E, maxg, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* wsb = (char*)ws;
  float* img = (float*)(wsb + L.img);
  _Float16 *fg = (_Float16*)(wsb + L.fg), *y = (_Float16*)(wsb + L.y);
  int32_t* pos = (int32_t*)(wsb + L.pos);
  Tile2* tiles = (Tile2*)(wsb + L.tiles);
  int32_t* hdr = (int32_t*)(wsb + L.hdr);
  hipStream_t st = (hipStream_t)stream;
  auto lin = [&](int i) { return Lin{p->w[i], (const _Float16*)p->b[i]}; };
  // launch grids: the number of tiles is known on the device only; every tile but one is more than half full
  int64_t grid_tiles = cdiv64(E, PM2_R / 2 + 1) + 2;
  if (grid_tiles > L.max_tiles) grid_tiles = L.max_tiles;
  int rc;
#define FU(...) do { rc = (__VA_ARGS__); if (rc) return rc; } while (0)
  hipLaunchKernelGGL(pm2_prepare_kernel, dim3(1), dim3(1024), 0, st, plan + PL.patch_off, plan + PL.counts, tiles, hdr, (int)L.max_tiles);
  DPVO_LAUNCH_CHECK();
  {
    PA2 a;
    a.c0 = lin(DPVO_UF_C0); a.c2 = lin(DPVO_UF_C2); a.c5 = lin(DPVO_UF_C5);
    a.c1a = lin(DPVO_UF_C1_0); a.c1b = lin(DPVO_UF_C1_2); a.c2a = lin(DPVO_UF_C2N_0); a.c2b = lin(DPVO_UF_C2N_2);
    a.fk = lin(DPVO_UF_AKK_F); a.gk = lin(DPVO_UF_AKK_G); a.hk = lin(DPVO_UF_AKK_H);
    a.fi = lin(DPVO_UF_AIJ_F); a.gi = lin(DPVO_UF_AIJ_G);
    a.cln_g = p->ln_g[0]; a.cln_b = p->ln_b[0]; a.norm_g = p->ln_g[1]; a.norm_b = p->ln_b[1];
    a.corr = (const _Float16*)corr; a.ld_corr = ld_corr; a.net = net; a.net_rows = net_rows; a.n_kept = n_kept;
    a.inp = (const _Float16*)inp; a.inp_rows = inp_rows; a.inp_mod = inp_mod;
    a.perm_k = plan + PL.perm_k; a.patch_off = plan + PL.patch_off; a.ix = plan + PL.ix; a.jx = plan + PL.jx;
    a.tiles = tiles; a.hdr = hdr; a.img = img; a.fg = fg; a.pos = pos; a.E = E;
    FU(launch<ka2_kernel>(grid_tiles, GeoA2::LDS_BYTES, a, st));
  }
  {
    int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
    const unsigned grid = (unsigned)(ngp < 8192 ? ngp : 8192);
    hipLaunchKernelGGL(softagg_pos_kernel, dim3(grid), dim3(384), 0, st, (const _Float16*)fg, (int64_t)768, plan + PL.perm_p,
                       (const int32_t*)pos, plan + PL.pair_off, plan + PL.counts + 1, y);
    DPVO_LAUNCH_CHECK();
  }
  {
    PB2 a;
    a.h = lin(DPVO_UF_AIJ_H);
    a.gate[0] = lin(DPVO_UF_G0_GATE); a.res0[0] = lin(DPVO_UF_G0_RES0); a.res2[0] = lin(DPVO_UF_G0_RES2);
    a.gate[1] = lin(DPVO_UF_G1_GATE); a.res0[1] = lin(DPVO_UF_G1_RES0); a.res2[1] = lin(DPVO_UF_G1_RES2);
    a.ln_g[0] = p->ln_g[2]; a.ln_b[0] = p->ln_b[2]; a.ln_g[1] = p->ln_g[3]; a.ln_b[1] = p->ln_b[3];
    a.d_w = (const _Float16*)p->d_w; a.d_b = (const _Float16*)p->d_b; a.w_w = (const _Float16*)p->w_w; a.w_b = (const _Float16*)p->w_b;
    a.y = y; a.pu = plan + PL.pu; a.perm_k = plan + PL.perm_k; a.patch_off = plan + PL.patch_off; a.ix = plan + PL.ix; a.jx = plan + PL.jx;
    a.tiles = tiles; a.hdr = hdr; a.img = img;
    a.coords = coords; a.pp = P * P;
    a.net_out = net_out; a.delta = delta; a.weight = weight; a.target = target; a.E = E;
    FU(launch<kb2_kernel>(grid_tiles, GeoA2::LDS_BYTES, a, st));
  }
  if (status) {
    if (hipMemcpyAsync(status, hdr + H2_ERR, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return DPVO_E_INVALID;
  }
#undef FU
  return DPVO_OK;
}

#ifdef FU_TRACE
extern "C" int dpvo_debug_pm2_trace_buffer(void* buf) {      // trace builds only: device buffer of 8*1024*4*16 u64, or NULL
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fu_trace), &p, sizeof(p));
}
#endif
