// update_pm.hip -- the update operator in FOUR launches, edges in per-patch order ("patch-major").  COMPARATOR library
// (libdpvo_hip_cmp.so): slower than the seven-launch operator of the product (711 vs 560 us: its k-loop is register-starved),
// kept as the structure that removes 60 % of the operator's memory traffic and as a second, independently written
// implementation the tests compare against.  Device building blocks: update_fused_dev.h.
#undef FU_TRACE
#include "update_fused_dev.h"
#include "../../include/dpvo_hip_cmp.h"

namespace {
namespace fu {


// ================================================================================================================
// Patch-major path: FOUR launches.
//
// With the edges taken in the plan's per-patch order (perm_k: sorted by patch, then target frame) two of the three
// exchanges of the operator stop crossing tiles when a tile holds whole patches:
//   * the neighbour rows of c1 / c2 (fastba.neighbors, ba.cpp:59-97: previous / next edge of the same patch in jj order) are
//     the rows q - 1 / q + 1 of the sorted order: the GEMM just reads the LDS tile one row up or down (a zero row at a patch
//     boundary), no gather at all;
//   * agg_kk (SoftAgg over the edges of a patch, net.py:87) is a segmented softmax over contiguous rows of the tile.
// Only agg_ij (groups by frame pair) still needs every tile.  So:
//     pm_prepare   tiles of <= 96 rows made of whole patches (greedy, 16 waves), inverse of perm_k
//     KA           corr MLP + norm, c1, c2, agg_kk (f, g, softmax-sum, h), f | g of agg_ij       -- f32 state in registers
//     SA           agg_ij softmax-sum over the frame-pair groups (rows addressed through the inverse permutation)
//     KB           agg_ij.h, 2 x (LayerNorm, gated residual), heads
// HBM traffic per edge: corr 1792 + net 1536 in, image 1536 + f | g 1536 out (KA); 1536 in (SA); image 1536 in, net 1536 out
// (KB) = ~11 KB instead of ~26 KB for the seven-launch path, and the workgroups (persistent, tiles handed out by an atomic
// counter) drift apart after their first tile, so the memory phases of one CU overlap the MFMA phases of the others.
// ================================================================================================================
struct Tile { int32_t row0, nrows, p0, np; };
enum { HDR_NTILES = 0, HDR_CTR_A = 1, HDR_CTR_B = 2, HDR_ERR = 3, HDR_INTS = 8 };
constexpr int PM_MAX_TILES = 2048;

// block 0: greedy packing (wave w packs the patches [w C, (w + 1) C)); every block: invk[perm_k[q]] = q
__global__ __launch_bounds__(1024) void pm_prepare_kernel(const int32_t* __restrict__ perm_k, const int32_t* __restrict__ patch_off,
                                                          const int32_t* __restrict__ counts, int64_t E, int R,
                                                          int32_t* __restrict__ invk, Tile* __restrict__ tiles,
                                                          int32_t* __restrict__ hdr, int max_tiles) {
  for (int64_t q = blockIdx.x * 1024ll + threadIdx.x; q < E; q += gridDim.x * 1024ll) invk[perm_k[q]] = (int32_t)q;
  if (blockIdx.x != 0) return;
  __shared__ Tile s_tiles[16][128];
  __shared__ int s_cnt[16], s_err;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int np = counts[0];
  const int C = (np + 15) / 16;
  const int pbeg = w * C, pend = (pbeg + C < np) ? pbeg + C : np;
  if (threadIdx.x == 0) s_err = 0;
  __syncthreads();
  int cnt = 0, err = 0;
  for (int p = pbeg; p < pend;) {
    const int row0 = patch_off[p];
    const int pl = p + lane;
    const int endl = (pl < pend) ? patch_off[pl + 1] : 0x7fffffff;
    const bool ok = (pl < pend) && (endl - row0 <= R) && (lane < 32);
    const unsigned long long m = __ballot(ok);
    int n = (~m == 0ull) ? 64 : __builtin_ctzll(~m);            // leading run of patches that fit (sizes are monotone)
    if (n < 1) { n = 1; err = 1; }                                // a patch with more than R edges: not representable here
    int nrows = patch_off[p + n] - row0;
    if (nrows > R) nrows = R;
    if (lane == 0) {
      if (cnt < 128) s_tiles[w][cnt] = Tile{row0, nrows, p, n};
      else err = 1;
    }
    ++cnt;
    p += n;
  }
  if (cnt > 128) cnt = 128;
  if (lane == 0) { s_cnt[w] = cnt; if (err) s_err = 1; }
  __syncthreads();
  int base = 0, total = 0;
  for (int i = 0; i < 16; ++i) { if (i < w) base += s_cnt[i]; total += s_cnt[i]; }
  for (int i = lane; i < cnt; i += 64) tiles[base + i] = s_tiles[w][i];
  // (more tiles than the launch grids of KA / KB cover: only possible if the caller's bound on the edges of a patch is wrong)
  if (threadIdx.x == 0) { hdr[HDR_NTILES] = total < max_tiles ? total : max_tiles; hdr[HDR_ERR] = (s_err || total > max_tiles) ? 1 : 0; }
}

// SoftAgg over groups whose member rows are addressed through an inverse permutation (row = invk[perm[p]])
__global__ __launch_bounds__(384) void softagg_inv_kernel(const _Float16* __restrict__ fg, int64_t ldfg,
                                                          const int32_t* __restrict__ perm, const int32_t* __restrict__ invk,
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
      const _Float16* rowp = fg + (int64_t)invk[perm[p]] * ldfg + 4 * cq;
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

// k-loop of a 384-wide layer as a ROLLED loop (DW k-steps per trip, DW even and a divisor of 24): ~1 KB of code per call
// site instead of 4 KB -- KA chains twelve GEMMs and must stay friendly to the 64 KB instruction cache.  Per-row-tile
// B-fragment addresses (bl[r]) so that a lane can read a shifted row or the zero row.  The prefetch index is clamped
// (the last trips re-request k-step 23, harmless).
template <int RT, int DW>
__device__ __forceinline__ void gemm384(f16v (&acc)[RT][3], h8 (&wf)[DW][3], const h8* __restrict__ wp, const char* const (&bl)[RT]) {
  static_assert(24 % DW == 0 && (DW & 1) == 0, "ring");
  h8 bf[2][RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl[r]);
#pragma unroll 1
  for (int s0 = 0; s0 < 24; s0 += DW) {
#pragma unroll
    for (int d = 0; d < DW; ++d) {
      const int s = s0 + d;
      const int sn = s + 1 < 24 ? s + 1 : 23;
#pragma unroll
      for (int r = 0; r < RT; ++r) bf[(d + 1) & 1][r] = *reinterpret_cast<const h8*>(bl[r] + sn * 32);
      __builtin_amdgcn_sched_barrier(0);        // (else the scheduler sinks these reads below the MFMAs to save 4 RT registers,
                                                //  and every k-step then waits a full LDS round trip)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r)
          acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[d][t], bf[d & 1][r], acc[r][t], 0, 0, 0);
      const int sw = s + DW < 24 ? s + DW : 23;
#pragma unroll
      for (int t = 0; t < 3; ++t) wf[d][t] = wp[(sw * 3 + t) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
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

struct PA {
  Lin c0, c2, c5, c1a, c1b, c2a, c2b, fk, gk, hk, fi, gi;
  const float *cln_g, *cln_b, *norm_g, *norm_b;
  const _Float16* corr; int64_t ld_corr;
  const float* net;
  const _Float16* inp; const int64_t* inp_rows; int64_t inp_mod;
  const int32_t *perm_k, *ix, *jx, *ku;
  const Tile* tiles; int32_t* hdr;
  float* img; _Float16* fg;
  int64_t E;
};

template <int RT> struct GeoA {
  static constexpr int R = 32 * RT;
  static constexpr int T1 = 0, T2 = R * PITCH, RED = 2 * R * PITCH, ZERO = RED + R * 32, META = ZERO + PITCH;
  static constexpr int LNP = META + R * 8 + 16;                  // two LayerNorms: 2 x [gamma | beta] f32
  static constexpr int LDS_BYTES = LNP + 2 * 2 * D * 4;
  static_assert(LDS_BYTES <= 163840, "LDS");
};

template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void ka_kernel(const PA p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA<RT>;
  constexpr int R = G::R;
  const Lane l = lane_of();
  char* t1 = smem + G::T1;
  char* t2 = smem + G::T2;
  float* red = reinterpret_cast<float*>(smem + G::RED);
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);          // edge id of row i (-1: no row)
  int32_t* meta_f = meta_e + R;                                            // patch index in the tile | first << 8 | last << 9
  int32_t* slot = meta_f + R;                                              // the tile this workgroup drew
  if ((int)blockIdx.x >= p.hdr[HDR_NTILES]) return;
  for (int i = l.tid; i < PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(smem + G::ZERO)[i] = 0u;
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.cln_g[i]; lnp[D + i] = p.cln_b[i]; lnp[2 * D + i] = p.norm_g[i]; lnp[3 * D + i] = p.norm_b[i];
  }
  // one tile per workgroup, the grid is the host's upper bound on the number of tiles (surplus workgroups leave at once): the
  // hardware dispatcher hands out tiles as CUs become free.  (A persistent loop around this body made the compiler hoist
  // ~400 loop-invariant addresses out of it and spill them: 4x slower.)
  (void)slot;
  const int n_tiles = p.hdr[HDR_NTILES];
  {
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    FU_T(5, 0);
    const Tile T = p.tiles[tile];
    const int64_t row0 = T.row0;
    if (l.tid < R) {
      const int i = l.tid;
      int e = -1, f = (1 << 8) | (1 << 9);
      if (i < T.nrows) {
        e = p.perm_k[row0 + i];
        f = (p.ku[e] - T.p0) | ((p.ix[e] < 0) << 8) | ((p.jx[e] < 0) << 9);
      }
      meta_e[i] = e;
      meta_f[i] = f;
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
    gemm384<RT, DW>(acc, wf, wp, bl0);
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
      gemm384<RT, DW>(x, wf, wp, bl0);
      wp = w_base(p.c1a.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.c1a.b, l);
      round_f16<RT>(x);
      // rows net[e] (f32) and inp[kk[e] % mod] (f16) in feature order, one row tile at a time (72 registers in flight)
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        int64_t ir = eg[r];
        if (p.inp_rows) { ir = p.inp_rows[eg[r]]; if (p.inp_mod > 0) ir %= p.inp_mod; }
        f4 nv[3][4];
        h4 iv[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
            nv[t][j] = *reinterpret_cast<const f4*>(p.net + (int64_t)eg[r] * D + f);
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
        const int i = r * 32 + l.n;
        const int f = meta_f[i];
        const bool none = k == 0 ? ((f >> 8) & 1) : ((f >> 9) & 1);
        bls[r] = none ? smem + G::ZERO + 16 * l.h : bl0[r] + (k == 0 ? -PITCH : PITCH);
      }
      acc_init<RT>(acc, bias);
      gemm384<RT, DW>(acc, wf, wp, bls);
      wp = w_base(k == 0 ? p.c1b.w : p.c2b.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, k == 0 ? p.c1b.b : p.c2b.b, l);
      to_lds<RT, 1>(acc, const_cast<char*>(al) + G::T2, l);
      __syncthreads();
      acc_init<RT>(acc, bias);
      gemm384<RT, DW>(acc, wf, wp, bl2);
      wp = w_base(k == 0 ? p.c2a.w : p.fk.w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, k == 0 ? p.c2a.b : p.fk.b, l);
      residual_add<RT>(x, acc);
      to_lds<RT, 0>(x, const_cast<char*>(al), l);          // (T1 was last read before the barrier above)
      __syncthreads();
    }
    FU_T(5, 4);
    // ---- agg_kk: f -> T2, g -> T1, segmented softmax-sum over the rows of every patch, y -> T1 rows [0, np)   (blocks.py:40-43)
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.gk.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gk.b, l);
    to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.hk.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.hk.b, l);
    __syncthreads();
    to_lds<RT, 0>(acc, const_cast<char*>(al), l);
    __syncthreads();
    FU_T(5, 5);
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
    // ---- h on the (<= 32) patch rows, expanded back to the edges: x += h(y)[patch of the row]       (blocks.py:45-48)
    {
      f16v hy[1][3];
      const char* blh[1] = {al};
      acc_init<1>(hy, bias);
      gemm384<1, DW>(hy, wf, wp, blh);
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
    // ---- f | g of agg_ij for every edge row, stored at the sorted position
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    wp = w_base(p.gi.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.gi.b, l);
    to_lds<RT, 0>(acc, const_cast<char*>(al) + G::T2, l);               // (T2 = h(y): consumed before the barrier above)
    acc_init<RT>(acc, bias);
    gemm384<RT, DW>(acc, wf, wp, bl0);
    __syncthreads();
    to_lds<RT, 0>(acc, const_cast<char*>(al), l);
    __syncthreads();
    tile_to_rows<RT>(t2, p.fg, 768, row0, T.nrows, l.tid);
    tile_to_rows<RT>(t1, p.fg + D, 768, row0, T.nrows, l.tid);
    FU_T(5, 8);
  }
}

struct PB {
  Lin h;
  Lin gate[2], res0[2], res2[2];
  const float *ln_g[2], *ln_b[2];
  const _Float16 *d_w, *d_b, *w_w, *w_b;
  const _Float16* y; const int32_t *pu, *perm_k;
  const Tile* tiles; int32_t* hdr;
  const float* img;
  const float* coords; int pp;
  float *net_out, *delta, *weight, *target;
  int64_t E;
};

template <int RT, int DW>
__global__ __launch_bounds__(256, 1) void kb_kernel(const PB p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using G = GeoA<RT>;
  constexpr int R = G::R;
  const Lane l = lane_of();
  char* t1 = smem + G::T1;
  float* red = reinterpret_cast<float*>(smem + G::RED);          // (the heads need R x 64 B: RED + ZERO + META are contiguous, R x 32 + 784 + ...)
  int32_t* meta_e = reinterpret_cast<int32_t*>(smem + G::META);
  int32_t* slot = meta_e + 2 * R;
  const int n_tiles = p.hdr[HDR_NTILES];
  if ((int)blockIdx.x >= n_tiles) return;
  float* lnp = reinterpret_cast<float*>(smem + G::LNP);
  for (int i = l.tid; i < D; i += 256) {
    lnp[i] = p.ln_g[0][i]; lnp[D + i] = p.ln_b[0][i]; lnp[2 * D + i] = p.ln_g[1][i]; lnp[3 * D + i] = p.ln_b[1][i];
  }
  char* al = t1 + l.n * PITCH + 16 * l.h;
  char* gl = al + G::T2;
  const char* bl0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) bl0[r] = al + r * 32 * PITCH;

  (void)slot;
  {
    const int tile = blockIdx.x;
    FU_T(6, 0);
#ifdef FU_TRACE
    if (g_fu_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)
      g_fu_trace[(((size_t)6 * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + 12] = clock64();
#endif
    const Tile T = p.tiles[tile];
    const int64_t row0 = T.row0;
    if (l.tid < R) meta_e[l.tid] = l.tid < T.nrows ? p.perm_k[row0 + l.tid] : -1;
    __syncthreads();

    f16v x[RT][3];
    h8 wf[DW][3];
    Bias bias;
    const h8* wp = w_base(p.h.w, KS384, l);
    w_preload<DW>(wf, wp);
    bias_load(bias, p.h.b, l);
    float* ip = img_ptr<RT>(const_cast<float*>(p.img), tile, l);
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
    __syncthreads();
    acc_init<RT>(x, bias);
    gemm384<RT, DW>(x, wf, wp, bl0);
    round_f16<RT>(x);
    // x = image + h(y); from here on the f32 state is in registers only BETWEEN the GEMMs of a gated residual: it is written
    // back to its (lane-private) image slot after each LayerNorm and re-read, one row tile at a time, for the residual add.
    // Holding it across the three GEMMs (144 + 144 accumulator registers > the 256-entry accumulation file) made the compiler
    // spill ~125 registers, and a scratch reload costs a full memory round trip with one wave per SIMD.
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 m[3][4];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) x[r][t][4 * j + q] += m[t][j][q];
      __builtin_amdgcn_sched_barrier(0);
    }
    FU_T(6, 1);
#pragma unroll
    for (int Gi = 0; Gi < 2; ++Gi) {
      if (Gi == 0) FU_T(7, 0);
      layernorm_tile_lds<RT>(x, red, lnp + 2 * D * Gi, l);
      if (Gi == 0) FU_T(7, 1);
      wp = w_base(p.gate[Gi].w, KS384, l);
      w_preload<DW>(wf, wp);
      bias_load(bias, p.gate[Gi].b, l);
      img_store<RT>(x, ip);
      if (Gi == 0) FU_T(7, 2);
      to_lds<RT, 0>(x, al, l);
      __syncthreads();
      if (Gi == 0) FU_T(7, 3);
      {
        f16v acc[RT][3];
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 4);
        wp = w_base(p.res0[Gi].w, KS384, l);
        w_preload<DW>(wf, wp);
        bias_load(bias, p.res0[Gi].b, l);
        to_lds<RT, 2>(acc, gl, l);
        if (Gi == 0) FU_T(7, 5);
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 6);
        wp = w_base(p.res2[Gi].w, KS384, l);
        w_preload<DW>(wf, wp);
        bias_load(bias, p.res2[Gi].b, l);
        __syncthreads();
        to_lds<RT, 1>(acc, al, l);
        __syncthreads();
        if (Gi == 0) FU_T(7, 7);
        acc_init<RT>(acc, bias);
        gemm384<RT, DW>(acc, wf, wp, bl0);
        if (Gi == 0) FU_T(7, 8);
        // x = x(image) + gate * res   (half * half -> half, blocks.py:28-29)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          f4 m[3][4];
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[t][j] = *reinterpret_cast<const f4*>(ip + r * IMG_RT_STRIDE + (t * 4 + j) * 256);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const h8 gt = *reinterpret_cast<const h8*>(gl + r * 32 * PITCH + ((3 * l.w + t) * 2 + c) * 32);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int k = 8 * c + i;
                const _Float16 rv = (_Float16)acc[r][t][k];
                x[r][t][k] = m[t][k >> 2][k & 3] + (float)(_Float16)(gt[i] * rv);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FU_T(6, 2 + Gi);
    }
    // ---- hidden state out (rows net_out[e], feature order) and the heads
    int eg[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) eg[r] = meta_e[r * 32 + l.n];
    float dsum[RT][4];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int o = 0; o < 4; ++o) dsum[r][o] = 0.f;
    h4 wv[3][4][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = 96 * l.w + 32 * t + 8 * j + 4 * l.h;
        wv[t][j][0] = *reinterpret_cast<const h4*>(p.d_w + f);
        wv[t][j][1] = *reinterpret_cast<const h4*>(p.d_w + D + f);
        wv[t][j][2] = *reinterpret_cast<const h4*>(p.w_w + f);
        wv[t][j][3] = *reinterpret_cast<const h4*>(p.w_w + D + f);
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
          for (int q = 0; q < 4; ++q) {
            const float v = x[r][t][4 * j + q];
            o4[q] = v;
            const float a = (float)(_Float16)(v > 0.f ? v : 0.f);
#pragma unroll
            for (int o = 0; o < 4; ++o) dsum[r][o] += a * (float)wv[t][j][o][q];
          }
          if (eg[r] >= 0) *reinterpret_cast<f4*>(p.net_out + (int64_t)eg[r] * D + f) = o4;
        }
      }
    __syncthreads();                                  // (the LayerNorm partials in `red` are dead; T1 reads retired)
    float* hred = reinterpret_cast<float*>(t1);       // [R][4 waves][4]: the tile is free now
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f4 s;
#pragma unroll
      for (int o = 0; o < 4; ++o) { s[o] = dsum[r][o]; s[o] += xhalf(s[o]); }
      if (l.h == 0) *reinterpret_cast<f4*>(hred + ((r * 32 + l.n) * 4 + l.w) * 4) = s;
    }
    __syncthreads();
    if (l.tid < R) {
      const int e = meta_e[l.tid];
      if (e >= 0) {
        f4 s = *reinterpret_cast<const f4*>(hred + (l.tid * 4 + 0) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const f4 q = *reinterpret_cast<const f4*>(hred + (l.tid * 4 + w) * 4);
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
#ifdef FU_TRACE
    if (g_fu_trace && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)
      g_fu_trace[(((size_t)6 * 1024 + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + 13] = clock64();
#endif
  }
}

}  // namespace fu
}  // namespace

#define FU_RT 3
#define FU_DW 6
#define FU_DWPM 4
#define FU_DWKB 6

// ------------------------------------------------------------------------------------------------ patch-major entry
namespace {
namespace fu {
struct WsPm { size_t img, fg, y, invk, tiles, hdr, total; int64_t max_tiles; };
template <int RT>
void ws_layout_pm(int64_t E, int64_t maxg, WsPm* w) {
  const size_t e = (size_t)(E > 0 ? E : 1), g = (size_t)(maxg > 0 ? maxg : 1);
  int64_t mt = 2 * cdiv64((int64_t)e, 32 * RT) + 48;          // (greedy tiles of whole patches: two consecutive ones hold > R rows, + 16 wave tails)
  if (mt > PM_MAX_TILES) mt = PM_MAX_TILES;
  w->max_tiles = mt;
  size_t o = 0;
  w->img = o; o += al256((size_t)mt * 32 * RT * D * 4);
  w->fg = o; o += al256(e * 2 * D * 2);
  w->y = o; o += al256(g * D * 2);
  w->invk = o; o += al256(e * 4);
  w->tiles = o; o += al256((size_t)PM_MAX_TILES * sizeof(Tile));
  w->hdr = o; o += al256(HDR_INTS * 4);
  w->total = o;
}
}  // namespace fu
}  // namespace

extern "C" size_t dpvo_update_pm_workspace_bytes(int64_t E, int64_t max_groups) {
  if (E < 0 || max_groups < 0) return 0;
  fu::WsPm w;
  fu::ws_layout_pm<FU_RT>(E, max_groups, &w);
  return w.total;
}

extern "C" int dpvo_update_forward_pm(const dpvo_update_fused_params_t* p, const float* net, const void* inp,
                                      const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr,
                                      const int32_t* plan, int64_t n_patches_ub, int64_t n_pairs_ub, int64_t patch_edges_ub,
                                      const float* coords, int P, float* net_out, float* delta, float* weight, float* target,
                                      int64_t E, void* ws, size_t ws_bytes, int32_t* status, void* stream) {
  using namespace fu;
  constexpr int RT = FU_RT, DW = FU_DW;
  if (E < 0 || !p) return DPVO_E_INVALID;
  if (E == 0) return DPVO_OK;
  if (!net || !inp || !corr || !plan || !net_out || !delta || !weight || !ws) return DPVO_E_INVALID;
  if (target && (!coords || P <= 0)) return DPVO_E_INVALID;
  if (ld_corr < 896 || (ld_corr % 8) || E >= (1ll << 31)) return DPVO_E_UNSUPPORTED;
  // tiles hold whole patches: every patch must fit (32 RT rows), and the tile table has PM_MAX_TILES entries
  if (patch_edges_ub <= 0 || patch_edges_ub > 32 * RT) return DPVO_E_UNSUPPORTED;
  if (2 * cdiv64(E, 32 * RT) + 48 > PM_MAX_TILES) return DPVO_E_UNSUPPORTED;
  for (int i = 0; i < DPVO_UF_NLIN; ++i)
    if (!p->w[i] || !p->b[i]) return DPVO_E_INVALID;
  const int64_t maxg = n_patches_ub > n_pairs_ub ? n_patches_ub : n_pairs_ub;
  WsPm L;
  ws_layout_pm<RT>(E, maxg, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* wsb = (char*)ws;
  float* img = (float*)(wsb + L.img);
  _Float16 *fg = (_Float16*)(wsb + L.fg), *y = (_Float16*)(wsb + L.y);
  int32_t* invk = (int32_t*)(wsb + L.invk);
  Tile* tiles = (Tile*)(wsb + L.tiles);
  int32_t* hdr = (int32_t*)(wsb + L.hdr);
  hipStream_t st = (hipStream_t)stream;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  auto lin = [&](int i) { return Lin{p->w[i], (const _Float16*)p->b[i]}; };
  // upper bound on the number of tiles: a closed tile plus the next patch exceed R rows, so every tile but the last of each of
  // the 16 packing waves holds more than R - patch_edges_ub rows
  int64_t grid_tiles = E / (32 * RT - patch_edges_ub + 1) + 17;
  if (grid_tiles > L.max_tiles) grid_tiles = L.max_tiles;
  (void)n_cu;
  int rc;
#define FU(...) do { rc = (__VA_ARGS__); if (rc) return rc; } while (0)
  {
    int64_t g = cdiv64(E, 1024);
    if (g > 64) g = 64;
    hipLaunchKernelGGL(pm_prepare_kernel, dim3((unsigned)g), dim3(1024), 0, st, plan + PL.perm_k, plan + PL.patch_off,
                       plan + PL.counts, E, 32 * RT, invk, tiles, hdr, (int)grid_tiles);
    DPVO_LAUNCH_CHECK();
  }
  {
    PA a;
    a.c0 = lin(DPVO_UF_C0); a.c2 = lin(DPVO_UF_C2); a.c5 = lin(DPVO_UF_C5);
    a.c1a = lin(DPVO_UF_C1_0); a.c1b = lin(DPVO_UF_C1_2); a.c2a = lin(DPVO_UF_C2N_0); a.c2b = lin(DPVO_UF_C2N_2);
    a.fk = lin(DPVO_UF_AKK_F); a.gk = lin(DPVO_UF_AKK_G); a.hk = lin(DPVO_UF_AKK_H);
    a.fi = lin(DPVO_UF_AIJ_F); a.gi = lin(DPVO_UF_AIJ_G);
    a.cln_g = p->ln_g[0]; a.cln_b = p->ln_b[0]; a.norm_g = p->ln_g[1]; a.norm_b = p->ln_b[1];
    a.corr = (const _Float16*)corr; a.ld_corr = ld_corr; a.net = net; a.inp = (const _Float16*)inp; a.inp_rows = inp_rows;
    a.inp_mod = inp_mod;
    a.perm_k = plan + PL.perm_k; a.ix = plan + PL.ix; a.jx = plan + PL.jx; a.ku = plan + PL.ku;
    a.tiles = tiles; a.hdr = hdr; a.img = img; a.fg = fg; a.E = E;
    FU(launch<ka_kernel<RT, FU_DWPM>>(grid_tiles, GeoA<RT>::LDS_BYTES, a, st));
  }
  {
    int64_t ngp = n_pairs_ub < 1 ? 1 : (n_pairs_ub > E ? E : n_pairs_ub);
    const unsigned grid = (unsigned)(ngp < 8192 ? ngp : 8192);
    hipLaunchKernelGGL(softagg_inv_kernel, dim3(grid), dim3(384), 0, st, (const _Float16*)fg, (int64_t)768, plan + PL.perm_p,
                       (const int32_t*)invk, plan + PL.pair_off, plan + PL.counts + 1, y);
    DPVO_LAUNCH_CHECK();
  }
  {
    PB a;
    a.h = lin(DPVO_UF_AIJ_H);
    a.gate[0] = lin(DPVO_UF_G0_GATE); a.res0[0] = lin(DPVO_UF_G0_RES0); a.res2[0] = lin(DPVO_UF_G0_RES2);
    a.gate[1] = lin(DPVO_UF_G1_GATE); a.res0[1] = lin(DPVO_UF_G1_RES0); a.res2[1] = lin(DPVO_UF_G1_RES2);
    a.ln_g[0] = p->ln_g[2]; a.ln_b[0] = p->ln_b[2]; a.ln_g[1] = p->ln_g[3]; a.ln_b[1] = p->ln_b[3];
    a.d_w = (const _Float16*)p->d_w; a.d_b = (const _Float16*)p->d_b; a.w_w = (const _Float16*)p->w_w; a.w_b = (const _Float16*)p->w_b;
    a.y = y; a.pu = plan + PL.pu; a.perm_k = plan + PL.perm_k; a.tiles = tiles; a.hdr = hdr; a.img = img;
    a.coords = coords; a.pp = P * P;
    a.net_out = net_out; a.delta = delta; a.weight = weight; a.target = target; a.E = E;
    FU(launch<kb_kernel<RT, FU_DWKB>>(grid_tiles, GeoA<RT>::LDS_BYTES, a, st));
  }
  if (status) {      // device int32: 1 if a patch had more than 32 RT edges (results then undefined but memory safe)
    if (hipMemcpyAsync(status, hdr + HDR_ERR, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return DPVO_E_INVALID;
  }
#undef FU
  return DPVO_OK;
}

