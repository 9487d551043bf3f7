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
//   gba_scatter_kernel   edge records -> Ecol[pair][6][slot]: a block per pair, a thread per patch slot sums that slot's
//                        edges in list order (duplicates of a (patch, frame) edge fold in a fixed order)
//   gba_patch_kernel     per patch (CSR): C, u, Ei -> Q, u, Eself[frame][6][slot]
//   gba_index_kernel     pairs by TARGET pose (tgt_off / tgt_list, ascending pair index) and the pair run of every source frame
//   gba_row_kernel       one workgroup per free pose p builds block row p of S = B - E Q E^T and y[p] = v - E Q u: the B / v
//                        terms of the pairs that touch p, then, for every source frame that sees p (ascending), the Schur
//                        terms  sum_slot Q ea eb^T  against all of that frame's blocks.  Every entry of the row is updated by
//                        ONE wave (column pose mod 4) in program order: no atomics, fixed summation order (the mirror blocks
//                        (p,q) / (q,p) are formed from the same commutative products in the same order), so the whole call
//                        is BIT-REPEATABLE (rounds 1-3 accumulated S with float
//                        atomics like the reference, ba_cuda.cu:335-373 / block_e.cu:147-283: repeatable to rounding only)
//   dpvo_gba_solve       S += I*(1e-4*S+1); blocked Cholesky + both substitutions on the device (chol.hip)
//   gba_retr_kernel      dZ = Q (u - e^T dX), depth + pose retraction
#include "ba_common.h"

namespace {
using namespace ba;

struct GbaWs {
  size_t pairbuf, edgebuf, run_lo, tgt_off, tgt_cnt, tgt_list, Q, u, Ecol, Eself, total;
};
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
inline void gba_layout(int64_t E, int64_t n_pairs, int64_t n_frames, int M, int64_t n_free, GbaWs* L) {
  const size_t n = (size_t)(E > 0 ? E : 1), g = (size_t)(n_pairs > 0 ? n_pairs : 1), f = (size_t)(n_frames > 0 ? n_frames : 1);
  size_t o = 0;
  L->pairbuf = o; o += al(g * kPairStride * 4);
  L->edgebuf = o; o += al(n * kEdgeStride * 4);
  const size_t nf = (size_t)(n_free > 0 ? n_free : 1);
  L->run_lo = o; o += al((f + 2) * 4);
  L->tgt_off = o; o += al((nf + 2) * 4);
  L->tgt_cnt = o; o += al((nf + 2) * 4);
  L->tgt_list = o; o += al(g * 4);
  L->Q = o; o += al(f * M * 4);
  L->u = o; o += al(f * M * 4);
  L->Eself = o; o += al(f * M * 6 * 4);
  L->Ecol = o; o += al(g * M * 6 * 4);          // (last: the only one of the four that is written in full, so it is not cleared)
  L->total = o;
}

// Ecol[pair][kk[e] % M][0..5] = sum over the pair's edges e with that slot of Ej_e, in list order (perm_p is the stable sort by
// (ii, jj): ascending edge number inside a pair).  128 threads per pair, a thread per patch slot; the slots of a chunk of 128 edges go
// through LDS.  `half` of `nhalf` 128-thread groups share a workgroup (the fused launch below runs two pairs per 256-thread
// workgroup): every group runs `nch` chunk steps -- the larger count of the workgroup's pairs -- because the barriers are the
// workgroup's; a group whose pair has fewer chunks (or no pair at all: g >= ng) idles through the surplus steps.
__device__ __forceinline__ void gba_scatter_body(const int64_t* __restrict__ kk, const int32_t* __restrict__ perm_p,
                                                 const int32_t* __restrict__ pair_off, const int32_t* __restrict__ n_pairs,
                                                 const float* __restrict__ edgebuf, float* __restrict__ Ecol, int M, int bid, int nblk,
                                                 int nhalf, int (*sl)[128], int (*se)[128]) {
  const int ng = *n_pairs, half = (int)threadIdx.x >> 7, t = (int)threadIdx.x & 127;
  for (int gb = bid * nhalf; gb < ng; gb += nblk * nhalf) {            // (uniform over the workgroup)
    const int g = gb + half;
    const bool on = g < ng;
    const int p0 = on ? pair_off[g] : 0, p1 = on ? pair_off[g + 1] : 0;
    int nch = 0;                                                       // chunk steps of the workgroup = the largest of its pairs'
    for (int h = 0; h < nhalf; ++h)
      if (gb + h < ng) { const int len = pair_off[gb + h + 1] - pair_off[gb + h]; const int c = (len + 127) >> 7; nch = c > nch ? c : nch; }
    for (int s0 = 0; s0 < M; s0 += 128) {
      const int slot = s0 + t;
      float acc[6] = {0, 0, 0, 0, 0, 0};
      for (int ch = 0; ch < nch; ++ch) {
        const int c0 = p0 + 128 * ch;
        __syncthreads();
        if (c0 + t < p1) { const int e = perm_p[c0 + t]; se[half][t] = e; sl[half][t] = (int)(kk[e] % M); } else sl[half][t] = -1;
        __syncthreads();
        const int nq = p1 - c0 < 128 ? p1 - c0 : 128;                  // (<= 0 for a group that is past its pair's last chunk)
        if (slot < M)
          for (int q = 0; q < nq; ++q)
            if (sl[half][q] == slot) {
              const float* eb = edgebuf + (int64_t)se[half][q] * kEdgeStride + 8;
#pragma unroll
              for (int a = 0; a < 6; ++a) acc[a] += eb[a];
            }
      }
      if (on && slot < M) {
        float* dst = Ecol + (int64_t)g * 6 * M + slot;                  // [pair][6][M]: component-major (see gba_row_kernel)
#pragma unroll
        for (int a = 0; a < 6; ++a) dst[(int64_t)a * M] = acc[a];
      }
    }
  }
}

// per patch k (index into kx): Q, u and the i-side block, stored by (frame - f0, slot); thread gt of nthr
__device__ __forceinline__ void gba_patch_body(const int32_t* __restrict__ perm_k, const int32_t* __restrict__ patch_off,
                                               const int32_t* __restrict__ kx, const int32_t* __restrict__ n_patches,
                                               const float* __restrict__ edgebuf, float lmbda, int M, int f0, int n_frames,
                                               float* __restrict__ Q, float* __restrict__ U, float* __restrict__ Eself, int gt, int nthr) {
  const int np = *n_patches;
  for (int k = gt; k < np; k += nthr) {
    float C = 0.f, u = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0};
    // four edges' records in flight (edge number, then its two 16-byte loads); the sums stay in list order.  One edge at a time this
    // loop was two dependent round trips per edge, ~28 edges per patch at the global BA's size: 44 us of every linearisation
    const int p0 = patch_off[k], p1 = patch_off[k + 1];
    for (int pb_ = p0; pb_ < p1; pb_ += 4) {
      int e4[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) e4[v] = pb_ + v < p1 ? perm_k[pb_ + v] : 0;
      f4 r0[4], r1[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f4* eb = reinterpret_cast<const f4*>(edgebuf + (int64_t)e4[v] * kEdgeStride);
        r0[v] = eb[0]; r1[v] = eb[1];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (pb_ + v < p1) {
          C += r0[v][0]; u += r0[v][1];
          Ei[0] += r0[v][2]; Ei[1] += r0[v][3]; Ei[2] += r1[v][0]; Ei[3] += r1[v][1]; Ei[4] += r1[v][2]; Ei[5] += r1[v][3];
        }
    }
    const int patch = kx[k];
    const int fr = patch / M - f0, slot = patch % M;
    if (fr < 0 || fr >= n_frames) continue;
    const int64_t o = (int64_t)fr * M + slot;
    Q[o] = 1.0f / (C + lmbda);
    U[o] = u;
#pragma unroll
    for (int a = 0; a < 6; ++a) Eself[((int64_t)fr * 6 + a) * M + slot] = Ei[a];          // [frame][6][M]
  }
}

// The two are independent (both read the edge records of ba_pair_kernel, they write different arrays): ONE launch instead of two
// dependent ones of ~30 us each (round 5, last session).  The first n_patch workgroups are the patches' (few, latency bound: they
// start at once and run beside the others), the rest take two pairs each.
struct GbaSP {
  const int64_t* kk; const int32_t *perm_p, *pair_off, *n_pairs, *perm_k, *patch_off, *kx, *n_patches;
  const float* edgebuf; float *Ecol, *Q, *U, *Eself; float lmbda; int M, f0, n_frames, n_patch;
};
__global__ __launch_bounds__(256) void gba_scatter_patch_kernel(GbaSP A) {
  __shared__ int sl[2][128];
  __shared__ int se[2][128];
  if ((int)blockIdx.x < A.n_patch)
    gba_patch_body(A.perm_k, A.patch_off, A.kx, A.n_patches, A.edgebuf, A.lmbda, A.M, A.f0, A.n_frames, A.Q, A.U, A.Eself,
                   (int)blockIdx.x * 256 + (int)threadIdx.x, A.n_patch * 256);
  else
    gba_scatter_body(A.kk, A.perm_p, A.pair_off, A.n_pairs, A.edgebuf, A.Ecol, A.M, (int)blockIdx.x - A.n_patch, (int)gridDim.x - A.n_patch, 2, sl, se);
}

// Q, u, Eself = 0 in front of a linearisation (hipMemsetAsync of these ~0.3 MB is a 23 us fill on this runtime,
// profiles/r05_e_lc_timeline.txt; a plain store kernel is a launch)
__global__ __launch_bounds__(256) void gba_zero_kernel(f4* __restrict__ p, int64_t n4) {
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) p[i] = z;
}

__device__ __forceinline__ int lower_bound_i(const int32_t* pair_ij, int ng, int f) {
  int lo = 0, hi = ng;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (pair_ij[2 * mid] < f) lo = mid + 1; else hi = mid; }
  return lo;
}

// Index structures of the row kernel, one workgroup: run_lo[fr] = first pair of source frame f0 + fr (fr = 0 .. n_frames);
// tgt_off / tgt_list = the pairs whose TARGET is free pose p, ascending pair index (= ascending source frame).
__global__ __launch_bounds__(1024) void gba_index_kernel(const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ n_pairs,
                                                         int f0, int n_frames, int t0, int N, int32_t* __restrict__ run_lo,
                                                         int32_t* __restrict__ tgt_off, int32_t* __restrict__ tgt_cnt,
                                                         int32_t* __restrict__ tgt_list) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int ng = *n_pairs, t = threadIdx.x;
  for (int fr = t; fr <= n_frames; fr += 1024) run_lo[fr] = lower_bound_i(pair_ij, ng, f0 + fr);
  for (int p = t; p < N; p += 1024) tgt_cnt[p] = 0;
  __syncthreads();
  for (int g = t; g < ng; g += 1024) {
    const int jx = pair_ij[2 * g + 1] - t0;
    if (jx >= 0 && jx < N) atomicAdd(&tgt_cnt[jx], 1);            // (integer counts: order-independent)
  }
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {                     // exclusive scan, 1024 poses at a time
    const int v = base + t < N ? tgt_cnt[base + t] : 0;
    part[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int x = t >= o ? part[t - o] : 0;
      __syncthreads();
      part[t] += x;
      __syncthreads();
    }
    if (base + t < N) tgt_off[base + t] = carry + part[t] - v;
    __syncthreads();
    if (t == 1023) carry += part[1023];
    __syncthreads();
  }
  if (t == 0) tgt_off[N] = carry;
  for (int p = t; p < N; p += 1024) tgt_cnt[p] = 0;
  __syncthreads();
  for (int g = t; g < ng; g += 1024) {
    const int jx = pair_ij[2 * g + 1] - t0;
    if (jx >= 0 && jx < N) tgt_list[tgt_off[jx] + atomicAdd(&tgt_cnt[jx], 1)] = g;
  }
  __syncthreads();
  // The claims above land in any order: every list is put in ascending pair order.  A WAVE per list (round 4: a thread per list ran
  // an insertion sort, O(n^2) dependent memory steps for the ~30-100 pairs that target a pose -- 90 of the kernel's 93 us at N = 100,
  // profiles/r05_lc_timeline.txt): the pair indices are distinct, so an entry's place is the number of smaller entries; lane l counts
  // it for entries l, l + 64, ... against the whole list (read through the cache), then the wave writes the list back in place.
  {
    const int lane = t & 63, wv = t >> 6;
    for (int p = wv; p < N; p += 16) {
      int32_t* L = tgt_list + tgt_off[p];
      const int n = tgt_off[p + 1] - tgt_off[p];
      constexpr int kMaxPer = 8;                                   // lists up to 512 entries in registers; longer ones: the serial sort
      if (n <= 64 * kMaxPer) {
        int v[kMaxPer], r[kMaxPer];
#pragma unroll
        for (int a = 0; a < kMaxPer; ++a) { v[a] = lane + 64 * a < n ? L[lane + 64 * a] : 0x7fffffff; r[a] = 0; }
        // every entry against the whole list, handed from register to register (v_readlane with a running lane number).  As broadcast
        // LOADS of L[b] this loop was a dependent memory round trip per entry -- most of the kernel's 68 us at N = 105
        // (profiles/r05_e_lc_timeline.txt)
#pragma unroll
        for (int a2 = 0; a2 < kMaxPer; ++a2) {
          const int nn = n - 64 * a2 < 64 ? n - 64 * a2 : 64;
          for (int l = 0; l < nn; ++l) {
            const int x = __builtin_amdgcn_readlane(v[a2], l);
#pragma unroll
            for (int a = 0; a < kMaxPer; ++a) r[a] += x < v[a];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                           // every lane has read the whole list before anyone overwrites it
#pragma unroll
        for (int a = 0; a < kMaxPer; ++a)
          if (lane + 64 * a < n) L[r[a]] = v[a];
      } else if (lane == 0) {
        for (int a = 1; a < n; ++a) {
          const int v = L[a];
          int b = a - 1;
          while (b >= 0 && L[b] > v) { L[b + 1] = L[b]; --b; }
          L[b + 1] = v;
        }
      }
    }
  }
}

// Block row p of S and y[p] (see the file header).  N * split workgroups (1-D launch, XCD-major numbering: top of the kernel), 1024 threads = 16 waves: the 16 * split waves of a row
// each own the column blocks whose pose q has q % (16 * split) == their number; wave 0 of the first workgroup also owns y.
// split is chosen by the launcher so that every workgroup of the launch is resident at once (a 1 024-thread workgroup
// of this kernel takes a CU): 4 for N <= 64, 2 for N <= 128, else 1.  (Round 4's first rule -- 4 up to 128, 2 up to 400 -- put 396
// workgroups on 256 CUs at N = 99: a second round started 150 us in, tools/gba_trace.sh; linearise + Schur 1.02 -> 0.80 ms there.)
// A wave finds ITS blocks of a source frame with one 64-wide fetch of the frame's targets and a ballot (it used to read the targets
// one by one, a dependent round trip per block whether the block was its own or not).
// (A row is one dependent chain per wave -- ~28 source frames x its share of their ~28 blocks, each a round trip for the block
// operands and one for the read-modify-write: with 4 waves per row the kernel took 0.45-1.9 ms, with 16 0.3-1.9; measured
// alternative: the frame's blocks staged in LDS + the row in an LDS strip, 16 waves: slower (two barriers and a 64 KB copy per
// source frame); the atomics version this replaces: 0.28 ms.)  `S` and `y` must be zero on entry.
constexpr int kRowWaves = 16;
#ifdef GBA_TRACE
// instrumentation build (tools/gba_trace.sh): wave 0 of the workgroup of the middle pose stamps the 100 MHz wall clock
__device__ unsigned long long gba_trace_buf[96];
__device__ unsigned long long gba_wg_buf[4096][3];     // start, end, source frames walked -- of every workgroup (pose x split)
__device__ unsigned int gba_wave_buf[1024][16][2];     // per wave: ticks from the workgroup's start to the wave's end, blocks it owned
#define GT(i) do { if (threadIdx.x == 0 && part_ == 0 && p == N / 2 && (i) < 96) gba_trace_buf[i] = wall_clock64(); } while (0)
#else
#define GT(i) do {} while (0)
#endif
__global__ __launch_bounds__(64 * kRowWaves) void gba_row_kernel(const int32_t* __restrict__ pair_ij, const int32_t* __restrict__ n_pairs,
                                                      const int32_t* __restrict__ run_lo, const int32_t* __restrict__ tgt_off,
                                                      const int32_t* __restrict__ tgt_list, const float* __restrict__ pairbuf,
                                                      const float* __restrict__ Q, const float* __restrict__ U,
                                                      const float* __restrict__ Ecol, const float* __restrict__ Eself, int M,
                                                      int f0, int n_frames, int t0, int N, int split, float* __restrict__ S,
                                                      float* __restrict__ y) {
  // Workgroup -> (pose, part).  The launch is 1-D; workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md), so the b-th
  // workgroup of an XCD takes logical index xcd * chunk + b / 8: the parts of a row and the rows of neighbouring poses -- which read
  // the blocks of the same ~40 source frames, 3-4 MB: an XCD's L2 -- share an XCD instead of being dealt out round robin over all
  // eight (the E blocks, 13 MB at the bench leg's size, then come from the Infinity Cache at twice the latency).  Placement only:
  // every entry is still written by the same wave in the same order.
  const int chunk_ = ((int)gridDim.x + 7) / 8;
  const int lidx_ = ((int)blockIdx.x % 8) * chunk_ + (int)blockIdx.x / 8;
  if (lidx_ >= N * split) return;
  const int p = lidx_ / split, part_ = lidx_ - p * split, j = p + t0;
  GT(0);
#ifdef GBA_TRACE
  const int wg_ = lidx_;
  if (threadIdx.x == 0 && wg_ < 4096) gba_wg_buf[wg_][0] = wall_clock64();
#endif
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) + kRowWaves * part_);
  const int kRowCls = kRowWaves * split;
  const int64_t n6 = 6 * (int64_t)N;
  const int r36 = lane / 6, c36 = lane - 6 * r36;                  // (lanes 0..35: entry (r36, c36) of a 6 x 6 block)
  float* Srow = S + (int64_t)(6 * p) * n6;
  const int tl0 = tgt_off[p], tl1 = tgt_off[p + 1];
  const int frj = j - f0;                                          // pose p as a SOURCE frame (it owns patches iff 0 <= frj < n_frames)
  const bool own = frj >= 0 && frj < n_frames;
  const int ja = own ? run_lo[frj] : 0, jb = own ? run_lo[frj + 1] : 0;
  // ---- B and v (ba_cuda.cu:335-349,363-368): pairs with source j, then pairs with target j, ascending pair index each
  // Round 5 (last session).  The two loops used to be one dependent chain per pair for EVERY wave of the row: the pair's target (loop 1)
  // resp. the list entry, then the pair's source (loop 2) loaded one at a time just to find out whose column it is, and the diagonal
  // block and y[p] read-modify-written in memory once per pair -- ~60 pairs x 2-3 round trips in front of the Schur terms.  Now the
  // indices come 64 at a time into the lanes and are handed out with readlane; the diagonal block and y[p] accumulate in registers in
  // the same order (S and y are zero on entry: 0 + a + b ... is what the read-modify-writes formed) and are written once; an
  // off-diagonal block of loop 1 is a plain store of 0 - b (zero on entry, the pairs of a source frame have distinct targets), its
  // mirror update in loop 2 stays a read-modify-write by the same wave, in program order; the self pair (j, j) of the frame's own edges
  // updates the diagonal block from both loops, in the register.  Same operations per entry, same order.
  {
    const bool dwave = wave == (p % kRowCls);
    const bool dl = dwave && lane < 36, yl = wave == 0 && lane < 6;
    float dacc = 0.f, yacc = 0.f;
    for (int g0 = ja; g0 < jb; g0 += 64) {
      const int gl = g0 + lane;
      const int jx_l = gl < jb ? pair_ij[2 * gl + 1] - t0 : -1;
      const int cnt = jb - g0 < 64 ? jb - g0 : 64;
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float dv[4], yv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                              // (the loads of four pairs in flight; the sums below stay in pair order)
          const float* pb = pairbuf + (int64_t)(g0 + i0 + u) * kPairStride;
          const bool on = i0 + u < cnt;
          dv[u] = (on && dl) ? pb[r36 * 16 + c36] : 0.f;
          yv4[u] = (on && yl) ? pb[lane * 16 + 12] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u < cnt) {
            const int jx = __builtin_amdgcn_readlane(jx_l, i0 + u);
            if (dl) dacc += dv[u];
            if (yl) yacc -= yv4[u];
            if (jx >= 0 && jx < N && wave == (jx % kRowCls) && lane < 36) {
              const float* pb = pairbuf + (int64_t)(g0 + i0 + u) * kPairStride;
              const float b = pb[r36 * 16 + 6 + c36];
              if (jx == p) dacc -= b;                              // a self pair (j, j): its "off-diagonal" block IS the diagonal one
              else Srow[(int64_t)r36 * n6 + 6 * jx + c36] = 0.f - b;
            }
          }
        }
      }
    }
    for (int q0 = tl0; q0 < tl1; q0 += 64) {
      const int ql_ = q0 + lane;
      const int g_l = ql_ < tl1 ? tgt_list[ql_] : 0;
      const int ix_l = ql_ < tl1 ? pair_ij[2 * g_l] - t0 : -1;
      const int cnt = tl1 - q0 < 64 ? tl1 - q0 : 64;
      for (int i0 = 0; i0 < cnt; i0 += 4) {
        float dv[4], yv4[4];
        int gg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = i0 + u < cnt;
          gg[u] = __builtin_amdgcn_readlane(g_l, on ? i0 + u : 0);
          const float* pb = pairbuf + (int64_t)gg[u] * kPairStride;
          dv[u] = (on && dl) ? pb[(6 + r36) * 16 + 6 + c36] : 0.f;
          yv4[u] = (on && yl) ? pb[(6 + lane) * 16 + 12] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (i0 + u < cnt) {
            const int ix = __builtin_amdgcn_readlane(ix_l, i0 + u);
            if (dl) dacc += dv[u];
            if (yl) yacc += yv4[u];
            if (ix >= 0 && ix < N && wave == (ix % kRowCls) && lane < 36) {
              const float* pb = pairbuf + (int64_t)gg[u] * kPairStride;
              const float b = pb[c36 * 16 + 6 + r36];                               // mirror of block (ix, p)
              if (ix == p) dacc -= b;
              else Srow[(int64_t)r36 * n6 + 6 * ix + c36] -= b;
            }
          }
        }
      }
    }
    // (blocks this row never touches stay zero; the diagonal block and y[p] always exist)
    if (dl) Srow[(int64_t)r36 * n6 + 6 * p + c36] = dacc;
    if (yl) y[6 * p + lane] = yacc;
  }
  // ---- Schur terms: every source frame f that has a block with pose p -- the sources of the target list, and j itself (its
  //      self block; a pair (j, j) of self edges is in the target list too) -- in ascending f
  GT(1);
#ifdef GBA_TRACE
  int nblk_ = 0;
  const unsigned long long wstart_ = wall_clock64();
#endif
  int q = tl0, it_ = 0;
  bool self_done = !own;
  // the list's metadata -- pair, its source frame, that frame's pair run -- for 64 entries at a time in the lanes (three round trips per
  // 64 frames instead of three per frame), handed out with readlane
  int qbase = tl0 - 64, m_g = -1, m_f = 0x7fffffff, m_g0 = 0, m_g1 = 0;
  // the next source frame with patches: the frame, its pair run and, if it comes from the list, its pair with p
  auto advance = [&](int& f, int& ga_, int& g0, int& g1) -> bool {
    while (q < tl1 || !self_done) {
      if (q < tl1 && q - qbase >= 64) {
        qbase = q;
        const int qi_ = q + lane;
        m_g = qi_ < tl1 ? tgt_list[qi_] : -1;
        m_f = m_g >= 0 ? pair_ij[2 * m_g] : 0x7fffffff;
        const int mfr = m_f - f0;
        const bool fin = m_g >= 0 && mfr >= 0 && mfr < n_frames;
        m_g0 = fin ? run_lo[mfr] : 0; m_g1 = fin ? run_lo[mfr + 1] : 0;
      }
      const int ql = __builtin_amdgcn_readfirstlane(q < tl1 ? q - qbase : 0);
      const int gq = q < tl1 ? __builtin_amdgcn_readlane(m_g, ql) : -1;
      const int fq = gq >= 0 ? __builtin_amdgcn_readlane(m_f, ql) : 0x7fffffff;
      if (!self_done && j <= fq) { f = j; ga_ = (fq == j) ? gq : -1; self_done = true; if (fq == j) ++q; g0 = ja; g1 = jb; }
      else { f = fq; ga_ = gq; ++q; g0 = __builtin_amdgcn_readlane(m_g0, ql); g1 = __builtin_amdgcn_readlane(m_g1, ql); }
      const int fr_ = f - f0;
      if (fr_ < 0 || fr_ >= n_frames) continue;
      return true;
    }
    return false;
  };
  // the target poses of 64 of a frame's blocks (its pairs g0 .. g1, then its self block), one per lane, -1 beyond the last
  auto fetch = [&](int f, int g0, int g1, int b0) -> int {
    const int bl_ = b0 + lane, P_ = g1 - g0;
    return bl_ <= P_ ? (bl_ < P_ ? pair_ij[2 * (g0 + bl_) + 1] : f) - t0 : -1;
  };
  // Round 5 (last session): the frame loop is software pipelined by one frame -- the NEXT frame is picked and the targets of its
  // blocks are requested before this frame's tiles run, so that a frame no longer starts with a memory round trip of its own (the
  // rows of the loop-closure targets walk ~100 source frames, profiles/README.md round 5).
  int f = 0, ga_ = -1, g0 = 0, g1 = 0;
  bool have = advance(f, ga_, g0, g1);
  int pbl0 = have ? fetch(f, g0, g1, 0) : -1;
  while (have) {
    GT(2 + it_); ++it_;
    int nf_ = 0, nga_ = -1, ng0_ = 0, ng1_ = 0;
    const bool nhave = advance(nf_, nga_, ng0_, ng1_);
    const int npbl0 = nhave ? fetch(nf_, ng0_, ng1_, 0) : -1;
    const int fr = f - f0;
    const int P = g1 - g0;
    const float* Qf = Q + (int64_t)fr * M;
    const float* Uf = U + (int64_t)fr * M;
    // the (at most two) blocks of f whose pose is p: its pair (f, j) if any, and the self block when f == j
    for (int which = 0; which < 2; ++which) {
      const float* Ea;
      if (which == 0) { if (ga_ < 0) continue; Ea = Ecol + (int64_t)ga_ * M * 6; }
      else { if (f != j) continue; Ea = Eself + (int64_t)fr * M * 6; }
      // f's blocks that are THIS wave's (column pose mod #waves), in ascending block order: the lanes fetch the targets of 64 blocks at
      // once and the wave walks the set bits of a ballot (as a scalar loop over all P + 1 blocks every wave paid a dependent global
      // round trip per block just to find out that it was not its own).  The products run on the matrix core, two of the wave's
      // blocks per tile:  D (16 x 16) += A (16 x 4) B (4 x 16) over the M patch slots, four per v_mfma_f32_16x16x4_f32 -- rows 0..5 of A =
      // e_a of the slot, columns 0..5 / 6..11 of B = Q e_b of the first / second block, column 12 (wave 0 only) = Q u: the right-hand
      // side comes with it.  An exact f32 fma chain in slot order; every entry of the row is still written by one wave in program
      // order.  (One lane per slot with 36 accumulators and 36 wave reductions per block took 7.5 us per block, tools/gba_trace.sh.)
      const int li = lane & 15, lk = lane >> 4;
      auto tile = [&](const float* Eb0, int pb0, const float* Eb1, int pb1, bool rhs) {
        const float* Eb = li < 6 ? Eb0 : (li < 12 ? Eb1 : nullptr);
        const int cj = li < 6 ? li : li - 6;
        const bool rl = rhs && li == 12;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        // the row's current values travel with the operands (the read-modify-write's read is not a round trip of its own)
        // (the lane's up to four entries D[4 lk + r][li], r < nr, lie one row of S -- or one element of y -- apart: one pointer + a stride)
        float cur[4] = {0.f, 0.f, 0.f, 0.f};
        const int nr = lk == 0 ? 4 : (lk == 1 ? 2 : 0);            // rows 4 lk + r < 6
        float* d0 = nullptr;
        if (nr) {
          if (li < 6) { if (Eb0) d0 = Srow + (int64_t)(4 * lk) * n6 + 6 * pb0 + li; }
          else if (li < 12) { if (Eb1) d0 = Srow + (int64_t)(4 * lk) * n6 + 6 * pb1 + (li - 6); }
          else if (rl) d0 = y + 6 * p + 4 * lk;
        }
        const int64_t dstride = li < 12 ? n6 : 1;
        {
          // (loaded by every lane, from somewhere readable when the lane has no entry: straight-line code, so that these loads and the
          //  operands' below are issued back to back -- behind a branch each the compiler waited for them group by group)
          const float* cs = d0 ? d0 : Srow;
          const int64_t cst = d0 ? dstride : 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) cur[r] = cs[(r < nr ? r : 0) * cst];
        }
        // operands: the blocks are stored component-major ([6][M]), so a lane's operands of FOUR chain steps are one 16-byte load: in
        // step (u, c) the four k-lanes of the tile take the slots 16 u + 4 lk + c.  48 slots per trip: 9 loads in flight, then 12
        // MFMAs.  (Slot-major blocks meant 72 four-byte gathers per tile and wave; a CU's 16 waves push them through one texture
        // path: that, not the arithmetic, was what a block cost.)
        const bool vec = (M & 3) == 0;
        auto ld4 = [&](const float* row, int s_, bool on) -> f4 {
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (on) {
            if (vec) v = *reinterpret_cast<const f4*>(row + s_);
            else {
#pragma unroll
              for (int c = 0; c < 4; ++c) if (s_ + c < M) v[c] = row[s_ + c];
            }
          }
          return v;
        };
        const float* arow = Ea + (int64_t)(li < 6 ? li : 0) * M;
        const float* brow = Eb ? Eb + (int64_t)cj * M : (rl ? Uf : nullptr);
        // M = 96 (default.yaml's PATCHES_PER_FRAME): all 18 operand loads of the tile in flight at once, then the 24 MFMAs in the same
        // order -- one memory round trip per tile instead of one per 48 slots
        if (M == 96) {
          // (no zeroing of the lanes that carry no operand: row i of D depends on row i of A only and column j on column j of B only, and
          //  the rows / columns such lanes feed are never stored -- they only need an address that can be read: row 0 of e_a, Q for B)
          f4 a4[6], b4[6], q4[6];
          const float* bs = brow ? brow : Qf;
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int s_ = 16 * u + 4 * lk;
            a4[u] = *reinterpret_cast<const f4*>(arow + s_);
            q4[u] = *reinterpret_cast<const f4*>(Qf + s_);
            b4[u] = *reinterpret_cast<const f4*>(bs + s_);
          }
          __builtin_amdgcn_sched_barrier(0);                         // (all 18 + 4 loads in flight before the first MFMA: the scheduler otherwise
                                                                     //  sinks two thirds of them between the MFMAs to save registers)
#pragma unroll
          for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][c], q4[u][c] * b4[u][c], acc, 0, 0, 0);
        } else
        for (int s0 = 0; s0 < M; s0 += 48) {
          f4 a4[3], b4[3], q4[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int s_ = s0 + 16 * u + 4 * lk;
            const bool on = s_ < M;
            a4[u] = ld4(arow, s_, on && li < 6);
            q4[u] = ld4(Qf, s_, on);
            b4[u] = ld4(brow, s_, on && brow != nullptr);
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][c], q4[u][c] * b4[u][c], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)                                  // acc[r] = D[4 lk + r][li]
          if (d0 && r < nr) d0[r * dstride] = cur[r] - acc[r];
      };
      bool rhs_due = wave == 0;                                      // right-hand side: y[p] -= sum_slot Q u e_a, once per (f, which)
      for (int b0 = 0; b0 <= P; b0 += 64) {
        const int bl_ = b0 + lane;
        const int pbl = b0 == 0 ? pbl0 : fetch(f, g0, g1, b0);
        unsigned long long todo = __ballot(bl_ <= P && pbl >= 0 && pbl < N && (pbl % kRowCls) == wave);
        while (todo) {
          const int bit0 = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
          todo &= todo - 1;
          const int bA = b0 + bit0, pA = __builtin_amdgcn_readlane(pbl, bit0);
          const float* EA = bA < P ? Ecol + (int64_t)(g0 + bA) * M * 6 : Eself + (int64_t)fr * M * 6;
          const float* EB = nullptr;
          int pB = 0;
          if (todo) {                                                // a second block for the same tile -- unless it is the same pose
            const int bit1 = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);     // (a self pair + the self block)
            const int pC = __builtin_amdgcn_readlane(pbl, bit1);
            if (pC != pA) {
              todo &= todo - 1;
              const int bB = b0 + bit1;
              pB = pC;
              EB = bB < P ? Ecol + (int64_t)(g0 + bB) * M * 6 : Eself + (int64_t)fr * M * 6;
            }
          }
#ifdef GBA_TRACE
          nblk_ += EB ? 2 : 1;
#endif
          tile(EA, pA, EB, pB, rhs_due);
          rhs_due = false;
        }
      }
      if (rhs_due) tile(nullptr, 0, nullptr, 0, true);
    }
    f = nf_; ga_ = nga_; g0 = ng0_; g1 = ng1_; pbl0 = npbl0; have = nhave;
  }
  GT(90);
#ifdef GBA_TRACE
  if (lane == 0 && wg_ < 1024) { gba_wave_buf[wg_][threadIdx.x >> 6][0] = (unsigned int)(wall_clock64() - wstart_); gba_wave_buf[wg_][threadIdx.x >> 6][1] = nblk_; }
  __syncthreads();
  if (threadIdx.x == 0 && wg_ < 4096) { gba_wg_buf[wg_][1] = wall_clock64(); gba_wg_buf[wg_][2] = it_; }
  if (threadIdx.x == 0 && part_ == 0 && p == N / 2) gba_trace_buf[95] = it_;
#endif
}

// dZ = Q (u - sum_blocks e_block . dX[pose(block)]) per (frame, slot) that owns edges; then retractions
__global__ __launch_bounds__(128) void gba_retr_kernel(float* __restrict__ poses, float* __restrict__ patches,
                                                       const int32_t* __restrict__ pair_ij,
                                                       const int32_t* __restrict__ n_pairs, const int32_t* __restrict__ kx,
                                                       const int32_t* __restrict__ n_patches, const float* __restrict__ Q,
                                                       const float* __restrict__ U, const float* __restrict__ Ecol,
                                                       const float* __restrict__ Eself, const float* __restrict__ dX,
                                                       const int32_t* __restrict__ run_lo,
                                                       int M, int f0, int n_frames, int t0, int N, int P) {
  const int np = *n_patches;
  const int PP = P * P;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = gt; k < np; k += gridDim.x * blockDim.x) {
    const int patch = kx[k];
    const int f = patch / M, fr = f - f0, slot = patch % M;
    if (fr < 0 || fr >= n_frames) continue;
    // the frame's pair run: run_lo[fr] = first pair of source frame f0 + fr (gba_index_kernel: the same lower bounds this kernel
    // used to search for itself, two dependent 12-step binary searches per patch)
    const int ga = run_lo[fr], gb = run_lo[fr + 1];
    float s = 0.f;
    const int ix = f - t0;
    if (ix >= 0 && ix < N) {
      const float* e = Eself + (int64_t)fr * 6 * M + slot;
#pragma unroll
      for (int r = 0; r < 6; ++r) s += e[(int64_t)r * M] * dX[6 * ix + r];
    }
    // four pairs in flight: their targets and E entries first, then the targets' steps; the products are added in pair order as before
    // (pair by pair the loop was two dependent round trips for each of a frame's ~30-40 pairs)
    for (int g0 = ga; g0 < gb; g0 += 4) {
      int jx4[4];
      float e4[4][6], d4[4][6];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int g = g0 + v < gb ? g0 + v : ga;
        jx4[v] = g0 + v < gb ? pair_ij[2 * g + 1] - t0 : -1;
        const float* e = Ecol + (int64_t)g * 6 * M + slot;
#pragma unroll
        for (int r = 0; r < 6; ++r) e4[v][r] = e[(int64_t)r * M];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bool on = jx4[v] >= 0 && jx4[v] < N;
#pragma unroll
        for (int r = 0; r < 6; ++r) d4[v][r] = on ? dX[6 * jx4[v] + r] : 0.f;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (jx4[v] >= 0 && jx4[v] < N) {
#pragma unroll
          for (int r = 0; r < 6; ++r) s += e4[v][r] * d4[v][r];
        }
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

extern "C" size_t dpvo_gba_workspace_bytes(int64_t E, int64_t n_pairs, int64_t n_frames, int M, int64_t n_free) {
  if (E < 0 || n_pairs < 0 || n_frames < 0 || M <= 0 || n_free < 0) return 0;
  GbaWs L;
  gba_layout(E, n_pairs, n_frames, M, n_free, &L);
  return L.total;
}

// Linearise: fills S [6N,6N] (B - E Q E^T, WITHOUT the damping) and y [6N] (v - E Q u); both must be zeroed by the
// caller.  Frames f0 .. f0+n_frames-1 are the source frames that own patches (n_frames*M Q/u slots).
static int gba_linearize_impl(const float* poses, const float* patches, const float* intrinsics, const float* target,
                              const float* weight, float lmbda, const int64_t* ii, const int64_t* jj,
                              const int64_t* kk, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                              int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, float* S, float* y,
                              void* ws, size_t ws_bytes, void* stream, bool reuse_index) {
  if (E <= 0 || P <= 0 || M <= 0 || t1 <= t0 || n_frames <= 0 || n_pairs_h <= 0 || n_patches_h <= 0) return DPVO_E_INVALID;
  if (!poses || !patches || !intrinsics || !target || !weight || !ii || !jj || !kk || !plan || !S || !y || !ws) return DPVO_E_INVALID;
  GbaWs L;
  gba_layout(E, n_pairs_h, n_frames, M, t1 - t0, &L);
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
  // Q, u, Eself: patch slots without an edge must read as zero.  Ecol is NOT cleared (round 5: it was 8-13 MB per call at the bound on
  // the pair count, 24 us of every linearisation): gba_scatter_kernel writes all 6 M entries of every existing pair, and every reader
  // (row kernel, retraction) reaches it through an existing pair's index
  {
    const int64_t n4 = (int64_t)((L.Ecol - L.Q) / 16);          // (every region is 256-byte aligned: gba_layout)
    const int64_t zg = (n4 + 255) / 256;
    hipLaunchKernelGGL(gba_zero_kernel, dim3((unsigned)(zg < 1 ? 1 : (zg > 2048 ? 2048 : zg))), dim3(256), 0, st, (f4*)(w + L.Q), n4);
  }
  int32_t* run_lo = (int32_t*)(w + L.run_lo);
  int32_t* tgt_off = (int32_t*)(w + L.tgt_off);
  int32_t* tgt_cnt = (int32_t*)(w + L.tgt_cnt);
  int32_t* tgt_list = (int32_t*)(w + L.tgt_list);
  const unsigned pair_grid = (unsigned)(n_pairs_h < 65535 ? n_pairs_h : 65535);
  hipLaunchKernelGGL(ba_pair_kernel, dim3(pair_grid), dim3(128), 0, st, poses, patches, intrinsics, target, weight, kk,
                     plan + PL.perm_p, plan + PL.pair_off, plan + PL.pair_ij, n_pairs, pairbuf, edgebuf, P);
  // (the index structures depend on the plan and the frame ranges only: a later Gauss-Newton iteration of the same call reuses them)
  if (!reuse_index)
    hipLaunchKernelGGL(gba_index_kernel, dim3(1), dim3(1024), 0, st, plan + PL.pair_ij, n_pairs, f0, n_frames, t0, N, run_lo,
                       tgt_off, tgt_cnt, tgt_list);
  {
    const int n_patch_blocks = (int)((n_patches_h + 255) / 256);
    const GbaSP A = {kk, plan + PL.perm_p, plan + PL.pair_off, n_pairs, plan + PL.perm_k, plan + PL.patch_off, plan + PL.kx, n_patches,
                     edgebuf, Ecol, Q, U, Eself, lmbda, M, f0, n_frames, n_patch_blocks};
    hipLaunchKernelGGL(gba_scatter_patch_kernel, dim3((unsigned)n_patch_blocks + (pair_grid + 1) / 2), dim3(256), 0, st, A);
  }
  {
    const int split = N <= 64 ? 4 : (N <= 128 ? 2 : 1);
    const unsigned nwg = (unsigned)((N * split + 7) / 8 * 8);       // (a multiple of 8: the XCD-major numbering of the kernel is a bijection on it)
    hipLaunchKernelGGL(gba_row_kernel, dim3(nwg), dim3(64 * kRowWaves), 0, st, plan + PL.pair_ij, n_pairs, run_lo, tgt_off, tgt_list,
                       pairbuf, Q, U, Ecol, Eself, M, f0, n_frames, t0, N, split, S, y);
  }
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

extern "C" int dpvo_gba_linearize(const float* poses, const float* patches, const float* intrinsics, const float* target,
                                  const float* weight, float lmbda, const int64_t* ii, const int64_t* jj,
                                  const int64_t* kk, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                                  int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, float* S, float* y,
                                  void* ws, size_t ws_bytes, void* stream) {
  return gba_linearize_impl(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, plan, n_patches_h, n_pairs_h, E, P, M, f0,
                            n_frames, t0, t1, S, y, ws, ws_bytes, stream, false);
}

extern "C" int dpvo_gba_relinearize(const float* poses, const float* patches, const float* intrinsics, const float* target,
                                    const float* weight, float lmbda, const int64_t* ii, const int64_t* jj,
                                    const int64_t* kk, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                                    int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, float* S, float* y,
                                    void* ws, size_t ws_bytes, void* stream) {
  return gba_linearize_impl(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, plan, n_patches_h, n_pairs_h, E, P, M, f0,
                            n_frames, t0, t1, S, y, ws, ws_bytes, stream, true);
}

extern "C" int dpvo_gba_retract(float* poses, float* patches, const int32_t* plan, int64_t n_patches_h, int64_t n_pairs_h,
                                int64_t E, int P, int M, int f0, int n_frames, int t0, int t1, const float* dX, void* ws,
                                size_t ws_bytes, void* stream) {
  if (E <= 0 || P <= 0 || M <= 0 || t1 <= t0 || n_frames <= 0 || n_pairs_h <= 0 || n_patches_h <= 0) return DPVO_E_INVALID;
  if (!poses || !patches || !plan || !dX || !ws) return DPVO_E_INVALID;
  GbaWs L;
  gba_layout(E, n_pairs_h, n_frames, M, t1 - t0, &L);
  if (ws_bytes < L.total) return DPVO_E_WORKSPACE;
  dpvo_plan_layout_t PL;
  dpvo_plan_layout(E, &PL);
  char* w = (char*)ws;
  const int N = t1 - t0;
  const int64_t work = n_patches_h > N ? n_patches_h : N;
  hipLaunchKernelGGL(gba_retr_kernel, dim3((unsigned)((work + 127) / 128)), dim3(128), 0, (hipStream_t)stream, poses,
                     patches, plan + PL.pair_ij, plan + PL.counts + 1, plan + PL.kx, plan + PL.counts + 0,
                     (const float*)(w + L.Q), (const float*)(w + L.u), (const float*)(w + L.Ecol),
                     (const float*)(w + L.Eself), dX, (const int32_t*)(w + L.run_lo), M, f0, n_frames, t0, N, P);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

#ifdef GBA_TRACE
extern "C" int dpvo_debug_gba_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gba_trace_buf), sizeof(gba_trace_buf)) == hipSuccess ? 0 : 1;
}
extern "C" int dpvo_debug_gba_wave_trace(unsigned int* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gba_wave_buf), sizeof(gba_wave_buf)) == hipSuccess ? 0 : 1;
}
extern "C" int dpvo_debug_gba_wg_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gba_wg_buf), sizeof(gba_wg_buf)) == hipSuccess ? 0 : 1;
}
#endif
