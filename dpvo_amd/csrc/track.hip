// track.hip -- one tracked frame with the host off the critical path.
//
//   dpvo_keyframe_step : DPVO.keyframe (reference dpvo/dpvo.py:266-310) decided and executed ON THE DEVICE: flow test ->
//                        decision, edges of the dropped keyframe removed + ids renumbered + ring buffers shifted, edges that
//                        left the optimisation window moved to the inactive store, the rest compacted in order.  The host
//                        only reads an 8-word result (decision + counts), one frame later if it likes.
//   dpvo_frame_update  : DPVO.update (dpvo.py:328-360) + that keyframe step as ONE C-ABI call: the launch sequence the
//                        Python front-end used to pace with ~10 ctypes calls (plan, reproject, correlation, update
//                        operator, BA, point cloud, flow test, removal) is issued back to back.
//
// Why: the reference resolves the keyframe decision with two `.item()` synchronisations and then rebuilds every edge tensor
// with boolean masks (dpvo.py:223-238); rounds 1-2 of this library kept a host mirror of the index arrays so that the masks
// could be computed while the GPU was busy, which still cost ~0.25 ms of numpy per frame and paced the start of the next frame.
// Here the masks never leave the device.
//
// Ordering guarantees of the compaction (bit-exact bookkeeping, tests/test_gpu_dpvo.py): both removals of the reference are
// order preserving, so ONE stable partition of the renumbered list into (kept, inactive, dropped) yields the same active and
// inactive lists as the reference's two passes.
#include "common.h"
#include "se3_dev.h"

namespace {

enum { RES_DECISION = 0, RES_KEEP = 1, RES_REM = 2, RES_E = 3, RES_OVERFLOW = 4, RES_LONG_RANGE = 5 };

struct Renum { int drop; int k; int M; };
__device__ __forceinline__ void renumber(const Renum& R, int64_t& i, int64_t& j, int64_t& k) {
  if (R.drop) {                                   // dpvo.py:281-283
    if (i > R.k) { k -= R.M; i -= 1; }
    if (j > R.k) j -= 1;
  }
}

// ---- 1. decision + stable partition in one launch of 1024-edge chunks (kf_decide_kernel below).  (One workgroup walking all
//         ~5 x 10^4 index triples took 95 us; a count launch + a select launch 41 us.)
__device__ __forceinline__ int kf_decide(const dpvo_keyframe_step_t& a, const float* flow4) {
  int d;
  if (a.forced >= 0) d = a.forced ? 1 : 0;
  else {
    // m = motionmag(i, j) + motionmag(j, i); drop iff m / 2 < KEYFRAME_THRESH (dpvo.py:266-271).  A direction without
    // edges gives mean = NaN in the reference (mean of an empty tensor) and the comparison is false.
    const float s0 = flow4[0], n0 = flow4[1], s1 = flow4[2], n1 = flow4[3];
    const float nan = __builtin_nanf("");
    const float m = (n0 > 0.f ? s0 / n0 : nan) + (n1 > 0.f ? s1 / n1 : nan);
    d = (m / 2.f < a.keyframe_thresh) ? 1 : 0;
  }
  const int k = a.n - a.keyframe_index;
  if (k < 1 || k >= a.n) d = 0;                   // (no such keyframe: cannot happen once the tracker is initialised)
  return d;
}
// class of edge e under decision d: 0 dropped with its keyframe, 1 kept, 2 moved to the inactive store
__device__ __forceinline__ int kf_class(const dpvo_keyframe_step_t& a, int d, int64_t e) {
  const Renum R = {d, a.n - a.keyframe_index, a.M};
  int64_t i = a.ii[e], j = a.jj[e], k = a.kk[e];
  if (R.drop && (i == R.k || j == R.k)) return 0;
  renumber(R, i, j, k);
  const int n_after = a.n - (d ? 1 : 0);
  bool rem = (k / a.M) < n_after - a.removal_window;         // ix[kk] < n - REMOVAL_WINDOW (dpvo.py:305): index_ rows hold their frame number
  if (a.loop_closure && rem) {
    rem = !(((j - i) > 30) && (j > (n_after - a.optimization_window)));      // dpvo.py:307-308
    if (!rem) return 3;       // kept ONLY because it serves loop closure: a long-range edge (source frame < n - REMOVAL_WINDOW).  While
  }                           // such edges are active the next update() runs the global BA (dpvo.py:348): the host needs their count
  return rem ? 2 : 1;
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifdef KF_TRACE
// instrumentation build (tools/kf_trace.sh): thread 0 of the first and the last chunk stamps the 100 MHz wall clock at every phase
__device__ unsigned long long kf_trace_buf[2][16];
#define KFT(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) kf_trace_buf[blockIdx.x == 0 ? 0 : 1][i] = wall_clock64(); } while (0)
#else
#define KFT(i) do {} while (0)
#endif
constexpr int KF_CHUNK = 1024;
constexpr unsigned long long KF_WAIT_TICKS = 100000000ull;   // a look-back gives up after ONE SECOND of the 100 MHz wall clock (an
                                                           // iteration count was 55-65 ms, which a time-sliced or profiled device can exceed)
constexpr int KF_HDR = 4;         // scratch behind the 8 result words: [0] generation, [1] workgroups done, then per chunk {kept, removed, long-range, flag}
// ONE launch (rounds 2-3: a count kernel and a select kernel, 21 + 20 us): every 1024-edge chunk classifies its edges, publishes its
// three counts with a flag (= the call's generation number, so nothing has to be cleared between calls) and reads the counts of the
// chunks BEFORE it (look-back: a workgroup only ever waits for lower-numbered ones, which were dispatched before it), writes its
// indices in order; the last chunk holds the totals and writes the result record.  The last workgroup to finish bumps the generation.
// F.poses != NULL: the flow test itself (DPVO.motionmag for (k - 1, k + 1) and back, dpvo.py:257-269) runs HERE, in every chunk's
// workgroup redundantly (one scan of the ~500 pair records + ~190 edge flows: less than the launch it saves), block 0 leaves the
// sums in F.out for the record and the host.  Same code, same reduction tree as dpvo_motionmag: same bits in every block.
__global__ __launch_bounds__(KF_CHUNK) void kf_decide_kernel(const dpvo_keyframe_step_t a, int32_t* __restrict__ sc, const MotionPlanArgs F) {
  __shared__ int wsum[3][16];
  __shared__ int pre[4];            // look-back prefixes {kept, removed, long-range}; [3]: a look-back gave up
  __shared__ float fl4[8];
  __shared__ __attribute__((aligned(16))) int rec[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int blk = blockIdx.x, last = (int)gridDim.x - 1;
  KFT(0);
  const int gen = __hip_atomic_load(&sc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const float* flow4 = a.flow4;
  if (tid < 4) pre[tid] = 0;
  float statv = 0.f;                              // (the plan counters of the host record: fetched here, used at the very end)
  if (F.poses) {
    if (blk == last && tid >= 4 && tid < 8) statv = (float)F.n_pairs[tid - 5];
    motionmag_plan_body(F.poses, F.patches, F.intr, F.kk, F.perm_p, F.pair_off, F.pair_ij, F.n_pairs, F.P, F.qi, F.qj, F.beta, fl4,
                        blk == 0 ? F.status : nullptr, F.flow);
    __syncthreads();
    if (blk == 0 && tid < 4) { F.out[tid] = fl4[tid]; __threadfence(); }     // (the last chunk's host copy reads them)
    flow4 = fl4;
  }
  KFT(1);
  const int d = kf_decide(a, flow4);
  const int64_t e = (int64_t)blk * KF_CHUNK + tid;
  int cls = e < a.E ? kf_class(a, d, e) : 0;
  const unsigned long long bl = __ballot(cls == 3);
  if (cls == 3) cls = 1;                          // (a long-range edge is a kept edge)
  const unsigned long long bk = __ballot(cls == 1), br = __ballot(cls == 2);
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int pk = __popcll(bk & below), pr = __popcll(br & below);
  if (lane == 0) { wsum[0][wv] = __popcll(bk); wsum[1][wv] = __popcll(br); wsum[2][wv] = __popcll(bl); }
  __syncthreads();
  KFT(2);
  int32_t* mine = sc + KF_HDR + 4 * blk;
  int tk = 0, tr = 0, tl = 0;
  if (tid == 0) {
    for (int w = 0; w < 16; ++w) { tk += wsum[0][w]; tr += wsum[1][w]; tl += wsum[2][w]; }
    mine[0] = tk; mine[1] = tr; mine[2] = tl;
    __hip_atomic_store(&mine[3], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (blk == 0 && tid == 64 && a.delta_pose) {    // (not thread 0: every other chunk waits for this chunk's counts)
    // dP = SE3(poses[k]) * SE3(poses[k-1]).inv() (dpvo.py:276), with lietorch's store / load between the two ops
    const int k = a.n - a.keyframe_index;
    if (k >= 1 && k < a.n) {
      float tmp[7];
      store_pose(tmp, se3_inv(load_pose(a.poses + 7 * (int64_t)(k - 1))));
      store_pose(a.delta_pose, se3_mul(load_pose(a.poses + 7 * (int64_t)k), load_pose(tmp)));
    }
  }
  KFT(3);
  // look-back: thread t collects chunk t, t + 1024, ... (< blk)
  {
    int ak = 0, ar = 0, al = 0;
    for (int b = tid; b < blk; b += KF_CHUNK) {
      const int32_t* o = sc + KF_HDR + 4 * b;
      // (bounded: a scratch that was cleared or reused between calls, or a launch that died half way, leaves flags this call never
      //  sees -- ~0.2 s of polling, then the chunk gives up and the record says so (RES_OVERFLOW bit 1) instead of hanging the queue)
      unsigned long long t_wait = 0;
      while (__hip_atomic_load(&o[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        const unsigned long long now = wall_clock64();
        if (t_wait == 0) t_wait = now;
        if (now - t_wait > KF_WAIT_TICKS) { pre[3] = 1; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      ak += __hip_atomic_load(&o[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ar += __hip_atomic_load(&o[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      al += __hip_atomic_load(&o[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wv * 64 < blk) {                          // (waves without a chunk to collect have nothing to add)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { ak += __shfl_xor(ak, o); ar += __shfl_xor(ar, o); al += __shfl_xor(al, o); }
      if (lane == 0) { atomicAdd(&pre[0], ak); atomicAdd(&pre[1], ar); atomicAdd(&pre[2], al); }
    }
  }
  KFT(4);
  __syncthreads();
  KFT(5);
  int ok = pre[0], orr = pre[1];
  for (int w = 0; w < wv; ++w) { ok += wsum[0][w]; orr += wsum[1][w]; }
  if (cls == 1) { a.keep_idx[ok + pk] = (int32_t)e; if (a.keep_rows) a.keep_rows[ok + pk] = e; }
  if (cls == 2 && orr + pr < a.inac_room) a.rem_idx[orr + pr] = (int32_t)e;
  KFT(6);
  if (blk == last && wv == 0) {
    // the record: 8 result words; the host's copy (pinned memory) takes the 8 words in front of them along when host_words = 16
    // (flow sums + plan counters: block 0 writes them to device memory, this chunk has the same values at hand).  One 16-byte
    // store per lane -- as 16 volatile 4-byte stores from one thread the copy took 11 us of the frame's tail.  From here on the
    // record is final (the gather that follows only executes it); the HOST starts on the next frame behind an event recorded
    // right after this kernel, one point cloud + gather earlier than the end of the call.
    if (tid == 0) {
      int nrem = pre[1] + tr, ovf = 0;
      if (nrem > a.inac_room) { nrem = (int)a.inac_room; ovf = 1; }
      if (pre[3]) ovf |= 2;                       // (the look-back gave up: counts and index lists are not to be trusted)
      rec[8 + RES_DECISION] = d; rec[8 + RES_KEEP] = pre[0] + tk; rec[8 + RES_REM] = nrem; rec[8 + RES_E] = (int32_t)a.E;
      rec[8 + RES_OVERFLOW] = ovf; rec[8 + RES_LONG_RANGE] = pre[2] + tl; rec[8 + 6] = rec[8 + 7] = 0;
    }
    if (tid < 8 && a.result_host && a.host_words == 16) {
      if (F.poses) rec[tid] = tid < 4 ? __float_as_int(fl4[tid]) : __float_as_int(statv);
      else rec[tid] = (a.result - 8)[tid];
    }
    wave_lds_fence();
    typedef int i4v __attribute__((ext_vector_type(4)));
    if (tid >= 2 && tid < 4) reinterpret_cast<i4v*>(a.result)[tid - 2] = reinterpret_cast<const i4v*>(rec)[tid];
    if (a.result_host) {
      const int first = a.host_words == 16 ? 0 : 2;
      if (tid >= first && tid < 4) reinterpret_cast<i4v*>(a.result_host)[tid - first] = reinterpret_cast<const i4v*>(rec)[tid];
    }
  }
  if (tid == 0) {
    KFT(7);
    const int done = __hip_atomic_fetch_add(&sc[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == last) {                           // everybody has read the generation and every flag: next call, next number
      __hip_atomic_store(&sc[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sc[0], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  KFT(8);
}
#ifdef KF_TRACE
}  // namespace
extern "C" int dpvo_debug_kf_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kf_trace_buf), sizeof(kf_trace_buf)) == hipSuccess ? 0 : 1;
}
namespace {
#endif

// ---- 2. the two gathers (kept -> spare set, incl. the 1.5 KB hidden-state rows; inactive -> tail of the inactive store)
__device__ __forceinline__ void kf_shift_body(const dpvo_keyframe_step_t& a, int blk, int blocks_per_ring);
__global__ __launch_bounds__(256) void kf_gather_kernel(const dpvo_keyframe_step_t a, int nblk_keep, int nblk_gather, int blocks_per_ring) {
  if ((int)blockIdx.x >= nblk_gather) { kf_shift_body(a, (int)blockIdx.x - nblk_gather, blocks_per_ring); return; }
  const Renum R = {a.result[RES_DECISION], a.n - a.keyframe_index, a.M};
  const bool keep_job = (int)blockIdx.x < nblk_keep;
  const int64_t bid = keep_job ? blockIdx.x : blockIdx.x - nblk_keep, nblk = keep_job ? nblk_keep : nblk_gather - nblk_keep;
  const int64_t gt = bid * 256 + threadIdx.x, gs = nblk * 256;
  const int32_t* __restrict__ idx = keep_job ? a.keep_idx : a.rem_idx;
  const int64_t n = keep_job ? a.result[RES_KEEP] : a.result[RES_REM];
  int64_t* oi = keep_job ? a.ii_b : a.ii_inac; int64_t* oj = keep_job ? a.jj_b : a.jj_inac; int64_t* ok = keep_job ? a.kk_b : a.kk_inac;
  float* ot = keep_job ? a.target_b : a.target_inac; float* ow = keep_job ? a.weight_b : a.weight_inac;
  for (int64_t t = gt; t < n; t += gs) {
    const int64_t s = idx[t];
    int64_t i = a.ii[s], j = a.jj[s], k = a.kk[s];
    renumber(R, i, j, k);
    oi[t] = i; oj[t] = j; ok[t] = k;
    ot[2 * t] = a.target[2 * s]; ot[2 * t + 1] = a.target[2 * s + 1];
    ow[2 * t] = a.weight[2 * s]; ow[2 * t + 1] = a.weight[2 * s + 1];
  }
  if (keep_job && a.net_b) {
    const int dq = a.D / 4;
    for (int64_t q = gt; q < n * dq; q += gs) {
      const int64_t t = q / dq;
      const int c = (int)(q - t * dq);
      reinterpret_cast<f4*>(a.net_b)[t * dq + c] = reinterpret_cast<const f4*>(a.net)[(int64_t)idx[t] * dq + c];
    }
  }
}

// ---- 3. ring buffers: slot i <- slot i + 1 for i = k .. n - 2 (dpvo.py:289-299), only if the keyframe was dropped.  A thread
//         owns the same 16-byte piece of every slot, so the in-place chain needs no synchronisation.
__device__ __forceinline__ void kf_shift_body(const dpvo_keyframe_step_t& a, int blk, int blocks_per_ring) {
  if (!a.result[RES_DECISION]) return;
  const int r = blk / blocks_per_ring;
  if (r >= a.n_ring) return;
  const dpvo_ring_t G = a.ring[r];
  const int64_t units = G.slot_bytes / 16;
  const int k = a.n - a.keyframe_index;
  typedef unsigned int u4v __attribute__((ext_vector_type(4)));
  for (int64_t u = (int64_t)(blk - r * blocks_per_ring) * 256 + threadIdx.x; u < units; u += (int64_t)blocks_per_ring * 256) {
    for (int i = k; i < a.n - 1; ++i) {
      const int64_t src = G.ring ? (i + 1) % G.ring : (i + 1), dst = G.ring ? i % G.ring : i;
      const u4v v = *reinterpret_cast<const u4v*>((const char*)G.base + src * G.slot_bytes + u * 16);
      *reinterpret_cast<u4v*>((char*)G.base + dst * G.slot_bytes + u * 16) = v;
    }
  }
  // (slot sizes that are not a multiple of 16 bytes: the tail, byte by byte, by the first block of the ring)
  const int64_t tail0 = units * 16;
  if (blk == r * blocks_per_ring)
    for (int64_t b = tail0 + threadIdx.x; b < G.slot_bytes; b += 256)
      for (int i = k; i < a.n - 1; ++i) {
        const int64_t src = G.ring ? (i + 1) % G.ring : (i + 1), dst = G.ring ? i % G.ring : i;
        ((char*)G.base)[dst * G.slot_bytes + b] = ((const char*)G.base)[src * G.slot_bytes + b];
      }
}

inline unsigned blocks_for(int64_t n, int64_t cap) {
  int64_t g = cdiv64(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

namespace {
void kf_decide_launch(const dpvo_keyframe_step_t* a, const MotionPlanArgs& F, hipStream_t st);
void kf_apply_launch(const dpvo_keyframe_step_t* a, hipStream_t st);
}
extern "C" int dpvo_keyframe_step(const dpvo_keyframe_step_t* a, void* stream) {
  if (!a || a->E < 0 || a->E >= (1ll << 31) || a->M <= 0 || a->D <= 0 || (a->D % 4) || a->n_ring < 0 || a->n_ring > 8) return DPVO_E_INVALID;
  if (a->result_host && a->host_words != 8 && a->host_words != 16) return DPVO_E_INVALID;
  if (((uintptr_t)a->result & 15) || ((uintptr_t)a->result_host & 15)) return DPVO_E_INVALID;      // (the record is stored 16 bytes at a time)
  if (!a->result || !a->keep_idx || !a->rem_idx || (a->forced < 0 && !a->flow4) || !a->poses) return DPVO_E_INVALID;
  if (a->E > 0 && (!a->ii || !a->jj || !a->kk || !a->target || !a->weight || !a->ii_b || !a->jj_b || !a->kk_b || !a->target_b ||
                   !a->weight_b || !a->ii_inac || !a->jj_inac || !a->kk_inac || !a->target_inac || !a->weight_inac))
    return DPVO_E_INVALID;
  if ((a->net == nullptr) != (a->net_b == nullptr)) return DPVO_E_INVALID;
  for (int r = 0; r < a->n_ring; ++r)
    if (!a->ring[r].base || a->ring[r].slot_bytes <= 0 || a->ring[r].ring < 0) return DPVO_E_INVALID;
  kf_decide_launch(a, MotionPlanArgs{}, (hipStream_t)stream);
  kf_apply_launch(a, (hipStream_t)stream);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}

namespace {
// decision, counts, index lists, result record (device + host copy).  F.poses != NULL: the flow test runs inside the count kernel.
void kf_decide_launch(const dpvo_keyframe_step_t* a, const MotionPlanArgs& F, hipStream_t st) {
  // the look-back scratch lives behind the 8 result words: `result` has room for 8 + 4 + 4 * ceil(E / 1024) ints, zeroed ONCE by the caller
  int32_t* sc = a->result + 8;
  const unsigned chunks = (unsigned)(a->E > 0 ? cdiv64(a->E, KF_CHUNK) : 1);
  hipLaunchKernelGGL(kf_decide_kernel, dim3(chunks), dim3(KF_CHUNK), 0, st, *a, sc, F);
}
// the two gathers (kept edges -> spare set, removed edges -> inactive store) and the ring shifts of a dropped keyframe
void kf_apply_launch(const dpvo_keyframe_step_t* a, hipStream_t st) {
  {
    // the two gathers and the ring shifts are independent of each other: one launch (the shift blocks leave at once unless
    // the keyframe was dropped)
    unsigned gk = 0, gr = 0, bpr = 0;
    if (a->E > 0) { gk = blocks_for(a->net_b ? a->E * (a->D / 4) : a->E, 2048); gr = blocks_for(a->E, 16); }
    if (a->n_ring > 0) {
      int64_t mx = 0;
      for (int r = 0; r < a->n_ring; ++r) mx = a->ring[r].slot_bytes > mx ? a->ring[r].slot_bytes : mx;
      bpr = blocks_for(mx / 16, 512);
    }
    if (gk + gr + bpr * (unsigned)a->n_ring > 0)
      hipLaunchKernelGGL(kf_gather_kernel, dim3(gk + gr + bpr * (unsigned)a->n_ring), dim3(256), 0, st, *a, (int)gk, (int)(gk + gr), (int)bpr);
  }
}
}  // namespace

#ifdef FU_HOST_TRACE
// instrumentation build (tools/fu_host_trace.sh): host time between the steps of dpvo_frame_update, summed over the calls
#include <chrono>
static double fu_ht_sum[16]; static long fu_ht_calls;
static inline double fu_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define HT(i) do { const double t_ = fu_now(); fu_ht_sum[i] += t_ - ht_prev; ht_prev = t_; } while (0)
extern "C" int dpvo_debug_fu_host_trace(double* out) { for (int i = 0; i < 16; ++i) out[i] = fu_ht_calls ? fu_ht_sum[i] / fu_ht_calls : 0.0; for (int i = 0; i < 16; ++i) fu_ht_sum[i] = 0; fu_ht_calls = 0; return 0; }
#else
#define HT(i) do {} while (0)
#endif
extern "C" int dpvo_frame_update(const dpvo_frame_update_t* a, void* stream) {
#ifdef FU_HOST_TRACE
  double ht_prev = fu_now(); ++fu_ht_calls;
#endif
  if (!a || !a->upd || !a->result_dev) return DPVO_E_INVALID;
  const dpvo_keyframe_step_t& K = a->kf;
  const int64_t E = K.E;
  const int n = K.n, M = K.M;
  // (K.loop_closure: the removal keeps long-range edges per dpvo.py:307-308 and reports their count; the CALLER must not use this
  //  entry while such edges are active -- update() then owes a global BA over active + inactive edges, dpvo.py:348 -- nor in a frame
  //  that appends loop-closure edges: the window plan below assumes sources within REMOVAL_WINDOW and targets within PATCH_LIFETIME)
  if (E <= 0 || n < 2 || M <= 0 || a->P <= 0) return DPVO_E_INVALID;
  if (!a->poses || !a->patches || !a->intrinsics || !a->points || !a->ix || !a->gmap || !a->fmap1 || !a->fmap2 || !a->imap ||
      !a->coords || !a->corr || !a->delta || !a->plan || !a->ws_plan || !a->ws_update || !a->ws_ba || !a->net)
    return DPVO_E_INVALID;
  if (a->net_rows && (a->n_kept < 0 || a->n_kept > E)) return DPVO_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  int32_t* plan_clear = nullptr; int64_t plan_clear_n = 0;
#define STEP(...) do { rc = (__VA_ARGS__); if (rc) return rc; } while (0)
  // ---- the new frame's state (patch gathers, motion model, depth median, pyramid level 1, its edges): dpvo.py:400-459
  if (a->fs) {
    if (a->fs_auto) {
      // frame f = n - 1 is the one being added (the caller has already counted it): its ring slots, index rows and edges
      if (K.n_ring != 8 || !a->index_map) return DPVO_E_INVALID;
      dpvo_frame_state_t* F = a->fs;
      const int64_t f = n - 1;
      auto slot = [&](int r, int64_t i) { const dpvo_ring_t& G = K.ring[r]; return (char*)G.base + (G.ring ? i % G.ring : i) * G.slot_bytes; };
      F->colors_slot = slot(0, f); F->patches_slot = (float*)slot(2, f);
      F->intrinsics_slot = F->intrinsics ? (float*)slot(3, f) : nullptr;
      F->imap_slot = slot(4, f); F->gmap_slot = slot(5, f); F->fmap = slot(6, f); F->fmap2_slot = slot(7, f);
      F->index_row = const_cast<int64_t*>(a->ix) + (f + 1) * M; F->index_map = a->index_map + (f + 1);
      F->poses = a->poses; F->mm_n = (int)f; F->patches_all = a->patches; F->md_n = (int)f;
      F->ii = const_cast<int64_t*>(K.ii); F->jj = const_cast<int64_t*>(K.jj); F->kk = const_cast<int64_t*>(K.kk); F->ix = a->ix;
      const int r = a->patch_lifetime, jlo = n - r > 0 ? n - r : 0;
      const int64_t total = (int64_t)M * ((n - 1 > 0 ? n - 1 : 0) - jlo) + (int64_t)M * (n - jlo);      // dpvo.py:362-375
      if (total > E) return DPVO_E_INVALID;
      F->E0 = E - total; F->ap_n = n; F->ap_r = r; F->D = K.D;
      F->net = a->net_rows ? nullptr : a->net;            // (rows >= n_kept read as zeros when the compaction is deferred)
      F->frame_next = f + 1; F->m_next = a->m;            // index_[f + 1] = f + 1, index_map_[f + 1] = m (after the increment)
      F->M = M; F->P = a->P; F->h = a->H0; F->w = a->W0;
    }
    // ev_enc (the side stream's "encoders done"): only the feature gathers and the pyramid's level 1 need the encoders' outputs,
    // so the state stores, the new edges, the plan and the reprojection go first and the stream waits just in front of part 2
    if (a->ev_enc) {
      // (part 1 also clears the counters of the plan's window build: one launch less in front of the correlation)
      if (!(a->plan_stream && a->ev_plan_fork && a->ev_plan_done) &&
          dpvo_plan_window_counters(E, a->ws_plan, a->ws_plan_bytes, &plan_clear, &plan_clear_n) != DPVO_OK) plan_clear = nullptr;
      STEP(dpvo_frame_state_part_clear(a->fs, 1, plan_clear, plan_clear ? plan_clear_n : 0, stream));
    } else {
      STEP(dpvo_frame_state(a->fs, stream));
      if (a->ev_fs && hipEventRecord((hipEvent_t)a->ev_fs, st) != hipSuccess) return DPVO_E_INVALID;
    }
  }
  HT(0);     // frame state part 1
  // ---- graph plan: every active edge has its source frame in [n - REMOVAL_WINDOW - 1, n) and its target within
  //      PATCH_LIFETIME frames of it (dpvo.py:305,362-375): counting-sort build over that window, bounds on the group counts
  const int RW = K.removal_window, PL = a->patch_lifetime;
  const int64_t nf = n < RW + 2 ? n : RW + 2;
  const int64_t ub_p = nf * M, ub_g = nf * (2 * PL + 2);
  const int64_t np_ub = ub_p < E ? ub_p : E, ng_ub = ub_g < E ? ub_g : E;
  const int64_t flo = n - (RW + PL + 3) > 0 ? n - (RW + PL + 3) : 0;
  // Nothing in front of the update operator's second kernel reads the plan, so with a second stream at hand its five launches
  // (~36 us of dependent-launch latency, 45 workgroups each) run beside the reprojection, part 2 of the frame state and the start
  // of the correlation kernel instead of in front of them
  const bool plan_aside = a->plan_stream && a->ev_plan_fork && a->ev_plan_done;
  if ((a->plan_stream || a->ev_plan_fork || a->ev_plan_done) && !plan_aside) return DPVO_E_INVALID;
  // (with the flow test's frame pair (k - 1, k + 1), k = n - KEYFRAME_INDEX, extracted on the way: dpvo_plan_layout_t.flow)
  // On the compute stream the reprojection (independent of the plan, same edges) rides in the plan's histogram launch and the
  // counters were cleared by part 1 of the frame state: the front of the frame is 6 launches instead of 8 -- a launch costs ~5 us
  // here whatever it does (win_zero_kernel: 6 145 stores, 4.9 us)
  bool reprojected = false;
  auto build_plan = [&](void* pst, bool with_reproject) -> int {
    int r = dpvo_plan_build_window_job(K.ii, K.jj, K.kk, E, a->plan, a->ws_plan, a->ws_plan_bytes, flo, n - flo, flo * M, (n - flo) * (int64_t)M,
                                       n - K.keyframe_index - 1, n - K.keyframe_index + 1, plan_clear != nullptr && pst == stream,
                                       with_reproject ? a->poses : nullptr, a->patches, a->intrinsics, a->coords, a->P, pst);
    if (r == DPVO_OK) reprojected = with_reproject;
    if (r == DPVO_E_UNSUPPORTED)
      r = dpvo_plan_build_ranged(K.ii, K.jj, K.kk, E, a->plan, a->ws_plan, a->ws_plan_bytes, a->n_buffer, a->n_buffer * M, pst);
    return r;
  };
  // aside: only the fork point is marked here; the plan's launches are ISSUED behind the correlation kernel's -- the front of the
  // frame is short kernels that the GPU finishes faster than the host enqueues them, so whatever the host issues in front of the
  // correlation launch delays it, on whichever stream it runs
  if (plan_aside) { if (hipEventRecord((hipEvent_t)a->ev_plan_fork, st) != hipSuccess) return DPVO_E_INVALID; }
  else STEP(build_plan(stream, true));
  HT(1);     // fork record / plan on the compute stream
  // ---- reproject -> correlation -> update operator (dpvo.py:331-343)
  if (!reprojected) STEP(dpvo_reproject(a->poses, a->patches, a->intrinsics, K.ii, K.jj, K.kk, a->coords, E, a->P, 1, stream));
  HT(2);     // reproject
  if (a->fs && a->ev_enc) {
    if (hipStreamWaitEvent(st, (hipEvent_t)a->ev_enc, 0) != hipSuccess) return DPVO_E_INVALID;
    // the encoders wrote the frame's feature map where the host EXPECTED its ring slot to be; a keyframe dropped in between moved
    // the slot down by one (dpvo.py:289-299 shifts the ring, the new frame lands at n - 1)
    if (a->fmap_spec && a->fmap_spec != a->fs->fmap &&
        hipMemcpyAsync(const_cast<void*>(a->fs->fmap), a->fmap_spec, (size_t)a->H0 * a->W0 * 128 * 2 /* [H0][W0][128] f16, the map dpvo_corr_pyramid_forward reads below */, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return DPVO_E_INVALID;
    STEP(dpvo_frame_state_part(a->fs, 2, stream));
    if (a->ev_fs && hipEventRecord((hipEvent_t)a->ev_fs, st) != hipSuccess) return DPVO_E_INVALID;
  }
  HT(3);     // encoder join + part 2
  if (a->ev[0] && hipEventRecord((hipEvent_t)a->ev[0], st) != hipSuccess) return DPVO_E_INVALID;
  STEP(dpvo_corr_pyramid_forward(a->gmap, a->fmap1, a->fmap2, a->coords, K.kk, K.jj, nullptr, a->corr, 896, E, 128, a->P,
                                 (int64_t)a->pmem * M, a->mem, a->H0, a->W0, a->H1, a->W1, 3, stream));
  if (a->ev[1] && hipEventRecord((hipEvent_t)a->ev[1], st) != hipSuccess) return DPVO_E_INVALID;
  if (a->ev[2] && a->ev[2] != a->ev[1] && hipEventRecord((hipEvent_t)a->ev[2], st) != hipSuccess) return DPVO_E_INVALID;   // (one record serves both)
  HT(4);     // correlation (+ profiling events)
  if (plan_aside) {
    hipStream_t pst = (hipStream_t)a->plan_stream;
    if (hipStreamWaitEvent(pst, (hipEvent_t)a->ev_plan_fork, 0) != hipSuccess) return DPVO_E_INVALID;
    HT(5);   // plan stream waits for the fork
    STEP(build_plan(a->plan_stream, false));
    HT(6);   // plan launches
    if (hipEventRecord((hipEvent_t)a->ev_plan_done, pst) != hipSuccess) return DPVO_E_INVALID;
    HT(7);   // plan-done record
    if (hipStreamWaitEvent(st, (hipEvent_t)a->ev_plan_done, 0) != hipSuccess) return DPVO_E_INVALID;
    HT(8);   // compute stream waits for the plan
  }
  float* net = a->net;
  float* target = const_cast<float*>(K.target);
  float* weight = const_cast<float*>(K.weight);
  STEP(dpvo_update_forward_fused_rows(a->upd, net, a->net_rows, a->n_kept, a->imap, K.kk, (int64_t)a->pmem * M, a->corr, 896, a->plan, np_ub, ng_ub,
                                      a->coords, a->P, net, a->delta, weight, target, E, a->ws_update, a->ws_update_bytes, stream));
  HT(9);     // update operator
  if (a->ev[3] && hipEventRecord((hipEvent_t)a->ev[3], st) != hipSuccess) return DPVO_E_INVALID;
  if (a->ev_update_done && hipEventRecord((hipEvent_t)a->ev_update_done, st) != hipSuccess) return DPVO_E_INVALID;
  // ---- two local BA iterations over the last ba_window poses (dpvo.py:351-354), point cloud (:358-360)
  int t0 = n - a->ba_window;
  if (t0 < 1) t0 = 1;
  STEP(dpvo_ba(a->poses, a->patches, a->intrinsics, target, weight, a->lmbda, K.ii, K.jj, K.kk, a->plan, np_ub, ng_ub, E, a->P, t0, n,
               a->iterations, nullptr, a->ws_ba, a->ws_ba_bytes, stream));
  HT(10);    // update-done records + BA
  // ---- point cloud, and the keyframe's flow test between frames k - 1 and k + 1 (dpvo.py:266-269), in one launch; then everything
  //      else of the keyframe step on the device
  // Order (round 4): flow test + decision + index lists + RESULT RECORD first, the event the host waits for right behind them,
  // then the point cloud and the gathers / ring shifts that execute the decision -- the host reads the record and enqueues the
  // next frame while those two kernels (~36 us) still run, which is what used to be a bubble of the same length.  (The point cloud
  // has to precede the ring shifts: it indexes the frames as update() left them, dpvo.py:358-360.)
  const int k = n - K.keyframe_index;
  dpvo_keyframe_step_t kf = K;
  kf.flow4 = a->result_dev;
  kf.result = reinterpret_cast<int32_t*>(a->result_dev + 8);
  kf.poses = a->poses;
  kf.host_words = 16;               // the host's copy carries the flow sums and the plan counters in front of the 8 result words
  if ((kf.result_host && kf.host_words != 16) || !kf.keep_idx || !kf.rem_idx) return DPVO_E_INVALID;
  {
    dpvo_plan_layout_t PLy;
    dpvo_plan_layout(E, &PLy);
    const MotionPlanArgs F = {a->poses, a->patches, a->intrinsics, K.kk, a->plan + PLy.perm_p, a->plan + PLy.pair_off, a->plan + PLy.pair_ij,
                              a->plan + PLy.counts + 1, a->P, k - 1, k + 1, a->mm_beta, a->result_dev, a->result_dev + 4, a->plan + PLy.flow};
    kf_decide_launch(&kf, F, st);
  }
  if (a->ev_record && hipEventRecord((hipEvent_t)a->ev_record, st) != hipSuccess) return DPVO_E_INVALID;
  STEP(dpvo_point_cloud(a->poses, a->patches, a->intrinsics, a->ix, a->points, a->m, a->P, stream));
  kf_apply_launch(&kf, st);
  // ---- LOOP_CLOSURE: the candidate test PatchGraph.edges_loop will ask for at the start of the next frame (patchgraph.py:56-72), on the
  //      state this call leaves behind (poses after the BA, rings after the shifts), for the frame count the decision leaves behind
  if (a->loop_out) {
    STEP(dpvo_loop_flow_next(a->poses, a->patches, a->intrinsics, a->ix, kf.result + RES_DECISION, n, K.removal_window, K.keyframe_index,
                             a->loop_freq, a->loop_max_age, M, a->P, 0.5f, a->loop_out, stream));
    if (a->ev_loop && hipEventRecord((hipEvent_t)a->ev_loop, st) != hipSuccess) return DPVO_E_INVALID;
  }
  HT(11);    // keyframe step + record + point cloud
  DPVO_LAUNCH_CHECK();
#undef STEP
  return DPVO_OK;
}
