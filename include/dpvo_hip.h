/*
 * dpvo_hip.h -- C ABI of libdpvo_hip.so: the MI355X (gfx950) implementation of DPVO's per-frame
 * hot path (altcorr correlation -> update operator -> fastba bundle adjustment).
 *
 * This header is the drop-in boundary.  Each entry replaces one function the reference binds
 * through pybind11 (`cuda_corr`, `cuda_ba`, `lietorch_backends`; /root/reference/setup.py:13-36) or
 * one torch-level stage of `Update.forward`; the reference interface is cited per entry as
 * file:line relative to the reference checkout.  INTEGRATION.md shows the Python-side stub a
 * maintainer adds to dpvo/altcorr/correlation.py, dpvo/fastba/ba.py and dpvo/lietorch/group_ops.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator); the
 *     library never frees or retains a pointer after the call returns;
 *   - `stream` is a hipStream_t passed as void* (NULL = legacy default stream); calls are
 *     asynchronous with respect to the host and re-entrant per stream, no global mutable state;
 *   - return value: 0 = ok, <0 = invalid argument / unsupported configuration (DPVO_E_*),
 *     >0 = a hipError_t from the launch; nothing ever calls exit();
 *   - index tensors are int64 exactly as the reference passes them (torch.long);
 *   - workspaces: `*_workspace_bytes` returns the size the matching call needs in `ws`.
 */
#ifndef DPVO_HIP_H
#define DPVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPVO_OK 0
#define DPVO_E_INVALID (-1)      /* bad argument (null pointer, negative size, ...)            */
#define DPVO_E_UNSUPPORTED (-2)  /* configuration outside what this entry implements           */
#define DPVO_E_WORKSPACE (-3)    /* workspace too small                                        */

#define DPVO_F16 0
#define DPVO_F32 1

/* ABI version: bumped on ANY change of a signature, a struct layout or the exported set (round 3 changed all three without a
 * bump: ADVICE r3).  The Python binding refuses a library whose version differs from the one it was written against. */
#define DPVO_ABI_VERSION 8
int dpvo_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * altcorr  (replaces cuda_corr: dpvo/altcorr/correlation.cpp:57-63)
 * ---------------------------------------------------------------------------------------------- */

/* cuda_corr.forward(fmap1, fmap2, coords, ii, jj, radius) -- correlation.cpp:28-35,58 ->
 * corr_cuda_forward, correlation_kernel.cu:193-233 (kernel :82-136 + bilinear :221-232), batch 1.
 *   fmap1  [N1,C,P,P]   strides s1 = {n,c,i,j} in elements (any layout)
 *   fmap2  [N2,C,H2,W2] strides s2 = {n,c,h,w} in elements (any layout)
 *   coords [E,2,P,P] f32 contiguous, multiplied by coord_scale before use (DPVO.corr passes coords/4
 *          for level 1, dpvo/dpvo.py:206)
 *   us,vs  [E] int64 indices into fmap1 / fmap2
 *   out    [E, D-1 (y), D-1 (x), P, P] contiguous, dtype = feature dtype; the reference returns
 *          out.permute(0,1,3,2,4,5) of exactly this tensor (:232); D = 2*radius+2.
 * Accumulates in f32 (the reference accumulates in the feature dtype), rounds once on store. */
int dpvo_corr_forward(const void* fmap1, const int64_t* s1, const void* fmap2, const int64_t* s2,
                      const float* coords, float coord_scale, const int64_t* us, const int64_t* vs,
                      void* out, int dtype, int64_t E, int C, int P, int64_t N1, int64_t N2, int H2, int W2,
                      int radius, void* stream);

/* Fused two-level correlation == DPVO.corr (dpvo/dpvo.py:200-207): both altcorr.corr calls, the
 * bilinear blend, the permute, torch.stack(...,-1) and .view(1,E,-1) in one MFMA kernel.
 *   gmap   [N1, P*P, C]      f16 channels-last  (logical [N1,C,P,P])
 *   fmap0  [N2, H0, W0, C]   f16 channels-last  (logical [N2,C,H0,W0]), level 0 (coords/1)
 *   fmap1  [N2, H1, W1, C]   f16 channels-last, level 1 (coords/4)
 *   coords [E,2,P,P] f32; us,vs [E] int64, non-negative, < 2^31: template / frame indices, taken modulo N1 / N2 by the
 *          kernel (the ring-buffer reduction of dpvo.py:202-203; already reduced indices are unchanged)
 *   order  [E] int32 or NULL: processing order of edges (an L2 / XCD locality hint; any permutation
 *          of 0..E-1; results do not depend on it)
 *   out    [E, ld_out] f16, ld_out >= 2*49*P*P, feature index ((((x*7+y)*P+i0)*P+j0)*2+level)
 *          exactly as the reference's stacked view; columns >= 882 of a padded row are zeroed.
 * Supported: C == 128, P == 3, radius == 3 (the only configuration DPVO uses, net.py:53,
 * dpvo.py:205-206); anything else returns DPVO_E_UNSUPPORTED and callers use dpvo_corr_forward. */
int dpvo_corr_pyramid_forward(const void* gmap, const void* fmap0, const void* fmap1, const float* coords,
                              const int64_t* us, const int64_t* vs, const int32_t* order, void* out,
                              int64_t ld_out, int64_t E, int C, int P, int64_t N1, int64_t N2, int H0, int W0,
                              int H1, int W1, int radius, void* stream);

/* cuda_corr.patchify_forward(net, coords, radius) -- correlation.cpp:46-49,61 ->
 * correlation_kernel.cu:16-47,286-306.  Integer-floor gather of (2R+2)^2 windows, zeros when OOB.
 *   net [C,H,W] strides sn={c,h,w}; coords [M,2] f32; out [M,C,D,D] contiguous (D=2R+2). */
int dpvo_patchify_forward(const void* net, const int64_t* sn, const float* coords, void* out, int dtype,
                          int64_t M, int C, int H, int W, int radius, void* stream);

/* altcorr.patchify(net, coords, radius, mode='bilinear') -- dpvo/altcorr/correlation.py:51-68: the gather above
 * fused with the Python-side bilinear blend (dx,dy of the centroid).  out [M,C,d,d], d = 2R+1. */
int dpvo_patchify_bilinear(const void* net, const int64_t* sn, const float* coords, void* out, int dtype,
                           int64_t M, int C, int H, int W, int radius, void* stream);

/* ------------------------------------------------------------------------------------------------
 * projective ops  (replaces the lietorch + elementwise chain of pops.transform)
 * ---------------------------------------------------------------------------------------------- */

/* DPVO.reproject (dpvo/dpvo.py:209-213) == pops.transform(SE3(poses), patches, intrinsics, ii,jj,kk)
 * (dpvo/projective_ops.py:53-68: iproj :19-29, Gj*Gi^-1 via lietorch se3.h:36-56, act4, proj with
 * Z clamp 0.1 :43) followed by permute(0,1,4,2,3).
 *   poses [NP,7] f32 (t, q_xyzw), patches [NK,3,P,P] f32, intrinsics [NP,4] f32, ii,jj,kk [E] int64
 *   coords out [E,2,P,P] f32.  Also the exported-but-unused cuda_ba.reproject (ba.cpp:48-56) maps
 * here with clamp_z=0 (ba_cuda.cu:379-429 divides by raw Z and uses intrinsics[0] for all frames). */
int dpvo_reproject(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                   const int64_t* jj, const int64_t* kk, float* coords, int64_t E, int P, int clamp_z,
                   void* stream);

/* pops.flow_mag (projective_ops.py:120-130) reduced as DPVO.motionmag uses it (dpvo.py:257-264):
 * writes per-edge mean-over-patch flow to flow[E] and validity count to nvalid[E]. */
int dpvo_flow_mag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                  const int64_t* jj, const int64_t* kk, float beta, float* flow, float* valid, int64_t E, int P,
                  void* stream);

/* PatchGraph.edges_loop's candidate test (dpvo/patchgraph.py:56-72) in one launch: for every pair (target frame j in [j0, j0 + n_j),
 * source frame f in [i0, i0 + n_i)) the flow magnitude of the reference -- pops.flow_mag(beta) on the CENTRE pixel of the M patches
 * k = f M + p with i = ix[k], val = nvalid > 0.5, sum(flow val) / max(sum val, 1) when more than 0.75 M are valid, +inf otherwise.
 *   ix [.] int64 (PatchGraph.index_ flattened), patches [NK,3,P,P];  flow_mag out [n_j * n_i] f32, target-major (the order of the
 *   reference's flatmeshgrid), device memory or pinned host memory (the caller's one read-back). */
int dpvo_loop_flow(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, int64_t j0, int64_t n_j,
                   int64_t i0, int64_t n_i, int M, int P, float beta, float* flow_mag, void* stream);

/* DPVO.motionmag(i,j) + DPVO.motionmag(j,i) (dpvo/dpvo.py:257-264,269): out4 = {sum_ij, n_ij, sum_ji, n_ji} of
 * the per-edge pixel-mean flow over the edges i->j and j->i (mean = sum/n; 0/0 = NaN like torch's empty mean). */
int dpvo_motionmag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                   const int64_t* jj, const int64_t* kk, const int32_t* plan /* of (ii,jj,kk), or NULL: full scan */,
                   int64_t E, int P, int64_t i, int64_t j, float beta, float* out4, void* stream);
/* The same, and the plan's counters ride along: status4 = (n_patches, n_pairs, 0, flag) as floats, flag != 0 when
 * dpvo_plan_build_window saw a frame / patch id outside the window it was promised (groups were then clamped): the caller's
 * only per-frame read-back also tells it that its bounds were wrong.  Needs a plan. */
int dpvo_motionmag_status(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                          const int64_t* jj, const int64_t* kk, const int32_t* plan, int64_t E, int P, int64_t i, int64_t j,
                          float beta, float* out4, float* status4, void* stream);
/* dpvo_point_cloud and dpvo_motionmag_status (plan variant) in ONE launch: the tail of a frame (dpvo.py:358-360, 266-269). */
int dpvo_point_cloud_motionmag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, float* points,
                               int64_t m, const int64_t* kk, const int32_t* plan, int64_t E, int P, int64_t i, int64_t j, float beta,
                               float* out4, float* status4, void* stream);

/* pops.point_cloud centre pixel (projective_ops.py:115-117, dpvo.py:358-360): points[m,3]. */
int dpvo_point_cloud(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                     float* points, int64_t m, int P, void* stream);

/* PatchGraph.normalize (patchgraph.py:84-90; caller DPVO.__run_global_BA, dpvo.py:321): s = mean depth of the first n frames'
 * patches (patches [.,M,3,P,P], channel 2, all P*P entries), depths /= s, translations *= s, poses[:n] = poses[:n] * poses[0]^-1.
 * Two launches (the reference: a strided torch reduction + seven elementwise / lietorch launches); the mean is summed in f64 in
 * a fixed order.  scratch: dpvo_normalize_scratch_bytes() bytes, 8-byte aligned; scratch[0] = s afterwards (for the caller's
 * PatchGraph.delta rescaling, patchgraph.py:88-89).  The point cloud refresh of patchgraph.py:92-94 is dpvo_point_cloud. */
size_t dpvo_normalize_scratch_bytes(void);
int dpvo_normalize(float* poses, float* patches, int n, int M, int P, float* scratch, void* stream);

/* lietorch_backends.{inv,mul,act4,expm,logm}(group_id=3, ...) -- lietorch.cpp:286-316, SE3 forward
 * only (lietorch_gpu.cu:21-30,47-56,73-82,101-110,225-236; se3.h, so3.h).  f32, n elements. */
int dpvo_se3_inv(const float* X, float* Y, int64_t n, void* stream);
int dpvo_se3_mul(const float* X, const float* Y, float* Z, int64_t n, void* stream);
int dpvo_se3_act4(const float* X, const float* p, float* q, int64_t n, void* stream);
int dpvo_se3_exp(const float* a, float* X, int64_t n, void* stream);
int dpvo_se3_log(const float* X, float* a, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * patch-graph plan  (replaces torch::_unique + fastba.neighbors host round trips)
 * ---------------------------------------------------------------------------------------------- */

/* Device-resident index structures of an edge list (ii,jj,kk), all int32, laid out in one
 * caller-owned buffer of dpvo_plan_bytes(E) bytes.  Built entirely on the device (rocPRIM radix sort),
 * no host synchronisation.  Offsets (in int32 elements) are returned by dpvo_plan_layout.
 *   perm_k[E]     edge ids sorted by (kk, jj, edge id)           -- == per-patch stable_sort by jj (ba.cpp:80-82)
 *   ku[E]         rank of kk[e] among the sorted unique kk      -- torch::_unique inverse (ba_cuda.cu:447-449)
 *   kx[E]         sorted unique patch ids (first n_patches valid) -- torch::_unique values
 *   patch_off[E+1] CSR offsets of perm_k per unique patch
 *   ix[E], jx[E]  fastba.neighbors output (ba.cpp:59-97), -1 = none
 *   perm_p[E]     edge ids sorted by (ii, jj, edge id)
 *   pu[E]         rank of (ii,jj) among the sorted unique pairs  -- SoftAgg groups of agg_ij (net.py:88, blocks.py:41)
 *   pair_off[E+1] CSR offsets of perm_p per unique pair
 *   pair_ij[2E]   (i,j) of each unique pair
 *   counts[4]     {n_patches, n_pairs, 0, 0}
 */
typedef struct dpvo_plan_layout_t {
  int64_t perm_k, ku, kx, patch_off, ix, jx, perm_p, pu, pair_off, pair_ij, counts, total_ints;
  int64_t flow;   /* DPVO_PLAN_FLOW_INTS ints, 16-byte aligned: {qi, qj, n_ij, n_ji} + 2 x DPVO_PLAN_FLOW_CAP patch ids -- the edges of ONE
                     frame pair in both directions, in edge order, extracted while the plan is built (dpvo_plan_build_window_flow) so
                     that the keyframe flow test (dpvo.py:257-270) starts from a list instead of three dependent look-ups; every other
                     builder writes qi = qj = -1 (readers then walk pair_ij / pair_off / perm_p as before: same edges, same order) */
} dpvo_plan_layout_t;
#define DPVO_PLAN_FLOW_CAP 256
#define DPVO_PLAN_FLOW_INTS (4 + 2 * DPVO_PLAN_FLOW_CAP)

int dpvo_plan_layout(int64_t E, dpvo_plan_layout_t* layout);
size_t dpvo_plan_workspace_bytes(int64_t E);
int dpvo_plan_build(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan,
                    void* ws, size_t ws_bytes, void* stream);
/* Same, with bounds on the index ranges supplied by the caller (every ii, jj < n_frames, every kk < n_patch_ids, e.g.
 * BUFFER_SIZE and BUFFER_SIZE * PATCHES_PER_FRAME of dpvo/config.py:6): the composite sort keys then fit 32 bits and the
 * radix sorts visit only the bits that can be set.  Indices outside the bounds give an undefined (but memory-safe)
 * grouping.  n_frames = n_patch_ids = 0 means "unknown" (64-bit keys). */
int dpvo_plan_build_ranged(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                           size_t ws_bytes, int64_t n_frames, int64_t n_patch_ids, void* stream);
/* Same plan when every edge's frame ids (ii, jj) lie in [frame_lo, frame_lo + n_frames_win) and its patch id (kk) in
 * [patch_lo, patch_lo + n_patches_win): one stable counting-sort pass per ordering instead of two device radix sorts
 * (6 launches instead of 14).  Limits: n_frames_win^2 <= 2048 and n_patches_win <= 4096, else DPVO_E_UNSUPPORTED (use
 * dpvo_plan_build_ranged).  Ids outside the promised window are clamped (memory safe, plan contents then unspecified) and
 * reported in counts[3] = 1. */
int dpvo_plan_build_window(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                           size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo, int64_t n_patches_win,
                           void* stream);
/* dpvo_plan_build_window + the flow-test edge list of the frame pair (qi, qj) / (qj, qi) in the plan's `flow` region (qi < 0: none) */
int dpvo_plan_build_window_flow(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                                size_t ws_bytes, int64_t frame_lo, int64_t n_frames_win, int64_t patch_lo, int64_t n_patches_win,
                                int64_t qi, int64_t qj, void* stream);

/* Same plan when the ids are bounded by the tracker's frame count only (every ii, jj in [0, n_frames), every kk in [0, n_patch_ids)):
 * what DPVO.update builds while long-range loop-closure edges are active and what DPVO.__run_global_BA builds over all active +
 * inactive edges (dpvo.py:312-326,345-354; the reference re-derives these groupings with torch.unique / fastba.neighbors per call).
 * One bin per patch id and per (i, j) pair in memory, integer-atomic histogram, scan, placement, then every element ranks itself
 * inside its bin by (jj, edge) resp. edge: the stable sorts' result bit for bit, 6 launches instead of the radix build's 16-48.
 * Limits: n_frames <= 2048, n_patch_ids <= 2^22, E < 2^30, else DPVO_E_UNSUPPORTED (use dpvo_plan_build_ranged); the work is
 * proportional to sum over bins of (bin size)^2, i.e. meant for graphs whose patches / frame pairs hold tens to hundreds of edges.
 * Ids outside the promise are clamped (memory safe, plan contents then unspecified) and reported in counts[3] = 1.
 * ws: dpvo_plan_wide_workspace_bytes(E, n_frames, n_patch_ids) bytes (0 = unsupported ranges). */
size_t dpvo_plan_wide_workspace_bytes(int64_t E, int64_t n_frames, int64_t n_patch_ids);
int dpvo_plan_build_wide(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int32_t* plan, void* ws,
                         size_t ws_bytes, int64_t n_frames, int64_t n_patch_ids, void* stream);

/* cuda_ba.neighbors(kk, jj) -- ba.cpp:59-97,187: int64 outputs for API parity (device resident). */
size_t dpvo_neighbors_workspace_bytes(int64_t E);
int dpvo_neighbors(const int64_t* kk, const int64_t* jj, int64_t* ix, int64_t* jx, int64_t E, void* ws,
                   size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * update operator  (replaces Update.forward, dpvo/net.py:74-92, under autocast: dpvo.py:332)
 * ---------------------------------------------------------------------------------------------- */

/* The update operator of the product library is dpvo_update_forward_fused(_rows) (further down).  Its comparator -- the
 * launch-by-launch composite of generic pieces (dpvo_linear / dpvo_layernorm / dpvo_gather_add / dpvo_heads /
 * dpvo_update_forward) -- lives in libdpvo_hip_cmp.so and is declared in dpvo_hip_cmp.h: a test and measurement partner, not part
 * of the product. */

/* epilogue selectors of dpvo_linear */
#define DPVO_EPI_NONE 0        /* out_f16 = h(acc + bias)                                     */
#define DPVO_EPI_RELU 1        /* out_f16 = relu(h(acc + bias))                               */
#define DPVO_EPI_SIGMOID 2     /* out_f16 = sigmoid(h(acc + bias))                            */
#define DPVO_EPI_RESADD 3      /* res_f32[row] += float(h(acc + bias))  (residual into net)   */
#define DPVO_EPI_GATED 4       /* res_f32[row] += float(h(gate[row] * h(acc + bias)))         */
#define DPVO_EPI_RELU_SIG 5    /* columns < n_split: relu, else sigmoid (fused res.0 | gate.0) */


/* SoftAgg core (dpvo/blocks.py:31-48 with torch_scatter scatter_softmax / scatter_sum, pytorch-scatter
 * 2.1.2): for every group g (CSR off/perm from the plan) and channel c
 *   y[g,c] = sum_e softmax_e(gx[e,c]) * fx[e,c]       (f32 math, one rounding to f16)
 * fg [E, 2*D] f16 holds f(x) in columns [0,D) and g(x) in [D,2D) (one fused GEMM). */
int dpvo_softagg(const void* fg, int64_t ldfg, const int32_t* perm, const int32_t* off, const int32_t* n_groups,
                 int64_t max_groups, void* y, int D, void* stream);


/* ------------------------------------------------------------------------------------------------
 * fastba  (replaces cuda_ba.forward: dpvo/fastba/ba.cpp:32-45,184 -> cuda_ba, ba_cuda.cu:433-582)
 * ---------------------------------------------------------------------------------------------- */

/* Gauss-Newton bundle adjustment over poses [t0,t1) and all patch inverse depths, `iterations`
 * steps, IN PLACE on poses/patches exactly like the reference (ba_cuda.cu:570-577).
 *   poses [NP,7], patches [NK,3,P,P], intrinsics [NP,4] (only row 0 is used, ba_cuda.cu:253-259),
 *   target,weight [E,2] f32, lmbda scalar, ii,jj,kk [E] int64, plan from dpvo_plan_build(ii,jj,kk).
 * Residual/Jacobian/mask semantics: ba_cuda.cu:232-376; damping S += I*(1e-4*S+1) :546,560; depth
 * prior Q=1/(C+lmbda) :519; retractions :157-229.  Deterministic (no float atomics).
 * info (device int32[iterations], may be NULL) receives the Cholesky status per iteration.
 * n_patches_hint / n_pairs_hint: the host's copy of the plan's counts, used only to size the launches exactly
 * (<= 0 = unknown, launches are then sized by E); the device-side counts stay authoritative.
 * Dense Schur path for 6*(t1-t0) <= DPVO_BA_MAX_DIM; larger systems return DPVO_E_UNSUPPORTED
 * (global BA is routed by the host to the block-sparse path). */
#define DPVO_BA_MAX_DIM 120
size_t dpvo_ba_workspace_bytes(int64_t E, int n_free_poses);
int dpvo_ba(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
            float lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, const int32_t* plan,
            int64_t n_patches_hint, int64_t n_pairs_hint, int64_t E, int P, int t0, int t1, int iterations,
            int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * front-end helpers  (callers either side of the hot path, dpvo/dpvo.py:377-473; each replaces a chain of
 * small torch launches -- the frame is launch-bound above ~200 fps)
 * ---------------------------------------------------------------------------------------------- */

/* image = 2*(image/255) - 0.5 (dpvo.py:389): f32 and/or f16 output, n elements. */
int dpvo_normalize_image(const void* img_u8, float* out_f32, void* out_f16, int64_t n, void* stream);
/* colours of the new patches (dpvo.py:404-405 + net.py:143): out_u8 [M,3] (BGR->RGB swap, uint8 truncation). */
int dpvo_patch_colors(const void* img_u8, const float* coords, void* out_u8, int M, int H, int W, void* stream);
/* ring-buffer store (dpvo.py:437-438): fmap [C,h,w] -> channels-last slot [h,w,C] + 4x4 avg-pooled [h/4,w/4,C]. */
int dpvo_store_features(const void* fmap, void* f1_slot, void* f2_slot, int dtype, int C, int h, int w, void* stream);
/* append_factors(edges_forw) + append_factors(edges_back) (dpvo.py:215-221,362-375,458-459) for frame count n:
 * writes kk, jj, ii = ix[kk] at [E0, E0+n_new) and zeroes the new rows of net [.,D] (net may be NULL: skipped); *n_new (host) = count. */
int dpvo_append_edges(int64_t* ii, int64_t* jj, int64_t* kk, float* net, const int64_t* ix, int64_t E0, int n, int M,
                      int r, int D, int64_t* n_new, void* stream);
/* remove_factors compaction (dpvo.py:223-238): out[t] = in[idx[t]] for ii,jj,kk and (optional) net,target,weight. */
int dpvo_gather_edges(const int64_t* idx, int64_t n, const int64_t* ii, const int64_t* jj, const int64_t* kk,
                      const float* net, const float* target, const float* weight, int64_t* oii, int64_t* ojj,
                      int64_t* okk, float* onet, float* otarget, float* oweight, int D, void* stream);
/* Two compactions from the same source arrays in ONE launch (remove_factors, dpvo.py:223-238: the removed edges go to the
 * inactive store, the kept ones are compacted): job a = (idx_a, n_a -> a_*), job b = (idx_b, n_b -> b_*); net / target /
 * weight outputs optional per job; the two output sets must not overlap each other or the sources. */
int dpvo_gather_edges2(const int64_t* idx_a, int64_t n_a, int64_t* a_ii, int64_t* a_jj, int64_t* a_kk, float* a_net,
                       float* a_target, float* a_weight, const int64_t* idx_b, int64_t n_b, int64_t* b_ii, int64_t* b_jj,
                       int64_t* b_kk, float* b_net, float* b_target, float* b_weight, const int64_t* ii, const int64_t* jj,
                       const int64_t* kk, const float* net, const float* target, const float* weight, int D, void* stream);
/* Everything a new frame contributes besides its feature maps, in one launch: Patchifier.forward's gathers
 * (net.py:136-147: gmap = patchify(fmap, coords, 1), imap = patchify(imap, coords, 0), patches = patchify(grid, coords, 1),
 * clr = patchify(image, 4 (coords + 0.5), 0)) and the per-frame state stores of dpvo.py:401-438 (intrinsics / RES,
 * index_, index_map_, patches[:,:,2] = depth, colours as RGB uint8).  fmap [h,w,CF] / imap [h,w,CI] are NHWC f16 (the
 * encoders' output), img_u8 [3,H,W]; coords [M,2] f32 OR the two randint draws xs, ys [M] int64 (net.py:132-133);
 * depth [M] f32.  Slots: gmap [M,3,3,CF] f16 (channels-last), imap [M,CI] f16, patches [M,3,3,3] f32, colours [M,3] u8,
 * intrinsics_slot [4] (may be NULL), index_row [M] = frame_next and *index_map = m_next (may be NULL), coords_out [M,2]
 * (may be NULL).  P must be 3.  Each output group (gmap / imap / patches / colours) is skipped when its slot is NULL, so the
 * state stores (which need no feature map) and the feature gathers can be issued as two launches around other work. */
int dpvo_frame_patches(const void* fmap, const void* imap, const void* img_u8, const float* coords, const int64_t* xs,
                       const int64_t* ys, const float* depth, const float* intrinsics, float res, void* gmap_slot,
                       void* imap_slot, float* patches_slot, void* colors_slot, float* intrinsics_slot,
                       int64_t* index_row, int64_t* index_map, float* coords_out, int M, int h, int w, int H, int W, int CF,
                       int CI, int P, int64_t frame_next, int64_t m_next, void* stream);
/* damped-linear motion model (dpvo.py:410-421): poses[n] = Exp(scale * Log(P[n-1] * P[n-2]^-1)) * P[n-1]. */
int dpvo_motion_model(float* poses, int n, float scale, void* stream);
/* depth initialisation (dpvo.py:430-432): patches[n][:,2] = torch.median(patches[n-3:n,:,2]) (lower median). */
int dpvo_median_depth(float* patches, int n, int M, int P, void* stream);

/* Everything a tracked frame does between the encoders and the plan, as ONE call (the launches are those of the entries
 * named below, in this order; this part of a frame is paced by the host, so one marshalling instead of five matters):
 *   dpvo_frame_patches (always) -> dpvo_motion_model (poses != NULL) -> dpvo_median_depth (patches_all != NULL)
 *   -> dpvo_pool4_nhwc (fmap2_slot != NULL) -> dpvo_append_edges (ii != NULL; n_new receives the number of edges). */
typedef struct {
  const void *fmap, *imap, *img_u8;               /* dpvo_frame_patches */
  const float* coords; const int64_t *xs, *ys; const float *depth, *intrinsics;
  void *gmap_slot, *imap_slot; float* patches_slot; void* colors_slot; float* intrinsics_slot;
  int64_t *index_row, *index_map;
  float* poses;                                   /* dpvo_motion_model(poses, mm_n, mm_scale) */
  float* patches_all;                             /* dpvo_median_depth(patches_all, md_n, M, P) */
  void* fmap2_slot;                               /* dpvo_pool4_nhwc(fmap, fmap2_slot, h, w, CF) */
  int64_t *ii, *jj, *kk; float* net; const int64_t* ix;   /* dpvo_append_edges(..., E0, ap_n, M, ap_r, D, &n_new) */
  int64_t frame_next, m_next, E0, n_new;
  float res, mm_scale;
  int32_t M, h, w, H, W, CF, CI, P, mm_n, md_n, ap_n, ap_r, D;
} dpvo_frame_state_t;
int dpvo_frame_state(dpvo_frame_state_t* p, void* stream);
/* part 0: the same; part 1: only what does not read fmap / imap (coordinate + depth patches, colours, intrinsics, index rows, motion
 * model, depth median, the new edges); part 2: only what does (gmap / imap gathers, pyramid level 1). */
int dpvo_frame_state_part(dpvo_frame_state_t* p, int part, void* stream);


/* Update.forward (dpvo/net.py:74-92) as SEVEN launches of row-tile-resident MFMA kernels (dpvo_amd/csrc/update_fused.hip): a workgroup keeps
 * a tile of edge rows in LDS for a whole chain of Linear layers (LayerNorm, residual, gate and the heads in the epilogues),
 * the weights stream from L2 as pre-packed MFMA fragments.  Same inputs, outputs and precision contract as
 * dpvo_update_forward (results agree to f32 summation order: the k index of chained layers is permuted).
 *   dpvo_update_fused_pack: W [384, ldw] f16 row-major (torch Linear weight) with K columns (a multiple of 16; columns
 *     >= k_valid read as zero) -> the fragment image (dpvo_update_fused_pack_bytes(K) bytes) the kernels consume;
 *     chained = 0 for the layer fed by the correlation rows (corr.0, K = 896), 1 for every other layer (K = 384).
 *   w[i] / b[i]: packed image / f16 bias [384] of layer i (DPVO_UF_*); ln_g / ln_b: corr.3, norm, gru.0, gru.2 (f32);
 *   d_w, w_w [2,384] and d_b, w_b [2] f16 (the heads, feature order). */
enum {
  DPVO_UF_C0 = 0, DPVO_UF_C2, DPVO_UF_C5,                       /* Update.corr.0 / .2 / .5          (net.py:51-59) */
  DPVO_UF_C1_0, DPVO_UF_C1_2, DPVO_UF_C2N_0, DPVO_UF_C2N_2,     /* Update.c1.0 / .2, Update.c2.0 / .2 (net.py:31-39) */
  DPVO_UF_AKK_F, DPVO_UF_AKK_G, DPVO_UF_AKK_H,                  /* Update.agg_kk.f / .g / .h        (blocks.py:36-38) */
  DPVO_UF_AIJ_F, DPVO_UF_AIJ_G, DPVO_UF_AIJ_H,
  DPVO_UF_G0_GATE, DPVO_UF_G0_RES0, DPVO_UF_G0_RES2,            /* Update.gru.1.gate.0 / .res.0 / .res.2 (blocks.py:19-26) */
  DPVO_UF_G1_GATE, DPVO_UF_G1_RES0, DPVO_UF_G1_RES2,            /* Update.gru.3 ...                                 */
  DPVO_UF_NLIN
};
typedef struct {
  const void* w[DPVO_UF_NLIN];
  const void* b[DPVO_UF_NLIN];
  const float* ln_g[4];
  const float* ln_b[4];
  const void *d_w, *d_b, *w_w, *w_b;
  /* tuning knobs of the seven-launch path (no reference counterpart), per call -- the library holds no mutable state:
   * tiling: bit 0 = chain kernels, bit 1 = the correlation-MLP kernel use 64-row tiles with two workgroups per CU instead of
   *   96-row tiles with one; bits 2 / 3 / 4 = the last kernel / the correlation-MLP kernel / the chain kernels run 96-row tiles with
   *   TWELVE waves (three per SIMD, 32 output features per wave) instead of four (one per SIMD, 96 features per wave); bits 3 and 4
   *   take precedence over bits 1 and 0; < 0 = dpvo_update_fused_default_tiling().  start_skew: workgroup b begins (b & 3) * us / 4
   *   microseconds late (0 = off).  Results are bit-identical for every setting. */
  int32_t tiling, start_skew;
} dpvo_update_fused_params_t;
size_t dpvo_update_fused_pack_bytes(int K);
int dpvo_update_fused_pack(const void* W, int64_t ldw, int K, int k_valid, int chained, void* out, void* stream);
/* dpvo_update_forward_fused with the compaction of remove_factors (dpvo.py:223-238, `self.pg.net = self.pg.net[:,~m]`) folded
 * into its first kernel: state row g = net[net_rows[g]] for g < n_kept (ascending `keep` list), zero for g >= n_kept (the edges
 * appended since: append_factors starts them at zero, dpvo.py:219).  net_out may alias net.  net_rows == NULL: as above. */
int dpvo_update_forward_fused_rows(const dpvo_update_fused_params_t* p, const float* net, const int64_t* net_rows, int64_t n_kept,
                                   const void* inp, const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr,
                                   const int32_t* plan, int64_t n_patches_ub, int64_t n_pairs_ub, const float* coords, int P,
                                   float* net_out, float* delta, float* weight, float* target, int64_t E, void* ws,
                                   size_t ws_bytes, void* stream);
size_t dpvo_update_fused_workspace_bytes(int64_t E, int64_t max_groups);
int dpvo_update_fused_default_tiling(void);      /* 13 */
int dpvo_update_forward_fused(const dpvo_update_fused_params_t* params, const float* net, const void* inp,
                              const int64_t* inp_rows, int64_t inp_mod, const void* corr, int64_t ld_corr, const int32_t* plan,
                              int64_t n_patches_ub, int64_t n_pairs_ub, const float* coords, int P, float* net_out,
                              float* delta, float* weight, float* target, int64_t E, void* ws, size_t ws_bytes, void* stream);


/* ------------------------------------------------------------------------------------------------
 * One tracked frame, host off the critical path  (DPVO.update + DPVO.keyframe, dpvo/dpvo.py:266-360)
 * ---------------------------------------------------------------------------------------------- */

/* dpvo_keyframe_step: DPVO.keyframe (dpvo.py:266-310) WITHOUT a host decision.  On the device, in stream order:
 *   1. decision: drop keyframe k = n - keyframe_index iff (s0/n0 + s1/n1) / 2 < keyframe_thresh, with (s0, n0, s1, n1) =
 *      flow4 = the output of dpvo_motionmag for (k - 1, k + 1) (dpvo.py:266-270); `forced` >= 0 replaces the test (0 keep, 1 drop);
 *   2. if dropped: delta_pose = P[k] * P[k-1]^-1 (dpvo.py:276), the edges with ii == k or jj == k leave (remove_factors(...,
 *      store=False), :279), the ids above k are renumbered (:281-283) and ring slots k+1 .. n-1 of every per-frame buffer in
 *      `ring` move down one slot (:289-299);  n' = n - 1, else n' = n;
 *   3. the edges whose source frame kk / M < n' - removal_window leave to the inactive store (remove_factors(..., store=True),
 *      :305-310; with loop_closure != 0 the long-range edges of :307-308 stay), the others are compacted in order into the
 *      spare arrays (*_b).
 * result (device, 8 x int32 followed by 4 + 4 * ceil(E / 1024) ints of look-back scratch that the caller ZEROES ONCE when it allocates
 * the buffer and never touches again; the words are also written to result_host if not
 * NULL -- pinned host memory, see host_words):
 *   [0] decision, [1] edges kept, [2] edges moved to the inactive store, [3] E, [4] 1 if the inactive room was too small
 *   (nothing written beyond it), [5] how many of the kept edges are long-range ones that stayed only because of the loop-closure
 *   rule of dpvo.py:307-308 (0 unless loop_closure != 0): while that count is non-zero the next update() owes the global bundle
 *   adjustment of dpvo.py:348 and the caller must take the call-by-call path.  The caller swaps its array sets and updates its counters when it reads the result --
 *   one frame later, if it likes: nothing on the device waits for the host. */
typedef struct { void* base; int64_t slot_bytes; int64_t ring; } dpvo_ring_t;   /* ring = 0: slot i lives at i, else at i % ring */
typedef struct {
  const int64_t *ii, *jj, *kk; const float *net, *target, *weight;             /* active edges (E) */
  int64_t *ii_b, *jj_b, *kk_b; float *net_b, *target_b, *weight_b;             /* spare set: receives the kept edges */
  int64_t *ii_inac, *jj_inac, *kk_inac; float *target_inac, *weight_inac;      /* tail of the inactive store */
  int64_t inac_room;
  const float* flow4; const float* poses; float* delta_pose;
  int32_t *keep_idx, *rem_idx;                                                 /* scratch, E ints each */
  int64_t* keep_rows;               /* optional [E]: the kept edges' old row numbers as int64 -- with net == net_b == NULL the hidden
                                       state is NOT moved and this list is what dpvo_update_forward_fused_rows takes as net_rows */
  int32_t* result; void* result_host;   /* both 16-byte aligned */
  int32_t host_words;               /* 8: result_host receives the 8 result words; 16: also the 8 words stored in front of `result`
                                       (dpvo_frame_update: flow sums + plan counters).  result_host is device-visible pinned host
                                       memory: the kernel writes it itself, ordered before the completion of the call's last kernel */
  dpvo_ring_t ring[8]; int32_t n_ring;
  int64_t E;
  int32_t n, M, D, keyframe_index, removal_window, loop_closure, optimization_window, forced;
  float keyframe_thresh;
} dpvo_keyframe_step_t;
int dpvo_keyframe_step(const dpvo_keyframe_step_t* a, void* stream);

/* dpvo_frame_update: DPVO.update() + DPVO.keyframe() of one steady-state frame (dpvo.py:328-360,266-310) as ONE call -- with
 * LOOP_CLOSURE too (kf.loop_closure = 1) in every frame that neither appends loop-closure edges nor has long-range edges active
 * (result word [5] of the previous step == 0): the local-BA branch of dpvo.py:351-354 is the one this entry runs -- graph plan (window build, ranged fallback), reproject, two-level correlation, update operator (seven
 * launches), two local BA iterations, flow test + keyframe decision + result record (ev_record), point cloud, the keyframe step's
 * gathers / ring shifts.  Every pointer is a caller
 * buffer (capacity buffers + workspaces sized with the *_workspace_bytes functions for E); ev[0..3] (hipEvent_t or NULL) are
 * recorded before / after the correlation kernel and before / after the update operator (roofline measurement).
 * result_dev: 16 words -- [0..3] flow sums, [4..7] plan counters (float), [8..15] the dpvo_keyframe_step result -- followed by
 * that step's 4 + 4 * ceil(E / 1024) ints of scratch (zeroed once by the caller).  fs (may be NULL): a dpvo_frame_state to issue first (the new frame's patch
 * gathers, state stores and edges: everything between the encoders and the plan), with ev_fs (hipEvent_t or NULL) recorded
 * right behind it. */
typedef struct {
  dpvo_keyframe_step_t kf;          /* edge arrays, rings, decision parameters; kf.flow4 / kf.result are set by the call */
  dpvo_frame_state_t* fs; void* ev_fs;
  void* ev_enc; const void* fmap_spec;
                                    /* ev_enc: hipEvent_t or NULL.  Not NULL (needs fs): an event of ANOTHER stream behind the producer of
                                       fs->fmap / fs->imap (the encoders); the call then issues the part of the frame state that does not read
                                       them, the plan and the reprojection first, waits for the event, and only then gathers gmap / imap and
                                       pools pyramid level 1.  fmap_spec: where the encoders wrote the feature map if that is not the frame's
                                       ring slot (a keyframe dropped since they were enqueued): copied into the slot behind the wait */
  void* ev_record;                  /* hipEvent_t or NULL: recorded as soon as the keyframe step's RESULT RECORD is final (behind its
                                       select kernel; the point cloud and the gathers that execute the decision follow): what the host
                                       waits for before it reads result_host and enqueues the next frame */
  void* ev_update_done;             /* hipEvent_t or NULL: recorded behind the update operator (the caller's side stream may hold the
                                       next frame's encoders back until the two chip-filling kernels are through) */
  void *plan_stream, *ev_plan_fork, *ev_plan_done;
                                    /* all three set (a second hipStream_t of the caller and two hipEvent_t) or all NULL.  Set: the graph
                                       plan -- five launches that only the update operator's SECOND kernel, the BA and the flow test read
                                       -- is issued on plan_stream behind ev_plan_fork (recorded on `stream` behind the new frame's edges)
                                       and joined through ev_plan_done in front of the update operator: reprojection and correlation
                                       do not wait for it.  plan_stream may be the stream the encoders run on: the plan goes behind
                                       whatever that stream already holds */
  int32_t fs_auto;                  /* != 0: the call fills the fields of *fs that depend on the frame number (ring slots, index rows,
                                       edge arrays and counts) from kf.ring[] (order: colours, poses, patches, intrinsics, imap, gmap,
                                       fmap1, fmap2), index_map and the counters; the caller sets the per-frame inputs only */
  int64_t* index_map;
  float* net;                       /* hidden state [capacity, D], updated in place; kf.net / kf.net_b NULL: rows are not moved by the
                                       keyframe step, the previous step's keep_rows come back as net_rows / n_kept (NULL: compact) */
  const int64_t* net_rows; int64_t n_kept;
  float *poses, *patches, *intrinsics, *points; const int64_t* ix;
  const void *gmap, *fmap1, *fmap2, *imap;
  const dpvo_update_fused_params_t* upd;
  float* coords; void* corr; float* delta; int32_t* plan;
  void *ws_plan, *ws_update, *ws_ba; size_t ws_plan_bytes, ws_update_bytes, ws_ba_bytes;
  float* result_dev;
  void* ev[4];
  int64_t m, n_buffer;              /* patches so far; BUFFER_SIZE */
  int32_t P, pmem, mem, H0, W0, H1, W1, patch_lifetime, ba_window, iterations;
  float lmbda, mm_beta;
  float* loop_out; void* ev_loop;   /* loop_out != NULL (LOOP_CLOSURE): behind the keyframe step's gathers the call runs dpvo_loop_flow for the
                                       frame count the NEXT frame will ask PatchGraph.edges_loop for -- n_eval = (kf.n - decision) + 1, the
                                       decision read from the result record on the device: targets [n_eval - loop_freq, n_eval - keyframe_index),
                                       sources [max(l - loop_max_age, 0), l), l = n_eval - removal_window -- into loop_out (device or pinned host
                                       memory): [0] = n_eval, [1] = number of pairs as floats, the flow magnitudes from [2] on (target-major;
                                       capacity: 2 + (loop_freq - keyframe_index) * min(loop_max_age, kf.n + 1 - removal_window) floats).
                                       ev_loop (hipEvent_t or NULL) is recorded behind it.  The next frame's candidate test then costs
                                       no launch and no extra round trip */
  int32_t loop_freq, loop_max_age;  /* GLOBAL_OPT_FREQ, MAX_EDGE_AGE */
} dpvo_frame_update_t;
int dpvo_frame_update(const dpvo_frame_update_t* a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * feature encoders  (Patchifier.fnet / .inet: dpvo/extractor.py:200-264, called at dpvo/net.py:116-117)
 * ---------------------------------------------------------------------------------------------- */

/* Both BasicEncoder4 towers on one normalised f16 image [3,H,W] (planar), H, W multiples of 16:
 *   fmap_out [H/4][W/4][128] = fnet(image) / 4   (InstanceNorm tower)      -- NHWC f16
 *   imap_out [H/4][W/4][384] = inet(image) / 4   (no normalisation)        -- NHWC f16
 * weights: 44 device pointers (22 per tower, fnet first) to f16 tensors repacked by the host:
 *   [0] conv1.weight as [32][192]: K ordered (kh, c, kw) with kw padded 7->8 and K padded 168->192; [1] conv1.bias;
 *   then weight / bias pairs -- weight as [K/32][Cout][32] where K = (kh, kw, cin) is the flattened filter, i.e. the
 *   [Cout][kh][kw][Cin] tensor cut into 32-wide k-steps with the k-step index outermost -- of layer1.0.conv1, layer1.0.conv2, layer1.1.conv1, layer1.1.conv2,
 *   layer2.0.conv1 (stride 2), layer2.0.conv2, layer2.0.downsample.0 (1x1 stride 2), layer2.1.conv1, layer2.1.conv2,
 *   conv2 (1x1).  10 launches for both towers (MIOpen path: ~114): a residual block output is formed by the convolution that
 *   consumes it, layer2.0.downsample rides on the centre tap of layer2.0.conv1. */
size_t dpvo_encoders_workspace_bytes(int H, int W);
int dpvo_encoders_forward(const void* image_f16, const void* const* weights, void* fmap_out, void* imap_out, int H, int W,
                          void* ws, size_t ws_bytes, void* stream);
/* The same with the stream made to wait for hold_event (hipEvent_t or NULL) in front of launch number hold_at (0..9). */
int dpvo_encoders_forward_hold(const void* image_f16, const void* const* weights, void* fmap_out, void* imap_out, int H, int W,
                               void* ws, size_t ws_bytes, void* hold_event, int hold_at, void* stream);
/* F.avg_pool2d(fmap, 4, 4) on an NHWC f16 map (dpvo/dpvo.py:438). */
int dpvo_pool4_nhwc(const void* in, void* out, int h, int w, int C, void* stream);

/* Global BA == cuda_ba.forward(..., eff_impl=True) (ba_cuda.cu:475-478,538-550 with EfficentE, block_e.cu:43-300),
 * used by DPVO.__run_global_BA (dpvo/dpvo.py:312-326).  One Gauss-Newton iteration is
 *     zero S[6N,6N], y[6N];  dpvo_gba_linearize -> S = B - E Q E^T, y = v - E Q u  (block-sparse E, device plan);
 *     dpvo_gba_solve -> S += I*(1e-4*S + 1), dX = cholesky_solve(y, cholesky(S))   (blocked device Cholesky, chol.hip);
 *     dpvo_gba_retract(dX) -> dZ = Q (u - E^T dX), depth and pose retraction in place.
 * f0 / n_frames: first source frame that owns a patch in kk and the number of frames up to the last one
 * (patch p belongs to frame p / M); M = patches per frame (PPF); plan from dpvo_plan_build(ii,jj,kk);
 * the same `ws` (dpvo_gba_workspace_bytes) must be passed to both calls of one iteration (dpvo_gba_retract reads Q, u, the E
 * blocks and the pair run of every source frame that the linearisation left there, for the same f0 / n_frames / t0 / t1). */
size_t dpvo_gba_workspace_bytes(int64_t E, int64_t n_pairs, int64_t n_frames, int M, int64_t n_free);   /* n_free = t1 - t0 */
int dpvo_gba_linearize(const float* poses, const float* patches, const float* intrinsics, const float* target,
                       const float* weight, float lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk,
                       const int32_t* plan, int64_t n_patches, int64_t n_pairs, int64_t E, int P, int M, int f0,
                       int n_frames, int t0, int t1, float* S, float* y, void* ws, size_t ws_bytes, void* stream);
/* The second and later Gauss-Newton iterations of ONE global BA (ba_cuda.cu:538-550 loops `iterations` times over the same edge
 * lists): identical to dpvo_gba_linearize, but the index structures that depend only on (plan, f0, n_frames, t0, t1) -- left in `ws` by
 * the dpvo_gba_linearize call of the first iteration -- are reused instead of rebuilt.  Same arguments, same `ws`, nothing else may
 * have written `ws` in between except dpvo_gba_retract. */
int dpvo_gba_relinearize(const float* poses, const float* patches, const float* intrinsics, const float* target,
                         const float* weight, float lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk,
                         const int32_t* plan, int64_t n_patches, int64_t n_pairs, int64_t E, int P, int M, int f0,
                         int n_frames, int t0, int t1, float* S, float* y, void* ws, size_t ws_bytes, void* stream);
int dpvo_gba_retract(float* poses, float* patches, const int32_t* plan, int64_t n_patches, int64_t n_pairs, int64_t E,
                     int P, int M, int f0, int n_frames, int t0, int t1, const float* dX, void* ws, size_t ws_bytes,
                     void* stream);

/* Dense solve of the damped global-BA system on the device: dX = (S + diag(1e-4 S_ii + 1))^-1 y by a blocked Cholesky
 * factorisation (chol.hip).  Replaces `S += I*(1e-4*S+1); U = linalg_cholesky_ex(S); dX = cholesky_solve(y, U)` of the reference
 * (dpvo/fastba/ba_cuda.cu:546-548) -- the damping is applied here, S and y are left untouched.  S: [n,n] row-major symmetric
 * (the lower triangle is read), y: [n], dX: [n], n = 6 * free poses.  No atomics: bit-repeatable.  A matrix that is not positive
 * definite gives NaNs in dX (the reference ignores cholesky_ex's info the same way, ba_cuda.cu:547). */
size_t dpvo_gba_solve_workspace_bytes(int n);
int dpvo_gba_solve(const float* S, const float* y, int n, float* dX, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * classical loop closure, numerics only  
 * ---------------------------------------------------------------------------------------------- */

/* cuda_ba.solve_system(J_Ginv_i, J_Ginv_j, ii, jj, res, ep, lm, freen) -- dpvo/fastba/ba.cpp:120-180 (exported :188; caller
 * dpvo/loop_closure/optim_utils.py:229): ONE Levenberg-Marquardt step of the Sim(3) pose-graph optimisation.
 *   J_Ginv_i, J_Ginv_j [r,7,7] f32: per-edge Jacobian blocks w.r.t. nodes ii[x], jj[x] (int64 [r], ii[x] != jj[x]); res [r,7] f32.
 *   A = J^T J, b = -J^T res assembled in f64 (the reference: Eigen sparse, double); A.diag += A.diag * lm, then += ep (:152-153);
 *   delta = A^-1 b over the leading 7 freen x 7 freen block when 0 <= freen < n_nodes (the other nodes get 0, :102-118), over all nodes
 *   otherwise; delta [n_nodes,7] f32.  n_nodes = max(ii, jj) + 1 (the reference reads it back with .item(), :131).
 * Dense f64 blocked Cholesky on the device (pgo.hip); info (device int32, may be NULL): 0 ok, 1 an edge with ii == jj or a negative
 * index (the reference calls exit(1), :139-140; this library never exits), 2 the damped matrix is not positive definite. */
size_t dpvo_solve_system_workspace_bytes(int64_t n_nodes, int64_t freen);
int dpvo_solve_system(const float* J_Ginv_i, const float* J_Ginv_j, const int64_t* ii, const int64_t* jj, const float* res, int64_t r,
                      int64_t n_nodes, float ep, float lm, int64_t freen, float* delta, int32_t* info, void* ws, size_t ws_bytes,
                      void* stream);


/* Dev aid: a one-thread kernel that stores the 100 MHz wall clock into *slot (uint64) when it executes: stream-ordered time
 * stamps across streams without a profiler (tools/stream_stamps.py). */
int dpvo_debug_stamp(void* slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPVO_HIP_H */
