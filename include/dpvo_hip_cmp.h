/* dpvo_hip_cmp.h -- C ABI of libdpvo_hip_cmp.so: the COMPARATOR implementation of the update operator.
 *
 * Not part of the product (libdpvo_hip.so / dpvo_hip.h), never loaded by the tracker: a second, independently written
 * implementation of Update.forward (reference dpvo/net.py:74-92) that the parity tests and the measurement tools run beside
 * the product's seven-launch operator (dpvo_update_forward_fused, update_fused.hip):
 *   - update.hip     the launch-by-launch composite of generic pieces (round 1: 23 launches, 780 us), whose pieces double as
 *                    stand-alone checks of a Linear / LayerNorm / gather-add / heads against torch.
 * (Rounds 2-4 also carried two patch-major cuts, update_pm.hip / update_pm2.hip: 700 / 602 us against 561 for the seven launches;
 * measured, recorded in DESIGN.md 3.4 and profiles/README.md, removed in round 5.)
 * The library links against libdpvo_hip.so (dpvo_softagg, dpvo_plan_layout).  Conventions as in dpvo_hip.h. */
#ifndef DPVO_HIP_CMP_H
#define DPVO_HIP_CMP_H
#include "dpvo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* nn.Linear under autocast: y = half(x_half @ W_half^T + b_half), f32 accumulate on MFMA
 * (v_mfma_f32_16x16x32_f16).  A [M,K] f16 or f32 (converted to f16 on load, as autocast does) with
 * leading dimension lda; optional row gather `rows` (int32, -1 -> zero row: the mask_ix * net[:,ix]
 * of net.py:81-85); W [N,K] f16 row-major (torch Linear weight layout) with leading dimension ldw;
 * bias [N] f16; out f16 [M,ldo] or the f32 residual target, see DPVO_EPI_*; K % 32 == 0, N % 16 == 0.
 * out16 (optional, RESADD / GATED with f16 A only): also store the updated residual row as f16 [M,ld16], the
 * operand image of the next Linear (autocast would cast it on the fly).
 * f16 A runs the LDS-DMA kernel (global_load_lds_dwordx4, 3-stage ring); f32 A the register-staged one. */
int dpvo_linear(const void* A, int a_dtype, int64_t lda, const int32_t* rows, const void* W, int64_t ldw,
                const void* bias, void* out, int64_t ldo, const void* gate, int64_t ldg, void* out16, int64_t ld16,
                int epilogue, int n_split, int64_t M, int N, int K, void* stream);

/* Fused "net = LayerNorm(net + inp[inp_rows] + corr)" (net.py:77-78) and plain LayerNorm (eps 1e-3):
 *   x [M,384] f32 or f16 (x_dtype), optional add1 f16 [.,384] gathered by add1_rows (int64 indices
 *   taken modulo add1_mod, the ctx = imap[:, kk % (M*pmem)] of dpvo.py:334), optional add2 f16 [M,384];
 *   gamma/beta f32 [384]; y_f32 [M,384] (may alias x when x is f32) and/or y_f16 [M,384] (optionally
 *   relu'd: the LN -> ReLU -> Linear of Update.corr, net.py:55-59). */
int dpvo_layernorm(const void* x, int x_dtype, const void* add1, const int64_t* add1_rows, int64_t add1_mod,
                   const void* add2, const float* gamma, const float* beta, float eps, float* y_f32, void* y_f16,
                   int relu_f16, int64_t M, int D, void* stream);

/* net[e] += float(hy[group[e]])  -- the `self.h(y)[:,jx]` expand + residual of net.py:87-88; optional f16 image. */
int dpvo_gather_add(float* net, const void* hy, const int32_t* group, void* net16, int64_t E, int D, void* stream);

/* Heads: delta = d(net), weight = sigmoid(w(net)) (net.py:61-71,92) as one row-dot kernel;
 * Wd,Ww [2,D] f16, bd,bw [2] f16; outputs f32 [E,2] (the .float() of dpvo.py:339-340 folded in). */
int dpvo_heads(const float* net, const void* Wd, const void* bd, const void* Ww, const void* bw, float* delta,
               float* weight, int64_t E, int D, void* stream);
/* Same, and additionally target = coords[:, :, P/2, P/2] + delta (dpvo/dpvo.py:340; coords [E,2,P,P] f32, target [E,2]). */
int dpvo_heads_target(const float* net, const void* Wd, const void* bd, const void* Ww, const void* bw, const float* coords,
                      int P, float* delta, float* weight, float* target, int64_t E, int D, void* stream);

/* The whole update operator (dpvo/net.py:74-92, Update.forward) as one call: exactly the launch sequence a host would
 * issue through dpvo_linear / dpvo_layernorm / dpvo_softagg / dpvo_gather_add / dpvo_heads_target.  Weight images (f16
 * unless noted; the host packs them once, see dpvo_amd/net.py:Update.pack):
 *   c0 = corr.0 padded to K = 896, c2 = corr.2, cln = corr.3 LayerNorm (f32), c5 = corr.5, norm (f32),
 *   c1 / c2n = the two neighbour MLPs (.0 and .2), akk / aij = SoftAgg: wfg = [f; g] stacked (768 rows), wh = h,
 *   g0 / g1 = gru.0+gru.1 / gru.2+gru.3: LayerNorm (f32), wrg = [res.0; gate.0] stacked, w2 = res.2, d / w = the heads.
 * net [E,384] f32 in; inp f16 rows gathered by inp_rows modulo inp_mod (the imap ring, dpvo.py:334); corr [E,ld_corr] f16 with
 * columns 882..895 zero; plan from dpvo_plan_build(_ranged) with upper bounds on its two group counts; coords [E,2,P,P] and
 * target [E,2] optional (both or neither).  Outputs: net_out [E,384] f32 (may alias net), delta, weight [E,2] f32. */
typedef struct {
  const void *c0_w, *c0_b, *c2_w, *c2_b;
  const float *cln_g, *cln_b;
  const void *c5_w, *c5_b;
  const float *norm_g, *norm_b;
  const void *c1_w0, *c1_b0, *c1_w2, *c1_b2, *c2n_w0, *c2n_b0, *c2n_w2, *c2n_b2;
  const void *akk_wfg, *akk_bfg, *akk_wh, *akk_bh, *aij_wfg, *aij_bfg, *aij_wh, *aij_bh;
  const float *g0_g, *g0_b;
  const void *g0_wrg, *g0_brg, *g0_w2, *g0_b2;
  const float *g1_g, *g1_b;
  const void *g1_wrg, *g1_brg, *g1_w2, *g1_b2;
  const void *d_w, *d_b, *w_w, *w_b;
} dpvo_update_params_t;
size_t dpvo_update_workspace_bytes(int64_t E, int64_t max_groups);
int dpvo_update_forward(const dpvo_update_params_t* params, const float* net, const void* inp, const int64_t* inp_rows,
                        int64_t inp_mod, const void* corr, int64_t ld_corr, const int32_t* plan, int64_t n_patches_ub,
                        int64_t n_pairs_ub, const float* coords, int P, float* net_out, float* delta, float* weight,
                        float* target, int64_t E, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPVO_HIP_CMP_H */
