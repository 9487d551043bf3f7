/*
 * dpvo_oracle.c -- CPU oracle for the DPVO per-frame hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99 restatement of the reference's altcorr / fastba / lietorch-SE3 / projective_ops
 * arithmetic (see oracle_body.inc for per-function reference file:line citations) plus the
 * integer graph bookkeeping (neighbors, unique, reduce_edges).  Built by oracle/Makefile into
 * oracle/liboracle.so; only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke()
 * may load it, as the checker.  The product (dpvo_amd/) never links or imports it.
 *
 * Parity status: the reference has no tests / golden vectors for this path;
 * native-kernel parity is therefore "unpinned" by reference tests.  What pins this file:
 * lietorch's algebraic identities (run_tests.py:16-52), goldens produced by importing the
 * reference's own Python (tests/golden/make_golden.py: Update.forward, pops.transform, patchify
 * glue, reduce_edges, and the independent Python bundle adjustment dpvo/ba.py:86-182 for orc_ba)
 * and internal cross-checks.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SUFFIX _f32
#define SQRT sqrtf
#define SIN sinf
#define COS cosf
#define ATAN atanf
#define FABS fabsf
#define FLOOR floorf
#include "oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef SIN
#undef COS
#undef ATAN
#undef FABS
#undef FLOOR

#define REAL double
#define SUFFIX _f64
#define SQRT sqrt
#define SIN sin
#define COS cos
#define ATAN atan
#define FABS fabs
#define FLOOR floor
#include "oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef SQRT
#undef SIN
#undef COS
#undef ATAN
#undef FABS
#undef FLOOR


/* ---------------------------------------------------------------------------------------------
 * altcorr in the REFERENCE'S OWN ARITHMETIC for half features (scalar_t = c10::Half):
 *   corr_forward_kernel (correlation_kernel.cu:121-131): `scalar_t s = 0; ... s += f1[j] * f2[j];` over the channels in
 *   order -- c10::Half operators compute in float and round the result to half (c10/util/Half-inl.h), so every product
 *   and every partial sum is rounded to f16;
 *   the ATen blend (:221-230): dx, dy = (x - floor(x)).to(half); every elementwise op on half tensors computes in float
 *   and rounds its result to half: (1 - dx), (1 - dx) * (1 - dy), (.) * corr, out += (.), left to right.
 * Inputs: features as float arrays holding f16-representable values; coords float.  Output float (f16-representable),
 * layout of orc_corr_forward (before the final permute).  This is the "fp16-rounding emulation" of SURVEY.md 8c: what the
 * reference CUDA kernels return bit for bit, up to double rounding in the float -> half conversions (the float result of a
 * two-half sum can itself be inexact when the exponents differ by more than 13: a tie can then round the other way).
 * ------------------------------------------------------------------------------------------- */
static inline float orc_h16(float f) {
  /* round to nearest even to IEEE binary16, return the value as float (overflow -> inf, subnormals handled) */
  union { float f; uint32_t u; } v; v.f = f;
  const uint32_t sign = v.u & 0x80000000u;
  uint32_t a = v.u & 0x7fffffffu;
  if (a >= 0x7f800000u) return f;                              /* inf / nan */
  if (a >= 0x477ff000u) {                                      /* >= 65520: rounds to inf */
    v.u = sign | 0x7f800000u; return v.f;
  }
  if (a < 0x38800000u) {                                       /* < 2^-14: half subnormal, quantum 2^-24 */
    v.u = a;
    float q = v.f * 16777216.0f;                               /* exact scaling by 2^24 */
    q = nearbyintf(q);                                         /* RNE in the default rounding mode */
    q *= 5.9604644775390625e-8f;                               /* 2^-24 */
    v.f = q; v.u |= sign; return v.f;
  }
  /* normal: keep 10 mantissa bits, RNE on the 13 dropped bits */
  const uint32_t lsb = (a >> 13) & 1u;
  a += 0x0fffu + lsb;
  a &= ~0x1fffu;
  v.u = sign | a; return v.f;
}

/* the rounding itself, exported so that tests can pin it against numpy's float16 conversion */
void orc_round_h16(const float* x, int64_t n, float* y) { for (int64_t i = 0; i < n; i++) y[i] = orc_h16(x[i]); }

void orc_corr_forward_h16(const float* fmap1, const float* fmap2, const float* coords, const int64_t* us,
                          const int64_t* vs, int64_t E, int C, int P, int H2, int W2, int radius, float* out) {
  const int D = 2 * radius + 2;
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < E; m++) {
    float* raw = (float*)malloc(sizeof(float) * (size_t)D * D * P * P);
    const float* f1 = fmap1 + (int64_t)us[m] * C * P * P;
    const float* f2 = fmap2 + (int64_t)vs[m] * C * H2 * W2;
    for (int i0 = 0; i0 < P; i0++)
      for (int j0 = 0; j0 < P; j0++) {
        float x = coords[((m * 2 + 0) * P + i0) * P + j0];
        float y = coords[((m * 2 + 1) * P + i0) * P + j0];
        float fx = floorf(x), fy = floorf(y);
        for (int a = 0; a < D; a++)
          for (int b = 0; b < D; b++) {
            float s = 0;
            if (fx == fx && fy == fy && fabsf(fx) < 1e9f && fabsf(fy) < 1e9f) {
              long i1 = (long)fy + (a - radius);
              long j1 = (long)fx + (b - radius);
              if (i1 >= 0 && i1 < H2 && j1 >= 0 && j1 < W2)
                for (int c = 0; c < C; c++) {
                  const float pr = orc_h16(f1[(c * P + i0) * P + j0] * f2[((int64_t)c * H2 + i1) * W2 + j1]);
                  s = orc_h16(s + pr);
                }
            }
            raw[((a * D + b) * P + i0) * P + j0] = s;
          }
      }
    for (int a = 0; a < D - 1; a++)
      for (int b = 0; b < D - 1; b++)
        for (int i0 = 0; i0 < P; i0++)
          for (int j0 = 0; j0 < P; j0++) {
            float x = coords[((m * 2 + 0) * P + i0) * P + j0];
            float y = coords[((m * 2 + 1) * P + i0) * P + j0];
            const float dx = orc_h16(x - floorf(x)), dy = orc_h16(y - floorf(y));
            const float mx = orc_h16(1.0f - dx), my = orc_h16(1.0f - dy);
#define RAW(aa, bb) raw[(((aa) * D + (bb)) * P + i0) * P + j0]
            float o = orc_h16(orc_h16(mx * my) * RAW(a, b));
            o = orc_h16(o + orc_h16(orc_h16(dx * my) * RAW(a, b + 1)));
            o = orc_h16(o + orc_h16(orc_h16(mx * dy) * RAW(a + 1, b)));
            o = orc_h16(o + orc_h16(orc_h16(dx * dy) * RAW(a + 1, b + 1)));
#undef RAW
            out[(((m * (D - 1) + a) * (D - 1) + b) * P + i0) * P + j0] = o;
          }
    free(raw);
  }
}

/* ---------------------------------------------------------------------------------------------
 * Integer bookkeeping (bit-exact contract).
 * ------------------------------------------------------------------------------------------- */

typedef struct { int64_t key; int64_t idx; } kv_t;

static int kv_cmp(const void* a, const void* b) {
  const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);   /* stable */
}

/* torch::_unique(x, sorted=true, return_inverse=true) (ba_cuda.cu:447, ba.cpp:62, blocks.py:41).
 * uniq must hold n entries; returns the number of unique values. */
int64_t orc_unique(const int64_t* x, int64_t n, int64_t* uniq, int64_t* inverse) {
  if (n == 0) return 0;
  kv_t* a = (kv_t*)malloc(sizeof(kv_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) { a[i].key = x[i]; a[i].idx = i; }
  qsort(a, (size_t)n, sizeof(kv_t), kv_cmp);
  int64_t m = 0;
  for (int64_t i = 0; i < n; i++) {
    if (i == 0 || a[i].key != a[i - 1].key) uniq[m++] = a[i].key;
    inverse[a[i].idx] = m - 1;
  }
  free(a);
  return m;
}

/* fastba.neighbors (ba.cpp:59-97): for every edge, the previous / next edge of the same patch
 * (first argument, "ii" there == kk at the call site net.py:80) in stable jj order; -1 at ends. */
typedef struct { int64_t k, j, idx; } rec_t;

static int rec_cmp(const void* p, const void* q) {
  const rec_t* x = (const rec_t*)p; const rec_t* y = (const rec_t*)q;
  if (x->k != y->k) return x->k < y->k ? -1 : 1;
  if (x->j != y->j) return x->j < y->j ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

void orc_neighbors(const int64_t* kk, const int64_t* jj, int64_t n, int64_t* ix, int64_t* jx) {
  if (n == 0) return;
  /* sort by (kk, jj, original index): identical to grouping by unique(kk) then stable_sort by jj */
  rec_t* a = (rec_t*)malloc(sizeof(rec_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) { a[i].k = kk[i]; a[i].j = jj[i]; a[i].idx = i; }
  qsort(a, (size_t)n, sizeof(rec_t), rec_cmp);
  for (int64_t i = 0; i < n; i++) {
    ix[a[i].idx] = (i > 0 && a[i - 1].k == a[i].k) ? a[i - 1].idx : -1;
    jx[a[i].idx] = (i < n - 1 && a[i + 1].k == a[i].k) ? a[i + 1].idx : -1;
  }
  free(a);
}

/* reduce_edges (loop_closure/optim_utils.py:23-60): greedy NMS over candidate loop edges sorted by
 * flow magnitude.  argsort must be numpy's default (quicksort, not stable): the caller passes the
 * permutation `order` = np.argsort(flow_mag) so that tie-breaking is exactly numpy's.
 * out [max_num_edges+1, 2]; returns the number of edges written. */
int64_t orc_reduce_edges(const double* flow_mag, const int64_t* ii, const int64_t* jj, const int64_t* order,
                         int64_t n, int64_t max_num_edges, int64_t nms, int64_t* out) {
  if (n == 0) return 0;
  int64_t Ni = 0, Nj = 0;
  for (int64_t i = 0; i < n; i++) { if (ii[i] + 1 > Ni) Ni = ii[i] + 1; if (jj[i] + 1 > Nj) Nj = jj[i] + 1; }
  unsigned char* ignore = (unsigned char*)calloc((size_t)(Ni * Nj), 1);
  int64_t cnt = 0;          /* len(es) - 1 */
  for (int64_t t = 0; t < n; t++) {
    int64_t idx = order[t];
    if (cnt + 1 > max_num_edges) break;            /* len(es) > max_num_edges */
    int64_t i = ii[idx], j = jj[idx];
    double mag = flow_mag[idx];
    if ((j - i) < 30) continue;
    if (mag >= 1000) continue;
    if (ignore[i * Nj + j]) continue;
    out[2 * cnt + 0] = i; out[2 * cnt + 1] = j; cnt++;
    for (int64_t di = -nms; di <= nms; di++) {
      int64_t i1 = i + di;
      if (0 <= i1 && i1 < Ni) ignore[i1 * Nj + j] = 1;
    }
  }
  free(ignore);
  return cnt;
}
