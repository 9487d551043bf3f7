// Minimal reproducer for the packed-FP32 operand-select fault found in ba_pair_kernel (profiles/README.md, round 2).
// One wave per workgroup runs ONE packed instruction form in a self-checking loop; the host runs it beside another kernel
// (tools/probes/pk_opsel_probe.py drives it next to the encoders).  The surrounding arithmetic is inline
// scalar assembly so that the only packed instruction in the loop is the one under test:
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC pk_opsel_probe.hip -o libpk_opsel_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f2 __attribute__((ext_vector_type(2)));
// scalar arithmetic the compiler cannot re-pack into v_pk_* (the unit is compiled with packed ops available, the assembler
// needs the feature for the instruction under test)
__device__ __forceinline__ float M(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float A(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// out[0..3]: faults of the LOW result per lane quarter; out[4..7]: of the HIGH result; out[8]: low result equal to the
// value the UNSELECTED half would have given; out[9]: equal to the previous iteration's result; out[10]: anything else;
// out[11]: iterations executed (sanity); out[12..19]: first fault seen: flag, lane, iteration, bits of the result, a.lo, a.hi,
// b.lo, b.hi
template <int FORM> __global__ __launch_bounds__(64) void pk_probe(int iters, unsigned* out, int nop) {
  const int lane = threadIdx.x;
  f2 a = {float(lane + 3), float(2 * lane + 5)}, b = {float(lane % 7 + 2), float(lane % 5 + 11)}, c = {1.0f, 2.0f};
  unsigned bad_lo = 0, bad_hi = 0, k_unsel = 0, k_stale = 0, k_other = 0;
  f2 prev = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    a.x = A(a.x, 1.0f); a.y = A(a.y, 2.0f); b.x = A(b.x, 1.0f); b.y = A(b.y, 3.0f);
    if (a.x > 900.0f) { a.x = A(a.x, -800.0f); a.y = A(a.y, -1600.0f); b.x = A(b.x, -800.0f); b.y = A(b.y, -2400.0f); }   // products stay exact in f32
    f2 d, e, u;                                                       // d: instruction under test, e: expected, u: unselected
    if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.x, b.y), M(a.y, b.x)}; u = (f2){M(a.x, b.x), M(a.y, b.y)}; }
    if (FORM == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.y, b.x), M(a.x, b.y)}; u = (f2){M(a.x, b.x), M(a.y, b.y)}; }
    if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.x, b.x), M(a.y, b.y)}; u = (f2){M(a.x, b.y), M(a.y, b.x)}; }
    if (FORM == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.x, b.x), M(a.x, b.y)}; u = (f2){M(a.x, b.x), M(a.y, b.y)}; }
    if (FORM == 4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                     e = (f2){A(M(a.x, b.y), c.x), A(M(a.y, b.x), c.y)}; u = (f2){A(M(a.x, b.x), c.x), A(M(a.y, b.y), c.y)}; }
    if (FORM == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){A(a.x, b.y), A(a.y, b.x)}; u = (f2){A(a.x, b.x), A(a.y, b.y)}; }
    if (FORM == 6) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){a.y, b.x}; u = (f2){a.x, b.y}; }
    if (FORM == 7) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.y, b.y), M(a.x, b.x)}; u = (f2){M(a.x, b.x), M(a.y, b.y)}; }
    if (FORM == 8) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
                     e = (f2){M(a.x, b.y), M(a.y, b.y)}; u = (f2){M(a.x, b.x), M(a.y, b.y)}; }
    if (FORM == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                     e = (f2){A(M(a.x, b.x), c.y), A(M(a.y, b.y), c.x)}; u = (f2){A(M(a.x, b.x), c.x), A(M(a.y, b.y), c.y)}; }
    if (FORM == 10) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                     e = (f2){A(M(a.y, b.x), c.x), A(M(a.x, b.y), c.y)}; u = (f2){A(M(a.x, b.x), c.x), A(M(a.y, b.y), c.y)}; }
    if (FORM == 11) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                     e = (f2){A(M(a.x, b.y), c.y), A(M(a.y, b.x), c.x)}; u = (f2){A(M(a.x, b.x), c.x), A(M(a.y, b.y), c.y)}; }
    if (FORM == 12) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                     e = (f2){A(M(a.y, b.y), c.x), A(M(a.x, b.x), c.y)}; u = (f2){A(M(a.x, b.x), c.x), A(M(a.y, b.y), c.y)}; }
    if (nop) asm volatile("s_nop 3");
    if (d.x != e.x && atomicCAS(&out[12], 0u, 1u) == 0u) {
      out[13] = lane; out[14] = it; out[15] = __float_as_uint(d.x); out[16] = __float_as_uint(a.x); out[17] = __float_as_uint(a.y);
      out[18] = __float_as_uint(b.x); out[19] = __float_as_uint(b.y);
    }
    if (d.x != e.x) { ++bad_lo; if (d.x == u.x) ++k_unsel; else if (d.x == prev.x) ++k_stale; else ++k_other; }
    if (d.y != e.y) { ++bad_hi; }
    prev = d;
  }
  if (bad_lo) atomicAdd(&out[lane >> 4], bad_lo);
  if (bad_hi) atomicAdd(&out[4 + (lane >> 4)], bad_hi);
  if (k_unsel) atomicAdd(&out[8], k_unsel);
  if (k_stale) atomicAdd(&out[9], k_stale);
  if (k_other) atomicAdd(&out[10], k_other);
  if (lane == 0 && blockIdx.x == 0) out[11] = iters;
}

// Synthetic neighbours: one instruction class each, 256 threads per workgroup, a long unrolled loop.
template <int KIND> __global__ __launch_bounds__(256) void pk_neighbour(int iters, float* sink) {
  __shared__ float lds[256];
  const int t = threadIdx.x;
  float x = t * 0.5f, y = 1.0f + t, z = 0.25f; double dx = t; unsigned h = 0x3c003800u + t, sel = 0x07060100u;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  f4 acc = {0.f, 0.f, 0.f, 0.f}; h8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(t + i); fb[i] = (_Float16)(i); }
  lds[t] = x;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(x) : "v"(h), "v"(h));
      if (KIND == 1) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x) : "v"(h));
      if (KIND == 2) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"((t ^ 16) * 4));
      if (KIND == 3) asm volatile("v_pk_max_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(h) : "v"(sel));
      if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dx) : "v"(dx));
      if (KIND == 5) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
      if (KIND == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(h) : "v"(h), "v"(sel));
      if (KIND == 7) { float2 p = {x, y}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p)); x = p.x; y = p.y; }
      if (KIND == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      if (KIND == 9) asm volatile("v_accvgpr_write_b32 a0, %1\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(x) : "v"(y) : "a0");
      if (KIND == 10) asm volatile("v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(x) : "v"(h));
      if (KIND == 12) { typedef float f16v __attribute__((ext_vector_type(16))); static_assert(sizeof(f16v) == 64, ""); }
      if (KIND == 12) asm volatile("v_mfma_f32_32x32x16_f16 a[0:15], %0, %1, a[0:15]" :: "v"(fa), "v"(fb) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
      if (KIND == 13) asm volatile("v_mfma_f32_16x16x16_f16 a[0:3], %0, %1, a[0:3]" :: "v"(dx), "v"(dx) : "a0","a1","a2","a3");
      if (KIND == 14) asm volatile("v_mfma_f32_32x32x8_f16 a[0:15], %0, %1, a[0:15]" :: "v"(dx), "v"(dx) : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15");
      if (KIND == 15) asm volatile("v_mfma_f32_16x16x4_f32 a[0:3], %0, %1, a[0:3]" :: "v"(x), "v"(y) : "a0","a1","a2","a3");
      if (KIND == 16) asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], %0, %1, a[0:3]" :: "v"(fa), "v"(fb) : "a0","a1","a2","a3");
      if (KIND == 17) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
      if (KIND == 18) asm volatile("v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]" :: "v"(dx), "v"(dx) : "a0","a1","a2","a3","a4","a5","a6","a7");
      if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %0 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(dx));
    }
  }
  if (x == 123.456f || h == 77u || dx == 1.5 || acc[0] == 3.25f) sink[t] = x + acc[1];
}

extern "C" int pk_neighbour_launch(void* stream, int kind, int iters, int blocks, float* sink) {
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: pk_neighbour<0><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 1: pk_neighbour<1><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 2: pk_neighbour<2><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 3: pk_neighbour<3><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 4: pk_neighbour<4><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 5: pk_neighbour<5><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 6: pk_neighbour<6><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 7: pk_neighbour<7><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 8: pk_neighbour<8><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 9: pk_neighbour<9><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 10: pk_neighbour<10><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 11: pk_neighbour<11><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 12: pk_neighbour<12><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 13: pk_neighbour<13><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 14: pk_neighbour<14><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 15: pk_neighbour<15><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 16: pk_neighbour<16><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 17: pk_neighbour<17><<<blocks, 256, 0, s>>>(iters, sink); break;
    case 18: pk_neighbour<18><<<blocks, 256, 0, s>>>(iters, sink); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int pk_probe_launch(void* stream, int form, int iters, int blocks, unsigned* out, int nop) {
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(out, 0, 20 * sizeof(unsigned), s);
  switch (form) {
    case 0: pk_probe<0><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 1: pk_probe<1><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 2: pk_probe<2><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 3: pk_probe<3><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 4: pk_probe<4><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 5: pk_probe<5><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 6: pk_probe<6><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 7: pk_probe<7><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 8: pk_probe<8><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 9: pk_probe<9><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 10: pk_probe<10><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 11: pk_probe<11><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    case 12: pk_probe<12><<<blocks, 64, 0, s>>>(iters, out, nop); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
