// tools/probes/valu_issue_probe.hip -- how fast does ONE wave issue VALU instructions on gfx950, and does a second wave on the same SIMD
// run beside it for free?  (Round 5: the one-wave Cholesky chains came out at ~7 cycles per VALU instruction whatever the instruction;
// K1 / K7 of the update operator run one wave per SIMD.)  Each kernel: `waves` waves per workgroup (wave i -> SIMD i % 4), every wave
// executes N instructions of one kind in an unrolled loop; shader cycles (s_memtime) per instruction are printed per wave count.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe.bin valu_issue_probe.hip && ./valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  const float b = 1.0001f, c = 0.5f;
  const f2 b2 = {b, b}, c2 = {c, c};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) {            // one dependent chain of v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
      } else if (KIND == 1) {     // eight independent chains of v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "v"(c));
      } else if (KIND == 2) {     // eight independent chains of v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(b2), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(b2), "v"(c2));
      } else if (KIND == 3) {     // conversions: f32 -> packed f16 (what the update kernels' epilogues are made of)
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a0), "v"(a1)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a2), "v"(a3)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a4), "v"(a5)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a6), "v"(a7)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a1), "v"(a0)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a3), "v"(a2)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a5), "v"(a4)); asm volatile("" :: "v"(r));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a7), "v"(a6)); asm volatile("" :: "v"(r));
      } else {                    // v_exp_f32 (quarter rate), independent
        asm volatile("v_exp_f32 %0, %0" : "+v"(a0)); asm volatile("v_exp_f32 %0, %0" : "+v"(a1));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a2)); asm volatile("v_exp_f32 %0, %0" : "+v"(a3));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a4)); asm volatile("v_exp_f32 %0, %0" : "+v"(a5));
        asm volatile("v_exp_f32 %0, %0" : "+v"(a6)); asm volatile("v_exp_f32 %0, %0" : "+v"(a7));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 1024, per_iter = 16 * 8;
  for (int waves : {1, 4, 8, 16}) {
    for (int blocks : {1, 256}) {
      hipMemset(cyc, 0, 4096 * 16 * 8);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      // s_memtime / readcyclecounter counts at a fixed 100 MHz on gfx9: report wall time per instruction too
      printf("%-28s %2d wave(s) per workgroup (%d per SIMD), %3d workgroup(s): %6.2f ns per instruction per wave (kernel %.1f us, counter ticks %llu)\n",
             name, waves, (waves + 3) / 4, blocks, ms * 1e6 / (iters * per_iter), ms * 1e3, h[0]);
    }
  }
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 4096 * 16 * 8);
  run<0>("v_fma_f32 dependent chain", out, cyc);
  run<1>("v_fma_f32 8 independent", out, cyc);
  run<2>("v_pk_fma_f32 8 independent", out, cyc);
  run<3>("v_cvt_pk_f16_f32 independent", out, cyc);
  run<4>("v_exp_f32 8 independent", out, cyc);
  return 0;
}
