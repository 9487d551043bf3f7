// clock_probe.hip -- what shader clock does a box run at, idle and under lock-step MFMA load?  ("clock /
// power telemetry beside s_memrealtime" for the box-to-box spread of the one-workgroup-per-CU kernels.)
//
// A "meter" wave executes a fixed chain of dependent v_fma_f32 (N of them, a constant number of shader cycles whatever else the
// chip does on other SIMDs) between two reads of the 100 MHz wall clock: cycles / wall time = effective shader clock, up to the
// constant "cycles per dependent FMA", which the idle run calibrates.  Load: 4 waves per CU issuing back-to-back
// v_mfma_f32_32x32x16_f16 (the instruction of the update operator), all CUs starting together; the meter rides on SIMD 0 of
// every CU as a fifth wave.  Reported: meter time alone on one CU / alone on all CUs / beside the MFMA load, in consecutive
// 50 us slices from the start of the load (a power-management response shows up as slices getting slower), and the MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/probes/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CHAIN = 2048;          // dependent FMAs per slice
constexpr int SLICES = 64;

__global__ __launch_bounds__(320) void probe_kernel(unsigned long long* __restrict__ stamps, unsigned long long* __restrict__ load_t, float* __restrict__ sink, int mfma_iters,
                                                    int with_load, int chain_reps) {
  const int w = threadIdx.x >> 6;
  if (w == 4) {
    // the meter
    float x = (float)threadIdx.x * 1e-9f, a = 1.0000001f, b = 1e-9f;
    unsigned long long* out = stamps + (size_t)blockIdx.x * (SLICES + 1);
    unsigned long long t = wall_clock64();
    if ((threadIdx.x & 63) == 0) out[0] = t;
    for (int s = 0; s < SLICES; ++s) {
      for (int r = 0; r < chain_reps; ++r) {
#pragma unroll 64
        for (int i = 0; i < CHAIN; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
      }
      t = wall_clock64();
      if ((threadIdx.x & 63) == 0) out[s + 1] = t;
    }
    if (x == 12345.f) sink[0] = x;
  } else if (with_load) {
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f16v acc0, acc1, acc2, acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
    const unsigned long long tl0 = wall_clock64();
    for (int it = 0; it < mfma_iters; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    if (s == 12345.f) sink[1] = s;
    if (threadIdx.x == 0) { load_t[2 * blockIdx.x] = tl0; load_t[2 * blockIdx.x + 1] = wall_clock64(); }
  }
}

static bool g_json = false;
static double g_last_ghz = 0, g_last_meter = 0;
static void run(const char* name, int grid, int with_load, int mfma_iters, int chain_reps, unsigned long long* d_st, float* d_sink) {
  std::vector<unsigned long long> st((size_t)grid * (SLICES + 1));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(320), 0, 0, d_st, d_st + (size_t)1024 * (SLICES + 1), d_sink, mfma_iters, with_load, chain_reps);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(320), 0, 0, d_st, d_st + (size_t)1024 * (SLICES + 1), d_sink, mfma_iters, with_load, chain_reps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  // per slice: median over the workgroups of the slice time (100 MHz ticks -> us)
  if (!g_json) printf("%-44s kernel %.1f us;", name, ms * 1e3);
  std::vector<double> med(SLICES);
  for (int s = 0; s < SLICES; ++s) {
    std::vector<double> v(grid);
    for (int g = 0; g < grid; ++g) v[g] = (double)(st[(size_t)g * (SLICES + 1) + s + 1] - st[(size_t)g * (SLICES + 1) + s]) * 0.01;
    std::sort(v.begin(), v.end());
    med[s] = v[grid / 2];
  }
  double all = 0; for (double m : med) all += m;
  g_last_meter = CHAIN * chain_reps / (all / SLICES) * 1e-3;          // dependent FMAs per ns
  if (!g_json) printf(" meter slice (%d dependent FMAs): mean %.2f us = %.0f MHz-equivalent at 4 cycles per FMA\n   every 4th slice:", CHAIN * chain_reps,
         all / SLICES, CHAIN * chain_reps * 4.0 / (all / SLICES));
  if (!g_json) { for (int s = 0; s < SLICES; s += 4) printf(" %.2f", med[s]); printf("\n"); }
  if (with_load) {
    // the load's own duration per workgroup (wave 0's stamps): 4 independent MFMAs per iteration, 8 passes = 32 cycles each when
    // the pipe is kept full -> shader clock while the load ran
    std::vector<unsigned long long> lt((size_t)2 * grid);
    hipMemcpy(lt.data(), d_st + (size_t)1024 * (SLICES + 1), lt.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> d(grid);
    for (int g = 0; g < grid; ++g) d[g] = (double)(lt[2 * g + 1] - lt[2 * g]) * 0.01;
    std::sort(d.begin(), d.end());
    const double us = d[grid / 2];
    g_last_ghz = (double)mfma_iters * 4 * 32 / us * 1e-3;
    const double flops = (double)grid * 4 * mfma_iters * 4 * 2.0 * 32 * 32 * 16;
    if (!g_json) printf("   MFMA load: median workgroup %.1f us (min %.1f, max %.1f) -> %.2f GHz at 32 cycles per MFMA, %.0f TFLOP/s while it ran\n", us, d[0],
                        d[grid - 1], g_last_ghz, flops / (us * 1e-6) * 1e-12);
  }
}

int main(int argc, char** argv) {
  g_json = argc > 1 && std::string(argv[1]) == "--json";
  unsigned long long* d_st; float* d_sink;
  hipMalloc(&d_st, ((size_t)1024 * (SLICES + 1) + 2048) * 8);
  hipMalloc(&d_sink, 64);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  if (g_json) {
    run("", 256, 0, 0, 1, d_st, d_sink); const double m_idle = g_last_meter;
    run("", 64, 1, 6000, 1, d_st, d_sink); const double q = g_last_ghz;
    run("", 256, 1, 6000, 1, d_st, d_sink); const double f = g_last_ghz;
    run("", 256, 1, 30000, 5, d_st, d_sink); const double f5 = g_last_ghz;
    printf("{\"mfma_clock_ghz_64_cus\": %.3f, \"mfma_clock_ghz_256_cus\": %.3f, \"mfma_clock_ghz_256_cus_3ms\": %.3f, \"dependent_fma_per_ns_idle\": %.4f, \"nominal_ghz\": %.1f}\n",
           q, f, f5, m_idle, p.clockRate * 1e-6);
    return 0;
  }
  printf("%s, %d CUs, clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run("meter alone, 1 workgroup", 1, 0, 0, 1, d_st, d_sink);
  run("meter alone, 256 workgroups", 256, 0, 0, 1, d_st, d_sink);
  run("meter beside MFMA load, 256 workgroups", 256, 1, 6000, 1, d_st, d_sink);
  run("meter beside MFMA load, 64 workgroups", 64, 1, 6000, 1, d_st, d_sink);
  run("meter beside MFMA load, 256 workgroups, 5x longer", 256, 1, 30000, 5, d_st, d_sink);
  run("meter beside MFMA load, 512 workgroups (2 per CU)", 512, 1, 6000, 1, d_st, d_sink);
  run("meter alone, 256 workgroups (after the load)", 256, 0, 0, 1, d_st, d_sink);
  return 0;
}
