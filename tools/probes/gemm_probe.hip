// gemm_probe.hip -- what bounds the k-loop of update_fused.hip's tile GEMM (RT x 32 rows, 4 waves x 96 features, K = 384)?
// Variants switch off the weight stream (L2 -> VGPR), the B-fragment LDS reads or the MFMAs.  Dev tool:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_probe tools/probes/gemm_probe.hip && /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int PITCH = 784;

template <int RT, int DW, bool WLOAD, bool BREAD, bool MFMA, int DSAHEAD>
__global__ __launch_bounds__(256, 1) void probe(const h8* __restrict__ W, int ncopy, float* out, int layers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 31, h = lane >> 5;
  for (int i = tid; i < RT * 32 * PITCH / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  const char* bl = smem + n * PITCH + 16 * h;
  f16v acc[RT][3];
  for (int r = 0; r < RT; ++r) for (int t = 0; t < 3; ++t) for (int k = 0; k < 16; ++k) acc[r][t][k] = 0.f;
  const h8* wl = W + (size_t)(blockIdx.x % ncopy) * (384 * 384 / 8) + (size_t)w * 24 * 3 * 64 + lane;
  h8 wf[DW][3];
  for (int d = 0; d < DW; ++d) for (int t = 0; t < 3; ++t) wf[d][t] = wl[(d * 3 + t) * 64];
  h8 dummy = wf[0][0];
  for (int L = 0; L < layers; ++L) {
    h8 bf[2][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) bf[0][r] = *reinterpret_cast<const h8*>(bl + r * 32 * PITCH);
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      if (BREAD && s + 1 < 24 && DSAHEAD == 0) {
#pragma unroll
        for (int r = 0; r < RT; ++r) bf[(s + 1) & 1][r] = *reinterpret_cast<const h8*>(bl + r * 32 * PITCH + (s + 1) * 32);
      }
      if (MFMA) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % DW][t], bf[BREAD ? (s & 1) : 0][r], acc[r][t], 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) dummy += wf[s % DW][t];
#pragma unroll
        for (int r = 0; r < RT; ++r) dummy += bf[BREAD ? (s & 1) : 0][r];
      }
      if (WLOAD) {
        const int sn = (s + DW) % 24;
#pragma unroll
        for (int t = 0; t < 3; ++t) wf[s % DW][t] = wl[(sn * 3 + t) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = (float)dummy[0];
  for (int r = 0; r < RT; ++r) for (int t = 0; t < 3; ++t) for (int k = 0; k < 16; ++k) s += acc[r][t][k];
  if (s == 12345.678f) out[0] = s;
}

template <typename K>
void run(const char* name, K kern, int RT, const h8* W, int ncopy, float* out, int grid, int layers) {
  const int lds = RT * 32 * PITCH;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, W, ncopy, out, layers);
  hipEventRecord(a);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, W, ncopy, out, layers);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  const double per = ms * 1e3 / layers / ((grid + 255) / 256);      // us per layer-tile (per round)
  const double flops = 2.0 * RT * 32 * 384 * 384 * (double)grid * layers;
  printf("%-44s RT=%d grid=%4d ncopy=%3d  %8.2f us/layer-tile  (%6.0f clk @2.4GHz per k-step)  %7.1f TFLOP/s-equiv\n", name, RT, grid, ncopy,
         per, per * 2400 / 24, flops / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t wn = (size_t)384 * 384 * 64;     // 64 copies
  h8* W; hipMalloc(&W, wn * 2);
  std::vector<_Float16> hw(wn); for (size_t i = 0; i < wn; ++i) hw[i] = (_Float16)(0.01f * (float)((i * 7) % 13));
  hipMemcpy(W, hw.data(), wn * 2, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, 4);
  const int L = 32;
  for (int grid : {256, 512}) {
    run("full  DW=6", probe<3, 6, true, true, true, 0>, 3, W, 1, out, grid, L);
    run("full  DW=6, 64 weight copies", probe<3, 6, true, true, true, 0>, 3, W, 64, out, grid, L);
    run("full  DW=3", probe<3, 3, true, true, true, 0>, 3, W, 1, out, grid, L);
    run("full  DW=10", probe<3, 10, true, true, true, 0>, 3, W, 1, out, grid, L);
    run("no W stream", probe<3, 6, false, true, true, 0>, 3, W, 1, out, grid, L);
    run("no B reads", probe<3, 6, true, false, true, 0>, 3, W, 1, out, grid, L);
    run("MFMA only", probe<3, 6, false, false, true, 0>, 3, W, 1, out, grid, L);
    run("W stream + B reads, no MFMA", probe<3, 6, true, true, false, 0>, 3, W, 1, out, grid, L);
    run("W stream only", probe<3, 6, true, false, false, 0>, 3, W, 1, out, grid, L);
    run("W stream only, 64 copies", probe<3, 6, true, false, false, 0>, 3, W, 64, out, grid, L);
  }
  run("full RT=4 DW=6", probe<4, 6, true, true, true, 0>, 4, W, 1, out, 256, L);
  run("full RT=6 DW=4", probe<6, 4, true, true, true, 0>, 6, W, 1, out, 256, L);
  run("full RT=2 DW=6", probe<2, 6, true, true, true, 0>, 2, W, 1, out, 256, L);
  run("MFMA only RT=6", probe<6, 4, false, false, true, 0>, 6, W, 1, out, 256, L);
  return 0;
}
