// tools/probes/graph_probe.hip -- would capturing the frame's ~30 dependent launches in a hipGraph shorten them?  A chain of N small
// dependent kernels (each reads what the previous one wrote; `work` blocks of 256 threads) issued (a) as N stream launches, (b) as one
// hipGraphLaunch of the captured chain: device time of the chain (HIP events) and host time to issue it.  On the frame's path a launch
// costs ~5 us whatever it does; this shows how much of that a graph takes away on this ROCm.
//   hipcc --offload-arch=gfx950 -O3 -o graph_probe.bin graph_probe.hip && ./graph_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void blocker(float* p, int iters) {          // ~0.5 ms of dependent FMAs on one wave: lets the host pre-fill the queue
  float x = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 1.0f;
  p[threadIdx.x] = x;
}
__global__ void link(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(i + 1) % n] * 1.0001f + 1.0f;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int N = 30, reps = 200;
  for (int blocks : {1, 64, 2048}) {
    const int n = blocks * 256;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(link, dim3(blocks), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n); };
    // (a) stream launches
    chain(); CK(hipStreamSynchronize(st));
    double host_a = 0; float dev_a = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st));
      const double t0 = now_us(); chain(); host_a += now_us() - t0;
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); dev_a += ms;
    }
    // (b) the same chain captured once, replayed
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal)); chain(); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    double host_b = 0; float dev_b = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st));
      const double t0 = now_us(); CK(hipGraphLaunch(ge, st)); host_b += now_us() - t0;
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); dev_b += ms;
    }
    // (c) both again BEHIND a 0.5 ms kernel, so that the whole chain is in the queue before its first kernel may start (the situation
    //     of the frame's tail: the host is a frame ahead of the device there)
    float dev_c = 0, dev_d = 0;
    for (int r = 0; r < reps / 4; ++r) {
      hipLaunchKernelGGL(blocker, dim3(1), dim3(64), 0, st, a, 150000);
      CK(hipEventRecord(e0, st)); chain(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); dev_c += ms;
      hipLaunchKernelGGL(blocker, dim3(1), dim3(64), 0, st, a, 150000);
      CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1)); dev_d += ms;
    }
    printf("%4d blocks: queued behind a 0.5 ms kernel: stream launches %.2f us per kernel on the device, hipGraphLaunch %.2f\n", blocks,
           1e3 * dev_c / (reps / 4) / N, 1e3 * dev_d / (reps / 4) / N);
    printf("%4d blocks x 256 threads, chain of %d dependent kernels: stream launches %.2f us per kernel on the device (host %.2f us per launch); "
           "hipGraphLaunch %.2f us per kernel on the device (host %.1f us per graph launch = %.2f per kernel)\n",
           blocks, N, 1e3 * dev_a / reps / N, host_a / reps / N, 1e3 * dev_b / reps / N, host_b / reps, host_b / reps / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(a)); CK(hipFree(b)); CK(hipStreamDestroy(st));
  }
  return 0;
}
