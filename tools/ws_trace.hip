// Dev tool: per-workgroup timeline of linear_ws_kernel (build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_TRACE
// -I include tools/ws_trace.hip -o gpurun_out/ws_trace).  Prints the 100 MHz wall-clock stamps of wave 0..3 of a few blocks.
#include "../dpvo_amd/csrc/update.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int epi = argc > 1 ? atoi(argv[1]) : 0;
  const int64_t M = 47712; const int N = 384, K = 384;
  _Float16 *A, *W, *b, *o16, *gate; void* out;
  hipMalloc(&A, M * K * 2); hipMalloc(&W, N * K * 2); hipMalloc(&b, N * 2); hipMalloc(&out, M * N * 4);
  hipMalloc(&o16, M * N * 2); hipMalloc(&gate, M * N * 2);
  hipMemset(A, 0, M * K * 2); hipMemset(W, 0, N * K * 2); hipMemset(b, 0, N * 2); hipMemset(out, 0, M * N * 4);
  hipMemset(gate, 0, M * N * 2);
  const bool rmw = epi == DPVO_EPI_RESADD || epi == DPVO_EPI_GATED;
  for (int it = 0; it < 3; ++it)
    dpvo_linear(A, DPVO_F16, K, nullptr, W, K, b, out, N, epi == DPVO_EPI_GATED ? gate : nullptr, N, rmw ? o16 : nullptr, N,
                epi, 0, M, N, K, nullptr);
  hipDeviceSynchronize();
  static unsigned long long h[512][4][32];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_trace), sizeof(h));
  unsigned long long t0 = ~0ull;
  for (int b2 = 0; b2 < 256; ++b2) for (int w = 0; w < 4; ++w) if (h[b2][w][0] < t0) t0 = h[b2][w][0];
  const char* names[] = {"entry", "bias", "w0", "w1", "w2", "w3", "w4", "w5", "prolog", "b0 bar", "b0 mma", "b0 e1", "b1 bar", "b1 mma",
                         "b1 e1", "b2 bar", "b2 mma", "b2 e1", "b3 bar", "b3 mma", "b3 e1", "b4 bar", "b4 mma", "b4 e1",
                         "b5 bar", "b5 mma", "b5 e1", "warm0", "warm8", "pre-w", "t0 iss", "t1 iss"};
  for (int b2 : {0, 100, 233}) {
    printf("block %d (10 ns ticks since first entry)\n", b2);
    for (int i = 0; i < 32; ++i) {
      printf("  %-7s", names[i]);
      for (int w = 0; w < 4; ++w) printf(" %8lld", (long long)(h[b2][w][i] - t0));
      printf("\n");
    }
  }
  unsigned long long tmax = 0;
  for (int b2 = 0; b2 < 256; ++b2) for (int w = 0; w < 4; ++w) for (int i = 0; i < 27; ++i) if (h[b2][w][i] > tmax && h[b2][w][i] - t0 < 100000) tmax = h[b2][w][i];
  printf("last stamp %lld\n", (long long)(tmax - t0));
  return 0;
}
