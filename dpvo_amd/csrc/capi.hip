// capi.hip -- ABI bookkeeping of libdpvo_hip.so
#include "common.h"
extern "C" int dpvo_abi_version(void) { return DPVO_ABI_VERSION; }

// dpvo_debug_stamp: a one-thread kernel that writes the 100 MHz wall clock into slot[0] when it EXECUTES -- a stream-ordered
// time stamp for timelines across streams without a profiler attached (tools/stream_stamps.py).  Dev aid, not on any hot path.
namespace { __global__ void stamp_kernel(unsigned long long* slot) { *slot = (unsigned long long)wall_clock64(); } }
extern "C" int dpvo_debug_stamp(void* slot, void* stream) {
  if (!slot) return DPVO_E_INVALID;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
  DPVO_LAUNCH_CHECK();
  return DPVO_OK;
}
