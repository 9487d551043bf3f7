// capi.hip -- ABI bookkeeping of libdpvo_hip.so
#include "common.h"
extern "C" int dpvo_abi_version(void) { return 1; }
