#!/bin/bash
# tools/gba_trace.sh: builds dpvo_amd/libdpvo_hip_gbt.so = the shipped objects with ba_global.hip recompiled with -DGBA_TRACE (phase
# stamps of one wave of gba_row_kernel) and runs tools/gba_bench.py on it.  Dev tool.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DGBA_TRACE -c ba_global.hip -o /tmp/bag_gbt.o 2>&1 | grep -v "not a recognized" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_gbt.so corr.o geom.o graph.o update_fused.o update_fused_k7.o ba.o /tmp/bag_gbt.o chol.o frontend.o encoder.o track.o capi.o
cd $root
if [ "$1" != "build" ]; then GBA_TRACE=1 GBA_SIZES=${GBA_SIZES:-100,400} DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_gbt.so python tools/gba_bench.py; fi
