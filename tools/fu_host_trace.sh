#!/bin/bash
# tools/fu_host_trace.sh: builds dpvo_amd/libdpvo_hip_fht.so = the shipped objects with track.hip recompiled with -DFU_HOST_TRACE (host
# time between the steps of dpvo_frame_update) and prints the mean per step over the steady-state frames, plan aside off / on.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DFU_HOST_TRACE -c track.hip -o /tmp/track_fht.o 2>&1 | grep -v "not a recognized" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_fht.so corr.o geom.o graph.o update_fused.o update_fused_k7.o ba.o ba_global.o chol.o frontend.o encoder.o /tmp/track_fht.o capi.o
cd $root
if [ "$1" != "build" ]; then for g in 0 1; do PLAN_ASIDE=$g DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_fht.so python tools/fu_host_trace.py; done; fi
