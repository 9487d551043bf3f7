#!/bin/bash
# tools/kf_trace.sh: builds dpvo_amd/libdpvo_hip_kft.so = the shipped objects with track.hip recompiled with -DKF_TRACE (phase stamps
# inside kf_decide_kernel), runs the bench loop on it and prints the phases of the last frame's keyframe kernel.  Dev tool.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DKF_TRACE -c track.hip -o /tmp/track_kft.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_kft.so corr.o geom.o graph.o update_fused.o ba.o ba_global.o chol.o frontend.o encoder.o /tmp/track_kft.o capi.o
cd $root
if [ "$1" != "build" ]; then DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_kft.so python tools/kf_trace.py; fi
