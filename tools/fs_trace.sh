#!/bin/bash
# tools/fs_trace.sh: builds dpvo_amd/libdpvo_hip_fst.so = the shipped objects with frontend.hip recompiled with -DFS_TRACE (start / end
# stamps of every workgroup of frame_state_kernel) and prints the longest workgroups by role for both parts of a steady-state frame.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DFS_TRACE -c frontend.hip -o /tmp/frontend_fst.o 2>&1 | grep -v "not a recognized" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_fst.so corr.o geom.o graph.o update_fused.o ba.o ba_global.o chol.o /tmp/frontend_fst.o encoder.o track.o capi.o
cd $root
if [ "$1" != "build" ]; then DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_fst.so python tools/fs_trace.py; fi
