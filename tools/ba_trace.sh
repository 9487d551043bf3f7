#!/bin/bash
# tools/ba_trace.sh: builds dpvo_amd/libdpvo_hip_bat.so = the shipped objects with ba.hip recompiled with -DBA_TRACE (start / end stamps
# of every workgroup of ba_patch_kernel) and prints the longest workgroups by role (per-patch blocks / one B-row block per free pose).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops -DBA_TRACE -c ba.hip -o /tmp/ba_bat.o 2>&1 | grep -v "not a recognized" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_bat.so corr.o geom.o graph.o update_fused.o /tmp/ba_bat.o ba_global.o chol.o frontend.o encoder.o track.o capi.o
cd $root
if [ "$1" != "build" ]; then DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_bat.so python tools/ba_trace.py; fi
