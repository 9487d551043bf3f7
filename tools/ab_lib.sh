#!/bin/bash
# tools/ab_lib.sh "<extra hipcc flags for the update operator's units>" [suffix=ab] : builds dpvo_amd/libdpvo_hip_<suffix>.so = the shipped
# objects with update_fused.hip and update_fused_k7.hip recompiled with the extra flags (an A/B partner for tools/update_bench.py on ONE
# box: DPVO_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../dpvo_amd/csrc"
sfx=${2:-ab}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $1 -c update_fused.hip -o /tmp/uf_$sfx.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $1 -c update_fused_k7.hip -o /tmp/uf7_$sfx.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_$sfx.so corr.o geom.o graph.o /tmp/uf_$sfx.o /tmp/uf7_$sfx.o ba.o ba_global.o chol.o frontend.o encoder.o track.o capi.o
echo built ../libdpvo_hip_$sfx.so with "$1"
