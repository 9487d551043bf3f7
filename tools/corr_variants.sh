#!/bin/bash
# tools/corr_variants.sh: corr_pyramid_kernel with parts switched off / made cache-hot (tools/probes/corr_variant.hip -- the measurement
# kernel's own translation unit, linked beside the product objects into dpvo_amd/libdpvo_hip_corrvar.so), timed by tools/corr_bench.py on
# ONE box.  Variant 0 must reproduce the product's checksum; the results of the others are wrong on purpose.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c $root/tools/probes/corr_variant.hip -o /tmp/corr_variant.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_corrvar.so /tmp/corr_variant.o corr.o geom.o graph.o update_fused.o update_fused_k7.o ba.o ba_global.o chol.o frontend.o encoder.o track.o capi.o
[ "$1" = build ] && exit 0
cd $root
echo "product:"; python tools/corr_bench.py 2>&1 | grep "per launch"
for v in 0 1 2 4 5 6 7 8 9 10 11 12; do echo "CORR_VARIANT=$v:"; CORR_VARIANT=$v DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_corrvar.so python tools/corr_bench.py 2>&1 | grep "per launch"; done
echo "product:"; python tools/corr_bench.py 2>&1 | grep "per launch"
