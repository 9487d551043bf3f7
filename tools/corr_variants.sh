#!/bin/bash
# tools/corr_variants.sh: corr_pyramid_kernel with parts switched off / made cache-hot (CORR_VARIANT in corr.hip), each as its own
# library next to the product one, timed by tools/corr_bench.py on ONE box.  Results of the variants are wrong on purpose.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
for v in 1 2 4 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DCORR_VARIANT=$v -c corr.hip -o /tmp/corr_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_corrv$v.so /tmp/corr_v$v.o geom.o graph.o update_fused.o ba.o ba_global.o chol.o frontend.o encoder.o track.o capi.o
done
cd $root
echo "product:"; python tools/corr_bench.py 2>&1 | grep "per launch"
for v in 1 2 4 5; do echo "CORR_VARIANT=$v:"; DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_corrv$v.so python tools/corr_bench.py 2>&1 | grep "per launch"; done
echo "product:"; python tools/corr_bench.py 2>&1 | grep "per launch"
