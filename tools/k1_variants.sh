#!/bin/bash
# tools/k1_variants.sh [build]: what would the update operator's first kernel gain if the correlation rows did not come from memory
# (correlation fused into K1 through LDS)?  Builds dpvo_amd/libdpvo_hip_k1nc.so = the product objects with update_fused.hip
# recompiled from a PATCHED COPY in which K1's correlation chunk loads are replaced by register values (results wrong on purpose; the
# product source carries no measurement switch), then times K1 in both libraries with tools/update_bench.py under rocprofv3.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
sed 's|d\[i\] = \*reinterpret_cast<const h8\*>(p.corr + g \* p.ld_corr + kc \* KCH + ch \* 8);|d[i] = (h8)(_Float16)(0.001f * (float)((int)g \& 15));|' update_fused.hip > /tmp/update_fused_k1nc.hip
grep -q "0.001f \* (float)((int)g" /tmp/update_fused_k1nc.hip || { echo "patch did not apply"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I$root/dpvo_amd/csrc -c /tmp/update_fused_k1nc.hip -o /tmp/uf_k1nc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_k1nc.so corr.o geom.o graph.o /tmp/uf_k1nc.o update_fused_k7.o ba.o ba_global.o chol.o frontend.o encoder.o track.o capi.o
[ "$1" = build ] && exit 0
cd /tmp && export TMPDIR=/tmp
for v in "" k1nc; do
  rm -rf /tmp/k1v$v; DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip${v:+_$v}.so WHICH=fused REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1v$v -- python $root/tools/update_bench.py > /dev/null 2>&1
  f=$(find /tmp/k1v$v -name "*kernel_stats.csv" | xargs ls -S | head -1); echo "== ${v:-product}"; python $root/tools/kstats.py $f 3
done
