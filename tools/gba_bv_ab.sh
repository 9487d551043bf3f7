#!/bin/bash
# tools/gba_bv_ab.sh [build]: the row kernel of this round (B / v part in registers, a tile's operands all in flight at M = 96) against the one of rounds 4-5
# (-DGBA_BV_REG=0 -DGBA_TILE96=0 -DGBA_XCD=0, or whatever GBA_AB_DEFS names; dpvo_amd/libdpvo_hip_bv0.so): bit comparison of the system and of the result (tools/gba_bits.py), then the times
# (tools/gba_bench.py) under both.  `build` only builds the comparison library (hipcc, no GPU).  Dev tool.
set -e
DEFS=${GBA_AB_DEFS:--DGBA_BV_REG=0 -DGBA_TILE96=0 -DGBA_XCD=0}      # the comparison build (default: the row kernel of rounds 4-5)
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/dpvo_amd/csrc
if [ ! -f ../libdpvo_hip_bv0.so ] || [ "$1" == "build" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $DEFS -c ba_global.hip -o /tmp/bag_bv0.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdpvo_hip_bv0.so corr.o geom.o graph.o update_fused.o update_fused_k7.o ba.o /tmp/bag_bv0.o chol.o frontend.o encoder.o track.o capi.o
fi
cd $root
[ "$1" == "build" ] && exit 0
F='amdgpu\|Warning\|autocast\|warnings.warn'
echo "== bits: product"; python tools/gba_bits.py 2>&1 | grep -v "$F" | tee /tmp/bits_a.txt
echo "== bits: $DEFS"; DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_bv0.so python tools/gba_bits.py 2>&1 | grep -v "$F" | tee /tmp/bits_b.txt
cmp /tmp/bits_a.txt /tmp/bits_b.txt && echo "BIT-IDENTICAL" || echo "DIFFERENT"
echo "== times: product"; GBA_SIZES=${GBA_SIZES:-50,100,200,400} python tools/gba_bench.py 2>&1 | grep -v "$F"
echo "== times: $DEFS"; GBA_SIZES=${GBA_SIZES:-50,100,200,400} DPVO_HIP_LIB=$root/dpvo_amd/libdpvo_hip_bv0.so python tools/gba_bench.py 2>&1 | grep -v "$F"
